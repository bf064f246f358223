"""Seeded synthetic corpus / queries of the benchmark configs.

Corpus X ~ N(0,1) float32, generated in fixed 1M-row chunks -- chunk ``c`` comes from
``np.random.default_rng([1234, c])`` -- so any shard or row range is reproducible without
holding N x d on the host.  Queries come from ``default_rng(4321)``.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Iterator, Tuple

import numpy as np

CHUNK_ROWS = 1_000_000
CORPUS_SEED = 1234
QUERY_SEED = 4321


def corpus_chunk(c: int, dim: int, rows: int = CHUNK_ROWS) -> np.ndarray:
    """First ``rows`` rows of chunk ``c`` (a prefix of the chunk is the same numbers)."""
    return np.random.default_rng([CORPUS_SEED, int(c)]).standard_normal((rows, dim), dtype=np.float32)


def corpus_rows(first: int, n: int, dim: int) -> np.ndarray:
    """Rows [first, first+n) of the synthetic corpus (may span chunks)."""
    out = np.empty((n, dim), dtype=np.float32)
    done = 0
    while done < n:
        row = first + done
        c, off = divmod(row, CHUNK_ROWS)
        take = min(n - done, CHUNK_ROWS - off)
        out[done:done + take] = corpus_chunk(c, dim, off + take)[off:]
        done += take
    return out


def iter_corpus(first: int, n: int, dim: int, piece_rows: int = 250_000, threads: int = 8
                ) -> Iterator[Tuple[int, np.ndarray]]:
    """Yield (row_offset, rows) pieces of rows [first, first+n), generated ahead by a thread pool.

    Pieces never cross a 1M-row chunk boundary; each chunk is generated once and sliced.
    """
    jobs = []
    row = first
    end = first + n
    while row < end:
        c, off = divmod(row, CHUNK_ROWS)
        take = min(end - row, CHUNK_ROWS - off)
        jobs.append((c, off, take, row - first))
        row += take
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        window = max(2, threads)
        futures = []
        it = iter(jobs)

        def submit():
            j = next(it, None)
            if j is not None:
                futures.append((j, pool.submit(corpus_chunk, j[0], dim, j[1] + j[2])))

        for _ in range(window):
            submit()
        while futures:
            (c, off, take, base), fut = futures.pop(0)
            chunk = fut.result()
            submit()
            for s in range(off, off + take, piece_rows):
                e = min(s + piece_rows, off + take)
                yield base + (s - off), chunk[s:e]
            del chunk


def queries(nq: int, dim: int, batch_index: int = 0) -> np.ndarray:
    """Query batch ``batch_index``: batch 0 is SURVEY 8d's ``default_rng(4321)``; batches 1, 2, ... come from
    ``default_rng([4321, i])`` (bench.py rotates through several, so that thresholds, candidate counts and branch
    behaviour are not those of one draw)."""
    seed = QUERY_SEED if batch_index == 0 else [QUERY_SEED, int(batch_index)]
    return np.random.default_rng(seed).standard_normal((nq, dim), dtype=np.float32)
