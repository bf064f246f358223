#!/usr/bin/env python3
"""Headline benchmark: queries/sec of exact cosine kNN (k=10) over a row-sharded 768-d corpus.

  python bench.py --gpus N --steps K --warmup W

N=1 runs BASELINE.json configs[2] (10M x 768, batch 256, 1 x MI355X).  N>1 runs one process per GPU: either the
caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) or a bare
`python bench.py --gpus N` starts them itself as a child `torch.distributed.run` and relays rank 0's JSON line and the
exit code (launch_plan / spawn_ranks); a WORLD_SIZE that contradicts --gpus is an error, never a silent 1-GPU run.
Every rank holds 10M rows (weak scaling; N=8 is configs[4], 80M rows, batch 1024), scans its shard for the whole query
wave, and rank 0 merges the per-shard top-k on the host.

A step = one query wave through the hot path with queries and corpus resident in HBM:
query prep -> int8- (cosine) or bf16-MFMA filter scan of the shard -> threshold updates -> exact fp64 rescoring
(-> gather + host merge when N>1).  Prints ONE JSON line (rank 0).  roofline.achieved: algorithmic
bytes of the scan launches / their HIP-event durations (events around filter_scan_asm_kernel only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)


def kernel_source_sha16() -> str:
    """Hash of what the default scan kernels are built from -- the generated int8 bodies, the shared device helpers and
    the C++ wrapper of the assembly (addresses, in-kernel scatter) -- so that a committed PMC measurement is tied to a
    kernel version; variant / diagnostic plumbing elsewhere in the sources does not invalidate it."""
    import hashlib

    import importlib.util

    h = hashlib.sha256()
    csrc = ROOT / "mlvectordb_amd" / "csrc"
    # the generated bodies are build products (not tracked): their text is regenerated here from tools/gen_scan_asm.py,
    # exactly as `make` writes scan_asm_<space>_i8_va.inc
    spec = importlib.util.spec_from_file_location("_mlvdb_gen_scan_asm", ROOT / "tools" / "gen_scan_asm.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for space in ("cosine", "l2", "ip"):
        h.update(gen.default_i8_body(space).encode())
    h.update((csrc / "scan_common.h").read_bytes())
    src = (csrc / "kernels_filter.hip").read_text()
    a = src.index("void filter_scan_asm_kernel(")
    b = src.index("// ------------------------------------------------------------------ threshold update + compaction")
    h.update(src[a:b].encode())
    return h.hexdigest()[:16]


def log(msg: str) -> None:
    print(f"[bench r{os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=0, help="queries per wave (default 256 at 1 GPU, 1024 beyond)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--space", default="cosine")
    ap.add_argument("--strategy", default="auto")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-2 (1M x 768, batch 1) side measurement")
    ap.add_argument("--tombstones", type=float, default=0.0,
                    help="secondary run (SURVEY 8d): tombstone this fraction of every shard's rows, default_rng(99), before searching")
    ap.add_argument("--gen-threads", type=int, default=0)
    ap.add_argument("--wave-mode", choices=["synchronised", "back_to_back"], default="synchronised",
                    help="N=1: what the timed K steps do between waves (both are measured; this one is `ms_per_step` / `value`)")
    ap.add_argument("--query-batches", type=int, default=8,
                    help="distinct seeded query batches the timed loops rotate through (each gated once, untimed)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="N=1: also time >= this many seconds (and >= 1000 waves) in both wave modes (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (gloo for single-device rehearsals)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--in-process", action="store_true",
                    help="ONE process, Index(devices=[...]): the row-sharded path behind the Protocol surface (not the "
                         "driver's mode: that is one process per GPU under torch.distributed.run)")
    ap.add_argument("--devices", default="", help="--in-process: comma-separated device list (a device may repeat: "
                                                  "logical shards); default 0..gpus-1")
    return ap.parse_args()


def in_process_main(args, json_fd) -> None:
    """`--in-process`: every namespace shard is an mlvdb_index of THIS process (mlvectordb_amd/multi_device.py): one host
    thread per shard issues the scans, the per-shard top-k (fp64 distances) are merged on the host.  A step = one wave of
    `batch` queries through Index.search_many (host-pointer entries: PCIe of queries and results included).  Same
    synthetic rows and fatal parity gate as the default mode; weak scaling (rows_per_gpu rows per shard)."""
    from mlvectordb_amd import Index, synth

    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    g = len(devices)
    n_local, d, k = args.rows_per_gpu, args.dim, args.k
    batch = args.batch or (256 if g == 1 else 1024)
    t0 = time.perf_counter()
    index = Index(space=args.space, devices=devices, strategy=args.strategy, capacity_hint=g * n_local)
    for off, rows in synth.iter_corpus(0, g * n_local, d, threads=args.gen_threads or 16):
        index.add_arrays(rows, "bench")
    load_s = time.perf_counter() - t0
    eng = index._ns["bench"].engine
    shards = getattr(eng, "shards", [eng])
    log(f"{g} shards on devices {devices}: {[s.counts()[0] for s in shards]} rows after {load_s:.1f} s")
    q_host = synth.queries(batch, d)
    hits = index.search_many(q_host, k, "bench", args.space)
    for s in shards:
        s.set_strategy("exact")
    exact = index.search_many(q_host, k, "bench", args.space)
    for s in shards:
        s.set_strategy(args.strategy)
    verify = {"queries_compared": int(batch), "merged_ids_equal_merged_exact_scans": bool(np.array_equal(hits.labels, exact.labels)),
              "max_abs_score_err": float(np.abs(hits.scores - exact.scores).max())}
    out = {"metric": "queries/sec, exact cosine kNN k=10 over a row-sharded Nx768 fp32 corpus (10M rows per GPU)",
           "unit": "queries/s (each query scanned against one shard; whole-corpus QPS = value / n_gpus)",
           "n_gpus": g, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "data": "synthetic", "mode": "in-process: Index(devices=...) behind the Protocol surface",
           "config": {"workload": f"{g * n_local} x {d} fp32 N(0,1) rows ({n_local}/shard), {args.space} kNN k={k}, batch={batch}",
                      "devices": devices, "rows_per_gpu": n_local, "dim": d, "k": k, "batch": batch},
           "parity_gate": verify, "load_s": round(load_s, 1)}
    if not verify["merged_ids_equal_merged_exact_scans"] or verify["max_abs_score_err"] > 1e-5:
        log(f"PARITY GATE FAILED: {verify}")
        out.update({"value": None, "error": "parity gate failed"})
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        sys.exit(1)
    for _ in range(args.warmup):
        index.search_many(q_host, k, "bench", args.space)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        index.search_many(q_host, k, "bench", args.space)
    unpipelined = time.perf_counter() - t_start
    # throughput mode: Index.search_stream -- the shard scans of wave i+1 are queued before wave i is merged (round 3)
    t_start = time.perf_counter()
    n_waves = sum(1 for _ in index.search_stream((q_host for _ in range(args.steps)), k, "bench", args.space))
    elapsed = time.perf_counter() - t_start
    assert n_waves == args.steps
    out.update({"value": round(g * batch * args.steps / elapsed, 1), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                "ms_per_step_unpipelined_search_many": round(unpipelined / args.steps * 1e3, 3),
                "whole_corpus_qps": round(batch * args.steps / elapsed, 1),
                "dtype": "i8 / bf16 (MFMA bounds) + f64 (exact rescoring of the f32 rows)"})
    index.close()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def config1_side(device: int) -> dict:
    """BASELINE configs[0]: 10k x 128 random vectors, cosine k=5, batch 1 through QueryProcessor.find_similar
    (reference query_processor.py:19-49): the same Protocol-level flow on the HIP engine and, as the CPU baseline of
    this config, on the NumPy oracle engine (the reference itself cannot run: hnswlib is absent).  Part of the
    cpu_baseline leg (rank 0, N=1)."""
    from mlvectordb_amd import Index, InMemoryStorage, QueryProcessor, VectorDTO
    from oracle.engine import OracleScanEngine

    n, d, k, nq = 10_000, 128, 5, 120
    rows = np.random.default_rng(1234).standard_normal((n, d), dtype=np.float32)
    queries = np.random.default_rng(4321).standard_normal((nq, d), dtype=np.float32)

    def run(index):
        qp = QueryProcessor(InMemoryStorage(), index)
        qp.upsert_many([VectorDTO(values=r, metadata={"i": i}) for i, r in enumerate(rows)], namespace="bench")
        lat, out = [], []
        for i in range(nq):
            t0 = time.perf_counter()
            hits = qp.find_similar(VectorDTO(values=queries[i], metadata={}), top_k=k, namespace="bench", metric="cosine")
            lat.append(time.perf_counter() - t0)
            out.append([(h["metadata"]["i"], h["score"]) for h in hits])
        index.close()
        lat = np.array(lat[20:])
        return {"p50_ms": round(float(np.median(lat)) * 1e3, 4), "qps": round(1.0 / float(np.mean(lat)), 1)}, out

    res = {}
    res["hip"], got = run(Index(space="cosine", device=device))
    res["cpu_numpy_oracle_engine"], want = run(Index(space="cosine", engine_factory=OracleScanEngine))
    res["ids_equal"] = all([g[0] for g in a] == [w[0] for w in b] for a, b in zip(got, want))
    res["max_abs_score_err"] = float(max(abs(g[1] - w[1]) for a, b in zip(got, want) for g, w in zip(a, b)))
    return res


def config4_side(eng_l2, q_host: np.ndarray, k: int) -> dict:
    """BASELINE configs[3]: squared-l2 kNN + range query (radius = mean k-th neighbour distance) over the
    same 10M x 768 rows, batch 256, host-pointer entries (PCIe of queries / results included).  Parity: kNN and range
    hits of every query against the exact fp64 scans."""
    nq = q_host.shape[0]
    labels, dist, _ = eng_l2.search(q_host, k)
    st = eng_l2.last_stats()
    eng_l2.set_strategy("exact")
    ex_l, ex_d, _ = eng_l2.search(q_host, k)
    eng_l2.set_strategy("auto")
    t = []
    for _ in range(35):  # (5 unrecorded: the first waves after the exact scan above run at another clock; then as the cosine leg)
        ts = time.perf_counter()
        eng_l2.search(q_host, k)
        t.append(time.perf_counter() - ts)
    knn_ms = float(np.median(t[5:])) * 1e3
    radius = float(dist[:, k - 1].mean())
    hits = eng_l2.range(q_host, radius, 8192)
    st_r = eng_l2.last_stats()
    t = []
    for _ in range(25):
        ts = time.perf_counter()
        eng_l2.range(q_host, radius, 8192)
        t.append(time.perf_counter() - ts)
    range_ms = float(np.median(t[5:])) * 1e3
    eng_l2.set_strategy("exact")
    hits_exact = eng_l2.range(q_host, radius, 8192)
    eng_l2.set_strategy("auto")
    nh = [len(h[0]) for h in hits]
    range_ok = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(hits, hits_exact))
    return {"knn_ms_per_wave_host_io": round(knn_ms, 3), "knn_qps": round(nq / knn_ms * 1e3, 1),
            "knn_bound_dtype": {1: "bf16", 2: "i8"}.get(st.get("bound_dtype"), "?"),
            "range_ms_per_wave_host_io": round(range_ms, 3), "range_qps": round(nq / range_ms * 1e3, 1),
            "range_over_knn": round(range_ms / knn_ms, 3), "range_fallback_queries": int(st_r["fallback_queries"]),
            "radius_squared_l2": radius, "mean_hits_per_query": float(np.mean(nh)), "max_hits": int(max(nh)),
            "parity": {"knn_ids_equal_exact_scan_all_queries": bool(np.array_equal(labels, ex_l)),
                       "knn_max_abs_err": float(np.abs(dist - ex_d).max()),
                       "range_hits_equal_exact_range_scan_all_queries": bool(range_ok)}}


def launch_plan(args, environ, argv):
    """What `python bench.py --gpus N` has to do before anything touches the GPU (north star: "reported at
    1, 2, 4 and 8 GPUs").  Returns ("run", None) when this process is a rank (or the one-GPU / in-process job),
    ("spawn", cmd) when it was started bare with --gpus N > 1 and has to start its N ranks itself, and ("error", msg)
    when the environment contradicts the command line -- a run that asked for N GPUs never reports fewer."""
    world = environ.get("WORLD_SIZE")
    if args.in_process:
        return "run", None
    if world is not None:
        if int(world) != args.gpus:
            return "error", (f"--gpus {args.gpus} but WORLD_SIZE={world}: start the job with "
                             f"--nproc-per-node {args.gpus} (or run `python bench.py --gpus {args.gpus}` bare)")
        return "run", None
    if args.gpus <= 1:
        return "run", None
    import socket

    with socket.socket() as s:  # a free rendezvous port on the loopback (the container hostname may not resolve)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    return "spawn", cmd


def spawn_ranks(cmd) -> int:
    """Start the N ranks as a CHILD process (never exec: this process may not be replaced once a GPU runtime could have
    been initialised), relay the one JSON line rank 0 prints and return the child's exit code."""
    import subprocess

    log("bare --gpus N: starting " + " ".join(cmd[1:8]) + " ...")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return proc.returncode if (lines or proc.returncode) else 1


def main() -> None:
    args = parse_args()
    what, arg = launch_plan(args, os.environ, sys.argv[1:])
    if what == "error":
        log(arg)
        sys.exit(2)
    if what == "spawn":
        sys.exit(spawn_ranks(arg))
    # stdout carries exactly one JSON line: libraries that print there (gloo's rendezvous banner does) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.in_process:
        return in_process_main(args, json_fd)
    import torch
    import torch.distributed as dist

    from mlvectordb_amd import synth
    from mlvectordb_amd.engine import HipScanEngine
    from mlvectordb_amd.sharded import merge_topk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus  # launch_plan() refused anything else
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    host_group = None
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            host_group = dist.new_group(backend="gloo")  # the per-shard candidates are merged on the host
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    # n_gpus of the JSON line = the ranks that really joined (one collective over the data-path group), never the flag
    ranks_ran, rccl_ranks, rank_devices = 1, 0, [local_rank]
    if world > 1:
        one = torch.ones(1, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_ran = int(round(one.item()))
        rccl_ranks = dist.get_world_size() if args.backend == "nccl" else 0
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, local_rank, group=host_group)
        if ranks_ran != args.gpus:
            log(f"{ranks_ran} ranks joined but --gpus {args.gpus}")
            sys.exit(2)

    n_local, d, k = args.rows_per_gpu, args.dim, args.k
    batch = args.batch or (256 if world == 1 else 1024)
    row0 = rank * n_local
    # generator threads = 1M-row chunks (3 GB each) in flight per rank: bounded so that 8 ranks on one host stay under
    # ~75 GB of generation buffers in total (all 10 chunks of a shard at once would be 30 GB per rank)
    threads = args.gen_threads or max(2, min(16, (os.cpu_count() or 8) // max(1, min(world, 8)), max(2, 24 // world)))

    # ---- shard: seeded N(0,1) rows, generated on the host in 1M-row chunks, appended to HBM
    t0 = time.perf_counter()
    # The shard is filled through the Protocol-level objects (QueryProcessor -> ArrayStorage + Index.add_arrays: ids
    # minted, rows appended to HBM, no host copy kept); the engine-level measurements below drive the namespace's
    # mlvdb_index directly, the protocol-level one (protocol_qps) goes back through QueryProcessor.
    from mlvectordb_amd import ArrayStorage, Index, QueryProcessor

    index = Index(space=args.space, device=local_rank, strategy=args.strategy, capacity_hint=n_local)
    qp = QueryProcessor(ArrayStorage(), index)
    # BASELINE configs[3] (squared-l2 kNN + range query on the same rows) is measured beside the headline at N=1:
    # its index is filled from the same generated pieces (one more upload per piece, no second generation)
    want_cfg4 = world == 1 and not args.no_extras and args.space == "cosine" and args.tombstones == 0.0
    eng_l2 = HipScanEngine(d, "l2", device=local_rank, capacity_hint=n_local) if want_cfg4 else None
    sample_rows = None
    for off, rows in synth.iter_corpus(row0, n_local, d, threads=threads):
        qp.upsert_arrays(rows, "bench", keep_host_copy=False)
        if eng_l2 is not None:
            eng_l2.append(rows)
        if rank == 0 and off == 0:
            sample_rows = rows.copy()  # first 250k rows, reused for the parity gate
    eng = index._ns["bench"].engine
    if args.tombstones > 0.0:
        dead = np.nonzero(np.random.default_rng([99, rank]).random(n_local) < args.tombstones)[0].astype(np.int64)
        eng.tombstone(dead)
        if sample_rows is not None:
            sample_dead = np.zeros(sample_rows.shape[0], dtype=bool)
            sample_dead[dead[dead < sample_rows.shape[0]]] = True
        log(f"tombstoned {dead.size} of {n_local} rows")
    load_s = time.perf_counter() - t0
    log(f"shard rows [{row0}, {row0 + n_local}) resident after {load_s:.1f} s ({threads} generator threads)")

    # Query batches: batch 0 is SURVEY 8d's default_rng(4321); every timed loop rotates through all of them, so thresholds,
    # candidate counts and branch behaviour are not those of one draw.  Each batch passes the parity gate once, untimed.
    nqb = max(1, args.query_batches)
    q_hosts = [synth.queries(batch, d, i) for i in range(nqb)]
    q_devs = [torch.from_numpy(q).to(dev) for q in q_hosts]
    q_host, q_dev = q_hosts[0], q_devs[0]
    lab = torch.empty((batch, k), dtype=torch.int64, device=dev)
    dst = torch.empty((batch, k), dtype=torch.float32, device=dev)
    cnt = torch.empty(batch, dtype=torch.int32, device=dev)
    d64 = torch.empty((batch, k), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    wave_no = [0]

    def local_wave(qi=None):
        if qi is None:  # the timed loops: the next batch in rotation
            qi = wave_no[0] % nqb
            wave_no[0] += 1
        eng.search_device(q_devs[qi].data_ptr(), batch, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(),
                          d64.data_ptr(), stream)

    def step(qi=None):
        local_wave(qi)
        if world == 1:
            return None
        torch.cuda.current_stream().synchronize()
        l_host = lab.cpu()
        l_host = torch.where(l_host >= 0, l_host + row0, l_host)
        d_host = d64.cpu()
        if rank == 0:
            gl = [torch.empty_like(l_host) for _ in range(world)]
            gd = [torch.empty_like(d_host) for _ in range(world)]
        else:
            gl = gd = None
        dist.gather(l_host, gl, dst=0, group=host_group)
        dist.gather(d_host, gd, dst=0, group=host_group)
        if rank == 0:
            return merge_topk([t.numpy() for t in gl], [t.numpy() for t in gd], k)
        return None

    # N > 1, throughput mode: wave i+1 is enqueued before wave i's candidates are copied out (second stream),
    # gathered and merged, so the host-side exchange overlaps the next scan instead of idling the GPU
    if world > 1:
        copy_stream = torch.cuda.Stream(device=dev)
        slots = [{"lab": torch.empty_like(lab), "dst": torch.empty_like(dst), "cnt": torch.empty_like(cnt),
                  "d64": torch.empty_like(d64), "ev": torch.cuda.Event(),
                  "l_host": torch.empty((batch, k), dtype=torch.int64).pin_memory(),
                  "d_host": torch.empty((batch, k), dtype=torch.float64).pin_memory()} for _ in range(2)]

        def enqueue(i):
            b = slots[i % 2]
            eng.search_device(q_devs[i % nqb].data_ptr(), batch, k, b["lab"].data_ptr(), b["dst"].data_ptr(), b["cnt"].data_ptr(),
                              b["d64"].data_ptr(), stream)
            b["ev"].record()

        def finish(i):
            b = slots[i % 2]
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(b["ev"])
                b["l_host"].copy_(b["lab"], non_blocking=True)
                b["d_host"].copy_(b["d64"], non_blocking=True)
            copy_stream.synchronize()
            l_host = torch.where(b["l_host"] >= 0, b["l_host"] + row0, b["l_host"])
            if rank == 0:
                gl = [torch.empty_like(l_host) for _ in range(world)]
                gd = [torch.empty_like(b["d_host"]) for _ in range(world)]
            else:
                gl = gd = None
            dist.gather(l_host, gl, dst=0, group=host_group)
            dist.gather(b["d_host"], gd, dst=0, group=host_group)
            if rank == 0:
                return merge_topk([t.numpy() for t in gl], [t.numpy() for t in gd], k)
            return None

    # ---- parity gate (untimed): ids must equal the exact fp64 GPU scan, and the oracle on a sample
    verify = {"query_batches": nqb, "filter_equals_exact_scan_ids": True, "filter_equals_exact_scan_max_abs_err": 0.0}
    for qi in range(nqb - 1, -1, -1):  # (batch 0 last: its answer stays in fast_ids for the protocol-level check below)
        local_wave(qi)
        torch.cuda.synchronize()
        stats0 = eng.last_stats()
        fast_ids = lab.cpu().numpy().copy()
        fast_dist = dst.cpu().numpy().copy()
        eng.set_strategy("exact")
        ex_l, ex_d, _ = eng.search(q_hosts[qi], k)  # every query of the wave against the exact fp64 scan of the whole shard
        eng.set_strategy(args.strategy)
        verify["filter_equals_exact_scan_ids"] &= bool(np.array_equal(fast_ids, ex_l))
        verify["filter_equals_exact_scan_max_abs_err"] = max(verify["filter_equals_exact_scan_max_abs_err"],
                                                             float(np.abs(fast_dist - ex_d).max()))
    verify["queries_compared"] = int(batch) * nqb
    if rank == 0 and sample_rows is not None:
        from oracle import exact_scan

        small = HipScanEngine(d, args.space, device=local_rank, strategy="filter" if d % 64 == 0 else "exact")
        small.append(sample_rows)
        if args.tombstones > 0.0:
            small.tombstone(np.nonzero(sample_dead)[0].astype(np.int64))
        sl, sd, _ = small.search(q_host[:64], k)
        small.close()
        ol, od, _ = exact_scan.knn(q_host[:64], sample_rows, k, args.space,
                                   deleted=sample_dead if args.tombstones > 0.0 else None)
        verify["oracle_sample"] = f"{sample_rows.shape[0]} rows x 64 queries"
        verify["oracle_ids_equal"] = bool(np.array_equal(sl, ol))
        verify["oracle_max_abs_err"] = float(np.abs(sd - od).max())
    if world > 1:
        # sharded path: the host-merged answer of the filter strategy must equal the merged exact scans
        merged_fast = step(0)
        eng.set_strategy("exact")
        merged_exact = step(0)
        eng.set_strategy(args.strategy)
        if rank == 0:
            verify["sharded_merge_equals_exact_ids"] = bool(np.array_equal(merged_fast[0], merged_exact[0]))
            verify["sharded_merge_max_abs_err"] = float(np.abs(merged_fast[1] - merged_exact[1]).max())
    gate_red = (not verify["filter_equals_exact_scan_ids"] or verify["filter_equals_exact_scan_max_abs_err"] > 1e-5
                or verify.get("oracle_ids_equal") is False or verify.get("oracle_max_abs_err", 0.0) > 1e-5
                or verify.get("sharded_merge_equals_exact_ids") is False)
    if world > 1:  # one red rank fails the whole job: every rank learns it and exits non-zero after this exchange
        flag = torch.tensor([1.0 if gate_red else 0.0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        gate_red = bool(flag.item() > 0)
    if gate_red:
        log(f"PARITY GATE FAILED: {verify}")
        if rank == 0:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps({"metric": "queries/sec, exact cosine kNN k=10 over a row-sharded Nx768 fp32 corpus (10M rows per GPU)",
                                           "value": None, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
                                           "warmup": args.warmup, "error": "parity gate failed: results differ from the exact scan / oracle",
                                           "parity_gate": verify}) + "\n").encode())
        if world > 1:
            dist.destroy_process_group()
        sys.exit(1)

    # ---- timed region
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.last_stats()  # start a fresh statistics window
    eng.set_profiling(os.environ.get("MLVDB_BENCH_NO_EVENTS") != "1")  # HIP events on the launch stream around every scan-kernel launch
    #                                                                     (=1: tuning only -- what the events themselves cost; no roofline then)
    # N = 1 times the K steps twice, each region bracketed by barrier + synchronize: once with the waves enqueued back to back
    # (the host never waits inside the region) and once with the host synchronising after every wave, as a caller that consumes
    # each wave's answer does.  The synchronised loop is the FASTER one on this hardware -- the ~30 us the GPU idles between waves
    # let its clock recover, the MFMA-heavy scan then runs 5-6 % shorter -- and it is the line's `ms_per_step` /
    # `value` (--wave-mode back_to_back swaps the two); the other region's figure stays beside it, scan-kernel events included.
    host_enqueue_s = None
    other = None
    if world == 1:
        def region(synchronised):
            eng.last_stats()  # fresh statistics window
            barrier()
            t0 = time.perf_counter()
            enq = None
            for _ in range(args.steps):
                step()
                if synchronised:
                    torch.cuda.current_stream().synchronize()
            if not synchronised:
                enq = time.perf_counter() - t0  # how long the host took to enqueue the K waves (it must stay ahead of the GPU)
            barrier()
            return time.perf_counter() - t0, enq, eng.last_stats()

        primary_sync = args.wave_mode == "synchronised"
        el_o, enq_o, st_o = region(not primary_sync)
        host_enqueue_s = enq_o
        other = {"wave_mode": "back_to_back" if primary_sync else "synchronised", "ms_per_step": round(el_o / args.steps * 1e3, 3),
                 "value": round(batch * args.steps / el_o, 1),
                 "scan_avg_launch_ms": round(st_o["scan_ms"] / max(1, st_o["scan_launches"]), 4)}
    barrier()
    t_start = time.perf_counter()
    if world == 1:
        for _ in range(args.steps):
            step()
            if primary_sync:
                torch.cuda.current_stream().synchronize()  # the caller has this wave's answer before it sends the next
        if not primary_sync:
            host_enqueue_s = time.perf_counter() - t_start
    else:
        last = None
        enqueue(0)
        for i in range(1, args.steps):
            enqueue(i)
            last = finish(i - 1)
        last = finish(args.steps - 1)  # every wave is copied out, gathered and merged inside the timed region
    barrier()
    elapsed = time.perf_counter() - t_start
    st = eng.last_stats()  # accumulated over the K steps (HIP events of every scan launch)
    scan_ms, scan_launches, rows_scanned = st["scan_ms"], st["scan_launches"], st["rows_scanned"]
    rescored, fallbacks = st["candidates_rescored"], st["fallback_queries"]
    # ---- sustained rate (N = 1): the K timed steps above are a 40 ms burst on a power-limited kernel; a server gets the
    # rate the chip holds for seconds.  Both wave modes again, >= --sustained-seconds and >= 1000 waves each, same rotation of
    # query batches, scan-kernel events on (what the first seconds of load cost in clock shows in scan_avg_launch_ms).
    sustained = None
    if world == 1 and args.sustained_seconds > 0:
        n_sus = max(1000, int(args.sustained_seconds * 1.1 / max(elapsed / args.steps, 1e-4)) + 1)
        sustained = {"waves_per_mode": n_sus, "query_batches": nqb,
                     "effective_clock_ghz": None,
                     "effective_clock_note": "GRBM_GUI_ACTIVE is a rocprofv3 PMC counter: not readable in-process; "
                                             "profiles/r04/ holds the counter passes of this command"}
        for mode in ("synchronised", "back_to_back"):
            eng.last_stats()
            barrier()
            ts = time.perf_counter()
            for _ in range(n_sus):
                step()
                if mode == "synchronised":
                    torch.cuda.current_stream().synchronize()
            barrier()
            el = time.perf_counter() - ts
            ss = eng.last_stats()
            sustained[mode] = {"ms_per_step": round(el / n_sus * 1e3, 3), "value": round(batch * n_sus / el, 1),
                               "seconds": round(el, 2),
                               "scan_avg_launch_ms": round(ss["scan_ms"] / max(1, ss["scan_launches"]), 4)}
    eng.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # latency of a single wave (host-synchronised), measured outside the throughput region.  Device-resident: queries
    # and outputs already in HBM.  host_io (SURVEY 8d's latency definition): the host-pointer entry, i.e. H2D of the
    # queries and D2H of labels / distances / counts included.
    per_step, per_step_io = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        step()
        torch.cuda.current_stream().synchronize()
        per_step.append(time.perf_counter() - ts)
    if world == 1:
        for _ in range(35):
            ts = time.perf_counter()
            eng.search(q_host, k)
            per_step_io.append(time.perf_counter() - ts)
        per_step_io = per_step_io[5:]

    ms_per_step = elapsed / args.steps * 1e3
    shard_queries_per_s = world * batch * args.steps / elapsed  # one unit = one query against one 10M-row shard
    # algorithmic bytes of one wave over one shard: corpus once + queries + results + row norms
    passes = (batch + 255) // 256
    alg_bytes_wave = n_local * d * 4 + batch * d * 4 + batch * k * 12 + n_local * 4
    alg_bytes_scan = float(rows_scanned) * (d * 4 + 4) + args.steps * passes * (256 * d * 2)
    roofline = None
    if scan_ms > 0:
        scan_s = scan_ms * 1e-3
        filt = stats0["strategy_used"] == 2
        i8 = filt and stats0.get("bound_dtype") == 2  # the int8 shadow computed the bounds (dim % 256 == 0)
        # Two physical floors of the scan kernel as built: the matrix-core time of 2*rows*d*256 multiply-adds
        # at the dense peak of the operand type, and the HBM time of the bytes the kernel has to move -- the shadow it
        # streams (1 B / 2 B per element) + 8 B / 4 B of row constants per row + the query image.  The roof that binds
        # is the slower floor; `frac` = that floor / measured time (<= 1 by construction).  SURVEY 8d's accounting (the
        # fp32 corpus bytes, which this kernel does not read) stays beside it as alg_hbm_*.
        elem = 1 if i8 else (2 if filt else 4)
        rowc = 8 if i8 else 4
        min_bytes = float(rows_scanned) * (d * elem + rowc) + args.steps * passes * 3 * (256 * d * elem)
        mfma_peak = (5.0e15 if i8 else 2.5e15) if filt else 157.3e12
        flops = 2.0 * float(rows_scanned) * d * 256 if filt else 2.0 * float(rows_scanned) * d * min(batch, 8)
        t_mfma, t_hbm = flops / mfma_peak, min_bytes / (HBM_PEAK_GBS * 1e9)
        bound = "mfma" if (filt and t_mfma >= t_hbm) else "hbm"
        # measured HBM traffic: rocprofv3 PMC passes of this same command (separate runs; counters cannot be read from
        # inside the process).  The file records the hash of the kernel sources it was measured on: stale = ignored.
        traffic, tsrc = None, None
        tfile = ROOT / "profiles" / "r04" / ("pmc_traffic_i8.json" if i8 else "pmc_traffic.json")
        if filt and n_local == 10_000_000 and d == 768 and tfile.exists():
            tj = json.loads(tfile.read_text())
            if tj.get("kernel_source_sha16") == kernel_source_sha16():
                traffic = tj["traffic_bytes_per_launch_avg"]
                tsrc = f"profiles/r04/{tfile.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same kernel sources)"
            else:
                tsrc = f"profiles/r04/{tfile.name} is stale (kernel sources changed since it was measured): not reported"
        if bound == "mfma":
            achieved, peak, unit = flops / scan_s / 1e12, mfma_peak / 1e12, "TFLOP/s"
        else:
            achieved, peak, unit = min_bytes / scan_s / 1e9, HBM_PEAK_GBS, "GB/s"
        roofline = {"bound": bound, "achieved": round(achieved, 1), "peak": peak, "unit": unit,
                    "frac": round(achieved / peak, 4),
                    # what `frac` counts: the bytes this kernel streams -- the int8 / bf16 shadow + row constants
                    # + query image -- not SURVEY 8(d)'s fp32 corpus bytes, whose fraction is `alg_frac` beside it
                    "frac_basis": (("int8" if i8 else "bf16") + " shadow bytes") if filt else "fp32 corpus bytes (= SURVEY 8d)",
                    "alg_frac": round(alg_bytes_scan / scan_s / 1e9 / HBM_PEAK_GBS, 4),
                    "alg_frac_basis": "SURVEY 8(d): rows x (d x 4 + 4) B of the fp32 corpus per scan launch (> 1: a narrower shadow is streamed)",
                    "traffic": traffic, "traffic_source": tsrc,
                    "kernel": "filter_scan_asm_kernel" if filt else "exact_scan_kernel",
                    "operand_dtype": ("i8" if i8 else "bf16") if filt else "f32->f64",
                    "avg_launch_ms": round(scan_ms / max(1, scan_launches), 4), "launches": scan_launches,
                    "floors_ms_per_launch": {"mfma": round(t_mfma / max(1, scan_launches) * 1e3, 4),
                                             "hbm_min_bytes": round(t_hbm / max(1, scan_launches) * 1e3, 4)},
                    "min_bytes_per_launch": round(min_bytes / max(1, scan_launches)),
                    "mfma_frac": round(flops / scan_s / mfma_peak, 4) if filt else None,
                    "hbm_min_bytes_frac": round(min_bytes / scan_s / 1e9 / HBM_PEAK_GBS, 4),
                    "hbm_actual_frac": round(traffic * scan_launches / scan_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                    # SURVEY 8d accounting (fp32 corpus bytes per wave; > 1 means: the kernel streams a narrower shadow)
                    "alg_bytes_per_launch": round(alg_bytes_scan / max(1, scan_launches)),
                    "alg_hbm_frac": round(alg_bytes_scan / scan_s / 1e9 / HBM_PEAK_GBS, 4),
                    "alg_hbm_whole_wave_frac": round(alg_bytes_wave * passes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    if world == 1 else None}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        "metric": "queries/sec, exact cosine kNN k=10 over a row-sharded Nx768 fp32 corpus (10M rows per GPU)",
        "value": round(shard_queries_per_s, 1),
        "unit": "queries/s (each query scanned against one 10M-row shard; whole-corpus QPS = value / n_gpus)",
        "n_gpus": ranks_ran, "rccl_ranks": rccl_ranks, "rank_devices": rank_devices,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": (("i8" if stats0.get("bound_dtype") == 2 else "bf16") + " (MFMA bounds) + f64 (exact rescoring of the f32 rows)")
                 if stats0["strategy_used"] == 2 else "f64 (exact scan of the f32 rows)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{2 if world == 1 else 4}]: {world * n_local} x {d} fp32 N(0,1) "
                               f"rows ({n_local}/GPU), {args.space} kNN k={k}, batch={batch}, exact ids "
                               f"({'int8' if stats0.get('bound_dtype') == 2 else 'bf16'}-MFMA bound filter + fp64 rescoring)",
                   "rows_per_gpu": n_local, "dim": d, "k": k, "batch": batch, "space": args.space,
                   "tombstoned_fraction": args.tombstones,
                   "strategy": {1: "exact", 2: "filter"}.get(stats0["strategy_used"], "?"),
                   "sharding": f"row-wise over {world} ranks, host merge of per-shard top-k"},
        "whole_corpus_qps": round(batch * args.steps / elapsed, 1),
        "host_enqueue_ms_per_wave": round(host_enqueue_s / args.steps * 1e3, 3) if host_enqueue_s is not None else None,
        "wave_mode": (args.wave_mode if world == 1 else "pipelined (wave i+1 enqueued before wave i is gathered and merged)"),
        "other_wave_mode": other,
        # stable keys: the same two definitions every round, whichever of them `value` is.  `value` was the
        # back-to-back figure until round 2 and is the synchronised one since the end of round 3.
        "value_back_to_back": (round(shard_queries_per_s, 1) if args.wave_mode == "back_to_back" else other["value"]) if world == 1 else None,
        "value_synchronised": (round(shard_queries_per_s, 1) if args.wave_mode == "synchronised" else other["value"]) if world == 1 else None,
        "sustained": sustained,
        "p50_ms_per_wave": round(float(np.median(per_step)) * 1e3, 3),
        "p50_ms_per_wave_host_io": round(float(np.median(per_step_io)) * 1e3, 3) if per_step_io else None,
        "roofline": roofline,
        "candidates_rescored_per_query": round(rescored / max(1, args.steps * batch), 1),
        "fallback_queries": int(fallbacks),
        "parity_gate": verify,
        "load_s": round(load_s, 1),
    }

    # ---- CPU exact-scan baseline (rank 0, N=1): this repo's NumPy/OpenBLAS scan on a bounded sample
    if world == 1 and not args.no_cpu_baseline:
        from oracle.blas_scan import BlasScanEngine

        # the CPU scan is timed THROUGH the same surface as the GPU path -- QueryProcessor.find_similar_many ->
        # Index.search_many -> engine -- with this repo's NumPy / OpenBLAS scan as the engine (UUID tables, score
        # post-processing, enrichment of every hit included); the bare sgemm + argpartition figure stays beside it
        s_rows = min(args.cpu_sample_rows, n_local)
        sample = synth.corpus_rows(0, s_rows, d)
        cpu_qp = QueryProcessor(ArrayStorage(), Index(space=args.space, engine_factory=BlasScanEngine))
        cpu_qp.upsert_arrays(sample, "cpu")
        cpu = cpu_qp._index._ns["cpu"].engine._built()
        cpu_qp.find_similar_many(q_host[:8], top_k=k, namespace="cpu", metric=args.space)
        reps, t_cpu, t_raw = 0, 0.0, 0.0
        while t_cpu < 10.0 and reps < 8:
            tc = time.perf_counter()
            cpu_hits = cpu_qp.find_similar_many(q_host, top_k=k, namespace="cpu", metric=args.space)
            t_cpu += time.perf_counter() - tc
            tc = time.perf_counter()
            cl, _ = cpu.search(q_host, k)
            t_raw += time.perf_counter() - tc
            reps += 1
        assert len(cpu_hits) == batch and len(cpu_hits[0]) == k
        per_wave_sample, per_wave_raw = t_cpu / reps, t_raw / reps
        try:
            from threadpoolctl import threadpool_info

            blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        except Exception:
            blas_threads = os.cpu_count() or 1
        out["cpu_baseline"] = {
            "value": round(batch / (per_wave_sample * n_local / s_rows), 2),
            "unit": "queries/s (extrapolated linearly in rows to the 10M-row corpus)",
            "cores": int(blas_threads), "kind": "port",
            "sample": f"{reps} waves of {batch} queries over the first {s_rows} rows through QueryProcessor.find_similar_many -> "
                      f"Index.search_many -> fp32 sgemm + argpartition engine (oracle/blas_scan.py), {per_wave_sample * 1e3:.0f} ms "
                      f"per wave on the sample",
            "raw_engine_value": round(batch / (per_wave_raw * n_local / s_rows), 2),
            "raw_engine_note": f"the engine's search alone (no ids, no enrichment): {per_wave_raw * 1e3:.0f} ms per wave on the sample",
        }
        cpu_qp._index.close()
        del cpu, cpu_qp
        # ---- side measurement, BASELINE configs[1]: 1M x 768, batch 1 (latency path).  "auto" = what the library
        # picks (the narrow bf16 bound filter + fp64 rescoring); "exact" = the fp32-row / fp64 scan, forced
        if not args.no_extras and s_rows >= 1_000_000:
            e2 = HipScanEngine(d, args.space, device=local_rank, capacity_hint=1_000_000)
            e2.append(sample[:1_000_000])
            q1 = q_dev[:1].contiguous()
            side, ids = {}, {}
            for strat in ("auto", "exact"):
                e2.set_strategy(strat)
                lat, scan2 = [], 0.0
                e2.set_profiling(False)  # the latency is measured WITHOUT the scan-kernel events (a dozen event records are
                for i in range(60):      # ~10 % of a 0.23 ms call); the kernels' own time comes from a second loop with them
                    ts = time.perf_counter()
                    e2.search_device(q1.data_ptr(), 1, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, stream)
                    torch.cuda.current_stream().synchronize()
                    if i >= 10:
                        lat.append(time.perf_counter() - ts)
                e2.set_profiling(True)
                e2.last_stats()
                for i in range(50):
                    e2.search_device(q1.data_ptr(), 1, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, stream)
                    torch.cuda.current_stream().synchronize()
                    scan2 += e2.last_stats()["scan_ms"]
                ids[strat] = lab[:1].cpu().numpy().copy()
                lat_io = []  # SURVEY 8d's latency: host-pointer entry, H2D of the query and D2H of the result included
                for i in range(60):
                    ts = time.perf_counter()
                    e2.search(q_host[:1], k)
                    if i >= 10:
                        lat_io.append(time.perf_counter() - ts)
                e2.last_stats()
                p50 = float(np.median(lat))
                alg = 1_000_000 * (d * 4 + 4)
                side[strat] = {
                    "strategy_used": {1: "exact", 2: "filter"}.get(e2.last_stats()["strategy_used"], "?"),
                    "qps": round(1.0 / p50, 1), "p50_ms": round(p50 * 1e3, 4),
                    "p50_ms_host_io": round(float(np.median(lat_io)) * 1e3, 4),
                    "scan_kernels_ms": round(scan2 / len(lat), 4),
                    "hbm_frac_p50_alg_bytes": round(alg / p50 / 1e9 / HBM_PEAK_GBS, 4),
                    "hbm_frac_scan_kernels_alg_bytes": round(alg / (scan2 / len(lat) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                }
            e2.close()
            side["ids_equal"] = bool((ids["auto"] == ids["exact"]).all())
            out["config2_1Mx768_batch1"] = side
        if not args.no_extras:
            out["config1_10kx128_find_similar"] = config1_side(local_rank)
    # ---- top_k = 100 (top_k 65..1024 stays on the filter path): one wave of the same batch, gated against
    # the exact scan for every query, timed device-resident like p50_ms_per_wave
    if world == 1 and not args.no_extras and args.tombstones == 0.0:
        k2 = 100
        lab2 = torch.empty((batch, k2), dtype=torch.int64, device=dev)
        dst2 = torch.empty((batch, k2), dtype=torch.float32, device=dev)
        eng.last_stats()
        t100 = []
        for i in range(8):
            torch.cuda.synchronize()
            ts = time.perf_counter()
            eng.search_device(q_devs[i % nqb].data_ptr(), batch, k2, lab2.data_ptr(), dst2.data_ptr(), cnt.data_ptr(), 0, stream)
            torch.cuda.current_stream().synchronize()
            t100.append(time.perf_counter() - ts)
        st100 = eng.last_stats()
        eng.search_device(q_dev.data_ptr(), batch, k2, lab2.data_ptr(), dst2.data_ptr(), cnt.data_ptr(), 0, stream)
        torch.cuda.current_stream().synchronize()
        l100, d100 = lab2.cpu().numpy(), dst2.cpu().numpy()
        eng.set_strategy("exact")
        xl, xd, _ = eng.search(q_host, k2)
        eng.set_strategy(args.strategy)
        ms100 = float(np.median(t100[2:])) * 1e3
        out["topk100_ms_per_wave"] = round(ms100, 3)
        out["topk100"] = {"over_k10_p50": round(ms100 / out["p50_ms_per_wave"], 3),
                          "strategy": {1: "exact (paged)", 2: "filter"}.get(st100["strategy_used"], "?"),
                          "candidates_rescored_per_query": round(st100["candidates_rescored"] / (8 * batch), 1),
                          "fallback_queries": int(st100["fallback_queries"]),
                          "ids_equal_exact_scan_all_queries": bool(np.array_equal(l100, xl)),
                          "max_abs_err": float(np.abs(d100 - xd).max())}
        if not out["topk100"]["ids_equal_exact_scan_all_queries"] or out["topk100"]["max_abs_err"] > 1e-5:
            log(f"PARITY GATE FAILED (top_k = 100): {out['topk100']}")
            out["value"] = None
            out["error"] = "top_k = 100 results differ from the exact scan"
            os.write(json_fd, (json.dumps(out) + "\n").encode())
            sys.exit(1)
    # ---- the same workload through the Protocol surface: QueryProcessor.find_similar_many over Index.search_many
    # (reference query_processor.py:26-49 batched).  Every hit comes back as {"id": UUID, "values": float32[d],
    # "metadata", "score"}; values are gathered from the index's rows in HBM.  "stream" = find_similar_stream: the scan
    # of wave i+1 overlaps the enrichment of wave i (one worker thread inside the GIL-free ctypes call).
    if world == 1 and args.tombstones == 0.0 and not args.no_extras:
        # A serving process with a 10M-row heap freezes what it has loaded (gc.freeze: the objects alive now are never walked
        # again by the cyclic collector); without it a full collection that lands inside the 20-wave stream below costs
        # tens of milliseconds and the figure measures the collector's phase, not the path (seen: 2.2 vs 8 ms per wave)
        import gc

        gc.collect()
        gc.freeze()
        wave_hits = qp.find_similar_many(q_host, top_k=k, namespace="bench", metric=args.space)  # warm-up + check
        table = index._ns["bench"].ids
        proto_ok = all([h["id"] for h in wave_hits[i]] == table.uuids_at(fast_ids[i]).tolist() for i in range(batch))
        proto_ok = proto_ok and all(np.array_equal(h["values"], eng.get_rows(int(l), 1)[0])
                                    for i in (0, batch - 1) for h, l in zip(wave_hits[i], fast_ids[i]))
        lat = []
        for _ in range(8):
            ts = time.perf_counter()
            qp.find_similar_many(q_host, top_k=k, namespace="bench", metric=args.space)
            lat.append(time.perf_counter() - ts)
        n_hits, ts = 0, time.perf_counter()
        for hits in qp.find_similar_stream((q_host for _ in range(args.steps)), top_k=k, namespace="bench", metric=args.space):
            n_hits += sum(len(h) for h in hits)
        t_stream = time.perf_counter() - ts
        out["protocol_qps"] = round(batch * args.steps / t_stream, 1)
        out["protocol"] = {"path": "QueryProcessor.find_similar_stream -> Index.search_many -> mlvdb_search_batch_ex; "
                                   "ArrayStorage (ids + metadata on the host, values gathered from HBM)",
                           "ms_per_wave_stream": round(t_stream / args.steps * 1e3, 3),
                           "p50_ms_find_similar_many": round(float(np.median(lat)) * 1e3, 3),
                           "protocol_over_engine_qps": round(batch * args.steps / t_stream / shard_queries_per_s, 3),
                           "hits_materialised_per_wave": n_hits // args.steps,
                           "ids_and_values_equal_engine_level_result": bool(proto_ok)}
        if not proto_ok:
            log("PARITY GATE FAILED (protocol path): ids / values differ from the engine-level result")
            out["value"] = None
            out["error"] = "protocol-level results differ from the engine-level results"
            os.write(json_fd, (json.dumps(out) + "\n").encode())
            sys.exit(1)
    if eng_l2 is not None:
        out["config4_10Mx768_l2_range"] = config4_side(eng_l2, q_host, k)
        eng_l2.close()
        c4 = out["config4_10Mx768_l2_range"]["parity"]
        if not (c4["knn_ids_equal_exact_scan_all_queries"] and c4["range_hits_equal_exact_range_scan_all_queries"]):
            log(f"PARITY GATE FAILED (configs[3]): {c4}")
            out["value"] = None
            out["error"] = "configs[3] parity failed"
            os.write(json_fd, (json.dumps(out) + "\n").encode())
            sys.exit(1)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
