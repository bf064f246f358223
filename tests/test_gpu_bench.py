"""bench.py end to end on the GPU box (small shards so it takes seconds): the bare `--gpus N` launch that starts its own ranks
(VERDICT r2 item 1), the sharded parity gate, and the one-GPU line's contract keys."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, timeout=timeout)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr.decode()[-2000:]
    return json.loads(lines[0])


def test_bare_gpus_2_starts_two_ranks_and_passes_the_sharded_gate():
    """No torchrun in the command: bench.py spawns `python -m torch.distributed.run --nproc-per-node 2` itself (two ranks on this
    box's one GPU: --single-device, gloo), every rank scans its own 300k-row shard for the 1024-query wave (four passes), rank 0
    merges on the host; the merged ids must equal the merged exact scans."""
    d = _run(["--gpus", "2", "--single-device", "--backend", "gloo", "--rows-per-gpu", "300000", "--steps", "3", "--warmup", "1",
              "--no-cpu-baseline", "--no-extras"])
    assert d["n_gpus"] == 2 and d["rank_devices"] == [0, 0] and d["scaling"] == "weak"
    gate = d["parity_gate"]
    assert gate["sharded_merge_equals_exact_ids"] is True and gate["filter_equals_exact_scan_ids"] is True
    assert gate["queries_compared"] == 1024 * gate["query_batches"] and gate["query_batches"] == 8 and d["config"]["batch"] == 1024
    assert d["value"] > 0 and abs(d["value"] - 2 * 1024 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_one_gpu_line_has_the_contract_keys_on_a_small_shard():
    d = _run(["--rows-per-gpu", "400000", "--steps", "3", "--warmup", "1", "--no-extras", "--cpu-sample-rows", "100000"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] <= 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and "find_similar_many" in d["cpu_baseline"]["sample"]
    assert d["parity_gate"]["filter_equals_exact_scan_ids"] and d["parity_gate"]["oracle_ids_equal"]
    # the K steps are timed twice at N = 1: synchronised wave by wave (the line's figure) and enqueued back to back (beside it)
    other = d["other_wave_mode"]
    assert d["wave_mode"] == "synchronised" and other["wave_mode"] == "back_to_back" and other["ms_per_step"] > 0
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    # round 4: stable keys for both definitions, the sustained rate (>= 1000 waves per mode, rotating query batches), and
    # SURVEY 8(d)'s fraction under its own name beside the shadow-bytes one
    assert d["value_synchronised"] == d["value"] and d["value_back_to_back"] == other["value"]
    sus = d["sustained"]
    assert sus["waves_per_mode"] >= 1000 and sus["query_batches"] == d["parity_gate"]["query_batches"] == 8
    assert sus["synchronised"]["ms_per_step"] > 0 and sus["back_to_back"]["scan_avg_launch_ms"] > 0
    r = d["roofline"]
    assert r["frac_basis"].endswith("shadow bytes") and r["alg_frac"] == r["alg_hbm_frac"] and r["alg_frac"] > r["frac"]
    # (--wave-mode back_to_back swaps which region is `value`: the same code path with the flag inverted; both regions ran here)
    assert d["host_enqueue_ms_per_wave"] > 0
