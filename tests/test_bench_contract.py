"""The bench line the driver parses (CPU): the committed final-build record under profiles/ must carry every key of the
contract with consistent values, and the PMC traffic file must belong to the kernel sources in the tree -- a stale file
would be a wrong number in the next bench line (bench.py ignores it, and says so, when the hash differs)."""
import json
from pathlib import Path

import bench

ROOT = Path(__file__).resolve().parents[1]


def _line(name):
    text = (ROOT / "profiles" / "r04" / name).read_text().strip().splitlines()[-1]
    return json.loads(text)


def test_final_bench_record_has_the_contract_keys_and_consistent_values():
    d = _line("bench_n1_final.json")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = queries per second of K timed waves of `batch` queries
    batch = d["config"]["batch"] if "batch" in d["config"] else 256
    assert abs(d["value"] - batch / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0.0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= r["min_bytes_per_launch"] * 0.99  # PMC bytes >= what the kernel must read
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    gate = d["parity_gate"]
    assert gate["filter_equals_exact_scan_ids"] is True and gate["oracle_ids_equal"] is True
    assert gate["query_batches"] == 8 and gate["queries_compared"] == 8 * 256  # every rotating batch is gated once
    # round 4: the sustained rate (>= 1000 waves and >= 2 s per wave mode), both headline definitions under stable keys,
    # SURVEY 8(d)'s fraction under its own name, top_k = 100 on the filter path within 2 x the k = 10 wave
    sus = d["sustained"]
    for mode in ("synchronised", "back_to_back"):
        assert sus["waves_per_mode"] >= 1000 and sus[mode]["seconds"] >= 2.0 and sus[mode]["ms_per_step"] > 0
    assert d["value_synchronised"] == d["value"] and d["value_back_to_back"] == d["other_wave_mode"]["value"]
    assert r["frac_basis"] == "int8 shadow bytes" and r["alg_frac"] > 1.0
    assert d["topk100"]["strategy"] == "filter" and d["topk100"]["ids_equal_exact_scan_all_queries"] is True
    assert d["topk100_ms_per_wave"] <= 2.0 * d["p50_ms_per_wave"]
    c4 = d["config4_10Mx768_l2_range"]["parity"]
    assert c4["knn_ids_equal_exact_scan_all_queries"] and c4["range_hits_equal_exact_range_scan_all_queries"]


def test_pmc_traffic_file_matches_the_kernel_sources_in_the_tree():
    import pytest

    pmc = json.loads((ROOT / "profiles" / "r04" / "pmc_traffic_i8.json").read_text())
    if pmc["kernel_source_sha16"] != bench.kernel_source_sha16():
        # not a failure of the code under test: bench.py then reports "traffic": null.  Shown as xfail so that it is seen.
        pytest.xfail("the scan kernel sources changed after profiles/r04/pmc_traffic_i8.json was measured: re-run the two "
                     "rocprofv3 --pmc passes and tools/pmc_traffic.py")
    assert pmc["traffic_bytes_per_launch_avg"] > 0


# ---- `python bench.py --gpus N` must run N ranks or fail: the launch decision, without a GPU
class _Args:
    def __init__(self, gpus, in_process=False):
        self.gpus, self.in_process = gpus, in_process


def test_launch_plan_runs_in_place_as_a_rank_or_as_the_one_gpu_job():
    assert bench.launch_plan(_Args(1), {}, ["--gpus", "1"]) == ("run", None)
    assert bench.launch_plan(_Args(4), {"WORLD_SIZE": "4", "RANK": "2"}, []) == ("run", None)
    assert bench.launch_plan(_Args(8, in_process=True), {}, []) == ("run", None)


def test_launch_plan_refuses_a_world_size_that_contradicts_the_flag():
    what, msg = bench.launch_plan(_Args(8), {"WORLD_SIZE": "1"}, [])
    assert what == "error" and "--gpus 8" in msg and "WORLD_SIZE=1" in msg
    assert bench.launch_plan(_Args(1), {"WORLD_SIZE": "2"}, [])[0] == "error"


def test_bare_gpus_n_spawns_n_ranks_under_torch_distributed_run():
    argv = ["--gpus", "2", "--steps", "3", "--single-device"]
    what, cmd = bench.launch_plan(_Args(2), {}, argv)
    assert what == "spawn"
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-len(argv) - 1].endswith("bench.py") and cmd[-len(argv):] == argv  # same arguments, same script


def test_spawn_relays_the_json_line_and_the_exit_code(tmp_path, capsys):
    import sys

    child = tmp_path / "child.py"
    child.write_text("import sys\nprint('banner')\nprint('{\"n_gpus\": 2}')\nsys.exit(int(sys.argv[1]))\n")
    assert bench.spawn_ranks([sys.executable, str(child), "0"]) == 0
    assert capsys.readouterr().out.strip() == '{"n_gpus": 2}'
    assert bench.spawn_ranks([sys.executable, str(child), "3"]) == 3
    silent = tmp_path / "silent.py"
    silent.write_text("pass\n")
    assert bench.spawn_ranks([sys.executable, str(silent)]) == 1  # no JSON line is a failure even at exit code 0


def test_a_contradicting_world_size_exits_non_zero_before_any_gpu_work():
    import os
    import subprocess
    import sys

    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8"], env=env, capture_output=True, timeout=120)
    assert r.returncode == 2 and b"WORLD_SIZE=1" in r.stderr and r.stdout.strip() == b""
