"""The bench.py output contract, checked on the lines recorded on the MI355X (profiles/r0*_bench_lines.jsonl) and on
bench.py's argument defaults — no GPU needed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_recorded_lines_follow_the_contract():
    import glob

    lines = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_lines.jsonl"))):
        lines += [json.loads(l) for l in open(f) if l.strip()]
    assert len(lines) >= 2
    for d in lines:
        for k in REQUIRED:
            assert k in d, k
        assert d["scaling"] in ("weak", "strong") and d["higher_is_better"] is True and d["vs_baseline"] is None
        assert d["data"] == "synthetic" and d["dtype"] in ("f64", "f32") and "workload" in d["config"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        # achieved = algorithmic bytes per launch / measured kernel time
        batch = d["config"]["batch_per_gpu"]
        assert abs(r["achieved"] - r["algorithmic_bytes_per_qp"] * batch / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["parity_status_equal"]
        # fp32 arithmetic under termination: the stop test sits on an fp32-noisy residual, so iteration counts equal the FLOAT
        # oracle's only statistically (tests/test_gpu_parity.py states the bar); everywhere else they are equal on every QP
        assert c["parity_iter_equal"] or (d["dtype"] == "f32" and d["config"]["mode"] != "fixed")
        if d["dtype"] == "f64":
            assert c["parity_max_rel_err_x"] < 1e-6 and c["parity_max_rel_err_y"] < 1e-6
        total = d["config"]["global_batch"]
        assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_refuses_without_a_device():
    import torch

    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and "no CPU path" in (p.stderr + p.stdout)


def test_bench_gpus_n_without_devices_fails_loudly():
    """`bench.py --gpus N` with fewer than N devices must not print an N-labelled (or silently 1-GPU) line (VERDICT r2 missing #2)"""
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "device(s) visible" in (p.stderr + p.stdout) and '"n_gpus"' not in p.stdout
    # a launcher-provided world that disagrees with --gpus is refused as well
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_bench_spawned_path_runs_the_rccl_gather():
    """bench.py starts its own ranks: `--spawn` takes the N>1 code path (torch.distributed.run, process group on the nccl = RCCL
    backend, ResultGather) on the one device of the box; the gathered records equal the resident state (asserted inside bench.py)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--batch-per-gpu", "1024"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["gather"] is True and d["config"]["global_batch"] == 1024
    # the N > 1 diagnostics (a first scaling run must be readable): ranks RCCL sees, per-rank kernel / solve / gather times
    mg = d["multi_gpu"]
    assert mg["rccl_ranks_seen"] == 1 and mg["backend"] == "nccl"
    for k in ("kernel_ms_avg_per_rank", "solve_ms_per_rank", "gather_ms_sync_per_rank"):
        assert len(mg[k]) == 1 and mg[k][0] > 0
    assert mg["record_bytes_per_rank"] == 1024 * (8 * 150 + 40)
    # strong-scaling split with a batch that does not divide evenly is fine on one rank too (padding path: rows == [1023])
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--global-batch", "1023"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["gather"] is True and d["config"]["global_batch"] == 1023 and d["scaling"] == "strong"


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device_over_gloo():
    """world size 2 with REAL kernels on a one-GPU box (RCCL refuses two ranks on one device; the gloo leg stages the gather through
    host buffers): different rank seeds, shard_bounds of an odd total (513 + 512), the padded ResultGather, and rank 0's re-solve
    check of sampled QPs of EVERY rank's shard (asserted inside bench.py) — the pieces an N = 1 run cannot exercise"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--global-batch", "1025"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 1025 and d["config"]["gather"] is True
    mg = d["multi_gpu"]
    assert mg["rccl_ranks_seen"] == 2 and mg["backend"] == "gloo"
    for k in ("kernel_ms_avg_per_rank", "solve_ms_per_rank", "gather_ms_sync_per_rank"):
        assert len(mg[k]) == 2 and all(v > 0 for v in mg[k])
    assert abs(d["value"] - 1025 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


EXTRA_KEYS_R4 = ("c2", "c3_default", "c3_sqp", "c3_whole_65536", "c5")  # round 4: the 8,192-QP shard was the headline
EXTRA_KEYS = ("c2", "c3_default", "c3_sqp", "c3_shard_8192", "c5")  # round 5 on: the whole 65,536 batch is


def check_extra(ex, keys=EXTRA_KEYS):
    for k in keys:
        e = ex[k]
        for f in ("workload", "ms_per_step", "kernel_ms_avg", "value", "frac", "traffic", "kernel", "algorithmic_bytes_per_qp", "parity"):
            assert f in e, (k, f)
        assert e["ms_per_step"] > 0 and e["value"] > 0 and 0 < e["frac"] < 1
        p = e["parity"]
        assert p["max_rel_err_x"] < 1e-6 and p["max_rel_err_y"] < 1e-6 and p["status_equal"] and p["iter_equal"], (k, p)
    assert "needed_bytes_per_qp" in ex["c5"] and ex["c5"]["frac_needed"] <= ex["c5"]["frac"]
    if keys is EXTRA_KEYS:  # configs[3], the batched SQP driver: cold (the reference's trajectories) and with warm-started subproblems
        c4 = ex["c4"]
        assert "error" not in c4, c4
        assert c4["instances"] == 1024 and c4["ms_per_batch"] > 0 and c4["launches"] > 0 and c4["strict_parity_with_serial_oracle"] >= 0.85 * 1024
        w = c4["warm_start_qp"]
        # (each is ONE timed run of a host-driven loop: the deterministic counters carry "the warm start pays", the wall clocks get slack for
        # a stall of a run — measured 6.4 against 14.6 ms)
        assert w["sum_qp_iter"] < c4["sum_qp_iter"] and w["solved"] >= c4["solved"] - 32 and w["ms_per_batch"] < 1.5 * c4["ms_per_batch"]


def test_recorded_default_line_carries_every_baseline_config():
    """the default `python bench.py` line (the one the driver records) measures the other BASELINE configs in its `extra` object"""
    f = os.path.join(ROOT, "profiles", "r04_bench_lines.jsonl")
    if not os.path.exists(f):
        pytest.skip("no round-4 lines recorded yet")
    lines = [json.loads(l) for l in open(f) if l.strip()]
    with_extra = [d for d in lines if "extra" in d]
    assert with_extra, "the default line of the round carries `extra`"
    for d in with_extra:
        check_extra(d["extra"], EXTRA_KEYS_R4)


def test_recorded_round5_default_line_is_the_whole_batch_with_every_config():
    """round 5: the driver-recorded default line is configs[2]'s whole 65,536 batch (strong scaling) and carries the shard, configs[1],
    [3] (the SQP driver, cold and warm-started) and [4] in `extra`, the PCIe-inclusive record and the traffic of the profiled kernels"""
    f = os.path.join(ROOT, "profiles", "r05_bench_lines.jsonl")
    if not os.path.exists(f):
        pytest.skip("no round-5 lines recorded yet")
    d = json.loads(open(f).readline())
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 65536 and d["n_gpus"] == 1
    check_extra(d["extra"])
    assert d["roofline"]["traffic"] and d["extra"]["c5"]["traffic"] and d["extra"]["c5"]["kernel"].startswith("csb_")
    assert d["pcie_inclusive"]["value"] < d["value"]
    assert d["cpu_baseline"]["threads_over_one_thread"] > 1
    sp = d["extra"].get("c5_sparse_P")  # configs[4]'s shape with P in compressed columns, read in place by the block-row kernel
    if sp is not None:
        assert sp["kernel"].endswith("_sp") and sp["algorithmic_bytes_per_qp"] < d["extra"]["c5"]["algorithmic_bytes_per_qp"] // 3
        p = sp["parity"]
        assert p["max_rel_err_x"] < 1e-6 and p["max_rel_err_y"] < 1e-6 and p["status_equal"] and p["iter_equal"], p


@pytest.mark.gpu
def test_default_bench_line_has_the_extra_configs():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-seconds", "2"],
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    for k in REQUIRED:
        assert k in d, k
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 65536 and d["config"]["batch_per_gpu"] == 65536
    check_extra(d["extra"])
    pc = d["pcie_inclusive"]
    assert pc["value"] > 0 and pc["value"] < d["value"] and pc["batch"] == 8192
    c = d["cpu_baseline"]
    assert c["threads_over_one_thread"] > 0 and "build" in c
