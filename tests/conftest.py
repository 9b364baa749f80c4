import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # the measurement-contract tests (bench.py subprocesses: the driver line with every extra, the spawned RCCL path) run after the
    # parity suites: under `-x` a hiccup of a timed subprocess must not hide them  (stable sort: everything else keeps its order)
    items.sort(key=lambda it: os.path.basename(str(it.fspath)) == "test_bench_contract.py")
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    if _has_gpu():
        import lds_poison

        lds_poison.install()  # (SQPH_TEST_POISON_LDS=0 turns it off)


def _hatch_summary():
    try:
        import cases
    except Exception:
        return None
    recs = list(cases.HATCH_COUNTS)
    if not recs:
        return None
    return {"calls": len(recs), "qps": int(sum(r["batch"] for r in recs)), "excused": int(sum(r["excused"] for r in recs)),
            "widened": int(sum(r["widened"] for r in recs)), "diag_unstable": int(sum(r["diag_unstable"] for r in recs)),
            "records": recs}


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Parity escape hatches taken in this session (tests/cases.py::parity_termination), so that a green run says how green."""
    import lds_poison

    if lds_poison.STATE["lib"] is not None:
        terminalreporter.write_line("LDS of every CU poisoned with NaN before each of %d solver calls (tests/lds_poison.hip)" % lds_poison.STATE["calls"])
    s = _hatch_summary()
    if s is None:
        return
    terminalreporter.write_line(
        "parity hatches: %d parity_termination calls, %d QPs: %d excused (reference path unstable), %d on the widened bar, "
        "%d with unstable reference diagnostics" % (s["calls"], s["qps"], s["excused"], s["widened"], s["diag_unstable"]))
    for r in s["records"]:
        if r["excused"] or r["widened"]:
            terminalreporter.write_line("  n=%d m=%d batch=%d adaptive=%s sqp=%s %s: excused %d widened %d" % (
                r["n"], r["m"], r["batch"], r["adaptive"], r["sqp_settings"], r["kw"], r["excused"], r["widened"]))
    if _has_gpu():
        try:
            import json

            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            json.dump(s, open(os.path.join(d, "parity_counts.json"), "w"), indent=1)
        except Exception:
            pass
