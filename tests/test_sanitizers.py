"""Host sanitizers: the plain-C oracle, the C++ facade's host side and the multi-GPU split arithmetic under
-fsanitize=address,undefined (g++ / gcc; no GPU needed — the facade test stops with exit code 3 at the first device call)."""
import os
import subprocess

import pytest

from sqp_solver_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def _run(exe, args=(), leaks=True):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=%d:abort_on_error=0" % int(leaks), UBSAN_OPTIONS="print_stacktrace=1")
    return subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=600, env=env)


def test_oracle_under_asan_ubsan():
    exe = os.path.join(CPP, "oracle_sanitize_test.bin")
    subprocess.check_call(["gcc", "-std=c11", "-fopenmp", "-ffp-contract=off"] + SAN + ["-o", exe, os.path.join(CPP, "oracle_sanitize_test.c"),
                                                                                        os.path.join(ROOT, "oracle", "qp_oracle.c"),
                                                                                        os.path.join(ROOT, "oracle", "sqp_oracle.c"), "-lm"])
    p = _run(exe)
    assert p.returncode == 0 and "oracle sanitize run passed" in p.stdout, p.stdout + p.stderr
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr


def _link():
    _capi.load()
    lib = _capi.lib_path()
    return [lib, "-ldl", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"]


def test_facade_host_side_under_asan_ubsan():
    """include/sqp_hip/qp.hpp + the C-ABI's argument checking (the HIP runtime's own allocations are not ours to judge: no leak check)"""
    import torch

    exe = os.path.join(CPP, "qp_facade_test_san.bin")
    subprocess.check_call(["g++", "-std=c++14", "-pthread"] + SAN + ["-o", exe, os.path.join(CPP, "qp_facade_test.cpp")] + _link())
    p = _run(exe, leaks=False)
    assert p.returncode == (0 if torch.cuda.is_available() else 3), (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr


def test_multi_gpu_split_arithmetic_under_asan_ubsan():
    exe = os.path.join(CPP, "multi_gpu_test_san.bin")
    subprocess.check_call(["g++", "-std=c++14", "-pthread"] + SAN + ["-o", exe, os.path.join(CPP, "multi_gpu_test.cpp")] + _link())
    p = _run(exe, ["split"], leaks=False)
    assert p.returncode == 0 and "split arithmetic passed" in p.stdout, p.stdout + p.stderr
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr


@pytest.mark.gpu
def test_facade_and_multi_gpu_on_the_device_under_asan_ubsan():
    """the same two binaries with a device: every facade case and the sharded solve + gather (RCCL sends-to-self included) with the
    host code instrumented"""
    for name, args in (("qp_facade_test", []), ("multi_gpu_test", [])):
        exe = os.path.join(CPP, name + "_san.bin")
        subprocess.check_call(["g++", "-std=c++14", "-pthread"] + SAN + ["-o", exe, os.path.join(CPP, name + ".cpp")] + _link())
        p = _run(exe, args, leaks=False)
        assert p.returncode == 0, (name, p.stdout[-3000:], p.stderr[-3000:])
        assert "runtime error" not in p.stderr and "ERROR: AddressSanitizer" not in p.stderr, p.stderr[-3000:]
