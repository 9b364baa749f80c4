"""Multi-GPU behind the C++ API (include/sqp_hip/multi_gpu.hpp): split arithmetic on the host, and on the GPU
box the sharded solve for every device count available against the single-device solve (bit-identical, unsharded order)."""
import os
import subprocess

import pytest

from sqp_solver_amd import _capi
from sqp_solver_amd.dist import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "multi_gpu_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "multi_gpu_test.bin")


def build():
    _capi.load()
    lib = _capi.lib_path()
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-pthread", "-o", EXE, SRC, lib, "-ldl", "-Wl,-rpath,$ORIGIN/../../sqp_solver_amd/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_split_arithmetic_host_only():
    p = subprocess.run([build(), "split"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "split arithmetic passed" in p.stdout, p.stdout + p.stderr


def test_c_abi_split_equals_the_python_split():
    import ctypes

    L = _capi.load()
    L.sqph_shard_bounds.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
    L.sqph_shard_bounds.restype = None
    for total in (1, 5, 8192, 65536, 65539):
        for world in (1, 2, 4, 8):
            for r in range(world):
                lo, hi = ctypes.c_longlong(), ctypes.c_longlong()
                L.sqph_shard_bounds(total, world, r, ctypes.byref(lo), ctypes.byref(hi))
                assert (lo.value, hi.value) == shard_bounds(total, world, r)


@pytest.mark.gpu
def test_multi_gpu_solver_matches_single_device():
    p = subprocess.run([build()], capture_output=True, text=True, timeout=600)
    print(p.stdout)
    assert p.returncode == 0 and "all passed" in p.stdout, p.stdout + p.stderr
