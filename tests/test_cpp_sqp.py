"""BASELINE config 4: the batched SQP host driver (include/sqp_hip/sqp.hpp) and the SQP oracle.

CPU: the serial SQP oracle (oracle/sqp_oracle.c) reproduces the known answers of the reference's own SQP tests
(tests/sqp_test.cpp, tests/sqp_test_autodiff.cpp).  GPU: sqp::BatchSQP — all QP subproblems of an outer iteration in
one libsqp_hip launch — agrees per instance with the oracle on the reference cases and on 1,024 SimpleNLP instances."""
import os
import subprocess

import pytest

import oracle
from sqp_solver_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "sqp_batch_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "sqp_batch_test.bin")


def build():
    _capi.load()
    oracle.build()
    libdir = os.path.dirname(_capi.lib_path())
    odir = os.path.join(ROOT, "oracle")
    deps = [SRC, os.path.join(ROOT, "include", "sqp_hip", "sqp.hpp"), os.path.join(ROOT, "include", "sqp_hip", "qp.hpp"),
            os.path.join(odir, "libqp_oracle.so"), _capi.lib_path()]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["g++", "-std=c++14", "-O1", "-pthread", "-o", EXE, SRC, "-L" + odir, "-lqp_oracle", "-L" + libdir, "-lsqp_hip",
           "-Wl,-rpath,$ORIGIN/../../oracle", "-Wl,-rpath,$ORIGIN/../../sqp_solver_amd/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_sqp_oracle_reproduces_reference_known_answers():
    exe = build()
    p = subprocess.run([exe, "oracle"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "oracle cases passed" in p.stdout
    # the reference's BFGS tests (tests/bfgs_test.cpp): the driver's and the oracle's damped update, bit-identical
    assert "bfgs  Test2D_posdef" in p.stdout and "bfgs  Test2D_indef" in p.stdout


def test_batch_sqp_host_driver_is_bit_exact_with_the_oracle_qp_backend():
    """sqp::BatchSQP<double, OracleBatchQP> (tests/cpp) == serial SQP oracle, bit for bit, on every instance of every
    workload (reference cases, 2 x 1,024 SimpleNLP, 256 Rosenbrock3, 256 SimpleNLP2)."""
    p = subprocess.run([build(), "exact"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "exact cases passed" in p.stdout


def test_batch_sqp_warm_started_subproblems_reach_the_known_answers_with_the_oracle_backend():
    """sqp_settings_t::warm_start_qp (update_qp(); solve() from the second outer iteration on, src/qp.cpp:46-62) over the oracle QP
    backend: the known answers of the reference's SQP tests, TestRosenbrock2 asserted in the state the source explains"""
    p = subprocess.run([build(), "warm-oracle"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "warm oracle cases passed" in p.stdout and p.stdout.count("known answer REACHED") == 7


def test_batch_sqp_refuses_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([build()], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr)


@pytest.mark.gpu
def test_batch_sqp_matches_serial_oracle():
    p = subprocess.run([build()], capture_output=True, text=True, timeout=600)
    print(p.stdout)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "all passed" in p.stdout
