"""Drop-in boundary with Eigen present: the facade's Eigen mode is compiled against a test-only stand-in for
Eigen's dense API (tests/cpp/eigen_stub — the image has no Eigen) through include/sqp_hip/compat, the mirror of the
reference's include paths.

* tests/cpp/qp_dropin_test.cpp, qp_dropin_legacy_test.cpp: this repository's own callers written in the reference's style
  (`#include "solvers/qp.hpp"`, `qp.P = &P;`, Eigen vectors from primal_solution(), QP<2,3> with Eigen members).
* the reference's OWN test files (tests/qp_solver_test.cpp, tests/unsupported/qp_solver_test.cpp,
  tests/qp_solver_sparse_test.cpp, tests/sqp_test.cpp, tests/sqp_test_autodiff.cpp, tests/bfgs_test.cpp + test_main.cpp), compiled
  UNCHANGED from where they lie under /root/reference against the facade + the stand-ins (Eigen, GoogleTest).  They can only
  be compiled where /root/reference exists (this container); the binaries land in tests/cpp/_ref/ (git-ignored, they travel
  to the GPU box like the built .so) and the GPU test runs them when they are there.  Nothing of the reference is copied.
"""
import os
import subprocess

import pytest

from sqp_solver_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
REFDIR = os.path.join(CPP, "_ref")
REFERENCE = "/root/reference"
INC = ["-I" + os.path.join(ROOT, "include", "sqp_hip", "compat"), "-I" + os.path.join(CPP, "eigen_stub")]  # (the stand-ins: Eigen/Dense, Eigen/Sparse,
# Eigen/Eigenvalues, unsupported/Eigen/AutoDiff)
OWN = ["qp_dropin_test", "qp_dropin_legacy_test", "qp_dropin_legacy_sparse_test", "sqp_dropin_test"]
REF = {  # binary -> reference sources (relative to /root/reference)
    "ref_qp_solver_test": ["tests/qp_solver_test.cpp", "tests/test_main.cpp"],
    "ref_legacy_qp_solver_test": ["tests/unsupported/qp_solver_test.cpp", "tests/test_main.cpp"],
    # the legacy class's sparse variant (QP_SOLVER_USE_SPARSE: Eigen::SparseMatrix members, CSR on the device)
    "ref_sparse_qp_solver_test": ["tests/qp_solver_sparse_test.cpp", "tests/test_main.cpp"],
    # the SQP class (compat/solvers/sqp.hpp: sqp::SQP<T> = the batched driver with one instance) and the BFGS update
    "ref_sqp_test": ["tests/sqp_test.cpp", "tests/test_main.cpp"],
    "ref_sqp_test_autodiff": ["tests/sqp_test_autodiff.cpp", "tests/test_main.cpp"],
    "ref_bfgs_test": ["tests/bfgs_test.cpp", "tests/test_main.cpp"],
}
# The reference's OWN outer loop (src/sqp.cpp, with its own include/solvers/sqp.hpp and bfgs.hpp), compiled unchanged, on top of the
# facade's QPSolver: the literal north_star boundary — SQP::solve() on the host, the GPU QP solver beneath.  `solvers/qp.hpp` must
# resolve to the facade while `solvers/sqp.hpp` / `bfgs.hpp` resolve to the reference's files: a generated include directory of
# one-line redirects (tests/cpp/_ref/inc, git-ignored).
REFSRC = {
    "refsrc_sqp_test": ["src/sqp.cpp", "tests/sqp_test.cpp", "tests/test_main.cpp"],
    "refsrc_sqp_test_autodiff": ["src/sqp.cpp", "tests/sqp_test_autodiff.cpp", "tests/test_main.cpp"],
}
HOST_ONLY = ["ref_bfgs_test"]  # no QP solve inside: runs without a device
# The one case of the reference's files that does not pass: the n = 3 half of SQPAutoDiff.TestRosenbrock.  The algorithm as
# written stalls at (1, 1, 0) there — the CPU oracle (double and x87 QP arithmetic) takes the identical path; DESIGN.md section 7.
KNOWN_FAILURES = {"ref_sqp_test_autodiff": ["SQPAutoDiff.TestRosenbrock"], "refsrc_sqp_test_autodiff": ["SQPAutoDiff.TestRosenbrock"]}


MANIFEST = os.path.join(CPP, "REF_MANIFEST")  # tracked: the binaries tests/cpp/_ref/ must hold on a GPU box (written where they are built)


def _write_manifest():
    with open(MANIFEST, "w") as f:
        f.write("# binaries built from the reference's own sources (tests/test_cpp_dropin.py); they travel untracked in tests/cpp/_ref/.\n")
        f.write("# A GPU box on which one of them is missing FAILS test_cpp_dropin's GPU tests instead of skipping them.\n")
        for k in list(REF) + list(REFSRC):
            f.write(k + ".bin\n")


def _require_manifest_binaries(names):
    """every binary the tracked manifest promises is there — a missing push must not quietly shrink the evidence"""
    if not os.path.exists(MANIFEST):
        return
    want = [l.strip() for l in open(MANIFEST) if l.strip() and not l.startswith("#")]
    missing = [w for w in want if w[:-4] in names and not os.path.exists(os.path.join(REFDIR, w))]
    assert not missing, "tests/cpp/_ref/ lacks %s (listed in tests/cpp/REF_MANIFEST): run __graft_entry__.build() where /root/reference exists" % missing


def _link_args(depth):
    _capi.load()
    lib = _capi.lib_path()
    up = "/".join([".."] * depth)
    return [lib, "-Wl,-rpath,$ORIGIN/%s/sqp_solver_amd/lib" % up, "-Wl,-rpath,/opt/rocm/lib"]


def build_own():
    out = []
    for name in OWN:
        exe = os.path.join(CPP, name + ".bin")
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror"] + INC + ["-o", exe, os.path.join(CPP, name + ".cpp")] + _link_args(2))
        out.append(exe)
    return out


def build_reference_tests():
    """Compile the reference's own QP test files against the facade (only where /root/reference exists)."""
    if not os.path.isdir(REFERENCE):
        return [os.path.join(REFDIR, k + ".bin") for k in REF if os.path.exists(os.path.join(REFDIR, k + ".bin"))]
    os.makedirs(REFDIR, exist_ok=True)
    out = []
    for name, srcs in REF.items():
        exe = os.path.join(REFDIR, name + ".bin")
        cmd = ["g++", "-std=c++14", "-O1"] + INC + ["-I" + os.path.join(CPP, "gtest_stub"), "-o", exe]
        cmd += [os.path.join(REFERENCE, s) for s in srcs] + _link_args(3)
        subprocess.check_call(cmd)
        out.append(exe)
    _write_manifest()
    return out


def build_reference_sqp_source():
    """src/sqp.cpp of the reference + its own tests, over compat/solvers/qp.hpp (only where /root/reference exists)."""
    if not os.path.isdir(REFERENCE):
        return [os.path.join(REFDIR, k + ".bin") for k in REFSRC if os.path.exists(os.path.join(REFDIR, k + ".bin"))]
    inc = os.path.join(REFDIR, "inc", "solvers")
    os.makedirs(inc, exist_ok=True)
    redirects = {"qp.hpp": os.path.join(ROOT, "include", "sqp_hip", "compat", "solvers", "qp.hpp"),
                 "sqp.hpp": os.path.join(REFERENCE, "include", "solvers", "sqp.hpp"),
                 "bfgs.hpp": os.path.join(REFERENCE, "include", "solvers", "bfgs.hpp")}
    for name, target in redirects.items():
        with open(os.path.join(inc, name), "w") as f:
            f.write("#pragma once\n#include \"%s\"\n" % target)
    out = []
    for name, srcs in REFSRC.items():
        exe = os.path.join(REFDIR, name + ".bin")
        cmd = ["g++", "-std=c++14", "-O1", "-I" + os.path.join(REFDIR, "inc"), "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(CPP, "eigen_stub"), "-I" + os.path.join(CPP, "gtest_stub"), "-o", exe]
        cmd += [os.path.join(REFERENCE, s) for s in srcs] + _link_args(3)
        subprocess.check_call(cmd)
        out.append(exe)
    return out


def _has_gpu():
    import torch

    return torch.cuda.is_available()


def test_eigen_mode_facade_compiles_and_refuses_without_device():
    for exe in build_own():
        p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        if _has_gpu():
            assert p.returncode == 0, p.stdout + p.stderr
        else:
            assert p.returncode == 3, (exe, p.returncode, p.stderr)  # host-only cases passed, then: no HIP device


def test_reference_test_files_compile_unchanged_against_the_facade():
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference tree is not on this machine (binaries are prebuilt where it is)")
    exes = build_reference_tests()
    assert len(exes) == len(REF)
    if not _has_gpu():
        # the one host-only case of the supported class's file passes even without a device
        p = subprocess.run([exes[0]], capture_output=True, text=True, timeout=120)
        assert "[       OK ] QPSolverTest.TestConstraint" in p.stdout, p.stdout
    for exe in exes:  # the reference's BFGS tests need no device at all
        if os.path.basename(exe)[:-4] in HOST_ONLY:
            p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
            assert p.returncode == 0 and " 0 failed" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_eigen_mode_facade_on_the_gpu():
    for exe in build_own():
        p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "all passed" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_reference_gtest_files_pass_against_the_facade():
    exes = build_reference_tests()
    _require_manifest_binaries(REF)
    if not exes:
        pytest.skip("tests/cpp/_ref/*.bin not built (needs /root/reference at build time) and no manifest promises them")
    for exe in exes:
        p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        print(p.stdout)
        name = os.path.basename(exe)[:-4]
        failed = sorted(l.split("]")[1].strip() for l in p.stdout.splitlines() if l.startswith("[  FAILED  ]"))
        assert failed == sorted(KNOWN_FAILURES.get(name, [])), p.stdout + p.stderr
        assert "tests ran" in p.stdout


def _solution_lines(exe):
    env = dict(os.environ, SQP_HIP_STUB_FULL_PRECISION="1")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    sols = [l for l in p.stdout.splitlines() if l.startswith(("primal solution", "dual solution"))]
    verdicts = [l for l in p.stdout.splitlines() if l.startswith(("[       OK ]", "[  FAILED  ]"))]
    return p, sols, verdicts


@pytest.mark.gpu
def test_reference_sqp_source_over_the_facade_equals_the_dropin_class_bit_for_bit():
    """The reference's src/sqp.cpp (unchanged, its own sqp.hpp / bfgs.hpp) driving the facade's QPSolver — the literal north_star
    boundary — against compat/solvers/sqp.hpp (the batched driver with one instance) on the reference's own SQP tests: same QP
    kernels, same host arithmetic order, so every printed primal / dual solution must agree to the last digit and the same tests
    must pass.  Any difference is a bug in one of the two outer loops."""
    build_reference_tests()
    exes = build_reference_sqp_source()
    _require_manifest_binaries(REFSRC)
    if not exes:
        pytest.skip("tests/cpp/_ref/refsrc_*.bin not built (needs /root/reference at build time) and no manifest promises them")
    for exe in exes:
        name = os.path.basename(exe)[:-4]
        twin = os.path.join(REFDIR, name.replace("refsrc_", "ref_") + ".bin")
        pa, sa, va = _solution_lines(exe)
        pb, sb, vb = _solution_lines(twin)
        print(pa.stdout)
        assert "tests ran" in pa.stdout and len(sa) >= 6, pa.stdout + pa.stderr
        assert sa == sb, "\n".join("%s | %s" % ab for ab in zip(sa, sb))
        assert va == vb
        failed = sorted(l.split("]")[1].strip() for l in va if l.startswith("[  FAILED  ]"))
        assert failed == sorted(KNOWN_FAILURES.get(name, [])), pa.stdout
