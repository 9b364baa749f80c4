"""GPU suite and soak tools: every solver call of the Python facade finds the LDS of all CUs filled with NaN (tests/lds_poison.hip), so
that a kernel reading LDS it did not write fails deterministically instead of depending on which kernel ran on the CU before.
LDS is cleared neither between kernels nor between processes.  SQPH_TEST_POISON_LDS=0 turns it off."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# every 32-bit word of LDS: 0xFFFFFFFF = NaN as doubles, -1 as ints; SQPH_TEST_POISON_PAT=7FF00000 = NaN as doubles, a huge positive int
# SQPH_TEST_POISON_PAT=RANDOM = a different pseudo-random word everywhere, re-seeded per call
PATTERN = 1 if os.environ.get("SQPH_TEST_POISON_PAT", "").upper() == "RANDOM" else int(os.environ.get("SQPH_TEST_POISON_PAT", "FFFFFFFF"), 16)
STATE = {"lib": None, "calls": 0}


def install():
    if STATE["lib"] is not None or os.environ.get("SQPH_TEST_POISON_LDS", "1") == "0":
        return False
    src, so = os.path.join(HERE, "lds_poison.hip"), os.path.join(HERE, "_lds_poison.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        try:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so + ".tmp", src],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            os.replace(so + ".tmp", so)
        except Exception as e:  # a test tool must not take the suite down: keep an older build, or run unpoisoned and say so
            print("lds_poison: build failed (%r); %s" % (e, "using the existing build" if os.path.exists(so) else "LDS poison OFF"))
            if not os.path.exists(so):
                return False
    try:
        lib = ctypes.CDLL(so)
    except OSError as e:
        print("lds_poison: cannot load %s (%r); LDS poison OFF" % (so, e))
        return False
    lib.lds_poison.argtypes = [ctypes.c_uint, ctypes.c_int]
    from sqp_solver_amd import qp as _qp

    def wrap(name):
        inner = getattr(_qp.QPSolverBatch, name)

        def call(self, *a, **k):
            import torch

            if torch.cuda.is_initialized() and torch.cuda.is_current_stream_capturing():  # (a device-wide synchronisation would invalidate the capture)
                return inner(self, *a, **k)
            self.synchronize()
            rc = lib.lds_poison(PATTERN, 160 * 1024)
            assert rc == 0, "lds_poison: hip error %d" % rc
            STATE["calls"] += 1
            return inner(self, *a, **k)

        setattr(_qp.QPSolverBatch, name, call)

    wrap("_call")
    wrap("_call_csr")
    STATE["lib"] = lib
    return True
