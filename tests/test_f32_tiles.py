"""SURVEY section 8 row f4 closed by measurement, tools/f32_tiles_measure.py:
* the Schur-ordered iteration entirely in fp32 is 10-50x less accurate than the reference's own float instantiation — no fp32
  arithmetic variant of the register-tiled kernels can match `QPSolver<float>`;
* fp32 STORAGE of the tiles with fp64 accumulation stays within ~2x of the float reference's own error — accurate enough, but
  gfx950 has no fp64 FMA with an fp32 operand: every use needs a v_cvt_f64_f32 (154 more VALU instructions on top of the 325 of
  a C3 iteration), and the set-up still needs the fp64 tiles, so the register high-water mark does not move.  Not shipped.
The committed record: profiles/r03_f32_tiles.json (emulator + MI355X)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _check(rec):
    r32 = rec["ref32_vs_truth"]
    st = rec["storage_vs_truth"]
    # storage-only: inside the acceptance rule (<= 4x the float reference's own error) — recorded, see the module docstring for
    # why it is not a kernel
    assert st["x"] <= 4 * r32["x"] and st["y"] <= 4 * r32["y"], rec
    assert rec["ship_fp32_tile_storage"] is True
    if "schur32_vs_truth" in rec:
        s32 = rec["schur32_vs_truth"]
        assert s32["x"] >= 10 * r32["x"] and s32["y"] >= 10 * r32["y"], rec  # fp32 arithmetic in the Schur form: not a QPSolver<float>


def test_f32_tile_options_under_the_emulator():
    import f32_tiles_measure as f

    for (n, m) in ((20, 40), (50, 100)):
        _check(f.measure("sim", n, m, 4, 100))


@pytest.mark.gpu
def test_f32_tile_storage_on_the_gpu(tmp_path):
    lib = os.path.join(ROOT, "sqp_solver_amd", "lib", "libsqp_hip_f32tiles.so")
    if not os.path.exists(lib):
        from sqp_solver_amd import build as b

        b.build_f32_tiles_experiment()
    out = os.path.join(ROOT, "gpurun_out", "f32_tiles_gpu.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    env = dict(os.environ, SQPH_LIB=lib)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "f32_tiles_measure.py"), "--backend", "gpu", "--out", out], env=env, timeout=900)
    recs = json.load(open(out))
    assert [r["kernel"] for r in recs] == ["wg1_8x8_5x3_w3", "wg2_16x8_7x7s_w2"]
    for rec in recs:
        _check(rec)
