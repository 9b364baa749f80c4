#!/bin/bash
# experiment (round 3): row-split kernel (admm_wgr_kernel.h) against the column-split one, same experiment build, same box.
cd $(dirname $0)/../..
mkdir -p gpurun_out
OUT=gpurun_out/xp_wgr.txt
: > $OUT
export SQPH_LIB=${XPLIB:-$PWD/sqp_solver_amd/lib/libsqp_hip_xp.so}
python - >> $OUT 2>&1 <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, cases
from sqp_solver_amd import QPSolverBatch
def mk(n, m, b, dtype=np.float64, **kw):
    return QPSolverBatch(n, m, b, dtype=dtype, device=0, keep_factor=kw.get("keep_factor", False), legacy_cold_start=kw.get("legacy_cold_start", False))
for (n, m) in ((50, 100), (56, 104), (40, 112), (33, 65)):
    s = mk(n, m, 4); s.setup_solve(*cases.random_qp_batch(4, n, m, seed=1)); print("kernel", n, m, s.kernel_name())
    print(" fixed", cases.parity_fixed_iters(mk, n, m, 64, iters=200))
    for kw in (dict(), dict(adaptive=True), dict(sqp_settings=True)):
        info = cases.parity_termination(mk, n, m, 64, **kw); print(" term", kw, cases.HATCH_COUNTS[-1]["excused"], cases.HATCH_COUNTS[-1]["widened"], int(info.iter.mean()))
cases.fused_then_solve(mk, n=50, m=100, batch=5); cases.soc_factor_reuse(mk, n=50, m=100, batch=5); cases.soc_reuse_after_failed_setup(mk, n=50, m=100, batch=5)
cases.warm_start_and_resolve(mk, n=50, m=100); print("state paths ok")
PY
run() { local label=$1; shift
  for mode in fixed default sqp; do
    echo -n "$label $mode: " >> $OUT
    env "$@" python bench.py --no-cpu-baseline --steps 30 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'], r['config']['admm_iters_per_qp'])" >> $OUT
  done
}
for i in 1 2; do
run wg SQPH_NO_WGR=1
run wgr SQPH_XP0=0
done
cat $OUT
