# fp32-product kernels against the fp64 ones at the BASELINE dense shapes (float interface both): tools/xp/f32_ab.sh <lib.so>
LIB=${1:-$PWD/sqp_solver_amd/lib/libsqp_hip.so}
for w in c3 c2; do
for mode in fixed default sqp; do
for f in "" "--f32-arith"; do
echo -n "$w $mode $f: "; SQPH_LIB=$LIB python bench.py --workload $w --dtype f32 $f --mode $mode --steps 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'], r['config']['admm_iters_per_qp'], r['cpu_baseline'].get('parity_max_rel_err_x'), r['cpu_baseline'].get('parity_max_rel_err_y'))"
done; done; done
