# A/B of slim experiment builds (tools/slim_build.sh): bench lines in the three modes, alternating, same box
# usage: tools/ab_slim.sh [variants, default "A B"]
V=${*:-A B}
for i in 1 2; do
for mode in fixed default sqp; do
for v in $V; do
echo -n "$mode $v: "; SQPH_LIB=$PWD/sqp_solver_amd/lib/libsqp_hip_slim$v.so python bench.py --no-cpu-baseline --steps 40 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['admm_iters_per_qp'])"
done; done; done
