# stacked operator (wg_stack.hip) against the padded one, same experiment build (-DSQPH_EXPERIMENTS), alternating
L=$PWD/sqp_solver_amd/lib/libsqp_hip_slimB.so
for i in 1 2; do for mode in fixed default sqp; do for v in 1 ""; do
echo -n "$mode nostack=$v: "; env ${v:+SQPH_NO_STACK=1} SQPH_LIB=$L python bench.py --no-cpu-baseline --steps 40 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])"
done; done; done
