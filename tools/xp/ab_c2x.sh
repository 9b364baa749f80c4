# C2 bench lines (fixed / default) of slim builds, alternating: tools/xp/ab_c2x.sh U0 U3 U4
L=$PWD/sqp_solver_amd/lib
for i in 1 2 3; do for mode in fixed default; do for v in "$@"; do echo -n "c2 $mode $v: "; SQPH_LIB=$L/libsqp_hip_slim$v.so python bench.py --workload c2 --mode $mode --no-extra --steps 100 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'], r['cpu_baseline'].get('parity_max_rel_err_x'), r['cpu_baseline'].get('parity_status_equal'))"; done; done; done
