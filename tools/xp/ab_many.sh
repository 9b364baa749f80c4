# bench lines of several slim builds in the three modes, alternating: tools/xp/ab_many.sh A I1 I2 ...
L=$PWD/sqp_solver_amd/lib
for i in 1 2 3; do for mode in fixed default sqp; do for v in "$@"; do
echo -n "$mode $v: "; SQPH_LIB=$L/libsqp_hip_slim$v.so python bench.py --no-cpu-baseline --no-extra --steps 40 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])"
done; done; done
SQPH_LIB=$L/libsqp_hip_slim${2:-A}.so timeout 300 python tools/xp/parity_c3.py 2>&1 | tail -6
