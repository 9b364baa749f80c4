# C2 (4,096 x (20,40)): bench lines of slim builds libsqp_hip_c2<V>_plain.so, fixed and default modes, alternating
L=$PWD/sqp_solver_amd/lib
for i in 1 2 3; do for mode in fixed default; do for v in ${*:-base new}; do echo -n "c2 $mode $v: "; SQPH_LIB=$L/libsqp_hip_c2${v}_plain.so python bench.py --workload c2 --mode $mode --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])"; done; done; done
