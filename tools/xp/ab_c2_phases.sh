# C2 (4,096 x (20,40)): set-up / iteration phase ticks and bench lines of two slim builds libsqp_hip_c2<V>_{setup,phase,plain}.so
L=$PWD/sqp_solver_amd/lib
for v in ${*:-base new}; do
echo "== $v"
SQPH_LIB=$L/libsqp_hip_c2${v}_setup.so python tools/setup_timing.py 20 40 4096
SQPH_LIB=$L/libsqp_hip_c2${v}_phase.so python tools/phase_timing.py 20 40 4096
done
for i in 1 2 3; do for v in ${*:-base new}; do echo -n "c2 $v: "; SQPH_LIB=$L/libsqp_hip_c2${v}_plain.so python bench.py --workload c2 --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])"; done; done
