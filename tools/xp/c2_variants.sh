# C2 (4,096 x (20,40), 200 fixed iterations) through the kernel families that can take it (experiment build with the environment knobs)
L=$PWD/sqp_solver_amd/lib/libsqp_hip_xpfull.so
for i in 1 2; do
for v in "" "SQPH_NO_G32=1" "SQPH_NO_G32=1 SQPH_WG_SKIP=1"; do
echo -n "c2 [$v]: "; env $v SQPH_LIB=$L python bench.py --workload c2 --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])"
done; done
for v in "" "SQPH_NO_G32=1"; do
echo -n "c2 x4 batch [$v]: "; env $v SQPH_LIB=$L python bench.py --n 20 --m 40 --batch-per-gpu 16384 --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])"
done
