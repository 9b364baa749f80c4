L=$PWD/sqp_solver_amd/lib/libsqp_hip_xpslim.so
for i in 1 2; do for v in "SQPH_NO_STACK=1" "SQPH_NO_STACK=1 SQPH_WG_ALWAYS_CHECKS=1"; do
echo -n "fixed [$v]: "; env $v SQPH_LIB=$L python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['config']['kernel'], r['config'].get('kernel_variant'))"
done; done
