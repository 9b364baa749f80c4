# same-box A/B of slim builds on the C3 shard (8,192 QPs) in the three modes and on C2: tools/xp/ab_c3.sh A B ...
V=${*:-A B}
for i in 1 2; do
for mode in fixed default sqp; do
for v in $V; do
echo -n "c3 $mode $v: "; SQPH_LIB=$PWD/sqp_solver_amd/lib/libsqp_hip_slim$v.so python bench.py --no-cpu-baseline --no-extra --global-batch 0 --steps 40 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['admm_iters_per_qp'], r['config']['kernel'])"
done; done
for v in $V; do
echo -n "c2 fixed $v: "; SQPH_LIB=$PWD/sqp_solver_amd/lib/libsqp_hip_slim$v.so python bench.py --no-cpu-baseline --no-extra --workload c2 --steps 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])"
done
done
