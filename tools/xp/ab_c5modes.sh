# same-box A/B of block-row kernel builds on config 5 under termination checks: tools/xp/ab_c5modes.sh lib1.so lib2.so ...   (PD=0.03: P sparse too)
for rep in 1 2; do for mode in default sqp; do for L in "$@"; do
  echo -n "c5 ${PD:+sparse-P }$mode $(basename $L): "; SQPH_LIB=$L python bench.py --workload c5 --mode $mode --no-cpu-baseline --no-extra --steps 3 --warmup 1 ${PD:+--p-density $PD} 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.3f %.3f %.1f' % (r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['admm_iters_per_qp']), r['config']['kernel'])"
done; done; done
