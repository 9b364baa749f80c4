#!/bin/bash
# time the fixed-iteration C3 bench with several experiment builds: tools/xp/wgr_variants.sh <suffix>...   (libsqp_hip_xp<suffix>.so)
cd $(dirname $0)/../..
mkdir -p gpurun_out
OUT=gpurun_out/xp_wgr_variants.txt
: > $OUT
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/sqp_solver_amd/lib/libsqp_hip_xp$v.so
  [ "$v" = "-" ] && lib=$PWD/sqp_solver_amd/lib/libsqp_hip_xp.so
  for mode in ${MODES:-fixed}; do
  echo -n "xp$v $mode: " >> $OUT
  SQPH_LIB=$lib python bench.py --no-cpu-baseline --steps 30 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])" >> $OUT
  done
done; done
echo -n "wg: " >> $OUT; SQPH_NO_WGR=1 SQPH_LIB=$PWD/sqp_solver_amd/lib/libsqp_hip_xp.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])" >> $OUT
cat $OUT
