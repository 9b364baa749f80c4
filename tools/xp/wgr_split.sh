#!/bin/bash
# set-up vs iteration time of experiment builds: fixed-iteration bench at 1, 101 and 201 iterations.  tools/xp/wgr_split.sh <suffix>...
cd $(dirname $0)/../..
mkdir -p gpurun_out
OUT=gpurun_out/xp_wgr_split.txt
: > $OUT
t() { env "$@" python bench.py --no-cpu-baseline --steps 20 --iters $IT 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['kernel_ms_avg'], r['config']['kernel'])"; }
for v in "$@"; do
  lib=$PWD/sqp_solver_amd/lib/libsqp_hip_xp$v.so
  [ "$v" = "-" ] && lib=$PWD/sqp_solver_amd/lib/libsqp_hip_xp.so
  for IT in 1 101 201; do echo "xp$v iters $IT: $(t SQPH_LIB=$lib)" >> $OUT; done
done
for IT in 1 101 201; do echo "wg iters $IT: $(t SQPH_NO_WGR=1 SQPH_LIB=$PWD/sqp_solver_amd/lib/libsqp_hip_xp.so)" >> $OUT; done
cat $OUT
