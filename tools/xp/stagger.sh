#!/bin/bash
# experiment (round 3): de-phase co-resident workgroups of the C3 kernel.  Needs sqp_solver_amd/lib/libsqp_hip_xp.so
# (tools/slim_build.sh sqp_solver_amd/lib/libsqp_hip_xp.so -DSQPH_EXPERIMENTS).  Output: gpurun_out/xp_stagger.txt
cd $(dirname $0)/../..
mkdir -p gpurun_out
OUT=gpurun_out/xp_stagger.txt
: > $OUT
export SQPH_LIB=$PWD/sqp_solver_amd/lib/libsqp_hip_xp.so
run() { # label, env...
  local label=$1; shift
  for mode in fixed; do
    echo -n "$label $mode: " >> $OUT
    env "$@" python bench.py --no-cpu-baseline --steps 30 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'])" >> $OUT
  done
}
# production library for reference
SQPH_LIB= python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('prod fixed:', r['ms_per_step'], r['roofline']['kernel_ms_avg'])" >> $OUT
run base SQPH_XP0=0
SQPH_XDBG=$PWD/gpurun_out/xp_dispatch_base.bin python bench.py --no-cpu-baseline --steps 1 --warmup 0 >/dev/null 2>&1
for D in 50000 100000 150000 200000 300000 400000; do
  run "hwslot D=$D first1024" SQPH_XP0=$D SQPH_XP1=1 SQPH_XP3=1024
  run "bit9 D=$D first1024" SQPH_XP0=$D SQPH_XP1=0 SQPH_XP2=9 SQPH_XP3=1024
  run "bit8 D=$D first1024" SQPH_XP0=$D SQPH_XP1=0 SQPH_XP2=8 SQPH_XP3=1024
  run "bit0 D=$D first1024" SQPH_XP0=$D SQPH_XP1=0 SQPH_XP2=0 SQPH_XP3=1024
done
for D in 50000 100000 150000 190000; do
  run "4phase bits8-9 D=$D first1024" SQPH_XP0=$D SQPH_XP1=2 SQPH_XP2=8 SQPH_XP3=1024
done
SQPH_XP0=150000 SQPH_XP1=1 SQPH_XP3=1024 SQPH_XDBG=$PWD/gpurun_out/xp_dispatch_hwslot.bin python bench.py --no-cpu-baseline --steps 1 --warmup 0 >/dev/null 2>&1
run base-again SQPH_XP0=0
cat $OUT
