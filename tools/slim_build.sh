#!/bin/bash
# Experiment build: libsqp_hip with the C3 wg shape only (+ generic fallback).  usage: tools/slim_build.sh <out.so> [extra flags]
OUT=$1; shift
cd $(dirname $0)/..
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSQPH_SLIM"
TMP=$(mktemp -d)
for u in capi wg_nocheck csr_nocheck wg_f32 csr_dense wg_stack csrb csrb_sp; do
  X=""   # flags of single units: sqp_solver_amd/build.py UNIT_FLAGS
  [ "${u#csrb}" != "$u" ] && X="-mllvm -simplifycfg-sink-common=false -mllvm -structurizecfg-skip-uniform-regions"
  [ "$u" = wg_nocheck -o "$u" = wg_stack ] && X="-mllvm -structurizecfg-skip-uniform-regions"
  /opt/rocm/bin/hipcc $F "$@" $X -c -o $TMP/$u.o sqp_solver_amd/csrc/$u.hip 2>&1 | grep -E "error" | head -5 &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $TMP/*.o 2>&1 | grep -E "error" | head -5
rm -rf $TMP
ls -la $OUT
