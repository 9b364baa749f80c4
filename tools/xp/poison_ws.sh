#!/bin/bash
# Full library with -DSQPH_EXPERIMENTS -> sqp_solver_amd/lib/libsqp_hip_xp.so (never shipped).  With SQPH_POISON_WS=1 every device
# allocation of the library starts as 0xFF bytes (capi.hip::ws_malloc); run the GPU suite and the soaks against it:
#   bash tools/xp/poison_ws.sh build            (here)
#   gpurun -- 'bash tools/xp/poison_ws.sh run'  (GPU box)
cd $(dirname $0)/../..
OUT=$PWD/sqp_solver_amd/lib/libsqp_hip_xp.so
if [ "$1" = build ]; then
  G="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSQPH_EXPERIMENTS"
  SK="-mllvm -structurizecfg-skip-uniform-regions"
  TMP=$(mktemp -d)
  for u in capi csr_nocheck wg_f32 wg_nocheck wg_stack csr_dense csrb csrb_sp; do
    X=""   # flags of single units: sqp_solver_amd/build.py UNIT_FLAGS
    [ "${u#csrb}" != "$u" ] && X="-mllvm -simplifycfg-sink-common=false $SK"
    [ "$u" = wg_nocheck -o "$u" = wg_stack ] && X="$SK"
    /opt/rocm/bin/hipcc $G $X -c -o $TMP/$u.o sqp_solver_amd/csrc/$u.hip 2>&1 | grep error &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $TMP/*.o; rm -rf $TMP; ls -la $OUT
else
  export SQPH_LIB=$OUT SQPH_POISON_WS=1
  python -m pytest tests -m gpu -q 2>&1 | tail -15
  for sd in 0 1; do SQPH_SOAK_SEED=$sd python tools/soak_shapes.py 2>&1 | tail -n 3 | cut -c1-600; done
  python tools/soak_shapes_csr.py 2>&1 | tail -n 3 | cut -c1-600
  python tools/soak_shapes_csr_sparse_P.py 2>&1 | tail -n 3 | cut -c1-600
fi
