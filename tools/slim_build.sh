#!/bin/bash
# Experiment build: libsqp_hip with the C3 wg shape only (+ generic fallback).  usage: tools/slim_build.sh <out.so> [extra flags]
OUT=$1; shift
cd $(dirname $0)/..
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DSQPH_SLIM "$@" -o $OUT sqp_solver_amd/csrc/capi.hip sqp_solver_amd/csrc/wg_nocheck.hip sqp_solver_amd/csrc/csr_nocheck.hip sqp_solver_amd/csrc/wg_f32.hip sqp_solver_amd/csrc/csr_dense.hip sqp_solver_amd/csrc/wg_stack.hip 2>&1 | grep -E "error" | head -5
ls -la $OUT
