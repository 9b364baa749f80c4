#!/bin/bash
# Profile bench.py on the GPU box with rocprofv3 (run through gpurun from the repo root):
#   1. --kernel-trace --stats                (per-kernel durations)
#   2. --pmc SQ_* pass                       (VALU utilisation, stalls, LDS)
#   3. --pmc FETCH_SIZE pass                 (HBM read side; gfx950 reports 1/2 for wide coalesced reads)
#   4. --pmc WRITE_SIZE pass                 (HBM write side)
# Counter passes are separate runs and never combined with trace domains other than kernel-trace.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# BENCH_CMD overrides the profiled command (e.g. the config-5 timing tool)
BENCH=${BENCH_CMD:-"python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra $*"}   # (the launch pattern of the driver's bench run: 5 + 20)
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merged-back directory small: per-dispatch CSVs can be tens of MB
find $OUT -type f -size +2M -delete
find $OUT -type f | head -50
