#!/bin/bash
# SQ counters of the C3 fixed-iteration bench for two configurations of an experiment build (A = row-split kernel, B = SQPH_NO_WGR=1)
cd $(dirname $0)/../..
export TMPDIR=/tmp
LIB=${XPLIB:-$PWD/sqp_solver_amd/lib/libsqp_hip_xp.so}
for cfg in A B; do
  OUT=gpurun_out/pmc_ab_$cfg; rm -rf $OUT; mkdir -p $OUT
  if [ $cfg = B ]; then export SQPH_NO_WGR=1; else unset SQPH_NO_WGR; fi
  BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-}"
  SQPH_LIB=$LIB rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
  SQPH_LIB=$LIB rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
  python - $OUT <<'PY' > $OUT/summary.txt
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_sq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "admm" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s avg %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
  cat $OUT/summary.txt
  find $OUT -type f -size +1M -delete
done
