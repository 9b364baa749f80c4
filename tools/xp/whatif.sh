#!/bin/bash
# time + LDS conflict counters of what-if builds: tools/xp/whatif.sh <suffix>...
cd $(dirname $0)/../..
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/xp_whatif.txt
: > $OUT
for v in "$@"; do
  lib=$PWD/sqp_solver_amd/lib/libsqp_hip_xp$v.so
  [ "$v" = "-" ] && lib=$PWD/sqp_solver_amd/lib/libsqp_hip_xp.so
  ms=$(SQPH_LIB=$lib python bench.py --no-cpu-baseline --steps 20 ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['kernel_ms_avg'], r['config']['kernel'])")
  D=gpurun_out/whatif_tmp; rm -rf $D
  SQPH_LIB=$lib rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $D -o pmc -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2>&1
  cnt=$(python - $D <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "admm" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(" ".join("%s %.3g" % (k.replace("SQ_", ""), sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
)
  echo "xp$v: $ms | $cnt" >> $OUT
  rm -rf $D
done
cat $OUT
