#!/bin/bash
# HBM traffic of one command under several library builds: tools/xp/traffic_lib.sh "<command>" lib1.so lib2.so ...
# (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, kernel-trace only; per-dispatch averages of the admm kernels, KiB)
export TMPDIR=/tmp PYTHONPATH=$PWD
CMD=$1; shift
for L in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_$$_$C; rm -rf $D
    SQPH_LIB=$L timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $D -o pmc -- $CMD > /dev/null 2>&1
    python - $D $C $(basename $L) <<'PY'
import csv, glob, os, sys
d, c, lib = sys.argv[1:4]
tot, ids = 0.0, set()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "admm" in r.get("Kernel_Name", "") and r["Counter_Name"] == c:
            tot += float(r["Counter_Value"]); ids.add(r.get("Dispatch_Id"))
print("%s %s %.4g KiB per dispatch (%d dispatches)%s" % (lib, c, tot / max(len(ids), 1), len(ids), "  [x2 for bytes on gfx950]" if c == "FETCH_SIZE" else ""))
PY
  done
done
