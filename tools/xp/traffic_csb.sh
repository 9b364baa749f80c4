#!/bin/bash
# HBM traffic of the block-row kernel for experiment builds: tools/xp/traffic_csb.sh lib1.so lib2.so ...   (FETCH_SIZE / WRITE_SIZE passes, KiB per launch)
export PYTHONPATH=$PWD TMPDIR=/tmp
for L in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C
    SQPH_LIB=$L timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmc_$C -o pmc -- python tools/bench_csr.py --steps 2 --check 0 > /dev/null 2>&1
    f=$(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1)
    python - "$f" "$L" $C <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'csrb' in r.get('Kernel_Name','')]
v=[float(r['Counter_Value']) for r in rows if r.get('Counter_Name')==sys.argv[3]]
print(sys.argv[2].split('/')[-1], sys.argv[3], 'per launch KiB: %.0f' % (sum(v)/max(1,len(set(r['Dispatch_Id'] for r in rows)))), 'dispatches', len(set(r['Dispatch_Id'] for r in rows)))
PY
  done
done
