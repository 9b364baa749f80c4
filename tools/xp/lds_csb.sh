#!/bin/bash
# LDS counters of the block-row kernel for experiment builds: tools/xp/lds_csb.sh lib1.so lib2.so ...   (per launch)
export PYTHONPATH=$PWD TMPDIR=/tmp
for L in "$@"; do
  rm -rf /tmp/pmc_lds
  SQPH_LIB=$L timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES -d /tmp/pmc_lds -o pmc -- python tools/bench_csr.py --steps 2 --check 0 --iters ${ITERS:-200} > /dev/null 2>&1
  f=$(find /tmp/pmc_lds -name "*counter_collection.csv" | head -1)
  python - "$f" "$L" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'csrb' in r.get('Kernel_Name','')]
nd=max(1,len(set(r['Dispatch_Id'] for r in rows)))
acc=collections.defaultdict(float)
for r in rows: acc[r['Counter_Name']]+=float(r['Counter_Value'])
print(sys.argv[2].split('/')[-1], {k:'%.3e'%(v/nd) for k,v in acc.items()}, 'conflict/active = %.3f' % (acc['SQ_LDS_BANK_CONFLICT']/max(1,acc['SQ_LDS_IDX_ACTIVE'])))
PY
done
