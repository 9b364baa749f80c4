#!/bin/bash
# same-box A/B of block-row kernel builds on config 5: tools/xp/ab_csb.sh lib1.so lib2.so ...   (three alternations)
export PYTHONPATH=$PWD
for rep in 1 2 3; do
  for L in "$@"; do
    r=$(SQPH_LIB=$L python tools/bench_csr.py --steps 3 --check 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%.3f' % d['kernel_ms'])")
    echo "$rep $(basename $L) $r"
  done
done
