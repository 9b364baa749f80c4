#!/bin/bash
# kernel time of the config-5 batch against the iteration count (set-up share, cost of the slot placement at max_iter >= 150)
export PYTHONPATH=$PWD
for L in "$@"; do for it in 1 2 50 100 149 150 200; do
  r=$(SQPH_LIB=$L python tools/bench_csr.py --steps 3 --check 0 --iters $it 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%.3f' % d['kernel_ms'])")
  echo "$(basename $L) iters=$it kernel_ms=$r"
done; done
