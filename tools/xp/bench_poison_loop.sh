#!/bin/bash
# the default bench line's contract test, each run started on LDS filled with the given pattern by another process (GPU box)
PAT=${1:-7FF00000}; N=${2:-6}
for i in $(seq 1 $N); do
  python tools/xp/poison_once.py $PAT > /dev/null
  SQPH_TEST_POISON_LDS=0 timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -q -x -k extra_configs 2>&1 | grep -v "^  n=\|amdgpu.ids" > /tmp/bp_$i.log
  if grep -q "1 passed" /tmp/bp_$i.log; then echo "run $i: passed"; else echo "run $i: FAILED"; grep -n "assert\|Error\|^E " /tmp/bp_$i.log | head -20; fi
done
