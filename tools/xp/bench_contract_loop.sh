#!/bin/bash
# the default bench line's contract test N times, failures with their assertion text (GPU box)
N=${1:-10}
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -q -x -k extra_configs 2>&1 | grep -v "^  n=\|amdgpu.ids" > /tmp/bc_$i.log
  if grep -q "1 passed" /tmp/bc_$i.log; then echo "run $i: passed"; else echo "run $i: FAILED"; grep -n "assert\|Error\|^E " /tmp/bc_$i.log | head -30; fi
done
