#!/bin/bash
# run a test binary N times, report non-zero exits with the tail of their output: tools/xp/repeat_bin.sh <binary> <N>
B=$1; N=${2:-50}; bad=0
for i in $(seq 1 $N); do
  $B > /tmp/rb_out.txt 2>&1; rc=$?
  if [ $rc -ne 0 ] || ! grep -q "tests ran\|all passed" /tmp/rb_out.txt; then bad=$((bad+1)); echo "run $i rc=$rc"; tail -15 /tmp/rb_out.txt; fi
done
echo "$B: $bad bad of $N"
