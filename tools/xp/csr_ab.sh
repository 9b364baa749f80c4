#!/bin/bash
# config-5 bench (kernel ms, parity vs the oracle sample) for experiment builds: tools/xp/csr_ab.sh <lib suffix>...   (libsqp_hip_csr<suffix>.so)
cd $(dirname $0)/../..
for v in "$@"; do
  echo -n "csr$v: "
  SQPH_LIB=$PWD/sqp_solver_amd/lib/libsqp_hip_csr$v.so python bench.py --workload c5 --steps 5 --warmup 1 --cpu-seconds 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'], r['cpu_baseline']['parity_max_rel_err_x'], r['cpu_baseline']['parity_max_rel_err_y'])"
done
