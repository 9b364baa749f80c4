#!/usr/bin/env python
"""Record the PMC traffic of a tools/profile_gpu.sh run in profiles/pmc_traffic.json, keyed by kernel, workload and the hash of
the kernel sources it was measured on (bench.py reports it as roofline.traffic only while that hash matches).
usage: tools/record_traffic.py <profile summary.txt> <kernel> <n> <m> <batch> <mode>"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

summary, kernel, n, m, batch, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
mt = re.search(r"TRAFFIC_BYTES_PER_LAUNCH (\d+)", open(summary).read())
if not mt:
    raise SystemExit("no TRAFFIC_BYTES_PER_LAUNCH line in " + summary)
path = os.path.join(bench.ROOT, "profiles", "pmc_traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {}
key = "%s|n=%d|m=%d|batch=%d|%s|src=%s" % (kernel, n, m, batch, mode, bench.kernel_source_hash())
table[key] = int(mt.group(1))
json.dump(table, open(path, "w"), indent=1, sort_keys=True)
print(key, table[key])
