"""Condense a tools/profile_gpu.sh output directory into a short text summary (kept under profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def main(out):
    print("# rocprofv3 summary for", out)
    for f in find(os.path.join(out, "trace"), "*kernel_stats.csv"):
        print("\n## kernel stats (%s)" % os.path.relpath(f, out))
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        for r in rows[:6]:
            d = {k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
            d["Name"] = d["Name"][:100]
            print(d)
    for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        files = find(os.path.join(out, sub), "*counter_collection.csv")
        for f in files:
            acc = defaultdict(lambda: defaultdict(float))
            cnt = defaultdict(set)
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = r.get("Kernel_Name", "?")
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[k].add(r.get("Dispatch_Id"))
            print("\n## %s (%s) — per-dispatch averages" % (sub, os.path.relpath(f, out)))
            for k in acc:
                n = max(len(cnt[k]), 1)
                if "admm" not in k:
                    continue
                print(k[:90], "dispatches", n)
                for c, v in sorted(acc[k].items()):
                    print("   %-28s %.6g" % (c, v / n))


def traffic(out):
    """(2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch of the admm kernel (see profiles/r01_fetch_size_calibration.txt)."""
    vals = {}
    for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for f in find(os.path.join(out, sub), "*counter_collection.csv"):
            tot, ids = 0.0, set()
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if "admm" in r.get("Kernel_Name", "") and r["Counter_Name"] == name:
                        tot += float(r["Counter_Value"])
                        ids.add(r.get("Dispatch_Id"))
            if ids:
                vals[name] = tot / len(ids)
    if len(vals) == 2:
        print("\n## HBM traffic per launch: (2*%.0f + %.0f) KB = %.1f MB" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"], (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / 1e6))
        print("TRAFFIC_BYTES_PER_LAUNCH %d" % int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024))


if __name__ == "__main__":
    main(sys.argv[1])
    traffic(sys.argv[1])
