#!/usr/bin/env python
"""After tools/round_profiles.sh ran on the GPU box: copy the summaries into profiles/, record the PMC traffic of each workload
under the current kernel-source hash, and move the bench lines.   usage: tools/collect_round_profiles.py r02 prof|bench"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, phase = sys.argv[1], sys.argv[2]
if phase == "bench":
    shutil.copy(os.path.join(ROOT, "gpurun_out", tag + "_bench_lines.jsonl"), os.path.join(ROOT, "profiles", tag + "_bench_lines.jsonl"))
    print("bench lines copied")
    sys.exit(0)
KERNELS = {"c3": "wg2_16x8_7x7s_w2", "c3_default": "wg2_16x8_7x7s_w2", "c3_sqp": "wg2_16x8_7x7s_w2", "c2": "wg1_8x8_5x3_w3", "c5": "csb_nb13", "c5_sp": "csb_nb13_sp", "lane": "lane_2x3_exact",
           "c3_full": "wg2_16x8_7x7s_w2", "c3_f32": "wg2_16x8_7x7s_w2_f32", "c2_f32": "wg1_8x8_5x3_w3_f32",
           "c5_default": "csb_nb13", "c5_sqp": "csb_nb13", "c5_sp_default": "csb_nb13_sp", "c5_sp_sqp": "csb_nb13_sp", "c3_fixed98": "wg2_16x8_7x7s_w2"}
W = {  # name -> (n, m, batch, mode)
    "c3": (50, 100, 8192, "fixed"), "c3_default": (50, 100, 8192, "default"), "c3_sqp": (50, 100, 8192, "sqp"),
    "c2": (20, 40, 4096, "fixed"), "c5": (200, 400, 8192, "fixed"), "c5_sp": (200, 400, 8192, "fixed"), "lane": (2, 3, 65536, "fixed"),
    "c3_full": (50, 100, 65536, "fixed"), "c3_f32": (50, 100, 8192, "fixed"), "c2_f32": (20, 40, 4096, "fixed"),
    "c5_default": (200, 400, 8192, "default"), "c5_sqp": (200, 400, 8192, "sqp"), "c5_sp_default": (200, 400, 8192, "default"), "c5_sp_sqp": (200, 400, 8192, "sqp"),
    "c3_fixed98": (50, 100, 8192, "fixed98"),
}
W = {k: v for k, v in W.items() if os.path.isdir(os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, k)))}  # (the workloads this round profiled)
for name, (n, m, batch, mode) in W.items():
    src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, name))
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(ROOT, "profiles", "%s_%s_summary.txt" % (tag, name)))
    for f in os.listdir(os.path.join(src, "trace")) if os.path.isdir(os.path.join(src, "trace")) else []:
        pass
    stats = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(src, "trace")) for f in fs if f.endswith("kernel_stats.csv")]
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, name)))
    kernel = KERNELS[name]
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "record_traffic.py"), os.path.join(src, "summary.txt"), kernel, str(n), str(m), str(batch), mode])
print("ok")
