"""experiment: per-workgroup start / end ticks and placement of a dump written with SQPH_XDBG=<file> (experiment builds)"""
import sys
import numpy as np
for name in sys.argv[1:]:
    a = np.fromfile(name, dtype=np.uint64).reshape(-1, 2, 4)
    hw = a[:, 0, 0]; t0 = a[:, 0, 1].astype(np.int64); t1 = a[:, 0, 2].astype(np.int64)
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    print(name, "workgroups", len(a), "duration ticks mean %.0f min %d max %d" % ((t1 - t0).mean(), (t1 - t0).min(), (t1 - t0).max()))
    for x in range(8):
        sel = xcc == x
        if sel.any():
            print("  xcc %d: %d workgroups, span %d ticks, sum of durations / span = %.2f resident workgroups" % (
                x, sel.sum(), t1[sel].max() - t0[sel].min(), (t1[sel] - t0[sel]).sum() / float(t1[sel].max() - t0[sel].min())))
