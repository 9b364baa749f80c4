"""LDS bank model (lds_bank_model.py) applied to every LDS access of one ADMM iteration of the stacked C3 kernel
(WgKernel<2,16,8,7,7,4>, STACK): predicted array cycles against the conflict-free count, per access class and wave."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lds_bank_model import cost, ideal
R, C, TR, TC, TW, TX = 16, 8, 7, 7, 4, 3
N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 100)
ev = lambda x: (x + 1) & ~1
TRp, TWp, TCp, Rp, Cp = ev(TR) + 2, ev(TW) + 2, ev(TC), R + 2, C + 2
NP = C * TC
O_ROWV = 0; O_COLV = R * TRp; O_COLV2 = O_COLV + C * TCp; O_WROW = O_COLV2 + C * TCp; O_STAGE = ev(O_WROW + R * TWp); O_STAGE_Y = O_STAGE + NP * Rp
SOFF = R * (TR + TX) - NP
tot_i = tot_c = 0
def acc(name, kind, fn, wave):
    global tot_i, tot_c
    addr = []
    for l in range(64):
        t = 64 * wave + l
        a = fn(t % R, t // R, t)
        addr.append(None if a is None else 8 * a)
    if all(a is None for a in addr): return
    c, i = cost(kind, addr), ideal(kind)
    tot_i += i; tot_c += c
    if c != i: print("  wave %d %-28s %-10s %d cycles (conflict-free %d)" % (wave, name, kind, c, i))
def sslot(sg):
    s, rr = sg // R, sg % R
    return O_ROWV + rr * TRp + s if s < TR else O_WROW + rr * TWp + (s - TR)
for w in range(2):
    for k in range(3): acc("get_rowv[%d]" % k, "read_b128", lambda r, c, t: O_ROWV + r * TRp + 2 * k, w)
    acc("get_rowv[6]", "read_b64", lambda r, c, t: O_ROWV + r * TRp + 6, w)
    acc("ur[0:2]", "read_b128", lambda r, c, t: O_WROW + r * TWp, w)
    acc("ur[2]", "read_b64", lambda r, c, t: O_WROW + r * TWp + 2, w)
    for k in range(TC): acc("stage1 store k=%d" % k, "write_b64", lambda r, c, t: O_STAGE + (TC * c + k) * Rp + r, w)
    for i in range(4): acc("y1 reduce read %d" % i, "read_b128", lambda r, c, t: (O_STAGE + (TC * c + (r & 7)) * Rp + 8 * (r >> 3) + 2 * i) if (r & 7) < TC else None, w)
    acc("put_colv2", "write_b64", lambda r, c, t: (O_COLV2 + c * TCp + r) if r < TC else None, w)
    for k in range(3): acc("get_colv2[%d]" % k, "read_b128", lambda r, c, t: O_COLV2 + c * TCp + 2 * k, w)
    acc("get_colv2[6]", "read_b64", lambda r, c, t: O_COLV2 + c * TCp + 6, w)
    for s in range(TR + TX): acc("stage2 store s=%d" % s, "write_b64", lambda r, c, t: O_STAGE_Y + (R * s + r) * Cp + ((c + (r >> 3)) & (C - 1)), w)
    for i in range(4): acc("owner z~ read %d" % i, "read_b128", lambda r, c, t: (O_STAGE_Y + t * Cp + 2 * i) if t < M else None, w)
    for i in range(4): acc("owner x~ read %d" % i, "read_b128", lambda r, c, t: (O_STAGE_Y + (SOFF + t) * Cp + 2 * i) if t < N else None, w)
    acc("put_rowv (w)", "write_b64", lambda r, c, t: (O_ROWV + r * TRp + c) if t < M else None, w)
    acc("put u", "write_b64", lambda r, c, t: sslot(SOFF + t) if t < N else None, w)
print("n=%d m=%d: array cycles per iteration (both waves) %d, conflict-free %d, conflict share %.1f %%" % (N, M, tot_c, tot_i, 100.0 * (tot_c - tot_i) / tot_c))
