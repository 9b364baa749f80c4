"""LDS bank model applied to one ADMM iteration of the fp32-product kernels (float staging), per access class.
usage: lds_model_wgf.py c3|c2 [variant]   variant 0 = the layout before round 4's fix, 1 = after"""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lds_bank_model import cost, ideal
WHICH = sys.argv[1] if len(sys.argv) > 1 else "c3"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if WHICH == "c3": NW, R, C, TR, TC, TW, N, M, STACK = 2, 16, 8, 7, 7, 4, 50, 100, 1
else: NW, R, C, TR, TC, TW, N, M, STACK = 1, 8, 8, 5, 3, 3, 20, 40, 0
TX = TW - 1 if STACK else TW
ev = lambda x: (x + 1) & ~1
r4 = lambda x: (x + 3) & ~3
odd16 = lambda x: x + 4 if (x // 4) % 2 == 0 else x
gs = lambda x: ev(x) if (ev(x) // 2) % 2 else ev(x) + 2
TRp, TWp, TCp, Rp, Cp = gs(TR), gs(TW), ev(TC), R + 2, C + 2
NP, MP, NR = C * TC, R * TR, R * TW
XR = 4 if (R == 8 and TC == 3) else TC
O_ROWV = 0; O_COLV = R * TRp; O_COLV2 = O_COLV + C * TCp; O_WROW = O_COLV2 + C * TCp; O_STAGE = ev(O_WROW + R * TWp); O_STAGE_Y = O_STAGE + C * XR * Rp
O_STX = O_STAGE_Y + max(NR, MP) * Cp  # (any 16-byte aligned place: the model only needs the stride)
TRf, TWf, TCf, Rf, Cf = odd16(r4(TR)), odd16(r4(TW)), r4(TC), r4(R) + 4, r4(C) + 4
SOFF = R * (TR + TX) - NP
HT = None
if V == 0 or R != 16:
    xrow = lambda c, k: TC * c + k
    half = lambda r, c: 8 * (r >> 3)
    xpos = lambda r: r
    pos = lambda r, c: c
if V == 1 and R == 16:
    xrow = lambda c, k: 15 * (c >> 1) + ((7 + ((k + 5) & 7)) if (c & 1) else k)  # WgLayout::xrowf
    xpos = lambda r: r
    pos = lambda r, c: (c + 2 * (r >> 3)) & (C - 1)
    half = lambda r, c: 8 * ((r >> 3) ^ ((((0x0f if (c & 1) else 0x50)) >> (r & 7)) & 1))  # WgLayout::y1_half_f
if V == 1 and R == 8:
    XRf = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    Rf = int(sys.argv[4]) if len(sys.argv) > 4 else Rf
    xrow = lambda c, k: XRf * c + k
tot_i = tot_c = 0
def acc(name, kind, fn, wave, quiet=False):
    global tot_i, tot_c
    addr = []
    for l in range(64):
        t = 64 * wave + l
        a = fn(t % R, t // R, t)
        addr.append(None if a is None else 4 * a)
    if all(a is None for a in addr): return 0
    c, i = cost(kind, addr), ideal(kind)
    if quiet: return c - i
    tot_i += i; tot_c += c
    if c != i: print("  wave %d %-28s %-10s %d cycles (conflict-free %d)" % (wave, name, kind, c, i))
    return c - i
def rd(name, n, fn, wave, quiet=False):
    x = 0
    for k in range((n + 3) // 4): x += acc("%s[%d:%d]" % (name, 4 * k, 4 * k + 4), "read_b128", lambda r, c, t: None if fn(r, c, t) is None else fn(r, c, t) + 4 * k, wave, quiet)
    return x
def sslot(sg):
    s, rr = sg // R, sg % R
    return 2 * O_ROWV + rr * TRf + s if s < TR else 2 * O_WROW + rr * TWf + (s - TR)
if HT == "search":
    best = None
    for h0 in itertools.product((0, 1), repeat=7):
        for h1 in itertools.product((0, 1), repeat=7):
            f = lambda r, c: 12 * ((r >> 3) ^ ((h0 + (0,))[r & 7] if (c & 1) == 0 else (h1 + (0,))[r & 7]))
            x = rd("y1", 8, lambda r, c, t: (2 * O_STAGE + xrow(c, r & 7) * Rf + f(r, c)) if (r & 7) < TC else None, 0, True)
            if best is None or x < best[0]: best = (x, h0, h1)
            if x == 0: break
        if best[0] == 0: break
    print("y1 half assignment: extra cycles %d, h_even %s h_odd %s (masks 0x%02x 0x%02x)" % (best + (sum(b << i for i, b in enumerate(best[1])), sum(b << i for i, b in enumerate(best[2])))))
    h0, h1 = best[1] + (0,), best[2] + (0,)
    half = lambda r, c: 12 * ((r >> 3) ^ (h0[r & 7] if (c & 1) == 0 else h1[r & 7]))
for w in range(NW):
    rd("getf_rowv", TR, lambda r, c, t: 2 * O_ROWV + r * TRf, w)
    rd("ur", TX, lambda r, c, t: 2 * O_WROW + r * TWf, w)
    for k in range(TC): acc("stage1 store k=%d" % k, "write_b32", lambda r, c, t: 2 * O_STAGE + xrow(c, k) * Rf + xpos(r), w)
    if R == 16: rd("y1 reduce", 8, lambda r, c, t: (2 * O_STAGE + xrow(c, r & 7) * Rf + half(r, c)) if (r & 7) < TC else None, w)
    else: rd("y1 reduce", R, lambda r, c, t: (2 * O_STAGE + xrow(c, r) * Rf) if r < TC else None, w)
    acc("putf_colv2", "write_b32", lambda r, c, t: (2 * O_COLV2 + c * TCf + r) if r < TC else None, w)
    rd("getf_colv2", TC, lambda r, c, t: 2 * O_COLV2 + c * TCf, w)
    stx = 2 * O_STAGE_Y + R * TR * Cf if STACK else 2 * O_STX
    for s in range(TR): acc("stage2 store s=%d" % s, "write_b32", lambda r, c, t: 2 * O_STAGE_Y + (R * s + r) * Cf + pos(r, c), w)
    for u in range(TX): acc("stage2 store u=%d" % u, "write_b32", lambda r, c, t: stx + (R * u + r) * Cf + pos(r, c), w)
    rd("owner z~", C, lambda r, c, t: (2 * O_STAGE_Y + t * Cf) if t < M else None, w)
    rd("owner x~", C, lambda r, c, t: ((2 * O_STAGE_Y + (SOFF + t) * Cf) if STACK else (2 * O_STX + t * Cf)) if t < N else None, w)
    if STACK:
        acc("putf_rowv (w)", "write_b32", lambda r, c, t: (2 * O_ROWV + r * TRf + c) if t < M else None, w)
        acc("put u", "write_b32", lambda r, c, t: sslot(SOFF + t) if t < N else None, w)
    else:
        acc("putf_rowv (w)", "write_b32", lambda r, c, t: (2 * O_ROWV + r * TRf + c) if t < MP else None, w)
        acc("putf_wrow (u)", "write_b32", lambda r, c, t: (2 * O_WROW + r * TWf + c) if t < NR else None, w)
print("f32 %s variant %d: array cycles per iteration %d, conflict-free %d, conflict share %.1f %%" % (WHICH, V, tot_c, tot_i, 100.0 * (tot_c - tot_i) / tot_c))
