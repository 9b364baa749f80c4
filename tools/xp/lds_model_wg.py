"""LDS bank model (lds_bank_model.py) applied to every LDS access of one ADMM iteration of a register-tiled workgroup kernel
(admm_wg_kernel.h, fp64 staging): predicted LDS-array cycles against the conflict-free count, per access class and wave.
usage: lds_model_wg.py NW R C TR TC TW stack(0/1) n m [y1fix(0/1)]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lds_bank_model import cost, ideal
NW, R, C, TR, TC, TW, STACK, N, M = [int(x) for x in sys.argv[1:10]]
Y1FIX = int(sys.argv[10]) if len(sys.argv) > 10 else 1
TX = TW - 1 if STACK else TW
ev = lambda x: (x + 1) & ~1
gs = lambda x: ev(x) if (ev(x) // 2) % 2 else ev(x) + 2
TRp, TWp, TCp, Rp, Cp = gs(TR), gs(TW), ev(TC), R + 2, C + 2
NT, MP, NP, NR = R * C, R * TR, C * TC, R * TW
O_ROWV = 0; O_COLV = R * TRp; O_COLV2 = O_COLV + C * TCp; O_WROW = O_COLV2 + C * TCp; O_STAGE = ev(O_WROW + R * TWp)
STAGE_X = NP * Rp; O_STAGE_Y = O_STAGE + STAGE_X; STAGE_Y = max(NR, MP) * Cp
SLOT = 4 if TC <= 4 else 8; SSTR = SLOT * C + 2; WSTR = SLOT * C
O_SJ = MP + NP + 2; O_AS = ev(O_SJ + NP); O_WL = O_AS + R * SSTR; CH = (C + 1) // 2
SETUP = O_WL + CH * TC * WSTR; STAGE = max(STAGE_X + STAGE_Y, MP + 2 * NP + 16 + 8 * NW); O_AS2 = NP * WSTR
O_STX = ev(max(max(O_STAGE + STAGE, SETUP), O_AS2 + R * SSTR - NR * Cp))
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
def rd(name, n, fn, wave):  # wg_read<n>: n/2 b128 + one b64 when n is odd
    for k in range(n // 2): acc("%s[%d:%d]" % (name, 2 * k, 2 * k + 2), "read_b128", lambda r, c, t: None if fn(r, c, t) is None else fn(r, c, t) + 2 * k, wave)
    if n & 1: acc("%s[%d]" % (name, n - 1), "read_b64", lambda r, c, t: None if fn(r, c, t) is None else fn(r, c, t) + n - 1, wave)
def sslot(sg):
    s, rr = sg // R, sg % R
    return O_ROWV + rr * TRp + s if s < TR else O_WROW + rr * TWp + (s - TR)
for w in range(NW):
    rd("get_rowv", TR, lambda r, c, t: O_ROWV + r * TRp, w)
    rd("ur", TX, lambda r, c, t: O_WROW + r * TWp, w)
    for k in range(TC): acc("stage1 store k=%d" % k, "write_b64", lambda r, c, t: O_STAGE + (TC * c + k) * Rp + r, w)
    if R == 16 and TC <= 8:
        hhf = (lambda r, c: ((r >> 3) ^ (r >> 2) ^ c) & 1) if Y1FIX else (lambda r, c: r >> 3)
        rd("y1 reduce", 8, lambda r, c, t: (O_STAGE + (TC * c + (r & 7)) * Rp + 8 * hhf(r, c)) if (r & 7) < TC else None, w)
    else:
        rd("y1 reduce", R, lambda r, c, t: (O_STAGE + (TC * c + r) * Rp) if r < TC else None, w)
    acc("put_colv2", "write_b64", lambda r, c, t: (O_COLV2 + c * TCp + r) if r < TC else None, w)
    rd("get_colv2", TC, lambda r, c, t: O_COLV2 + c * TCp, w)
    pos = lambda r, c: (c + (r >> 3)) & (C - 1)
    for s in range(TR): acc("stage2 store s=%d" % s, "write_b64", lambda r, c, t: O_STAGE_Y + (R * s + r) * Cp + pos(r, c), w)
    stx = O_STAGE_Y + R * TR * Cp if STACK else O_STX
    for u in range(TX): acc("stage2 store u=%d" % u, "write_b64", lambda r, c, t: stx + (R * u + r) * Cp + pos(r, c), w)
    rd("owner z~", C, lambda r, c, t: (O_STAGE_Y + t * Cp) if t < M else None, w)
    rd("owner x~", C, lambda r, c, t: ((O_STAGE_Y + (SOFF + t) * Cp) if STACK else (O_STX + t * Cp)) if t < N else None, w)
    if STACK:
        acc("put_rowv (w)", "write_b64", lambda r, c, t: (O_ROWV + r * TRp + c) if t < M else None, w)
        acc("put u", "write_b64", lambda r, c, t: sslot(SOFF + t) if t < N else None, w)
    else:
        acc("put_rowv (w)", "write_b64", lambda r, c, t: (O_ROWV + r * TRp + c) if t < MP else None, w)
        acc("put_wrow (u)", "write_b64", lambda r, c, t: (O_WROW + r * TWp + c) if t < NR else None, w)
print("(%d,%d,%d,%d,%d,%d)%s n=%d m=%d: array cycles per iteration %d, conflict-free %d, conflict share %.1f %%" % (NW, R, C, TR, TC, TW, " stacked" if STACK else "", N, M, tot_c, tot_i, 100.0 * (tot_c - tot_i) / tot_c))
