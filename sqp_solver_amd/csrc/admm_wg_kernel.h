// Workgroup-tiled ADMM kernel: NW wavefronts (64*NW lanes) per QP, matrices in VGPRs, vectors in LDS.
//
// Lanes form an R x C grid (t = c*R + r, R*C = 64*NW).  Lane (r,c) keeps two register tiles for the
// whole solve:
//     at[s][k] = A[R*s + r][C*k + c]      s < TR, k < TC      (rows cyclic over r, columns cyclic over c)
//     wt[u][k] = W[R*u + r][C*k + c]      u < TW, k < TC      (W lower triangular, S^-1 = W'W)
// Both index directions are cyclic, so the zeros of a triangular matrix fall on the SAME tile positions in every lane: entry (u, k)
// of the W tile is structurally zero when C k > R u + R - 1, entry (u, k) of the W' tile when C k + C - 1 < R u — those multiply-adds
// are never issued (the n^2 of the reference's two triangular solves, src/qp.cpp:90, instead of 2 n^2), the factorisation's products
// run over the non-zero half, and both wavefronts of a QP carry the same share of every triangular phase.  A column j = C k + c has
// the SLOT index TC c + k (its position among the tile columns of all lanes): staging areas and gather vectors are addressed by slot.
// so that A x / A'w and W b / W'y all run out of the same registers.  Vectors never live in more than
// one lane's registers: a product is   gather operand from LDS -> FMAs on the tile -> write partial sums
// to an LDS staging area -> owners (lane t owns element t) reduce them.   Measured on gfx950
// (tools/ubench): one wave issues one VALU instruction every ~5 cycles whatever the opcode, two waves on
// a SIMD bring that to ~3.5, four to ~2.5 — so the kernel is organised to (a) keep the instruction count
// per iteration close to the FMA count (LDS moves 16 B per lane per instruction, whole-register
// butterflies cost ~45 VALU instructions per 8-value exchange of doubles) and (b) fit two or more waves
// per SIMD (a 50x100 QP is split over two waves: 98 + 56 VGPRs of tiles per lane).
//
// Numerics: identical formulas to admm_generic.h (reference src/qp.cpp:84-144 on the Schur-ordered
// system, factor W of admm_generic.h:factor_schur), fp64 arithmetic, TIN inputs.
#pragma once
#include "block_ops.h"
#include "kargs.h"
#include "wave_ops.h"

#ifndef SQPH_OPAQUE_S
#ifdef SQPH_SIM
#define SQPH_OPAQUE_S(x) (void)0
#define SQPH_OPAQUE_V(x) (void)0
#else
#define SQPH_OPAQUE_S(x) asm volatile("" : "+s"(x))
#define SQPH_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif
#endif

// -DSQPH_SETUP_TIMING (debug builds, tools/setup_timing.py): s_memtime ticks of the set-up phases of wave 0, returned in x[0..8)
#ifdef SQPH_SETUP_TIMING
#define SQPH_STICK(k) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); sqph_stk[k] += tn_ - sqph_stp; sqph_stp = tn_; }
#define SQPH_STICK_ARGS , unsigned long long (&sqph_stk)[8], unsigned long long &sqph_stp
#define SQPH_STICK_PASS , sqph_stk, sqph_stp
#else
#define SQPH_STICK(k)
#define SQPH_STICK_ARGS
#define SQPH_STICK_PASS
#endif

// unroll factor of the residual check's streaming loops over A and P (global loads in flight per lane = factor x TR resp. TC)
#ifndef SQPH_CHECK_UNROLL
#define SQPH_CHECK_UNROLL 1
#endif

// Measurement only (tests/test_f32_tiles.py): what fp32 STORAGE of the B / W' register tiles with fp64
// accumulation would do to the results — the tiles are rounded through float once, when they are built.  Never in the shipped
// library: -DSQPH_F32_TILE_STORAGE experiment builds, or the emulator's run-time switch.
#if defined(SQPH_SIM)
#define SQPH_TILE_QUANT(x) (::sqph_sim::tile_quant() ? (double)(float)(x) : (double)(x))
#elif defined(SQPH_F32_TILE_STORAGE)
#define SQPH_TILE_QUANT(x) ((double)(float)(x))
#else
#define SQPH_TILE_QUANT(x) (x)
#endif

namespace sqph {

typedef double sqph_v2 __attribute__((vector_size(16)));

__device__ __forceinline__ double wg_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1/d for a positive finite normal d: v_rcp_f64 refined by two Newton steps (the sequence inside the compiler's IEEE division without
// its scaling and fix-up instructions); within 1 ulp of the quotient
__device__ __forceinline__ double fast_rcp(double d) {
#ifdef SQPH_SIM
    return 1.0 / d;
#else
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
#endif
}

// N contiguous doubles from a 16-byte aligned LDS address (N rounded up to even is read)
template <int N>
__device__ __forceinline__ void wg_read(const double *p, double (&v)[N]) {
    const sqph_v2 *q = reinterpret_cast<const sqph_v2 *>(__builtin_assume_aligned(p, 16));
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
        const sqph_v2 t = q[k];
        v[2 * k] = t[0];
        v[2 * k + 1] = t[1];
    }
    if constexpr (N & 1) v[N - 1] = p[N - 1];
}
template <int N>
__device__ __forceinline__ double wg_sum(const double *p) {
    double v[N];
    wg_read<N>(p, v);
    if constexpr (N >= 8 && (N % 4) == 0) {
        double s0 = v[0], s1 = v[1], s2 = v[2], s3 = v[3];  // four independent chains
#pragma unroll
        for (int k = 4; k < N; k += 4) {
            s0 += v[k];
            s1 += v[k + 1];
            s2 += v[k + 2];
            s3 += v[k + 3];
        }
        return (s0 + s1) + (s2 + s3);
    } else {
        double s0 = v[0], s1 = N > 1 ? v[1] : 0.0;
#pragma unroll
        for (int k = 2; k + 1 < N; k += 2) {
            s0 += v[k];
            s1 += v[k + 1];
        }
        if constexpr ((N & 1) && N > 1) s0 += v[N - 1];
        return s0 + s1;
    }
}

// ---- fp32 products (SQPH_FLAG_F32_ARITH on a QPSolver<float>): the B / W' tiles, the operand vectors and the partial sums of the two
// iteration stages are single precision, two columns per register pair (v_pk_fma_f32: two multiply-adds per lane and instruction);
// the factorisation that builds the tiles and the iterates x, z, y stay double.
typedef float sqph_f2 __attribute__((vector_size(8)));
typedef float sqph_f4 __attribute__((vector_size(16)));
__device__ __forceinline__ sqph_f2 wgf_fma2(sqph_f2 a, sqph_f2 b, sqph_f2 c) {
#ifdef SQPH_SIM
    sqph_f2 r;
    r[0] = __builtin_fmaf(a[0], b[0], c[0]);
    r[1] = __builtin_fmaf(a[1], b[1], c[1]);
    return r;
#else
    return __builtin_elementwise_fma(a, b, c);
#endif
}
// N floats from a 16-byte aligned LDS address (N rounded up to a multiple of four is read)
template <int N>
__device__ __forceinline__ void wgf_read(const float *p, float (&v)[N]) {
    const sqph_f4 *q = reinterpret_cast<const sqph_f4 *>(__builtin_assume_aligned(p, 16));
#pragma unroll
    for (int k = 0; k < (N + 3) / 4; k++) {
        const sqph_f4 t = q[k];
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (4 * k + e < N) v[4 * k + e] = t[e];
    }
}
template <int N>
__device__ __forceinline__ float wgf_sum(const float *p) {
    float v[N];
    wgf_read<N>(p, v);
    if constexpr (N >= 8 && (N % 4) == 0) {
        float s0 = v[0], s1 = v[1], s2 = v[2], s3 = v[3];
#pragma unroll
        for (int k = 4; k < N; k += 4) {
            s0 += v[k];
            s1 += v[k + 1];
            s2 += v[k + 2];
            s3 += v[k + 3];
        }
        return (s0 + s1) + (s2 + s3);
    } else {
        float s0 = v[0], s1 = N > 1 ? v[1] : 0.0f;
#pragma unroll
        for (int k = 2; k + 1 < N; k += 2) {
            s0 += v[k];
            s1 += v[k + 1];
        }
        if constexpr ((N & 1) && N > 1) s0 += v[N - 1];
        return s0 + s1;
    }
}

template <int NW, int R, int C, int TR, int TC, int TW>
struct WgLayout {
    // NW >= 1: the R x C lane grid is a workgroup of NW wavefronts.  NW == 0: a 4 x 4 grid of 16 lanes — four
    // independent QPs share one wavefront (small problems), see run_group().
    static_assert(NW == 0 ? (R * C == 16 || R * C == 32 || R * C == 64) : R * C == 64 * NW, "lane grid must cover the workgroup");
    static constexpr int NT = R * C;
    static constexpr int MP = R * TR;  // padded m
    static constexpr int NP = C * TC;  // padded n (columns)
    static constexpr int NR = R * TW;  // padded n (rows of W)
    static_assert(NW == 0 || (NP <= NT && NR >= NP && NR <= NT && MP <= NT), "owners: lane t owns n-element t and m-element t");
    static constexpr int ev(int x) { return (x + 1) & ~1; }
    // gather strides per r: an odd number of 16-byte units, so that the ds_read_b128 of the R lanes of a column group fall on distinct bank
    // quads (an even number puts lanes r and r + 8 — or r + 4 with R = 8 — on the same ones: C2's TR = 5 at stride 8 read every operand
    // at half rate, tools/xp/lds_model_wg.py)
    static constexpr int gstride(int x) { return (ev(x) / 2) % 2 ? ev(x) : ev(x) + 2; }
    static constexpr int TRp = gstride(TR);  // row-gather stride per r    (w, y)
    static constexpr int TWp = gstride(TW);  // W-row gather stride per r  (y1)
    static constexpr int TCp = ev(TC);      // column-gather stride per c (b, x~, x)
    static constexpr int Rp = R + 2;        // staging stride per output for reductions over r
    static constexpr int Cp = C + 2;        // staging stride per output for reductions over c
    // LDS map (doubles)
    static constexpr int O_ROWV = 0;                       // [R][TRp]  m-vector in row-gather order      (w ; y at checks)
    static constexpr int O_COLV = O_ROWV + R * TRp;        // [C][TCp]  n-vector in column-gather order   (u ; x at checks)
    static constexpr int O_COLV2 = O_COLV + C * TCp;       // [C][TCp]  second n-vector, column order     (y1)
    static constexpr int O_WROW = O_COLV2 + C * TCp;       // [R][TWp]  n-vector in W-row-gather order    (y1)
    static constexpr int O_STAGE = ev(O_WROW + R * TWp);   // staging X: partials reduced over r  [NP][Rp]
    static constexpr int mx(int a, int b) { return a > b ? a : b; }
    // rows of staging X per column group: TC, or 4 where R = 8 and TC = 3 — a ds_write_b64 is served 16 lanes at a time, two column
    // groups of 8 lanes then, whose rows must start 16 banks apart: 4 rows of Rp = 10 doubles do, 3 rows do not (C2: every stage-1 store
    // was two-way conflicted)
    static constexpr int XR = (R == 8 && TC == 3) ? 4 : TC;
    static constexpr int xrow(int c, int k) { return XR * c + k; }          // row of tile column k of lane group c
    static constexpr int xslot(int j) { return XR * (j % C) + j / C; }     // row of matrix column j
    static constexpr int STAGE_X = C * XR * Rp;
    static constexpr int O_STAGE_Y = O_STAGE + STAGE_X;    // staging Y: partials reduced over c  [max(NR,MP)][Cp]
    static constexpr int STAGE_Y = mx(NR, MP) * Cp;
    // set-up scratch, aliasing the staging areas:  rho[MP] | rowbuf[NP+2] | sj[NP] | As[R][SSTR] | Wl[NP][SSTR]
    // As = one block of R rows of A, Wl = W transposed (Wl[j][slot(i')] = W[i'][j]); column groups are
    // padded to SLOT = 8 (4 when TC <= 4) entries (slot(j) = SLOT*(j%C) + j/C) so a lane's TC tile columns are one aligned read.
    // (the set-up scratch starts at offset 0: no vector is live in LDS while a factor is being built)
    static constexpr int SLOT = TC <= 4 ? 4 : 8;
    static constexpr int SSTR = SLOT * C + 2;  // As row stride (padded: the R rows are written by different lanes)
    static constexpr int WSTR = SLOT * C;      // Wl row stride
    static constexpr int O_RHO = 0;
    static constexpr int O_ROWBUF = O_RHO + MP;
    static constexpr int O_SJ = O_ROWBUF + NP + 2;
    static constexpr int O_AS = ev(O_SJ + NP);
    static constexpr int O_WL = O_AS + R * SSTR;
    static constexpr int CH = (C + 1) / 2;  // W is staged half of its columns (CH column groups) at a time
    // P staged in LDS behind As while S = A'RA is accumulated (LDS-DMA issued ahead of the A tile's loads, see stage_P_async)
    static constexpr int O_PST = ev(O_AS + R * SSTR);
    static constexpr int SETUP = O_AS + R * SSTR;  // (rounds 1-3 staged W behind As during the factorisation; the lower-triangle factor of round 4 does not)
    static constexpr int O_RED = O_STAGE + MP + 2 * NP + 16;  // workgroup max scratch (residual checks)
    static constexpr int STAGE = mx(STAGE_X + STAGE_Y, MP + 2 * NP + 16 + 8 * NW);
    // x~ partials of stage 2 ([R u + r][Cp]): a region of their own — stage 2 of a fast wave must not overwrite the stage-1
    // partials a slower wave of the workgroup is still reducing (there is no workgroup barrier between the two stages)
    // (the aliased region below the owners' constants is also made large enough to stage all of W for the W -> W' transposition)
    // The staged transposed copy of W (build_B_inplace, load_vt_lds): row j holds column j of W.  Padded form: NP rows of WSTR.  The
    // 16 x 16 grids with 7 x 7 tiles (n <= 112) pack it — row j = C kj + cj keeps the entries W[C k + c][j] of the tile columns k >= kj
    // only (the others are zero in every lane), k-major: wf_row(j) + C (k - kj) + c — 7,168 instead of 14,336 doubles, which is what
    // lets two of these workgroups share a CU (they ran one wave per SIMD until round 4).  (Since the end of round 4 these grids take
    // the MFMA set-up — MSX below — whose blocks serve build_B and the W' tile; this form is what they fall back to without it.)
    static constexpr bool WPACK = NW > 0 && R == 16 && C == 16 && TC == 7 && TW == 7;
    static constexpr int wf_len(int kj) { return C * (TC - kj); }                                   // entries of a row of column block kj
    static constexpr int wf_blk(int kj) { return C * C * (kj * TC - kj * (kj - 1) / 2); }           // first row of column block kj
    static constexpr int WF_SIZE = WPACK ? wf_blk(TC) : NP * WSTR;
    static constexpr int O_AS2 = WF_SIZE;  // build_B: one block of R rows of A behind the staged W
    // The 16 x 16 / 7 x 7 grids with m <= 128 run their set-up on the MFMA (admm_wg_msetup.h) with seven block columns in unpadded,
    // swizzled 16 x 16 blocks: rho / diagonal / flags, 7 staging blocks, 28 lower blocks, 2 diagonal-block buffers
#ifdef SQPH_NO_MSX  // A/B experiment builds: these grids with the scalar set-up and the packed staged copy of W (WPACK)
    static constexpr bool MSX = false;
#else
    static constexpr bool MSX = NW == 4 && R == 16 && C == 16 && TC == 7 && TW == 7 && TR <= 8;
#endif
    static constexpr int MSX_END = ev(MP + 16 * 7 + 8) + (7 + 28 + 2) * 256;
    // the 32 x 16 grid of eight waves (m <= 224, n <= 112) takes the MFMA set-up too: its LDS has the room (one workgroup per CU anyway);
    // so do the tall grids (32 x 8 with four waves, 64 x 8 with eight: n <= 56), whose blocks may need a little more than the
    // scalar set-up's scratch (MST_END enters O_QV below)
#ifdef SQPH_NO_MST  // A/B experiment builds
    static constexpr bool MST = false;
#else
    // (not the 64 x 8 / 7 x 7 grid: with the MFMA code its no-check loop spills, 2,048 x (50,400) 3.61 -> 5.52 ms)
    static constexpr bool MST = (NW == 4 && R == 32 && C == 8) || (NW == 8 && R == 64 && C == 8 && TC <= 4);
#endif
    static constexpr bool MSR = (NW == 8 && R == 32 && C == 16 && TC == 7 && TW == 4) || MST;
    static constexpr int MST_NB = (NP + 15) / 16;
    static constexpr int MST_END = ev(MP + 16 * MST_NB + 8 + 2) + (2 * MST_NB + MST_NB * (MST_NB + 1) / 2 + 2) * (MST_NB > 4 ? 256 : 272);
    // (the set-up scratch and build_B's staging end below the owners' constants; the x~ staging region may lie inside them)
    static constexpr int O_STX = ev(mx(O_STAGE + STAGE, (MSX ? MSX_END : mx(mx(SETUP, O_AS2 + R * SSTR), MST ? MST_END : 0)) - NR * Cp));
    // per-element constants of the owners (q, l, u): LDS instead of VGPRs, after everything the set-up may alias
    static constexpr int O_QV = ev(O_STX + NR * Cp);
    static constexpr int O_LOV = O_QV + NP;
    static constexpr int O_UPV = O_LOV + MP;
    static constexpr int O_RINV = O_UPV + MP;  // 1/rho of the owned constraint (changes only at a refactorisation)
    static constexpr int TOTAL = ev(O_RINV + MP);
    static_assert(MSX || O_AS2 + R * SSTR <= O_QV, "build_B stages all of W and a block of A rows in [0, O_QV)");
    // the workgroup kernels whose scratch has room for an n x n block of doubles take P through LDS
    static constexpr bool P_STAGED = NW > 0 && O_PST + NP * NP <= O_QV;
    // fp32-product variant: the same regions viewed as floats (float offset = 2 x the double offset), rows padded to 16 bytes
    static constexpr int r4(int x) { return (x + 3) & ~3; }
    // gather strides (floats).  The R lanes of a column group read their rows with one ds_read_b128 each: a stride of an EVEN number of
    // 16-byte units puts lanes r and r + 8 on the same banks (TR = 7: 8 floats, two-way on every operand read), an odd number spreads
    // the 16 lanes over all 64
    static constexpr int odd16(int x) { return ((x / 4) % 2 == 0) ? x + 4 : x; }
    static constexpr int TRf = odd16(r4(TR)), TWf = odd16(r4(TW)), TCf = r4(TC);
    static constexpr int Rf = r4(R) + 4, Cf = r4(C) + 4;           // staging strides (floats)
    static constexpr int TC2 = (TC + 1) / 2;                       // column pairs of a tile row
    // rows of the float staging X.  The C3 grid (R = 16, C = 8, TC = 7): a ds_write_b32 is served 32 lanes at a time — the 16 lanes of an
    // even column group and of the odd one next to it, whose rows must start 16 banks apart (rows 4 mod 8 apart at Rf = 20) — and the two
    // ds_read_b128 of the y1 reduction 16 lanes at a time, 14 of them active, which need 14 different bank quads: the rows of an odd group
    // follow the even group's seven in the order 3 4 5 6 - 0 1 2 (one row unused), and which lane of a pair sums which half of a row is
    // y1_half_f().  Before: every store of the two stages and every read of the reduction two-way conflicted, 30 % of the kernel's LDS
    // cycles (tools/xp/lds_model_wgf.py)
    static constexpr bool XF_ROT = R == 16 && C == 8 && TC == 7;
    static constexpr int xrowf(int c, int k) { return XF_ROT ? 15 * (c >> 1) + ((c & 1) ? 7 + ((k + 5) & 7) : k) : TC * c + k; }
    static constexpr int y1_half_f(int r, int c) { return XF_ROT ? ((r >> 3) ^ ((((c & 1) ? 0x0f : 0x50) >> (r & 7)) & 1)) : (r >> 3); }
    static constexpr int XROWSF = XF_ROT ? 15 * (C / 2) : NP;
    static constexpr bool F32_FITS = R * TRf <= 2 * R * TRp && R * TWf <= 2 * R * TWp && C * TCf <= 2 * C * TCp && XROWSF * Rf <= 2 * STAGE_X &&
                                     mx(NR, MP) * Cf <= 2 * STAGE_Y && NR * Cf <= 2 * NR * Cp;
    // column distribution (cyclic over c): tile column k of lane group c is matrix column col(c, k); cslot(j) = its slot index
    static constexpr int col(int c, int k) { return C * k + c; }
    static constexpr int cslot(int j) { return TC * (j % C) + j / C; }
    static constexpr int slot(int j) { return SLOT * (j % C) + j / C; }  // the same with column groups padded to SLOT entries
    // structural zeros shared by all lanes: W tile entry (u, k) = W[R u + r][C k + c], W' tile entry (u, k) = W[C k + c][R u + r]
    static constexpr bool wt_zero(int u, int k) { return C * k > R * u + R - 1; }
    static constexpr bool vt_zero(int u, int k) { return C * k + C - 1 < R * u; }
};

}  // namespace sqph
#include "admm_wg_msetup.h"
namespace sqph {

template <typename TIN, int NW, int R, int C, int TR, int TC, int TW>
struct WgKernel {
    using T = double;
    using L = WgLayout<NW, R, C, TR, TC, TW>;
    // the matrix phases of the set-up on the f64 MFMA where the lane grid and the LDS budget allow (admm_wg_msetup.h)
    using MS = MSetup<NW, R, C, TR, TC, TW>;
#ifdef SQPH_NO_MSETUP  // A/B experiment builds only
    static constexpr bool MSET = false;
#else
    static constexpr bool MSET = MS::ENABLED;
#endif
    static constexpr int NT = L::NT;

    // barrier of the lanes that work on one QP: the workgroup, or (NW == 0) a 16-lane group of a wavefront, whose LDS
    // operations execute in program order anyway — only the compiler has to be kept from reordering them
    // ordering among the lanes of ONE wavefront (the R lanes of a column group always sit in one): LDS operations of a
    // wave execute in program order, so this only pins the compiler
    static __device__ __forceinline__ void wave_sync() {
#ifdef SQPH_SIM
        if constexpr (NW == 0 && NT < 64) wsync();
        else ::sqph_sim::yield_wait(2);
#else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
    }
    static __device__ __forceinline__ void wsync() {
        if constexpr (NW > 0) {
            __syncthreads();
        } else {
#ifdef SQPH_SIM
            if constexpr (NT == 16) ::sqph_sim::group_sync<16>();
            else if constexpr (NT == 32) ::sqph_sim::group_sync<32>();
            else ::sqph_sim::yield_wait(2);
#else
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
        }
    }

    // ------------------------------------------------------------------ gathers (operand from LDS)
    static __device__ __forceinline__ void put_rowv(T *lds, int r, int c, T v) { lds[L::O_ROWV + r * L::TRp + c] = v; }
    static __device__ __forceinline__ void get_rowv(const T *lds, int r, T (&w)[TR]) { wg_read<TR>(lds + L::O_ROWV + r * L::TRp, w); }
    static __device__ __forceinline__ void put_colv(T *lds, int j, T v) { lds[L::O_COLV + (j % C) * L::TCp + (j / C)] = v; }
    static __device__ __forceinline__ void get_colv(const T *lds, int c, T (&x)[TC]) { wg_read<TC>(lds + L::O_COLV + c * L::TCp, x); }
    static __device__ __forceinline__ void put_colv2(T *lds, int j, T v) { lds[L::O_COLV2 + (j % C) * L::TCp + (j / C)] = v; }
    static __device__ __forceinline__ void get_colv2(const T *lds, int c, T (&x)[TC]) { wg_read<TC>(lds + L::O_COLV2 + c * L::TCp, x); }
    static __device__ __forceinline__ void put_wrow(T *lds, int r, int c, T v) { lds[L::O_WROW + r * L::TWp + c] = v; }
    static __device__ __forceinline__ void get_wrow(const T *lds, int r, T (&y)[TW]) { wg_read<TW>(lds + L::O_WROW + r * L::TWp, y); }

    // ------------------------------------------------------------------ tile products -> staging
    // A' w : partial over my rows for my TC columns; staged for a reduction over r
    static __device__ __forceinline__ void stage_AT(const T (&at)[TR][TC], const T (&w)[TR], T *lds, int r, int c) {
        T pb[TC];
#pragma unroll
        for (int k = 0; k < TC; k++) pb[k] = 0;
#pragma unroll
        for (int s = 0; s < TR; s++)
#pragma unroll
            for (int k = 0; k < TC; k++) pb[k] = wg_fma(at[s][k], w[s], pb[k]);
        T *st = lds + L::O_STAGE;
#pragma unroll
        for (int k = 0; k < TC; k++) st[L::xrow(c, k) * L::Rp + r] = pb[k];
    }
    // A x : partial over my columns for my TR rows; staged for a reduction over c
    static __device__ __forceinline__ void stage_A(const T (&at)[TR][TC], const T (&x)[TC], T *lds, int r, int c) {
        T pz[TR];
#pragma unroll
        for (int s = 0; s < TR; s++) pz[s] = 0;
#pragma unroll
        for (int k = 0; k < TC; k++)
#pragma unroll
            for (int s = 0; s < TR; s++) pz[s] = wg_fma(at[s][k], x[k], pz[s]);
        T *st = lds + L::O_STAGE_Y;
#pragma unroll
        for (int s = 0; s < TR; s++) st[(R * s + r) * L::Cp + c] = pz[s];
    }
    // M x with the square tile (rows R*u + r): staged for a reduction over c
    static __device__ __forceinline__ void stage_W(const T (&wt)[TW][TC], const T (&x)[TC], T *lds, int r, int c) {
        T py[TW];
#pragma unroll
        for (int u = 0; u < TW; u++) {
            T acc = 0;
#pragma unroll
            for (int k = 0; k < TC; k++) acc = wg_fma(wt[u][k], x[k], acc);
            py[u] = acc;
        }
        T *st = lds + L::O_STAGE_Y;
#pragma unroll
        for (int u = 0; u < TW; u++) st[(R * u + r) * L::Cp + c] = py[u];
    }
    // M' y with the square tile: staged for a reduction over r
    static __device__ __forceinline__ void stage_WT(const T (&wt)[TW][TC], const T (&y)[TW], T *lds, int r, int c) {
        T px[TC];
#pragma unroll
        for (int k = 0; k < TC; k++) px[k] = 0;
#pragma unroll
        for (int u = 0; u < TW; u++)
#pragma unroll
            for (int k = 0; k < TC; k++) px[k] = wg_fma(wt[u][k], y[u], px[k]);
        T *st = lds + L::O_STAGE;
#pragma unroll
        for (int k = 0; k < TC; k++) st[L::xrow(c, k) * L::Rp + r] = px[k];
    }
    // The iteration's two stages with the W' tile (vt[u][k] = W[TC c + k][R u + r]): both products of a stage share the
    // reduction direction, so stage 1 stages ONE set of partial sums (W u and B'w accumulate into the same registers)
    // and every owner sums 16 + 8 + 8 instead of 16 + 8 + 16 + 8 partials per iteration; 18 instead of 25 LDS stores.
    //   stage 1:  y1[TC c + k] += sum_s B[R s + r][.] w[R s + r] + sum_u W[.][R u + r] u[R u + r]      (reduced over r)
    // entry (u, k) of the second tile that no lane ever holds a non-zero in (W' is upper triangular, both tile directions cyclic)
    template <int TX, bool STACK>
    static constexpr bool xzero(int u, int k) {
        return STACK ? (C * k + C - 1 < R * (TR + u) - (R * (TR + TX) - L::NP)) : L::vt_zero(u, k);
    }
    template <int TX, bool STACK = false>
    static __device__ __forceinline__ void stage1(const T (&bt)[TR][TC], const T (&vt)[TX][TC], const T (&w)[TR], const T (&ur)[TX], T *lds,
                                                  int r, int c) {
        T pb[TC];
#pragma unroll
        for (int k = 0; k < TC; k++) pb[k] = 0;
#pragma unroll
        for (int s = 0; s < TR; s++)
#pragma unroll
            for (int k = 0; k < TC; k++) pb[k] = wg_fma(bt[s][k], w[s], pb[k]);
#pragma unroll
        for (int u = 0; u < TX; u++)
#pragma unroll
            for (int k = 0; k < TC; k++)
                if (!xzero<TX, STACK>(u, k)) pb[k] = wg_fma(vt[u][k], ur[u], pb[k]);
        T *st = lds + L::O_STAGE;
#pragma unroll
        for (int k = 0; k < TC; k++) st[L::xrow(c, k) * L::Rp + r] = pb[k];
    }
    //   stage 2:  z~[R s + r] = sum_k B[.][TC c + k] y1[TC c + k] ,  x~[R u + r] = sum_k W[TC c + k][.] y1[TC c + k]   (both over c)
    // STACK: the x~ partial sums continue the z~ array (one stacked (m+n)-vector of outputs, see run())
    template <int TX, bool STACK = false>
    static __device__ __forceinline__ void stage2(const T (&bt)[TR][TC], const T (&vt)[TX][TC], const T (&y1)[TC], T *lds, int r, int c) {
        T pz[TR];
#pragma unroll
        for (int s = 0; s < TR; s++) pz[s] = 0;
#pragma unroll
        for (int k = 0; k < TC; k++)
#pragma unroll
            for (int s = 0; s < TR; s++) pz[s] = wg_fma(bt[s][k], y1[k], pz[s]);
        T *sty = lds + L::O_STAGE_Y;
        // position inside a row rotated by r/8: with the even row stride lanes r and r+8 of a store would share banks;
        // the owner sums the whole row, so the order is free
        const int pos = (c + (r >> 3)) & (C - 1);
#pragma unroll
        for (int s = 0; s < TR; s++) sty[(R * s + r) * L::Cp + pos] = pz[s];
        T *stx = lds + (STACK ? L::O_STAGE_Y + R * TR * L::Cp : L::O_STX);
#pragma unroll
        for (int u = 0; u < TX; u++) {
            T acc = 0;
#pragma unroll
            for (int k = 0; k < TC; k++)
                if (!xzero<TX, STACK>(u, k)) acc = wg_fma(vt[u][k], y1[k], acc);
            stx[(R * u + r) * L::Cp + pos] = acc;
        }
    }
    static __device__ __forceinline__ T reduce_xt(const T *lds, int i) { return wg_sum<C>(lds + L::O_STX + i * L::Cp); }

    // ------------------------------------------------------------------ fp32-product variant of the two stages (see wgf_fma2)
    static __device__ __forceinline__ void putf_rowv(float *lf, int r, int c, float v) { lf[2 * L::O_ROWV + r * L::TRf + c] = v; }
    static __device__ __forceinline__ void getf_rowv(const float *lf, int r, float (&w)[TR]) { wgf_read<TR>(lf + 2 * L::O_ROWV + r * L::TRf, w); }
    static __device__ __forceinline__ void putf_wrow(float *lf, int r, int c, float v) { lf[2 * L::O_WROW + r * L::TWf + c] = v; }
    static __device__ __forceinline__ void getf_wrow(const float *lf, int r, float (&y)[TW]) { wgf_read<TW>(lf + 2 * L::O_WROW + r * L::TWf, y); }
    static __device__ __forceinline__ void putf_colv2(float *lf, int j, float v) { lf[2 * L::O_COLV2 + (j % C) * L::TCf + (j / C)] = v; }
    static __device__ __forceinline__ void getf_colv2(const float *lf, int c, float (&x)[TC]) { wgf_read<TC>(lf + 2 * L::O_COLV2 + c * L::TCf, x); }
    // double tiles -> float pairs (column 2 kp | 2 kp + 1; the odd tail column is paired with a zero)
    template <int NRW>
    static __device__ __forceinline__ void tile_to_f32(const T (&src)[NRW][TC], sqph_f2 (&dst)[NRW][L::TC2]) {
#pragma unroll
        for (int s = 0; s < NRW; s++)
#pragma unroll
            for (int kp = 0; kp < L::TC2; kp++) {
                dst[s][kp][0] = (float)src[s][2 * kp];
                dst[s][kp][1] = (2 * kp + 1 < TC) ? (float)src[s][2 * kp + 1 < TC ? 2 * kp + 1 : 0] : 0.0f;
            }
    }
    // column pair kp of the second tile's row u is zero in every lane (the zeros of a tile row are its first columns)
    template <int TX, bool STACK>
    static constexpr bool xzero2(int u, int kp) { return xzero<TX, STACK>(u, 2 * kp + 1 < TC ? 2 * kp + 1 : 2 * kp); }
    template <int TX, bool STACK = false>
    static __device__ __forceinline__ void stage1_f(const sqph_f2 (&bt)[TR][L::TC2], const sqph_f2 (&vt)[TX][L::TC2], const float (&w)[TR],
                                                    const float (&ur)[TX], float *lf, int r, int c) {
        sqph_f2 pb[L::TC2];
#pragma unroll
        for (int kp = 0; kp < L::TC2; kp++) pb[kp] = sqph_f2{0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < TR; s++) {
            const sqph_f2 ws = {w[s], w[s]};
#pragma unroll
            for (int kp = 0; kp < L::TC2; kp++) pb[kp] = wgf_fma2(bt[s][kp], ws, pb[kp]);
        }
#pragma unroll
        for (int u = 0; u < TX; u++) {
            const sqph_f2 us = {ur[u], ur[u]};
#pragma unroll
            for (int kp = 0; kp < L::TC2; kp++)
                if (!xzero2<TX, STACK>(u, kp)) pb[kp] = wgf_fma2(vt[u][kp], us, pb[kp]);
        }
        float *st = lf + 2 * L::O_STAGE;
#pragma unroll
        for (int k = 0; k < TC; k++) st[L::xrowf(c, k) * L::Rf + r] = pb[k / 2][k & 1];
    }
    template <int TX, bool STACK = false>
    static __device__ __forceinline__ void stage2_f(const sqph_f2 (&bt)[TR][L::TC2], const sqph_f2 (&vt)[TX][L::TC2], const float (&y1)[TC],
                                                    float *lf, int r, int c) {
        sqph_f2 yp[L::TC2];
#pragma unroll
        for (int kp = 0; kp < L::TC2; kp++) {
            yp[kp][0] = y1[2 * kp];
            yp[kp][1] = (2 * kp + 1 < TC) ? y1[2 * kp + 1 < TC ? 2 * kp + 1 : 0] : 0.0f;
        }
        float *sty = lf + 2 * L::O_STAGE_Y;
        float *stx = lf + (STACK ? 2 * L::O_STAGE_Y + R * TR * L::Cf : 2 * L::O_STX);
        // position inside a row rotated by 2 (r / 8): the row stride is a multiple of 4 floats, so lanes r and r + 8 of a store would share
        // banks; the owner sums the whole row, so the order is free
        const int pos = R >= 16 ? ((c + 2 * (r >> 3)) & (C - 1)) : c;
#pragma unroll
        for (int s = 0; s < TR; s++) {
            sqph_f2 acc = {0.0f, 0.0f};
#pragma unroll
            for (int kp = 0; kp < L::TC2; kp++) acc = wgf_fma2(bt[s][kp], yp[kp], acc);
            sty[(R * s + r) * L::Cf + pos] = acc[0] + acc[1];
        }
#pragma unroll
        for (int u = 0; u < TX; u++) {
            sqph_f2 acc = {0.0f, 0.0f};
#pragma unroll
            for (int kp = 0; kp < L::TC2; kp++)
                if (!xzero2<TX, STACK>(u, kp)) acc = wgf_fma2(vt[u][kp], yp[kp], acc);
            stx[(R * u + r) * L::Cf + pos] = acc[0] + acc[1];
        }
    }
    static __device__ __forceinline__ T reducef_xt(const float *lf, int i) { return (T)wgf_sum<C>(lf + 2 * L::O_STX + i * L::Cf); }
    static __device__ __forceinline__ T reducef_over_c(const float *lf, int t) { return (T)wgf_sum<C>(lf + 2 * L::O_STAGE_Y + t * L::Cf); }

    // owner-side reductions (lane t owns output t)
    static __device__ __forceinline__ T reduce_over_r(const T *lds, int t) { return wg_sum<R>(lds + L::O_STAGE + (t < L::NP ? L::xslot(t) : 0) * L::Rp); }  // t = column index
    static __device__ __forceinline__ T reduce_over_c(const T *lds, int t) { return wg_sum<C>(lds + L::O_STAGE_Y + t * L::Cp); }

    // ------------------------------------------------------------------ tile loads
    static __device__ __forceinline__ void load_A_tile(const TIN *__restrict__ gA, int n, int m, int r, int c, T (&at)[TR][TC]) {
#pragma unroll
        for (int k = 0; k < TC; k++) {
            const int j = L::col(c, k);
#pragma unroll
            for (int s = 0; s < TR; s++) {
                const int i = R * s + r;
                at[s][k] = (j < n && i < m) ? (T)gA[(long)j * m + i] : T(0);
            }
        }
    }
    template <typename TM>
    static __device__ __forceinline__ void load_sq_tile(const TM *__restrict__ M, int n, int r, int c, T (&wt)[TW][TC]) {
#pragma unroll
        for (int u = 0; u < TW; u++) {
            const int i = R * u + r;
#pragma unroll
            for (int k = 0; k < TC; k++) {
                const int j = L::col(c, k);
                wt[u][k] = (i < n && j < n) ? (T)M[(long)j * n + i] : T(0);
            }
        }
    }
    static __device__ __forceinline__ void store_sq_tile(T *__restrict__ M, int n, int r, int c, const T (&wt)[TW][TC]) {
#pragma unroll
        for (int u = 0; u < TW; u++) {
            const int i = R * u + r;
#pragma unroll
            for (int k = 0; k < TC; k++) {
                const int j = L::col(c, k);
                if (i < n && j < n) M[(long)j * n + i] = wt[u][k];
            }
        }
    }

    // B = A W' IN PLACE over the A tile (bt[s][k] = sum_j A[R s + r][j] * W[C k + c][j], W lower triangular).
    // W is staged once (transposed) in LDS from the register tile, A goes through LDS one block of R rows at
    // a time — nothing is re-read from global memory.
    // Triangular: column j = C kj + cj of A meets W[C k + c][j], which is zero for k < kj in EVERY lane (cyclic columns), so the
    // products of column block kj run over the tile columns k >= kj only — (TC + 1) / 2 TC of the full count, the same in both
    // wavefronts of a QP (with blocked columns the wave holding the last column groups ran all of them: no gain on the critical path).
    // STACK (stacked operator, see load_stacked_lds): the rows of W' that fall into tile row s of the B tile (stacked row R s + r = SOFF + j)
    // are added as soon as that tile row is done — from the same staged copy, while nothing else is live that is not live anyway
    static constexpr int TXS = TW - 1;
    // stacked row of W' row j is SOFF + j: a compile-time offset (the structural zeros of the W' tile rows must not depend on m);
    // the host launches the stacked kernels where m <= SOFF
    static constexpr int SOFF = R * (TR + TXS) - L::NP;
    template <bool STACK = false>
    static __device__ __forceinline__ void build_B_inplace(T (&at)[TR][TC], const T (&wt)[TW][TC], int n, T *lds, int r, int c, int m SQPH_STICK_ARGS) {
        // All of W is staged once, transposed, in [0, NP * WSTR): Wf[j][slot(i')] = W[i'][j].  It serves the B = A W' product below
        // (the W tile's registers are dead from here on) and, afterwards, the W -> W' transposition of load_vt_lds().  The A tile goes
        // through As one block of R rows at a time (natural column order: As[r][j]); nothing goes through global memory.
        T *Wf = lds, *As = lds + L::O_AS2;
        wsync();  // the factorisation's scratch in this region is dead
        if constexpr (L::WPACK) {
            static_assert(!STACK, "the packed copy serves the padded operator");
            // row j = C k + c (column block k, my column group), entry i = R u + r = C u + r of it: kept for u >= k only
#pragma unroll
            for (int u = 0; u < TW; u++)
#pragma unroll
                for (int k = 0; k <= u && k < TC; k++) Wf[L::wf_blk(k) + c * L::wf_len(k) + C * (u - k) + r] = wt[u][k];
        } else {
#pragma unroll
        for (int u = 0; u < TW; u++) {
            const int i = R * u + r;
            if (i < L::NP) {
#pragma unroll
                for (int k = 0; k < TC; k++) Wf[L::col(c, k) * L::WSTR + L::slot(i)] = wt[u][k];
            }
        }
        }
#pragma unroll
        for (int s = 0; s < TR; s++) {
            wsync();
#pragma unroll
            for (int k = 0; k < TC; k++) As[r * L::SSTR + L::col(c, k)] = at[s][k];
            wsync();
            T acc[TC];
#pragma unroll
            for (int k = 0; k < TC; k++) acc[k] = 0;
#pragma unroll
            for (int kj = 0; kj < TC; kj++) {
                const int cjn = n - C * kj < C ? n - C * kj : C;  // columns of this block inside the matrix (block-uniform)
                const T *ap = As + r * L::SSTR + C * kj;
                if constexpr (L::WPACK) {
                    const T *wp = Wf + L::wf_blk(kj) + c;
#pragma unroll 2
                    for (int cj = 0; cj < cjn; cj++) {
                        const T av = ap[cj];
                        T wv[TC];
#pragma unroll
                        for (int k = kj; k < TC; k++) wv[k] = wp[cj * L::wf_len(kj) + C * (k - kj)];
#pragma unroll
                        for (int k = kj; k < TC; k++) acc[k] = wg_fma(av, wv[k], acc[k]);
                    }
                } else {
                const T *wp = Wf + (C * kj) * L::WSTR + L::SLOT * c;
#pragma unroll 2
                for (int cj = 0; cj < cjn; cj++) {
                    const T av = ap[cj];
                    T wv[L::SLOT];
                    wg_read<L::SLOT>(wp + cj * L::WSTR, wv);  // the reads of the entries below kj are dead code
#pragma unroll
                    for (int k = kj; k < TC; k++) acc[k] = wg_fma(av, wv[k], acc[k]);
                }
                }
            }
#pragma unroll
            for (int k = 0; k < TC; k++) at[s][k] = SQPH_TILE_QUANT(acc[k]);
            if constexpr (STACK) {
                if (R * s + R - 1 >= SOFF) {  // (compile-time: tile rows below the W' rows skip this)
                    const int jp = R * s + r - SOFF;  // row of W' at stacked row R s + r
                    const bool in = jp >= 0 && jp < n;
                    T tmp[L::SLOT];
                    wg_read<L::SLOT>(Wf + (in ? jp : 0) * L::WSTR + L::SLOT * c, tmp);
#pragma unroll
                    for (int k = 0; k < TC; k++) at[s][k] += (in && L::col(c, k) < n) ? SQPH_TILE_QUANT(tmp[k]) : T(0);
                }
            }
        }
        // The W' tile (vt[u][k] = W[C k + c][R u + r]) is picked up from Wf by load_vt_lds() once the set-up block — and with it
        // the W tile's registers — has ended.  (Computing vt inside this block, next to the live W tile, put 20 tile registers of
        // the iteration loop into scratch.)
    }
    // second half of the W -> W' transposition: vt[u][k] = W[C k + c][R u + r] from the staged copy (the tile of W' in the tile
    // layout — rows cyclic over r, columns cyclic over c — that the iteration loop runs on).  Nothing goes through global memory.
    // Entries that are structurally zero (L::vt_zero) are never read by the stages: left at zero here.
    static __device__ __forceinline__ void load_vt_lds(T *lds, int n, int r, int c, T (&vt)[TW][TC]) {
        wsync();
        if constexpr (L::WPACK) {
            // vt[u][k] = W[C k + c][R u + r]: row j = C u + r of the packed copy (column block u, R = C), entry C k + c of it, k >= u
#pragma unroll
            for (int u = 0; u < TW; u++)
#pragma unroll
                for (int k = 0; k < TC; k++) {
                    T v = T(0);
                    if (k >= u) v = lds[L::wf_blk(u) + r * L::wf_len(u) + C * (k - u) + c];
                    vt[u][k] = (k >= u && R * u + r < n && L::col(c, k) < n) ? SQPH_TILE_QUANT(v) : T(0);
                }
        } else {
#pragma unroll
        for (int u = 0; u < TW; u++) {
            const int jp = R * u + r;
            T tmp[L::SLOT];
            wg_read<L::SLOT>(lds + (jp < L::NP ? jp : 0) * L::WSTR + L::SLOT * c, tmp);
#pragma unroll
            for (int k = 0; k < TC; k++) vt[u][k] = (!L::vt_zero(u, k) && jp < n && L::col(c, k) < n) ? SQPH_TILE_QUANT(tmp[k]) : T(0);
        }
        }
        wsync();
    }
    // STACKED operator (run<CHECKS, F32, STACK = true>, problems with m <= SOFF): instead of padding B to R*TR rows and W' to
    // R*TW, the rows of W' share the last tile rows of B — stacked row sigma = SOFF + j holds row j of W' — so the operator takes
    // TR + TXS tile rows (10 instead of 11 at the C3 shape, 14 VGPRs fewer), of which the multiply-adds with the structural zeros of W' are dropped
    // (stacked_zero: 9 of the 21 W'-only tile entries at the C3 shape — 122 instead of 154 multiply-adds per lane and iteration).
    // Row sigma = R s + r lives in tile row s of lane r as before; the tile rows s < TR are the B tile (rows >= m of it are zero)
    // PLUS the W' rows that fall into them, the TXS rows beyond are W' only.  Both come from the staged transposed copy of W.
    static constexpr bool stacked_zero(int u, int k) { return C * k + C - 1 < R * (TR + u) - SOFF; }  // entry (u, k) of the W'-only tile rows
    static __device__ __forceinline__ void load_stacked_lds(T *lds, int n, int r, int c, T (&xt)[TXS][TC]) {
        wsync();
#pragma unroll
        for (int u = 0; u < TXS; u++) {  // the tile rows beyond the B tile (the W' rows inside it were added by build_B_inplace<true>)
            const int jp = R * (TR + u) + r - SOFF;
            const bool in = jp >= 0 && jp < n;
            T tmp[L::SLOT];
            wg_read<L::SLOT>(lds + (in ? jp : 0) * L::WSTR + L::SLOT * c, tmp);
#pragma unroll
            for (int k = 0; k < TC; k++) xt[u][k] = (!stacked_zero(u, k) && in && L::col(c, k) < n) ? SQPH_TILE_QUANT(tmp[k]) : T(0);
        }
        wsync();
    }
    static __device__ __forceinline__ int stacked_slot_f(int sigma) {  // the same for the float views of the fp32-product variant
        const int s = sigma / R, rr = sigma % R;
        return s < TR ? 2 * L::O_ROWV + rr * L::TRf + s : 2 * L::O_WROW + rr * L::TWf + (s - TR);
    }
    // LDS slot of stacked row sigma in the operand vector: tile rows < TR in the row-gather region, the rest in the W-row region
    static __device__ __forceinline__ int stacked_slot(int sigma) {
        const int s = sigma / R, rr = sigma % R;
        return s < TR ? L::O_ROWV + rr * L::TRp + s : L::O_WROW + rr * L::TWp + (s - TR);
    }

    // residual check only: A x partials (staged for the reduction over c) and A' y partials (over r), with the A tile streamed from
    // global memory.  A check is a chain of dependent memory round trips (the register tiles hold B and W', so A and P come from
    // L2 / HBM), hence: loads are unconditional (indices clamped into the matrix; the padding needs no masking because x and y
    // are staged with zeros beyond n and m and the padded outputs are never read), and two columns are in flight at a time — as
    // many as fit next to the live tiles without raising the kernel's register high-water mark (the loop is NOT unrolled further).
    // m == 0: the caller passes any readable pointer (every index is clamped to 0 then).
    static __device__ __forceinline__ void stage_A_AT_gmem(const TIN *__restrict__ gA, int n, int m, int r, int c,
                                                           const T (&y)[TR], T *lds) {
        T pz[TR];
        int ii[TR];
#pragma unroll
        for (int s = 0; s < TR; s++) {
            pz[s] = 0;
            const int i = R * s + r;
            ii[s] = i < m ? i : (m > 0 ? m - 1 : 0);
        }
        T *stx = lds + L::O_STAGE;
        const T *xv = lds + L::O_COLV + c * L::TCp;
#pragma unroll 1
        for (int k = 0; k < TC; k += 2) {
            const int j0 = L::col(c, k), j1 = L::col(c, k + ((k + 1 < TC) ? 1 : 0));
            const TIN *p0 = gA + (long)(j0 < n ? j0 : n - 1) * m;
            const TIN *p1 = gA + (long)(j1 < n ? j1 : n - 1) * m;
            T a0[TR], a1[TR];
#pragma unroll
            for (int s = 0; s < TR; s++) a0[s] = (T)p0[ii[s]];
#pragma unroll
            for (int s = 0; s < TR; s++) a1[s] = (T)p1[ii[s]];
            {
                const T xk = xv[k];
                T pb = 0;
#pragma unroll
                for (int s = 0; s < TR; s++) {
                    pz[s] = wg_fma(a0[s], xk, pz[s]);
                    pb = wg_fma(a0[s], y[s], pb);
                }
                stx[L::xrow(c, k) * L::Rp + r] = pb;
            }
            if (k + 1 < TC) {
                const T xk = xv[k + 1];
                T pb = 0;
#pragma unroll
                for (int s = 0; s < TR; s++) {
                    pz[s] = wg_fma(a1[s], xk, pz[s]);
                    pb = wg_fma(a1[s], y[s], pb);
                }
                stx[L::xrow(c, k + 1) * L::Rp + r] = pb;
            }
        }
        T *sty = lds + L::O_STAGE_Y;
#pragma unroll
        for (int s = 0; s < TR; s++) sty[(R * s + r) * L::Cp + c] = pz[s];
    }
    // residual check only: P x partials with the P tile streamed two row blocks at a time (same scheme)
    static __device__ __forceinline__ void stage_P_gmem(const TIN *__restrict__ gP, int n, int r, int c, T *lds) {
        T xc[TC];
        get_colv(lds, c, xc);
        int jo[TC];
#pragma unroll
        for (int k = 0; k < TC; k++) {
            const int j = L::col(c, k);
            jo[k] = (j < n ? j : n - 1) * n;
        }
        T *sty = lds + L::O_STAGE_Y;
#pragma unroll 1
        for (int u = 0; u < TW; u += 2) {
            const int i0 = R * u + r, i1 = R * (u + ((u + 1 < TW) ? 1 : 0)) + r;
            const TIN *p0 = gP + (i0 < n ? i0 : n - 1);
            const TIN *p1 = gP + (i1 < n ? i1 : n - 1);
            T v0[TC], v1[TC];
#pragma unroll
            for (int k = 0; k < TC; k++) v0[k] = (T)p0[jo[k]];
#pragma unroll
            for (int k = 0; k < TC; k++) v1[k] = (T)p1[jo[k]];
            T acc = 0;
#pragma unroll
            for (int k = 0; k < TC; k++) acc = wg_fma(v0[k], xc[k], acc);
            sty[(R * u + r) * L::Cp + c] = acc;
            if (u + 1 < TW) {
                T acc1 = 0;
#pragma unroll
                for (int k = 0; k < TC; k++) acc1 = wg_fma(v1[k], xc[k], acc1);
                sty[(R * (u + 1) + r) * L::Cp + c] = acc1;
            }
        }
    }

    // P -> LDS, asynchronously (gfx950 LDS-DMA, global_load_lds_dwordx4: 64 lanes x 16 bytes land contiguously at a wave-uniform LDS
    // base, no VGPRs in between).  Issued ahead of the A tile's loads, so the two HBM streams of a set-up overlap, and every lane's
    // read is coalesced — the factor's own access P[min(i,j) n + max(i,j)] (lower triangle only, qp.cpp:159-189) is a stride-n
    // gather for half of the tile (measured: ~20 k of the set-up's 191 k cycles waiting for it).  Returns false (nothing issued)
    // when the block is not 16-byte aligned; the factor then reads P from global memory as before.
    static constexpr bool P_STAGED = MSET ? MS::PST : L::P_STAGED;
    static constexpr int O_PST = MSET ? MS::O_SB : L::O_PST;
    static __device__ __forceinline__ bool stage_P_async(const TIN *__restrict__ gP, int n, T *lds, int t) {
#ifdef SQPH_NO_P_STAGE  // A/B experiment builds only
        return false;
#endif
        if constexpr (!P_STAGED) {
            return false;
        } else {
            if ((reinterpret_cast<unsigned long long>(gP) & 15ull) != 0) return false;
            char *dst = reinterpret_cast<char *>(lds + O_PST);
#ifdef SQPH_SIM
            for (unsigned e = (unsigned)t; e < (unsigned)(n * n); e += (unsigned)NT) reinterpret_cast<TIN *>(dst)[e] = gP[e];
#else
            const unsigned bytes = (unsigned)(n * n) * (unsigned)sizeof(TIN);
            const char *src = reinterpret_cast<const char *>(gP);
            const unsigned full = bytes & ~15u;
            for (unsigned off = (unsigned)(t >> 6) * 1024u + (unsigned)(t & 63) * 16u; off < full; off += (unsigned)NT * 16u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(dst + (off - (unsigned)(t & 63) * 16u)), 16, 0, 0);
            if ((bytes & 15u) && t == 0) {  // n odd with 8-byte elements (or n n not a multiple of 4 floats): the tail by a plain load
                for (unsigned e = full / (unsigned)sizeof(TIN); e < (unsigned)(n * n); e++) reinterpret_cast<TIN *>(dst)[e] = gP[e];
            }
#endif
            return true;
        }
    }

    // ------------------------------------------------------------------ factor (see admm_generic.h factor_schur)
    // Scratch inside the staging area: rho[MP] | rowbuf[NP + 1] | sj[NP]   (doubles)
    // `at` is the A register tile (rows R s + r, columns C k + c); it is only read here.
    // p_staged: P lies in LDS at O_PST (stage_P_async was issued before the call; block-uniform)
    // Only the LOWER triangle of S is formed and eliminated (tile entries (u, k) with L::wt_zero(u, k) are never touched: both tile
    // directions are cyclic, so they are the same entries in every lane): S = P_lower + sigma I + A'RA over the non-zero tile entries;
    // at pivot k the entries right of the diagonal of row k — by symmetry column k below the diagonal — are published by the lanes
    // that own column k, the entries left of it (the L^-1 part) by the lanes that own row k.
    static constexpr int RC = R / C;  // pivot rows R u + C h + rr2 (h < RC, rr2 < C) lie in tile column RC u + h of lane group rr2
    static_assert(R % C == 0, "a tile row of pivots covers whole column groups");
    static __device__ __forceinline__ bool factor(const TIN *__restrict__ gP, const T (&at)[TR][TC], int n, int m, T sigma,
                                                  T *lds, int t, int r, int c, T (&wt)[TW][TC], bool p_staged SQPH_STICK_ARGS) {
        T *rho_l = lds + L::O_RHO;
        T *rowbuf = lds + L::O_ROWBUF;
        T *sjv = lds + L::O_SJ;
        T *As = lds + L::O_AS;
        int ir[TW], jc[TC];
#pragma unroll
        for (int u = 0; u < TW; u++) {
            const int i = R * u + r;
            ir[u] = i < n ? i : 0;
        }
#pragma unroll
        for (int k = 0; k < TC; k++) {
            const int j = L::col(c, k);
            jc[k] = j < n ? j : 0;
        }
#pragma unroll
        for (int u = 0; u < TW; u++)
#pragma unroll
            for (int k = 0; k < TC; k++) wt[u][k] = 0;
        // S = A' diag(rho) A : the A tile goes through LDS one block of R rows at a time (row R s + il of A is
        // As[il][.]); every lane then reads the TW + TC entries of each row it needs. No global re-reads.
        int sl[TW];
#pragma unroll
        for (int u = 0; u < TW; u++) sl[u] = L::slot(ir[u]);
#pragma unroll
        for (int s = 0; s < TR; s++) {
            wsync();
#pragma unroll
            for (int k = 0; k < TC; k++) As[r * L::SSTR + L::SLOT * c + k] = at[s][k];
            wsync();
#pragma unroll 2
            for (int il = 0; il < R; il++) {
                const int i = R * s + il;
                if (i >= m) break;
                const T ri = rho_l[i];
                const T *row = As + il * L::SSTR;
                T a2[L::SLOT], a1[TW];
                wg_read<L::SLOT>(row + L::SLOT * c, a2);
#pragma unroll
                for (int u = 0; u < TW; u++) a1[u] = row[sl[u]] * ri;
#pragma unroll
                for (int u = 0; u < TW; u++)
#pragma unroll
                    for (int k = 0; k < TC; k++)
                        if (!L::wt_zero(u, k)) wt[u][k] = wg_fma(a1[u], a2[k], wt[u][k]);
            }
        }
        SQPH_STICK(1)
#ifndef SQPH_SIM
        if (L::P_STAGED && p_staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my share of the P block has landed in LDS
#endif
        wsync();
        for (int e = t; e < L::NP + 2; e += NT) {
            rowbuf[e] = 0;
            if (e < L::NP) sjv[e] = T(1);
        }
        wsync();
        const TIN *Pst = reinterpret_cast<const TIN *>(lds + L::O_PST);
#pragma unroll
        for (int u = 0; u < TW; u++) {
            const int i = R * u + r;
#pragma unroll
            for (int k = 0; k < TC; k++) {
                if (L::wt_zero(u, k)) continue;
                const int j = L::col(c, k);
                const bool ok = i < n && j < n;
                const int lo = i > j ? i : j, hi = i > j ? j : i;
                // only the lower triangle of P reaches the reference's factor (Eigen::LDLT<.,Lower>)
                const T p = ok ? ((L::P_STAGED && p_staged) ? (T)Pst[hi * n + lo] : (T)gP[(long)hi * n + lo]) : T(0);
                wt[u][k] = ok ? (wt[u][k] + p + (i == j ? sigma : T(0))) : T(0);
                if (ok && i == j) sjv[j] = wt[u][k];  // the diagonal, for the Jacobi scaling
            }
        }
        wsync();
        // non-positive / non-finite diagonal => not SPD (block-uniform decision through LDS)
        bool bad = false;
        for (int j = 0; j < n; j++) {
            const T d = sjv[j];
            bad = bad || !(d > T(0)) || !(d * T(0) == T(0));
        }
        wsync();
        if (bad) return false;
        if constexpr (NW > 0) {
            if (t < n) sjv[t] = T(1) / (T)sqrt((double)sjv[t]);
        } else {
            for (int e = t; e < n; e += NT) sjv[e] = T(1) / (T)sqrt((double)sjv[e]);
        }
        wsync();
        T srow[TW], scol[TC];
#pragma unroll
        for (int u = 0; u < TW; u++) srow[u] = sjv[ir[u]];
#pragma unroll
        for (int k = 0; k < TC; k++) scol[k] = sjv[jc[k]];
#pragma unroll
        for (int u = 0; u < TW; u++)
#pragma unroll
            for (int k = 0; k < TC; k++)
                if (!L::wt_zero(u, k)) wt[u][k] = wt[u][k] * srow[u] * scol[k];
        SQPH_STICK(2)
        // forward elimination of [S~ | I] in place on the lower triangle; pivot row k = R u + C h + rr2 is published through LDS
        // (slot order): its columns <= k by the lanes that hold row k, its columns > k — column k below the diagonal — by the
        // lanes that hold column k (lane group rr2, tile column RC u + h)
        bool ok_all = true;
        T dsave[TW];
#pragma unroll
        for (int u = 0; u < TW; u++) dsave[u] = T(1);
#pragma unroll
        for (int u = 0; u < TW; u++) {
#pragma unroll
            for (int h = 0; h < RC; h++) {
                constexpr int kkmax = TC - 1;
                const int kk = RC * u + h < kkmax ? RC * u + h : kkmax;  // tile column of the pivots of this stretch (beyond TC: k >= n, not run)
#pragma unroll 1
                for (int rr2 = 0; rr2 < C; rr2++) {
                    const int rr = C * h + rr2;
                    const int k = R * u + rr;
                    if (k >= n || RC * u + h >= TC || !ok_all) break;
                    if (r == rr) {
#pragma unroll
                        for (int q = 0; q < TC; q++)
                            if (q <= kk && !L::wt_zero(u, q) && (q < kk || c <= rr2)) rowbuf[TC * c + q] = wt[u][q];
                    }
                    if (c == rr2) {
#pragma unroll
                        for (int v = u; v < TW; v++) {
                            const int i = R * v + r;
                            if (!L::wt_zero(v, kk) && i > k && i < L::NP) rowbuf[L::cslot(i)] = wt[v][kk];
                        }
                    }
                    wsync();
                    const T d = rowbuf[L::cslot(k)];
                    if (!(d > T(0)) || !(d * T(0) == T(0))) {
                        ok_all = false;
                        break;
                    }
                    const T dinv = T(1) / d;
                    T g[TC], f[TW];
#pragma unroll
                    for (int q = 0; q < TC; q++) g[q] = rowbuf[TC * c + q];
                    g[kk] = (c == rr2) ? d + T(1) : g[kk];
#pragma unroll
                    for (int v = u; v < TW; v++) {
                        const int i = R * v + r;
                        const T gi = rowbuf[i < L::NP ? L::cslot(i) : 0];
                        f[v] = (i > k && i < n) ? gi * dinv : T(0);
                    }
                    wsync();
#pragma unroll
                    for (int v = u; v < TW; v++)
#pragma unroll
                        for (int q = 0; q < TC; q++)
                            if (!L::wt_zero(v, q)) wt[v][q] = wg_fma(-f[v], g[q], wt[v][q]);
                    dsave[u] = (r == rr) ? d : dsave[u];
                }
            }
        }
        SQPH_STICK(3)
        // W = D^-1/2 L^-1 D_J^-1/2
#pragma unroll
        for (int u = 0; u < TW; u++) {
            const int i = R * u + r;
            const T rs = T(1) / (T)sqrt((double)dsave[u]);
#pragma unroll
            for (int k = 0; k < TC; k++) {
                const int j = L::col(c, k);
                const bool ok = i < n && j < n;
                const T v = i > j ? wt[u][k] * rs : (i == j ? rs : T(0));
                wt[u][k] = (ok && !L::wt_zero(u, k)) ? v * scol[k] : T(0);
            }
        }
        return ok_all;
    }

    // ------------------------------------------------------------------ kernel body
    // CHECKS = false: the instantiation for calls that never look at the residuals (check_termination == 0 and no adaptive rho).  The
    // check block is dead code there, but leaving it in costs the iteration loop its registers: with it the allocator keeps scratch
    // reloads and a store inside the loop (256 VGPRs, 27 spilled); without it nothing is spilled (241 VGPRs) — 3.01 -> 2.75 ms per
    // 8,192 x 200 iterations on the C3 shard.  Those kernels live in a translation unit of their own (wg_nocheck.hip): instantiated next
    // to the checking ones, they changed the register allocation of the latter (+3.7 % on the default-termination run).
    // F32 = true: the iteration's two stages in single precision (stage1_f / stage2_f), everything else unchanged
    // STACK = true: the stacked operator (load_stacked_lds); the host launches it only where m <= SOFF
    template <bool CHECKS = true, bool F32 = false, bool STACK = false>
    static __device__ __forceinline__ void run(const KArgs<T, TIN> &a, T *lds) {
        constexpr int TX = STACK ? TXS : TW;  // tile rows of the iteration's second tile
        static_assert(!F32 || L::F32_FITS, "float views must fit the regions of the double layout");
        static_assert(!(F32 && STACK) || R * (TR + TXS) * L::Cf <= 2 * L::STAGE_Y, "stacked float partial sums fit the z~ staging region");
        // the stacked fp64 operator stages R (TR + TXS) rows of partial sums from O_STAGE_Y on: past STAGE_Y they run over the gap and the
        // O_STX region (unused in the stacked mode: its x~ partial sums continue the z~ array) — but never into the owners' constants
        static_assert(!STACK || F32 || L::O_STAGE_Y + R * (TR + TXS) * L::Cp <= L::O_QV, "stacked partial sums end before the owners' constants");
        float *lf = reinterpret_cast<float *>(lds);
        const int t = threadIdx.x;
        const int r = t % R, c = t / R;
        const int qp = blockIdx.x;
        if (qp >= a.batch) return;
        const int n = a.n, m = a.m;
        const TIN *gP = a.P + (long)qp * a.sP;
        const TIN *gq = a.q + (long)qp * a.sq;
        const TIN *gA = a.A + (long)qp * a.sA;
        const TIN *gl = a.l + (long)qp * a.sl;
        const TIN *gu = a.u + (long)qp * a.su;
        T *sx = a.x + (long)qp * n;
        T *sz = a.z + (long)qp * m;
        T *sy = a.y + (long)qp * m;
        T *srho = a.rho_vec + (long)qp * m;
        int *sct = a.ctype + (long)qp * m;
        T *gW = a.Sinv + (long)qp * 2 * n * n;

        // Calls that never check (CHECKS = false) touch status / iter / rho_updates only: those three travel in scalar registers and
        // the record's doubles stay where they are.  (Kept whole in registers, the record was spilled: 32 B of scratch per lane, 33 MB
        // of HBM writes per C3 launch.  The checking kernels keep it whole: writing the residuals from the check block instead measured
        // +0.4 % default / +1 % SQP settings.)
        sqph_info info;
        if constexpr (CHECKS) {
            info = a.info[qp];
        } else {
#ifdef SQPH_SIM
            info = a.info[qp];
#else
            info.status = __builtin_amdgcn_readfirstlane(a.info[qp].status);
            info.iter = __builtin_amdgcn_readfirstlane(a.info[qp].iter);
            info.rho_updates = __builtin_amdgcn_readfirstlane(a.info[qp].rho_updates);
#endif
        }
        T rho_s = a.rho[qp];
        const int mode = a.mode;
        if (!(mode & (MODE_SETUP | MODE_UPDATE)) &&
            (info.status == SQPH_UNINITIALIZED || info.status == SQPH_NUMERICAL_ISSUES))
            return;  // qp.cpp:68-71 (block-uniform)
#if defined(SQPH_EXPERIMENTS) && !defined(SQPH_SIM)
        // experiment: de-phase co-resident workgroups.  xp[0] = delay in s_memtime ticks, xp[1] = selector (0: blockIdx bit xp[2];
        // 1: wave slot parity of this wave; 2: delay proportional to blockIdx bits [xp[2], xp[2]+2)), xp[3] = only blocks below
        unsigned long long xt0 = __builtin_amdgcn_s_memtime();
        {
            const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            if (a.xp[0] > 0 && (a.xp[3] <= 0 || qp < a.xp[3])) {
                unsigned long long d = 0;
                if (a.xp[1] == 0) d = ((qp >> a.xp[2]) & 1) ? a.xp[0] : 0;
                else if (a.xp[1] == 1) d = (hwid & 1) ? a.xp[0] : 0;
                else d = (unsigned long long)((qp >> a.xp[2]) & 3) * a.xp[0];
                while (__builtin_amdgcn_s_memtime() - xt0 < d) __builtin_amdgcn_s_sleep(32);
            }
            if (a.xdbg && (t & 63) == 0) {
                a.xdbg[8 * qp + 4 * (t >> 6) + 0] = hwid | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
                a.xdbg[8 * qp + 4 * (t >> 6) + 1] = xt0;
            }
        }
#endif

        // lane t owns x[t], q[t] (t < n) and z[t], y[t], l[t], u[t], rho[t] (t < m)
        const bool nown = t < n, mown = t < m;
        const T INF = T(1) / T(0);
        // q, l, u of the owned elements live in LDS (read once per iteration by the owner): 6 VGPRs that the
        // 2-waves-per-QP tiles cannot spare (a spilled VGPR costs a scratch round trip per iteration)
        T *qv = lds + L::O_QV, *lov = lds + L::O_LOV, *upv = lds + L::O_UPV, *rinvv = lds + L::O_RINV;
        if (t < L::NP) qv[t] = nown ? (T)gq[t] : T(0);
        if (t < L::MP) {
            lov[t] = mown ? (T)gl[t] : -INF;
            upv[t] = mown ? (T)gu[t] : INF;
        }
        __syncthreads();
        T x = 0, z = 0, y = 0;
        T rho = T(1);
        if (t < L::MP) rinvv[t] = T(1);  // 1/rho lives in LDS (see O_RINV)

        bool rho_differs = false;
        if (mode & (MODE_SETUP | MODE_UPDATE)) {
            rho_s = a.rho0;
            if (mown) {
                const T lo = lov[t], up = upv[t];
                int ctype = SQPH_INEQUALITY_CONSTRAINT;
                if (lo < -a.loose_thresh && up > a.loose_thresh)
                    ctype = SQPH_LOOSE_BOUNDS;
                else if (up - lo < a.eq_tol)
                    ctype = SQPH_EQUALITY_CONSTRAINT;
                rho = rho_for_type<T>(ctype, rho_s, a.rho_min, a.rho_eq_factor);
                rinvv[t] = T(1) / rho;
                rho_differs = !(rho == srho[t]);  // against the vector the resident factor was built with (MODE_SAME_MATRICES)
                sct[t] = ctype;
                srho[t] = rho;
            }
            info.rho_updates += 1;
            if (!(mode & MODE_SETUP)) {
                if (nown) x = sx[t];
                if (mown) {
                    z = sz[t];
                    y = sy[t];
                }
            }
        } else {
            if (nown) x = sx[t];
            if (mown) {
                z = sz[t];
                y = sy[t];
                rho = srho[t];
                rinvv[t] = T(1) / rho;
            }
        }

#ifdef SQPH_SETUP_TIMING
        unsigned long long sqph_stk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sqph_stp = __builtin_amdgcn_s_memtime();
#endif
        T vt[TX][TC];  // the tile of W' the iteration runs on (the W tile itself lives only inside the set-up block)
        T at[TR][TC];  // the A tile; turned into B = A W' in place once the factor is known
        bool need_factor = (mode & (MODE_SETUP | MODE_UPDATE | MODE_REFACTOR)) != 0;
        if ((mode & MODE_SAME_MATRICES) && (mode & (MODE_SETUP | MODE_UPDATE)) && !(mode & MODE_REFACTOR) &&
            info.status != SQPH_NUMERICAL_ISSUES && info.status != SQPH_UNINITIALIZED) {  // a failed set-up left no valid factor: refactor
            // sqph_setup_solve_reuse: same P and A as the resident factor; it is also the factor setup() would build if no lane's
            // rho differs from the vector it was built with (workgroup-wide OR through one LDS word of the idle staging area)
            T *flag = lds + L::O_STAGE;
            if (t == 0) *flag = T(0);
            __syncthreads();
            if (rho_differs) *flag = T(1);
            __syncthreads();
            if (*flag == T(0)) {
                need_factor = false;
                info.status = SQPH_UNSOLVED;  // qp.cpp:39-43
            }
            __syncthreads();
        }
        bool solving = false;
        bool state_dirty = (mode & MODE_SETUP) != 0;
        bool have_A = false;  // `at` currently holds A (as opposed to B)
        // A x of the owned constraint row by the recurrence  A x_{k+1} = alpha z~_{k+1} + (1 - alpha) A x_k  (z~ = A x~ is the stage-2
        // result, x_{k+1} = alpha x~ + (1 - alpha) x_k: qp.cpp:92-96) — exact in exact arithmetic, valid from x_0 = 0 (a solve that
        // starts from retained iterates computes A x at its checks by streaming A, as before).  With it the PRIMAL half of a
        // termination check needs no memory access at all, see the check block.
        T ax = 0;
        bool ax_valid = (mode & MODE_SETUP) != 0;
        const T alpha = a.alpha, sigma = a.sigma, oma = T(1) - a.alpha;
        int iter = 1;
        // countdowns to the next termination check / rho adaptation (0 or disabled: never fires)
        int next_check = a.check_termination > 0 ? a.check_termination : -1;
        int next_adapt = (a.adaptive_rho && a.adaptive_rho_interval > 0) ? a.adaptive_rho_interval : -1;
        for (;;) {
            {  // ---- set-up part of a pass; the W tile is a local of this block so that it is dead in the iteration loop
#ifndef SQPH_SIM
            __builtin_amdgcn_s_setprio(2);  // the set-up is one long dependent chain: ahead of the co-resident QPs' multiply-add blocks (note below)
#endif
            T wt[MSET ? 1 : TW][TC];  // (MFMA set-up: W lives in LDS blocks, never in a register tile)
            if (!need_factor) {   // solve() on a previously set-up instance
                if constexpr (MSET) {
                    __syncthreads();
                    MS::load_W(gW, n, lds, t);
                    __syncthreads();
                } else {
                    load_sq_tile<T>(gW, n, r, c, wt);
                }
            }
            if (need_factor) {
                __syncthreads();
                if (t < L::MP) lds[L::O_RHO + t] = mown ? rho : T(0);
                __syncthreads();
                int n_f = n, m_f = m, r_f = r, c_f = c, t_f = t;
                const TIN *gA_f = gA, *gP_f = gP;
                SQPH_OPAQUE_S(n_f); SQPH_OPAQUE_S(m_f); SQPH_OPAQUE_V(r_f); SQPH_OPAQUE_V(c_f); SQPH_OPAQUE_V(t_f);
                SQPH_OPAQUE_S(gA_f); SQPH_OPAQUE_S(gP_f);
                SQPH_STICK(7)
                const bool p_staged = stage_P_async(gP_f, n_f, lds, t_f);  // P -> LDS in flight next to the loads of A
                load_A_tile(gA_f, n_f, m_f, r_f, c_f, at);  // the only read of A from global memory per factorisation
                SQPH_STICK(0)
                bool ok;
                if constexpr (MSET) ok = MS::template factor<TIN, MS::O_SB, MS::PST>(gP_f, at, n_f, m_f, sigma, lds, t_f, p_staged SQPH_STICK_PASS);
                else ok = factor(gP_f, at, n_f, m_f, sigma, lds, t_f, r_f, c_f, wt, p_staged SQPH_STICK_PASS);
                SQPH_STICK(4)
                // the factor is kept for later solve() calls unless the host asked for a fused setup+solve without it
                if (!(mode & MODE_NO_FACTOR_STORE)) {
                    if constexpr (MSET) MS::store_W(gW, n_f, lds, t_f);
                    else store_sq_tile(gW, n_f, r_f, c_f, wt);
                }
                __syncthreads();
                need_factor = false;
                have_A = true;
                if (!solving) {
                    if (mode & (MODE_SETUP | MODE_UPDATE)) info.status = ok ? SQPH_UNSOLVED : SQPH_NUMERICAL_ISSUES;  // qp.cpp:39-43, 57-61
                    else if (!ok) info.status = SQPH_NUMERICAL_ISSUES;  // solve() rebuilding a factor that was not kept
                } else if (!ok) {
                    info.status = SQPH_NUMERICAL_ISSUES;  // qp.cpp:139-142: break, iter not advanced
                    break;
                } else {
                    iter++;  // the for-loop increment of the iteration that requested the new factor
                }
            }
            if (!(mode & MODE_SOLVE) || info.status == SQPH_NUMERICAL_ISSUES || info.status == SQPH_UNINITIALIZED) break;
            if (!solving) {
                solving = true;
                state_dirty = true;
                if ((mode & MODE_COLD_RESET) && !a.warm_start) {
                    x = z = y = 0;
                    ax_valid = true;
                }
            }
            // B = A W' replaces A in the iteration:  y1 = W u + B' w,  x~ = W' y1,  z~ = B y1   (u = sigma x - q)
            // => two dependent stages per iteration instead of four (A'w -> W -> W' -> A), i.e. 4 barriers, not 8.
            if (!have_A) {  // solve() on a previously set-up instance: the factor came from the workspace
                int n_t = n, m_t = m, r_t = r, c_t = c;
                const TIN *gA_t = gA;
                SQPH_OPAQUE_S(n_t); SQPH_OPAQUE_S(m_t); SQPH_OPAQUE_V(r_t); SQPH_OPAQUE_V(c_t); SQPH_OPAQUE_S(gA_t);
                load_A_tile(gA_t, n_t, m_t, r_t, c_t, at);
            }
            have_A = false;  // consumed: `at` holds B from here on
            {
                int n_t = n, r_t = r, c_t = c;
                SQPH_OPAQUE_S(n_t); SQPH_OPAQUE_V(r_t); SQPH_OPAQUE_V(c_t);
                SQPH_STICK(7)
                if constexpr (MSET) {
                    int m_t = m, t_t = t;
                    SQPH_OPAQUE_S(m_t); SQPH_OPAQUE_V(t_t);
                    MS::template build_B<STACK, SOFF>(at, n_t, m_t, lds, t_t SQPH_STICK_PASS);
                } else {
                    build_B_inplace<STACK>(at, wt, n_t, lds, r_t, c_t, m SQPH_STICK_PASS);
                }
                SQPH_STICK(5)
            }
#ifndef SQPH_SIM
            __builtin_amdgcn_s_setprio(0);
#endif
            }  // ---- end of the set-up part
            {
                int n_t = n, r_t = r, c_t = c;
                SQPH_OPAQUE_S(n_t); SQPH_OPAQUE_V(r_t); SQPH_OPAQUE_V(c_t);
                if constexpr (MSET) {
                    __syncthreads();
                    MS::template load_vt<TX, STACK, SOFF>(lds, n_t, r_t, c_t, vt);
                    __syncthreads();  // W in LDS is dead from here on: the operand vectors of the iteration take its place
                } else if constexpr (STACK) {
                    load_stacked_lds(lds, n_t, r_t, c_t, vt);
                } else {
                    load_vt_lds(lds, n_t, r_t, c_t, vt);
                }
                SQPH_STICK(6)
            }
            T (&bt)[TR][TC] = at;
            sqph_f2 btf[F32 ? TR : 1][L::TC2], vtf[F32 ? TX : 1][L::TC2];
            if constexpr (F32) {
                tile_to_f32<TR>(at, btf);
                tile_to_f32<TX>(vt, vtf);
            }
            // publish w = R (z - R^-1 y) [rhs tail of qp.cpp:275 pre-multiplied by R] and u = sigma x - q
            if constexpr (F32 && STACK) {
                if (mown) putf_rowv(lf, r, c, (float)(rho * (z - rinvv[t] * y)));
                if (nown) lf[stacked_slot_f(SOFF + t)] = (float)(sigma * x - qv[t]);
                for (int sg = t; sg < R * (TR + TX); sg += NT)
                    if ((sg >= m && sg < SOFF) || sg >= SOFF + n) lf[stacked_slot_f(sg)] = 0.0f;
            } else if constexpr (F32) {
                if (t < L::MP) putf_rowv(lf, r, c, mown ? (float)(rho * (z - rinvv[t] * y)) : 0.0f);
                if (t < L::NR) putf_wrow(lf, r, c, nown ? (float)(sigma * x - qv[t < L::NP ? t : 0]) : 0.0f);
            } else if constexpr (STACK) {
                // stacked operand vector [w ; 0 ; u ; 0]: w_i at stacked row i, u_j at row SOFF + j, zeros in between and beyond SOFF + n
                if (mown) put_rowv(lds, r, c, rho * (z - rinvv[t] * y));
                if (nown) lds[stacked_slot(SOFF + t)] = sigma * x - qv[t];
                for (int sg = t; sg < R * (TR + TX); sg += NT)
                    if ((sg >= m && sg < SOFF) || sg >= SOFF + n) lds[stacked_slot(sg)] = T(0);
            } else {
                if (t < L::MP) put_rowv(lds, r, c, mown ? rho * (z - rinvv[t] * y) : T(0));
                if (t < L::NR) put_wrow(lds, r, c, nown ? sigma * x - qv[t < L::NP ? t : 0] : T(0));
            }
#ifdef SQPH_PHASE_TIMING
            unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define SQPH_TICK(k) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tprev; tprev = tn_; }
#else
#define SQPH_TICK(k)
#endif
            // The owners' constants (1/rho, l, u, q of the owned elements) live in LDS for the padded fp64 operator, whose 154 tile
            // registers leave no room; the stacked operator (140), the fp32 tiles (70-77) and the small tiles do: there they are registers, re-read here
            // after every (re)factorisation — four LDS reads fewer per wave and iteration (C3 shard fixed-200 2.42 -> 2.39 ms, default
            // 2.02 -> 1.98).
            constexpr bool REGCONST = STACK || F32 || (!CHECKS && (TR + TW) * TC <= 48);  // ... and the small tiles of calls that never check (C2: 0.326 -> 0.309 ms)
            const T k_rinv = (REGCONST && t < L::MP) ? rinvv[t] : T(1), k_lo = (REGCONST && t < L::MP) ? lov[t] : T(0),
                    k_up = (REGCONST && t < L::MP) ? upv[t] : T(0), k_q = (REGCONST && t < L::NP) ? qv[t] : T(0);
            // Wave priorities: the two waves of a SIMD belong to different QPs; while one is in a latency-bound stretch (the wave-local
            // y1 reduction, the owners' reduction + update behind the barrier: short dependent chains of LDS round trips) the other is
            // usually issuing its multiply-add blocks.  s_setprio 3 for those stretches, 0 for the two FMA blocks, lets the short
            // chains through first: 2.71 -> 2.67 ms fixed-200, -1.1 % in the default and SQP-settings modes (tools/ab_slim.sh).  The
            // set-up block and the residual checks — dependent chains from end to end — run at priority 2 (1 or 3 measure the same):
            // 2.67 -> 2.57 ms fixed-200, 2.14 -> 2.10 default, 1.97 -> 1.93 SQP settings.
            // Segments: the iterations up to the next residual check run in a tight loop that contains no check code (the allocator then
            // keeps its spill code out of it); the check follows, between two segments.
            while (iter <= a.max_iter) {
                int seg = a.max_iter - iter + 1;
                if constexpr (CHECKS) {
                    if (next_check > 0 && next_check < seg) seg = next_check;
                    if (next_adapt > 0 && next_adapt < seg) seg = next_adapt;
                }
                for (int seg_i = 0; seg_i < seg; seg_i++) {
                    __syncthreads();
                    SQPH_TICK(0)
                    if constexpr (F32) {
                        float w[TR], ur[TX];
                        getf_rowv(lf, r, w);
                        wgf_read<TX>(lf + 2 * L::O_WROW + r * L::TWf, ur);
                        stage1_f<TX, STACK>(btf, vtf, w, ur, lf, r, c);
                    } else {   // stage 1 partials:  B' w + W u, both reduced over r
                        T w[TR], ur[TX];
                        get_rowv(lds, r, w);
                        wg_read<TX>(lds + L::O_WROW + r * L::TWp, ur);
                        stage1<TX, STACK>(bt, vt, w, ur, lds, r, c);
                    }
                    SQPH_TICK(1)
                    // y1 = W u + B' w: the R producers of a column group's TC outputs and their consumers in stage 2 are the
                    // same R lanes of one wavefront, so the reduction is wave-local — lane r < TC of group c sums output
                    // TC c + r and publishes it for its group; no workgroup barrier between the two stages
#ifndef SQPH_SIM
                    __builtin_amdgcn_s_setprio(3);  // see the note on wave priorities at the top of the segment loop
#endif
                    wave_sync();
                    SQPH_TICK(2)
                    if constexpr (R == 16 && TC <= 8) {
                        // two lanes per output: lane r < TC sums partials 0..7, lane r + 8 partials 8..15 of output TC c + r, one DPP
                        // rotation inside the 16-lane row combines them — half the LDS reads and adds in the wave's instruction
                        // stream and a shorter chain (C3 shard fixed-200 2.49 -> 2.42 ms)
                        // (which lane of a pair takes which half alternates with r / 4 and c: the 16-lane groups a ds_read_b128 is served
                        // in — lanes {0-3, 12-15, 20-27}, ... — then hold 14 different bank quads; with hh = r / 8 every read was two-way
                        // conflicted, 11.4 % of the kernel's LDS cycles: tools/xp/lds_model_wg.py)
                        const int o = r & 7, hh = F32 ? L::y1_half_f(r, c) : (((r >> 3) ^ (r >> 2) ^ c) & 1);
                        const int sj = F32 ? L::xrowf(c, o) : L::xrow(c, o), j = L::col(c, o);  // staging row and column of the output
                        if constexpr (F32) {
                            float part = (o < TC) ? wgf_sum<8>(lf + 2 * L::O_STAGE + sj * L::Rf + 8 * hh) : 0.0f;
                            part += xchg16<8>(part);
                            if (r < TC) putf_colv2(lf, j, j < n ? part : 0.0f);
                        } else {
                            T part = (o < TC) ? wg_sum<8>(lds + L::O_STAGE + sj * L::Rp + 8 * hh) : T(0);
                            part += xchg16<8>(part);
                            if (r < TC) put_colv2(lds, j, j < n ? part : T(0));
                        }
                    } else
                    if (r < TC) {
                        const int sj = F32 ? L::xrowf(c, r) : L::xrow(c, r), j = L::col(c, r);  // staging row and column of the output
                        if constexpr (F32) putf_colv2(lf, j, j < n ? wgf_sum<R>(lf + 2 * L::O_STAGE + sj * L::Rf) : 0.0f);
                        else put_colv2(lds, j, j < n ? wg_sum<R>(lds + L::O_STAGE + sj * L::Rp) : T(0));
                    }
                    SQPH_TICK(3)
                    wave_sync();
                    SQPH_TICK(4)
#ifndef SQPH_SIM
                    __builtin_amdgcn_s_setprio(0);
#endif
                    if constexpr (F32) {
                        float y1c[TC];
                        getf_colv2(lf, c, y1c);
                        stage2_f<TX, STACK>(btf, vtf, y1c, lf, r, c);
                    } else {   // stage 2 partials:  z~ = B y1  and  x~ = W' y1, both reduced over c
                        T y1c[TC];
                        get_colv2(lds, c, y1c);
                        stage2<TX, STACK>(bt, vt, y1c, lds, r, c);
                    }
                    SQPH_TICK(5)
                    // the owner's constants do not depend on the partial sums: fetched before the barrier, their LDS latency
                    // hides behind it
                    T c_rinv = T(1), c_lo = T(0), c_up = T(0), c_q = T(0);
                    if constexpr (REGCONST) {
                        c_rinv = k_rinv; c_lo = k_lo; c_up = k_up; c_q = k_q;
                    } else {
                    if (t < L::MP) {
                        c_rinv = rinvv[t];
                        c_lo = lov[t];
                        c_up = upv[t];
                    }
                    if (t < L::NP) c_q = qv[t];
                    }
#ifndef SQPH_SIM
                    __builtin_amdgcn_s_setprio(3);  // see the note on wave priorities at the top of the segment loop
#endif
                    __syncthreads();
                    SQPH_TICK(6)
                    // owner work sits behind wave-uniform branches on purpose: waves without owners skip it, and the
                    // SIMDs are issue-bound at two waves each (a branch-free variant measured 14 % slower)
                    // the owners' updates with fused multiply-adds, like the products (6 instructions fewer on the chain that follows
                    // the barrier: -0.7 % fixed, -1.3 % default / SQP settings; status and iteration counts still equal to the oracle's)
#define SQPH_OFMA(a_, b_, c_) wg_fma((a_), (b_), (c_))
                    if (nown) x = SQPH_OFMA(alpha, (F32 ? (STACK ? (T)wgf_sum<C>(lf + 2 * L::O_STAGE_Y + (SOFF + t) * L::Cf) : reducef_xt(lf, t)) : (STACK ? wg_sum<C>(lds + L::O_STAGE_Y + (SOFF + t) * L::Cp) : reduce_xt(lds, t))), oma * x);
                    if (mown) {
                        const T zt = F32 ? reducef_over_c(lf, t) : reduce_over_c(lds, t);
                        if constexpr (CHECKS) ax = SQPH_OFMA(alpha, zt, oma * ax);
                        const T zr = SQPH_OFMA(alpha, zt, oma * z);
                        T zn = SQPH_OFMA(c_rinv, y, zr);
                        const T lo = c_lo, up = c_up;
                        zn = zn < lo ? lo : zn;  // cwiseMax(l) then cwiseMin(u), qp.cpp:278-281
                        zn = zn > up ? up : zn;
                        y = SQPH_OFMA(rho, zr - zn, y);
                        z = zn;
                    }

                    // operands of the next iteration (the barrier at the loop top orders them before the gathers)
                    if constexpr (F32 && STACK) {
                        if (mown) putf_rowv(lf, r, c, (float)(rho * SQPH_OFMA(-c_rinv, y, z)));
                        if (nown) lf[stacked_slot_f(SOFF + t)] = (float)SQPH_OFMA(sigma, x, -c_q);
                    } else if constexpr (F32) {
                        if (t < L::MP) putf_rowv(lf, r, c, mown ? (float)(rho * SQPH_OFMA(-c_rinv, y, z)) : 0.0f);
                        if (t < L::NR) putf_wrow(lf, r, c, nown ? (float)SQPH_OFMA(sigma, x, -c_q) : 0.0f);
                    } else if constexpr (STACK) {
                        if (mown) put_rowv(lds, r, c, rho * SQPH_OFMA(-c_rinv, y, z));
                        if (nown) lds[stacked_slot(SOFF + t)] = SQPH_OFMA(sigma, x, -c_q);
                    } else {
                        if (t < L::MP) put_rowv(lds, r, c, mown ? rho * SQPH_OFMA(-c_rinv, y, z) : T(0));
                        if (t < L::NR) put_wrow(lds, r, c, nown ? SQPH_OFMA(sigma, x, -c_q) : T(0));
                    }
#undef SQPH_OFMA
#ifndef SQPH_SIM
                    __builtin_amdgcn_s_setprio(0);  // (held through the loop-top barrier and the operand gathers instead: +0.6 % fixed, -0.6 % default)
#endif
                    SQPH_TICK(7)
                }
                iter += seg;
                if constexpr (CHECKS) {
                    bool check = false, adapt = false;
                    if (next_check > 0 && (next_check -= seg) == 0) {
                        check = true;
                        next_check = a.check_termination;
                    }
                    if (next_adapt > 0 && (next_adapt -= seg) == 0) {
                        adapt = true;
                        next_adapt = a.adaptive_rho_interval;
                    }
                    // A termination check whose PRIMAL test fails cannot end the solve, and nothing else of it is observable: the
                    // reference overwrites info.res_prim / res_dual at every check (qp.cpp:316-331) and the caller sees the values of
                    // the last one.  With A x kept by recurrence the primal residual costs two reductions and no memory traffic, so
                    // the dual half (A'y and P x: 60 KB streamed per QP at C3) is only computed when the primal test passes, when
                    // rho is to be adapted (the estimate needs both residuals), or at the last check a solve can reach (its
                    // residuals are what a MAX_ITER_EXCEEDED solve reports).
                    bool full_check = check || adapt;
                    if (check && !adapt && ax_valid && iter - 1 + a.check_termination <= a.max_iter) {
                        __syncthreads();
                        T vp[2] = {0, 0};  // nrm_prim | res_prim
                        if (mown) {
                            vp[0] = nanmax(tabs(ax), tabs(z));
                            vp[1] = tabs(ax - z);
                        }
                        T *red = lds + L::O_RED;
#pragma unroll
                        for (int e = 0; e < 2; e++) vp[e] = wave_nanmax(vp[e]);
                        if constexpr (NW > 1) {
                            if ((t & 63) == 0) {
#pragma unroll
                                for (int e = 0; e < 2; e++) red[e * NW + (t >> 6)] = vp[e];
                            }
                            __syncthreads();
#pragma unroll
                            for (int e = 0; e < 2; e++) {
                                T mval = red[e * NW];
#pragma unroll
                                for (int wv_ = 1; wv_ < NW; wv_++) mval = nanmax(mval, red[e * NW + wv_]);
                                vp[e] = mval;
                            }
                            __syncthreads();
                        }
                        if (!(vp[1] <= a.eps_abs + a.eps_rel * vp[0])) {  // (a NaN residual fails the test as it does in the full check)
                            info.res_prim = (double)vp[1];
                            full_check = false;
                        }
                    }
                    if (full_check) {
#ifndef SQPH_SIM
                        __builtin_amdgcn_s_setprio(2);  // a chain of memory round trips
#endif
                        // update_state + residuals, qp.cpp:316-331, 353-361.  A and P are streamed from global
                        // memory here (the register tiles hold B and W); this block runs every check_termination
                        // iterations only.
                        __syncthreads();
                        if (t < L::NP) put_colv(lds, t, nown ? x : T(0));
                        if (t < L::MP) put_rowv(lds, r, c, mown ? y : T(0));
                        __syncthreads();
                        {
                            T yr[TR];
                            get_rowv(lds, r, yr);
                            int n_c = n, m_c = m, r_c = r, c_c = c;
                            const TIN *gA_c = m > 0 ? gA : gP;  // m == 0: nothing to read, any readable address will do
                            SQPH_OPAQUE_S(n_c); SQPH_OPAQUE_S(m_c); SQPH_OPAQUE_V(r_c); SQPH_OPAQUE_V(c_c); SQPH_OPAQUE_S(gA_c);
                            stage_A_AT_gmem(gA_c, n_c, m_c, r_c, c_c, yr, lds);  // A x (over c) and A' y (over r)
                        }
                        __syncthreads();
                        // (a variant streaming for A'y alone when A x is known, three columns in flight, measured equal: full checks are rare now)
                        const T Ax_streamed = mown ? reduce_over_c(lds, t) : T(0);
                        const T Ax = ax_valid ? (mown ? ax : T(0)) : Ax_streamed;
                        const T ATy = nown ? reduce_over_r(lds, t) : T(0);
                        __syncthreads();
                        {
                            int n_c = n, r_c = r, c_c = c;
                            const TIN *gP_c = gP;
                            SQPH_OPAQUE_S(n_c); SQPH_OPAQUE_V(r_c); SQPH_OPAQUE_V(c_c); SQPH_OPAQUE_S(gP_c);
                            stage_P_gmem(gP_c, n_c, r_c, c_c, lds);  // full P (both triangles), as qp.cpp:324
                        }
                        __syncthreads();
                        const T Px = nown ? reduce_over_c(lds, t) : T(0);
                        __syncthreads();
                        // the seven norms of qp.cpp:316-331 as four reductions: max(|Ax|, |z|) and max(|Px|, |A'y|, |q|) are only used
                        // combined, so each lane combines its own terms first (a NaN-propagating max is order-free)
                        T v[4] = {0, 0, 0, 0};  // nrm_prim | res_prim | nrm_dual | res_dual
                        if (mown) {
                            v[0] = nanmax(tabs(Ax), tabs(z));
                            v[1] = tabs(Ax - z);
                        }
                        if (nown) {
                            const T q = qv[t];
                            v[2] = nanmax(tabs(Px), nanmax(tabs(ATy), tabs(q)));
                            v[3] = tabs(Px + q + ATy);
                        }
                        {   // workgroup-wide NaN-propagating max: butterfly inside each wave, NW values through LDS
                            T *red = lds + L::O_RED;
#pragma unroll
                            for (int e = 0; e < 4; e++) v[e] = wave_nanmax(v[e]);
                            if constexpr (NW > 1) {
                                if ((t & 63) == 0) {
#pragma unroll
                                    for (int e = 0; e < 4; e++) red[e * NW + (t >> 6)] = v[e];
                                }
                                __syncthreads();
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    T mval = red[e * NW];
#pragma unroll
                                    for (int wv_ = 1; wv_ < NW; wv_++) mval = nanmax(mval, red[e * NW + wv_]);
                                    v[e] = mval;
                                }
                                __syncthreads();
                            }
                        }
                        const T nrm_prim = v[0];
                        const T nrm_dual = v[2];
                        info.res_prim = (double)v[1];
                        info.res_dual = (double)v[3];
                        if (check) {
                            if (v[1] <= a.eps_abs + a.eps_rel * nrm_prim && v[3] <= a.eps_abs + a.eps_rel * nrm_dual) {
                                info.status = SQPH_SOLVED;
                                iter--;  // the iteration the test passed at (the segment loop has already counted past it)
                                break;
                            }
                        }
                        if (adapt) {
                            const T eps = a.regul;
                            const T rp_norm = v[1] / (nrm_prim + eps);
                            const T rd_norm = v[3] / (nrm_dual + eps);
                            T new_rho = rho_s * (T)sqrt((double)(rp_norm / (rd_norm + eps)));
                            new_rho = new_rho < a.rho_max ? new_rho : a.rho_max;
                            new_rho = new_rho > a.rho_min ? new_rho : a.rho_min;
                            info.rho_estimate = (double)new_rho;
                            if (new_rho < rho_s / a.rho_tol || new_rho > rho_s * a.rho_tol) {
                                rho_s = new_rho;
                                if (mown) {
                                    rho = rho_for_type<T>(sct[t], rho_s, a.rho_min, a.rho_eq_factor);  // type re-read from the state array (rare)
                                    rinvv[t] = T(1) / rho;
                                }
                                info.rho_updates += 1;
                                need_factor = true;
                                iter--;
                                break;  // leave the iteration loop WITHOUT advancing iter; the factor block does it
                            }
                        }
#ifndef SQPH_SIM
                        __builtin_amdgcn_s_setprio(0);
#endif
                        // the check borrowed the row-gather vector for y: publish w again for the next segment
                        if constexpr (F32 && STACK) {
                            if (mown) putf_rowv(lf, r, c, (float)(rho * (z - rinvv[t] * y)));
                            if (nown) lf[stacked_slot_f(SOFF + t)] = (float)(sigma * x - qv[t]);
                            for (int sg = t; sg < R * TR; sg += NT)
                                if ((sg >= m && sg < SOFF) || sg >= SOFF + n) lf[stacked_slot_f(sg)] = 0.0f;
                        } else if constexpr (F32) {
                            if (t < L::MP) putf_rowv(lf, r, c, mown ? (float)(rho * (z - rinvv[t] * y)) : 0.0f);
                        } else if constexpr (STACK) {
                            // the check's y vector ran over the stacked rows of the row-gather region: w, the u entries and the zeros again
                            if (mown) put_rowv(lds, r, c, rho * (z - rinvv[t] * y));
                            if (nown) lds[stacked_slot(SOFF + t)] = sigma * x - qv[t];
                            for (int sg = t; sg < R * TR; sg += NT)
                                if ((sg >= m && sg < SOFF) || sg >= SOFF + n) lds[stacked_slot(sg)] = T(0);
                        } else {
                            if (t < L::MP) put_rowv(lds, r, c, mown ? rho * (z - rinvv[t] * y) : T(0));
                        }
                    }
                }
            }
#ifdef SQPH_PHASE_TIMING
            if (t < 8) x = (T)tacc[t];           // debug build only: wave 0's phase ticks instead of x[0..8)
            if (t >= 64 && t < 72) y = (T)tacc[t - 64];  // wave 1's phase ticks in y[64..72)
#endif
            if (!need_factor) break;  // converged, exhausted, or no refactor pending
        }
        if (solving) {
            if (iter > a.max_iter) info.status = SQPH_MAX_ITER_EXCEEDED;
            info.iter = iter;
        }

#ifdef SQPH_SETUP_TIMING
        if (t < 8) x = (T)sqph_stk[t];
#endif
        if (state_dirty) {
            if (nown) sx[t] = x;
            if (mown) {
                sz[t] = z;
                sy[t] = y;
                srho[t] = rho;
            }
        }
        if (t == 0) {
            if constexpr (CHECKS) {
                a.info[qp] = info;
            } else {
                a.info[qp].status = info.status;
                a.info[qp].iter = info.iter;
                a.info[qp].rho_updates = info.rho_updates;
            }
            a.rho[qp] = rho_s;
        }
#if defined(SQPH_EXPERIMENTS) && !defined(SQPH_SIM)
        if (a.xdbg && (t & 63) == 0) a.xdbg[8 * qp + 4 * (t >> 6) + 2] = __builtin_amdgcn_s_memtime();
#endif
    }

    // ------------------------------------------------------------------ NW == 0: four QPs per wavefront
    // Small problems (n <= 4 TC <= 24, m <= 4 TR <= 48): a 16-lane 4 x 4 grid per QP, four independent QPs in one
    // wavefront (threadIdx.x >> 4 selects the QP and its LDS slice).  Same tiles, staging and formulas as run(); no
    // s_barrier anywhere (wsync() is a compiler fence), a lane owns the elements t, t+16, t+32 of the n- and m-vectors,
    // the four groups diverge freely (different iteration counts, refactorisations).
    static constexpr int GL = NT;  // lanes per QP: 16 (four QPs per wavefront) or 64 (one)
    static constexpr int NON = (L::NP + GL - 1) / GL, NOM = (L::MP + GL - 1) / GL;
    static constexpr int GTOTAL = L::TOTAL + L::NP + 3 * L::MP + (L::NP & 1);  // LDS doubles per QP slice (layout + iterates)
    static __device__ __forceinline__ int rowv_at(int i) { return L::O_ROWV + (i % R) * L::TRp + i / R; }
    static __device__ __forceinline__ int wrow_at(int i) { return L::O_WROW + (i % R) * L::TWp + i / R; }

    static __device__ __forceinline__ void run_group(const KArgs<T, TIN> &a, T *lds_block) {
        static_assert(NW == 0, "group kernel");
        const int t = threadIdx.x % GL;
        const int slot = threadIdx.x / GL;
        const int r = t % R, c = t / R;
        const int qp = blockIdx.x * (64 / GL) + slot;
        if (qp >= a.batch) return;  // a whole group leaves: nobody waits for it
        T *lds = lds_block + slot * GTOTAL;
        // the iterates of the owned elements live in LDS as well: with 75 doubles of tiles per lane there is no room for
        // them in registers (kept there, they were spilled to scratch inside the loop: 12 scratch round trips per iteration)
        T *xs = lds + L::TOTAL, *zs = xs + L::NP, *ys = zs + L::MP, *rhos = ys + L::MP;
        const int n = a.n, m = a.m;
        // per-QP pointers are lane-varying here (four QPs per wave): derived where they are used instead of being kept
        // in 22 VGPRs for the whole solve
#define SQPH_GP (a.P + (long)qp * a.sP)
#define SQPH_GA (a.A + (long)qp * a.sA)
#define SQPH_GW (a.Sinv + (long)qp * 2 * n * n)
#define sx (a.x + (long)qp * n)
#define sz (a.z + (long)qp * m)
#define sy (a.y + (long)qp * m)
#define srho (a.rho_vec + (long)qp * m)
#define sct (a.ctype + (long)qp * m)

        // status / iter / rho_updates travel in registers; the three doubles of the record (written at checks only) are
        // stored straight to the info array
        sqph_info info;
        info.status = a.info[qp].status;
        info.iter = a.info[qp].iter;
        info.rho_updates = a.info[qp].rho_updates;
        T rho_s = a.rho[qp];
        const int mode = a.mode;
        if (!(mode & (MODE_SETUP | MODE_UPDATE)) && (info.status == SQPH_UNINITIALIZED || info.status == SQPH_NUMERICAL_ISSUES))
            return;  // qp.cpp:68-71

        const T INF = T(1) / T(0);
        T *qv = lds + L::O_QV, *lov = lds + L::O_LOV, *upv = lds + L::O_UPV, *rinvv = lds + L::O_RINV;
#pragma unroll
        for (int k = 0; k < NON; k++) {
            const int j = t + GL * k;
            if (j < L::NP) qv[j] = j < n ? (T)(a.q + (long)qp * a.sq)[j] : T(0);
        }
#pragma unroll
        for (int k = 0; k < NOM; k++) {
            const int i = t + GL * k;
            if (i < L::MP) {
                lov[i] = i < m ? (T)(a.l + (long)qp * a.sl)[i] : -INF;
                upv[i] = i < m ? (T)(a.u + (long)qp * a.su)[i] : INF;
                rinvv[i] = T(1);
            }
        }
        wsync();
#pragma unroll
        for (int k = 0; k < NON; k++) {
            const int j = t + GL * k;
            if (j < L::NP) xs[j] = 0;
        }
#pragma unroll
        for (int k = 0; k < NOM; k++) {
            const int i = t + GL * k;
            if (i < L::MP) {
                zs[i] = 0;
                ys[i] = 0;
                rhos[i] = T(1);
            }
        }
        if (mode & (MODE_SETUP | MODE_UPDATE)) {
            rho_s = a.rho0;
#pragma unroll
            for (int k = 0; k < NOM; k++) {
                const int i = t + GL * k;
                if (i < m) {
                    const T lo = lov[i], up = upv[i];
                    int ctype = SQPH_INEQUALITY_CONSTRAINT;
                    if (lo < -a.loose_thresh && up > a.loose_thresh)
                        ctype = SQPH_LOOSE_BOUNDS;
                    else if (up - lo < a.eq_tol)
                        ctype = SQPH_EQUALITY_CONSTRAINT;
                    rhos[i] = rho_for_type<T>(ctype, rho_s, a.rho_min, a.rho_eq_factor);
                    rinvv[i] = T(1) / rhos[i];
                    sct[i] = ctype;
                    srho[i] = rhos[i];
                }
            }
            info.rho_updates += 1;
        }
        if (!(mode & MODE_SETUP)) {
#pragma unroll
            for (int k = 0; k < NON; k++) {
                const int j = t + GL * k;
                if (j < n) xs[j] = sx[j];
            }
#pragma unroll
            for (int k = 0; k < NOM; k++) {
                const int i = t + GL * k;
                if (i < m) {
                    zs[i] = sz[i];
                    ys[i] = sy[i];
                    if (!(mode & MODE_UPDATE)) {
                        rhos[i] = srho[i];
                        rinvv[i] = T(1) / rhos[i];
                    }
                }
            }
        }

        T vt[TW][TC];
        T at[TR][TC];
        bool need_factor = (mode & (MODE_SETUP | MODE_UPDATE | MODE_REFACTOR)) != 0;
        bool solving = false;
        bool state_dirty = (mode & MODE_SETUP) != 0;
        bool have_A = false;
        const T alpha = a.alpha, sigma = a.sigma, oma = T(1) - a.alpha;
        int iter = 1;
        int next_check = a.check_termination > 0 ? a.check_termination : -1;
        int next_adapt = (a.adaptive_rho && a.adaptive_rho_interval > 0) ? a.adaptive_rho_interval : -1;
        for (;;) {
            {  // ---- set-up part of a pass (the W tile is local to it)
#ifndef SQPH_SIM
            __builtin_amdgcn_s_setprio(2);  // as in run(): -0.9 % at C2
#endif
            T wt[TW][TC];
            if (!need_factor) load_sq_tile<T>(SQPH_GW, n, r, c, wt);
            if (need_factor) {
                wsync();
#pragma unroll
                for (int k = 0; k < NOM; k++) {
                    const int i = t + GL * k;
                    if (i < L::MP) lds[L::O_RHO + i] = i < m ? rhos[i] : T(0);
                }
                wsync();
                bool ok;
                {
                    int n_f = n, m_f = m, r_f = r, c_f = c, t_f = t, qp_f = qp;
                    SQPH_OPAQUE_S(n_f); SQPH_OPAQUE_S(m_f); SQPH_OPAQUE_V(r_f); SQPH_OPAQUE_V(c_f); SQPH_OPAQUE_V(t_f); SQPH_OPAQUE_V(qp_f);
                    load_A_tile(a.A + (long)qp_f * a.sA, n_f, m_f, r_f, c_f, at);
                    ok = factor(a.P + (long)qp_f * a.sP, at, n_f, m_f, sigma, lds, t_f, r_f, c_f, wt, false);
                    if (!(mode & MODE_NO_FACTOR_STORE)) store_sq_tile(a.Sinv + (long)qp_f * 2 * n_f * n_f, n_f, r_f, c_f, wt);
                }
                wsync();
                need_factor = false;
                have_A = true;
                if (!solving) {
                    if (mode & (MODE_SETUP | MODE_UPDATE)) info.status = ok ? SQPH_UNSOLVED : SQPH_NUMERICAL_ISSUES;
                    else if (!ok) info.status = SQPH_NUMERICAL_ISSUES;
                } else if (!ok) {
                    info.status = SQPH_NUMERICAL_ISSUES;
                    break;
                } else {
                    iter++;
                }
            }
            if (!(mode & MODE_SOLVE) || info.status == SQPH_NUMERICAL_ISSUES || info.status == SQPH_UNINITIALIZED) break;
            if (!solving) {
                solving = true;
                state_dirty = true;
                if ((mode & MODE_COLD_RESET) && !a.warm_start) {
#pragma unroll
                    for (int k = 0; k < NON; k++) {
                        const int j = t + GL * k;
                        if (j < L::NP) xs[j] = 0;
                    }
#pragma unroll
                    for (int k = 0; k < NOM; k++) {
                        const int i = t + GL * k;
                        if (i < L::MP) zs[i] = ys[i] = 0;
                    }
                }
            }
            if (!have_A) {
                int n_t = n, m_t = m, r_t = r, c_t = c, qp_t = qp;
                SQPH_OPAQUE_S(n_t); SQPH_OPAQUE_S(m_t); SQPH_OPAQUE_V(r_t); SQPH_OPAQUE_V(c_t); SQPH_OPAQUE_V(qp_t);
                load_A_tile(a.A + (long)qp_t * a.sA, n_t, m_t, r_t, c_t, at);
            }
            have_A = false;
            {
                int n_t = n, r_t = r, c_t = c;
                SQPH_OPAQUE_S(n_t); SQPH_OPAQUE_V(r_t); SQPH_OPAQUE_V(c_t);
                build_B_inplace<false>(at, wt, n_t, lds, r_t, c_t, m);
            }
#ifndef SQPH_SIM
            __builtin_amdgcn_s_setprio(0);
#endif
            }  // ---- end of the set-up part
            {
                int n_t = n, r_t = r, c_t = c;
                SQPH_OPAQUE_S(n_t); SQPH_OPAQUE_V(r_t); SQPH_OPAQUE_V(c_t);
                load_vt_lds(lds, n_t, r_t, c_t, vt);
            }
            T (&bt)[TR][TC] = at;
#define SQPH_G_PUBLISH()                                                                                         \
    {                                                                                                            \
        _Pragma("unroll") for (int k = 0; k < NOM; k++) {                                                        \
            const int i = t + GL * k;                                                                            \
            if (i < L::MP) lds[rowv_at(i)] = i < m ? rhos[i] * (zs[i] - rinvv[i] * ys[i]) : T(0);                   \
        }                                                                                                        \
        _Pragma("unroll") for (int k = 0; k < (L::NR + GL - 1) / GL; k++) {                                      \
            const int j = t + GL * k;                                                                            \
            if (j < L::NR) lds[wrow_at(j)] = j < n ? sigma * xs[j] - qv[j] : T(0);                                \
        }                                                                                                        \
    }
            SQPH_G_PUBLISH()
            for (; iter <= a.max_iter; iter++) {
                wsync();
                {
                    T w[TR], ur[TW];
                    get_rowv(lds, r, w);
                    get_wrow(lds, r, ur);
                    stage1<TW>(bt, vt, w, ur, lds, r, c);
                }
                wsync();
#pragma unroll
                for (int k = 0; k < NON; k++) {
                    const int sj = t + GL * k;                     // slot of the output ...
                    const int j = C * (sj % TC) + sj / TC;         // ... and its column (slot TC c + k <-> column C k + c)
                    if (sj < L::NP) put_colv2(lds, j, j < n ? wg_sum<R>(lds + L::O_STAGE + sj * L::Rp) : T(0));
                }
                wsync();
                {
                    T y1c[TC];
                    get_colv2(lds, c, y1c);
                    stage2<TW>(bt, vt, y1c, lds, r, c);
                }
                wsync();
#pragma unroll
                for (int k = 0; k < NON; k++) {
                    const int j = t + GL * k;
                    if (j < n) xs[j] = alpha * reduce_xt(lds, j) + oma * xs[j];
                }
#pragma unroll
                for (int k = 0; k < NOM; k++) {
                    const int i = t + GL * k;
                    if (i < m) {
                        const T zt = wg_sum<C>(lds + L::O_STAGE_Y + i * L::Cp);
                        const T zr = alpha * zt + oma * zs[i];
                        T zn = zr + rinvv[i] * ys[i];
                        const T lo = lov[i], up = upv[i];
                        zn = zn < lo ? lo : zn;
                        zn = zn > up ? up : zn;
                        ys[i] = ys[i] + rhos[i] * (zr - zn);
                        zs[i] = zn;
                    }
                }
                bool check = false, adapt = false;
                if (--next_check == 0) {
                    check = true;
                    next_check = a.check_termination;
                }
                if (--next_adapt == 0) {
                    adapt = true;
                    next_adapt = a.adaptive_rho_interval;
                }
                if (check || adapt) {
                    wsync();
#pragma unroll
                    for (int k = 0; k < NON; k++) {
                        const int j = t + GL * k;
                        if (j < L::NP) put_colv(lds, j, j < n ? xs[j] : T(0));
                    }
#pragma unroll
                    for (int k = 0; k < NOM; k++) {
                        const int i = t + GL * k;
                        if (i < L::MP) lds[rowv_at(i)] = i < m ? ys[i] : T(0);
                    }
                    wsync();
                    {
                        T yr[TR];
                        get_rowv(lds, r, yr);
                        int n_c = n, m_c = m, r_c = r, c_c = c, qp_c = qp;
                        SQPH_OPAQUE_S(n_c); SQPH_OPAQUE_S(m_c); SQPH_OPAQUE_V(r_c); SQPH_OPAQUE_V(c_c); SQPH_OPAQUE_V(qp_c);
                        stage_A_AT_gmem(m_c > 0 ? a.A + (long)qp_c * a.sA : a.P + (long)qp_c * a.sP, n_c, m_c, r_c, c_c, yr, lds);
                    }
                    wsync();
                    T Ax[NOM], ATy[NON], Px[NON];
#pragma unroll
                    for (int k = 0; k < NOM; k++) {
                        const int i = t + GL * k;
                        Ax[k] = i < m ? wg_sum<C>(lds + L::O_STAGE_Y + i * L::Cp) : T(0);
                    }
#pragma unroll
                    for (int k = 0; k < NON; k++) {
                        const int j = t + GL * k;
                        ATy[k] = j < n ? wg_sum<R>(lds + L::O_STAGE + L::xslot(j) * L::Rp) : T(0);
                    }
                    wsync();
                    {
                        int n_c = n, r_c = r, c_c = c, qp_c = qp;
                        SQPH_OPAQUE_S(n_c); SQPH_OPAQUE_V(r_c); SQPH_OPAQUE_V(c_c); SQPH_OPAQUE_V(qp_c);
                        stage_P_gmem(a.P + (long)qp_c * a.sP, n_c, r_c, c_c, lds);
                    }
                    wsync();
#pragma unroll
                    for (int k = 0; k < NON; k++) {
                        const int j = t + GL * k;
                        Px[k] = j < n ? wg_sum<C>(lds + L::O_STAGE_Y + j * L::Cp) : T(0);
                    }
                    wsync();
                    T v[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int k = 0; k < NOM; k++) {
                        const int i = t + GL * k;
                        if (i < m) {
                            v[0] = nanmax(v[0], tabs(Ax[k]));
                            v[1] = nanmax(v[1], tabs(zs[i]));
                            v[2] = nanmax(v[2], tabs(Ax[k] - zs[i]));
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NON; k++) {
                        const int j = t + GL * k;
                        if (j < n) {
                            const T q = qv[j];
                            v[3] = nanmax(v[3], tabs(Px[k]));
                            v[4] = nanmax(v[4], tabs(ATy[k]));
                            v[5] = nanmax(v[5], tabs(q));
                            v[6] = nanmax(v[6], tabs(Px[k] + q + ATy[k]));
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 7; e++) {
                        if constexpr (GL == 16) v[e] = group16_nanmax(v[e]);
                        else if constexpr (GL == 32) v[e] = group32_nanmax(v[e]);
                        else v[e] = wave_nanmax(v[e]);
                    }
                    const T nrm_prim = nanmax(v[0], v[1]);
                    const T nrm_dual = nanmax(v[3], nanmax(v[4], v[5]));
                    if (t == 0) {
                        a.info[qp].res_prim = (double)v[2];
                        a.info[qp].res_dual = (double)v[6];
                    }
                    if (check) {
                        if (v[2] <= a.eps_abs + a.eps_rel * nrm_prim && v[6] <= a.eps_abs + a.eps_rel * nrm_dual) {
                            info.status = SQPH_SOLVED;
                            break;
                        }
                    }
                    if (adapt) {
                        const T eps = a.regul;
                        const T rp_norm = v[2] / (nrm_prim + eps);
                        const T rd_norm = v[6] / (nrm_dual + eps);
                        T new_rho = rho_s * (T)sqrt((double)(rp_norm / (rd_norm + eps)));
                        new_rho = new_rho < a.rho_max ? new_rho : a.rho_max;
                        new_rho = new_rho > a.rho_min ? new_rho : a.rho_min;
                        if (t == 0) a.info[qp].rho_estimate = (double)new_rho;
                        if (new_rho < rho_s / a.rho_tol || new_rho > rho_s * a.rho_tol) {
                            rho_s = new_rho;
#pragma unroll
                            for (int k = 0; k < NOM; k++) {
                                const int i = t + GL * k;
                                if (i < m) {
                                    rhos[i] = rho_for_type<T>(sct[i], rho_s, a.rho_min, a.rho_eq_factor);
                                    rinvv[i] = T(1) / rhos[i];
                                }
                            }
                            info.rho_updates += 1;
                            need_factor = true;
                            break;
                        }
                    }
                }
                SQPH_G_PUBLISH()
            }
#undef SQPH_G_PUBLISH
            if (!need_factor) break;
        }
        if (solving) {
            if (iter > a.max_iter) info.status = SQPH_MAX_ITER_EXCEEDED;
            info.iter = iter;
        }
        if (state_dirty) {
#pragma unroll
            for (int k = 0; k < NON; k++) {
                const int j = t + GL * k;
                if (j < n) sx[j] = xs[j];
            }
#pragma unroll
            for (int k = 0; k < NOM; k++) {
                const int i = t + GL * k;
                if (i < m) {
                    sz[i] = zs[i];
                    sy[i] = ys[i];
                    srho[i] = rhos[i];
                }
            }
        }
        if (t == 0) {
            a.info[qp].status = info.status;
            a.info[qp].iter = info.iter;
            a.info[qp].rho_updates = info.rho_updates;
            a.rho[qp] = rho_s;
        }
#undef SQPH_GP
#undef SQPH_GA
#undef SQPH_GW
#undef sx
#undef sz
#undef sy
#undef srho
#undef sct
    }
};

// WPE = waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument)
template <typename TIN, int NW, int R, int C, int TR, int TC, int TW, int WPE>
__global__ __launch_bounds__(64 * NW, WPE) void admm_wg_kernel(KArgs<double, TIN> a) {
    __shared__ __attribute__((aligned(16))) double lds[WgLayout<NW, R, C, TR, TC, TW>::TOTAL];
#ifdef SQPH_SIM
    ::sqph_sim::poison_static_lds(lds, sizeof(lds));
#endif
    WgKernel<TIN, NW, R, C, TR, TC, TW>::template run<true>(a, lds);
}
// the same without the residual-check block (see WgKernel::run); instantiated in wg_nocheck.hip only
template <typename TIN, int NW, int R, int C, int TR, int TC, int TW, int WPE>
__global__ __launch_bounds__(64 * NW, WPE) void admm_wg_nocheck_kernel(KArgs<double, TIN> a) {
    __shared__ __attribute__((aligned(16))) double lds[WgLayout<NW, R, C, TR, TC, TW>::TOTAL];
#ifdef SQPH_SIM
    ::sqph_sim::poison_static_lds(lds, sizeof(lds));
#endif
    WgKernel<TIN, NW, R, C, TR, TC, TW>::template run<false>(a, lds);
}

// waves per SIMD of the no-check instantiation where it can hold more than the shape's checking kernel: the C2 shape (one wave per QP)
// fits four per SIMD without the check block — all 4,096 QPs of BASELINE configs[1] resident at once instead of 3,072 (0.345 -> 0.317 ms;
// the checking kernel spills at that bound and stays at three: 0.78 against 0.84 ms under the reference's default settings)
constexpr int wg_nocheck_wpe(int NW, int R, int C, int TR, int TC, int W) {
    return (NW == 1 && R == 8 && C == 8 && TR == 5 && TC == 3) ? 4 : (NW == 4 && R == 16 && C == 16 && TR == 8 && TC == 4) ? 3 : W;
}

// the stacked operator (WgKernel::run<CHECKS, false, STACK = true>) for problems with m <= R (TR + TW - 1) - C TC (WgKernel::SOFF); instantiated in
// wg_stack.hip only
template <typename TIN, int NW, int R, int C, int TR, int TC, int TW, int WPE>
__global__ __launch_bounds__(64 * NW, WPE) void admm_wgs_kernel(KArgs<double, TIN> a) {
    __shared__ __attribute__((aligned(16))) double lds[WgLayout<NW, R, C, TR, TC, TW>::TOTAL];
#ifdef SQPH_SIM
    ::sqph_sim::poison_static_lds(lds, sizeof(lds));
#endif
    WgKernel<TIN, NW, R, C, TR, TC, TW>::template run<true, false, true>(a, lds);
}
template <typename TIN, int NW, int R, int C, int TR, int TC, int TW, int WPE>
__global__ __launch_bounds__(64 * NW, WPE) void admm_wgs_nocheck_kernel(KArgs<double, TIN> a) {
    __shared__ __attribute__((aligned(16))) double lds[WgLayout<NW, R, C, TR, TC, TW>::TOTAL];
#ifdef SQPH_SIM
    ::sqph_sim::poison_static_lds(lds, sizeof(lds));
#endif
    WgKernel<TIN, NW, R, C, TR, TC, TW>::template run<false, false, true>(a, lds);
}
// shapes of the stacked variant {NW, R, C, TR, TC, TW, WPE}: the C3 shape (m <= 104, n <= 56)
#define SQPH_WGS_SHAPES(X) X(2, 16, 8, 7, 7, 4, 2)

// fp32 products (SQPH_FLAG_F32_ARITH with QPSolver<float>): the same kernels with the iteration's two stages in single precision;
// instantiated in wg_f32.hip only
template <typename TIN, int NW, int R, int C, int TR, int TC, int TW, int WPE, bool STACK = false>
__global__ __launch_bounds__(64 * NW, WPE) void admm_wgf_kernel(KArgs<double, TIN> a) {
    __shared__ __attribute__((aligned(16))) double lds[WgLayout<NW, R, C, TR, TC, TW>::TOTAL];
#ifdef SQPH_SIM
    ::sqph_sim::poison_static_lds(lds, sizeof(lds));
#endif
    WgKernel<TIN, NW, R, C, TR, TC, TW>::template run<true, true, STACK>(a, lds);
}
template <typename TIN, int NW, int R, int C, int TR, int TC, int TW, int WPE, bool STACK = false>
__global__ __launch_bounds__(64 * NW, WPE) void admm_wgf_nocheck_kernel(KArgs<double, TIN> a) {
    __shared__ __attribute__((aligned(16))) double lds[WgLayout<NW, R, C, TR, TC, TW>::TOTAL];
#ifdef SQPH_SIM
    ::sqph_sim::poison_static_lds(lds, sizeof(lds));
#endif
    WgKernel<TIN, NW, R, C, TR, TC, TW>::template run<false, true, STACK>(a, lds);
}
// shapes of the fp32-product variant {NW, R, C, TR, TC, TW, WPE}: the BASELINE dense shapes (20,40) and (50,100)
#define SQPH_WGF_SHAPES(X)   \
    X(1, 8, 8, 5, 3, 3, 3)   \
    X(2, 16, 8, 7, 7, 4, 2)

// four QPs per wavefront (run_group): block = one wavefront, LDS = 4 slices
template <typename TIN, int TR, int TC, int WPE>
__global__ __launch_bounds__(64, WPE) void admm_g16_kernel(KArgs<double, TIN> a) {
    __shared__ __attribute__((aligned(16))) double lds[4 * WgKernel<TIN, 0, 4, 4, TR, TC, TC>::GTOTAL];
#ifdef SQPH_SIM
    ::sqph_sim::poison_static_lds(lds, sizeof(lds));
#endif
    WgKernel<TIN, 0, 4, 4, TR, TC, TC>::run_group(a, lds);
}
// (the same path with one QP per wavefront — 8 x 8 grid, TR = 13, TC = 7, 481 VGPRs, one wave per SIMD — measured
// 4.99 ms on the C3 shard against 3.72 ms for the two-waves-per-QP workgroup kernel: without a second wave per SIMD
// nothing hides the LDS round trips)
// (two QPs per wavefront — an 8 x 4 grid of 32 lanes per QP, `admm_g32_kernel`, rounds 1-2 — is retired: at its one shape, n <= 20 and
// m <= 40 with fixed iteration counts, the one-wave-per-QP kernel is the faster one since round 3 (wave priorities): 0.332 against
// 0.347 ms per 4,096 x 200 iterations, 1.19 against 1.30 ms per 16,384)

// shapes compiled into the library: {NW, R, C, TR, TC, TW, WPE}; first fit (m <= R*TR, n <= C*TC) wins.  The 32 x 8 grids are for
// problems with many more constraints than variables (m <= 224 with n <= 16 / 32 / 56): measured 4,096 x (10,150) 1.43 ms against
// 7.73 ms in the 16 x 16 / 13 x 7 shape it fell into before, 4,096 x (50,150) 2.94 against 8.68 ms; the 64 x 8 grids (8 waves) carry
// m <= 448 with n <= 32 / 56: 2,048 x (50,400) 3.98 ms against 45.5 ms in the generic kernel; the 16 x 16 grids with 2 / 4 / 8 tile rows
// serve 56 < n <= 112 with m <= 32 / 64 / 128 (fewer products per iteration than the 13-row shape: 1.95x / 1.7x / 1.36x).  Until round 4
// their staged copy of W (112 rows of 128 doubles) took 115 KB of LDS, i.e. one workgroup per CU, one wave per SIMD; packed to its lower
// tile-triangle (WgLayout::WPACK) the four 16 x 16 / 7 x 7 shapes take 74-78 KB: two workgroups per CU, WPE 2 — 2,048 x (100,100)
// 2.92 -> 1.86 ms, (100,30) 2.11 -> 1.29 — and 1.58 / 1.14 ms with the MFMA set-up in swizzled blocks that replaced the packed copy at
// the end of the round (WgLayout::MSX, admm_wg_msetup.h: 81.7 KB, still two per CU).  m <= 224 with 56 < n <= 112 takes a 32 x 16 grid of eight waves (7 x 7 + 4 x 7 doubles of
// tiles per lane, two waves per SIMD) since round 4: the four-wave 16 x 16 grid with 13 x 7 + 7 x 7 doubles per lane that served it
// ran one wave per SIMD with the AGPRs as spill space (bound to 256 registers it spilled into the loop, 4.05 -> 8.4 ms):
// 2,048 x (100,200) 4.01 -> 3.65 ms, under the default settings 4.33 -> 3.06; 3.01 / 2.61 ms with the MFMA set-up, WgLayout::MSR)
// n <= 32 with 64 < m <= 128 has a 16 x 8 grid of its own since round 4 (8 x 4 + 2 x 4 doubles of tiles per lane, three waves per SIMD,
// MFMA set-up): until then it ran in the C3 grid padded to 56 columns or, beyond m = 112, in the four-wave 16 x 16 grid —
// 4,096 x (24,96) 1.11 -> 0.75 ms, 4,096 x (32,128) ~1.5 -> 0.79 ms (tools/xp/shape_sweep.py found the gap)
// (SQPH_SLIM: experiment builds with the C3 shape only — seconds instead of minutes to compile; never shipped)
#ifdef SQPH_SLIM
#ifdef SQPH_SLIM_C2  // ... plus the C2 shape
#define SQPH_WG_SHAPES(X) X(1, 8, 8, 5, 3, 3, 3) X(2, 16, 8, 7, 7, 4, 2)
#elif defined(SQPH_SLIM_SHAPES)  // ... or a list given on the command line: -D'SQPH_SLIM_SHAPES(X)=X(4,32,8,7,4,1,3) X(...)'
#define SQPH_WG_SHAPES(X) SQPH_SLIM_SHAPES(X)
#elif defined(SQPH_SLIM_W4)  // ... plus the four-wave 16 x 16 shape
#define SQPH_WG_SHAPES(X) X(2, 16, 8, 7, 7, 4, 2) X(4, 16, 16, 8, 4, 4, 2)
#else
#define SQPH_WG_SHAPES(X) X(2, 16, 8, 7, 7, 4, 2)
#endif
#define SQPH_G16_SHAPES(X)
#else
#define SQPH_WG_SHAPES(X)        \
    X(1, 8, 8, 1, 1, 1, 4)       \
    X(1, 8, 8, 3, 2, 2, 4)       \
    X(1, 8, 8, 5, 3, 3, 3)       \
    X(1, 8, 8, 8, 4, 4, 2)       \
    X(2, 16, 8, 8, 4, 2, 3)      \
    X(2, 16, 8, 7, 7, 4, 2)      \
    X(4, 16, 16, 8, 4, 4, 2)     \
    X(4, 32, 8, 7, 2, 1, 4)      \
    X(4, 32, 8, 7, 4, 1, 3)      \
    X(4, 32, 8, 7, 7, 2, 2)      \
    X(4, 16, 16, 2, 7, 7, 2)     \
    X(4, 16, 16, 4, 7, 7, 2)     \
    X(4, 16, 16, 8, 7, 7, 2)     \
    X(8, 32, 16, 7, 7, 4, 2)     \
    X(8, 64, 8, 7, 4, 1, 2)      \
    X(8, 64, 8, 7, 7, 1, 2)

// shapes {TR, TC, WPE}: m <= 4 TR, n <= 4 TC; first fit wins
#define SQPH_G16_SHAPES(X) \
    X(1, 1, 4)             \
    X(3, 2, 4)             \
    X(6, 3, 2)
#endif  // SQPH_SLIM

#ifdef SQPH_SIM
template <typename TIN>
inline int sim_run_g16(const KArgs<double, TIN> &a) {
#define SQPH_SIM_CASE(TR_, TC_, W_)                                                                        \
    if (a.m <= 4 * TR_ && a.n <= 4 * TC_) {                                                                \
        ::sqph_sim::launch(admm_g16_kernel<TIN, TR_, TC_, W_>, dim3((a.batch + 3) / 4), dim3(64), 0, a);   \
        return 0;                                                                                          \
    }
    SQPH_G16_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

#ifdef SQPH_SIM
template <typename TIN>
inline int sim_run_wgs(const KArgs<double, TIN> &a) {
#define SQPH_SIM_CASE(NW_, R_, C_, TR_, TC_, TW_, W_)                                                         \
    if (a.m <= R_ * TR_ && a.n <= C_ * TC_ && a.m <= R_ * (TR_ + TW_ - 1) - C_ * TC_) {                            \
        if (a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0))                      \
            ::sqph_sim::launch(admm_wgs_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_>, dim3(a.batch), dim3(64 * NW_), 0, a); \
        else                                                                                                  \
            ::sqph_sim::launch(admm_wgs_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_>, dim3(a.batch), dim3(64 * NW_), 0, a); \
        return 0;                                                                                             \
    }
    SQPH_WGS_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

#ifdef SQPH_SIM
template <typename TIN>
inline int sim_run_wgf(const KArgs<double, TIN> &a) {
    const bool nochk = a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0);
#define SQPH_SIM_CASE(NW_, R_, C_, TR_, TC_, TW_, W_)                                                         \
    if (a.m <= R_ * TR_ && a.n <= C_ * TC_) {                                                                 \
        if (NW_ == 2 && a.m <= R_ * (TR_ + TW_ - 1) - C_ * TC_) {  /* the stacked operator, as the host dispatch */ \
            if (nochk) ::sqph_sim::launch(admm_wgf_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_, NW_ == 2>, dim3(a.batch), dim3(64 * NW_), 0, a); \
            else ::sqph_sim::launch(admm_wgf_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_, NW_ == 2>, dim3(a.batch), dim3(64 * NW_), 0, a); \
        } else if (nochk)                                                                                     \
            ::sqph_sim::launch(admm_wgf_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_>, dim3(a.batch), dim3(64 * NW_), 0, a); \
        else                                                                                                  \
            ::sqph_sim::launch(admm_wgf_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_>, dim3(a.batch), dim3(64 * NW_), 0, a); \
        return 0;                                                                                             \
    }
    SQPH_WGF_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

#ifdef SQPH_SIM
template <typename TIN>
inline int sim_run_wg(const KArgs<double, TIN> &a) {
#define SQPH_SIM_CASE(NW_, R_, C_, TR_, TC_, TW_, W_)                                                         \
    if (a.m <= R_ * TR_ && a.n <= C_ * TC_) {                                                                 \
        if (a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0))                      \
            ::sqph_sim::launch(admm_wg_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_>, dim3(a.batch), dim3(64 * NW_), 0, a); \
        else                                                                                                  \
            ::sqph_sim::launch(admm_wg_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_>, dim3(a.batch), dim3(64 * NW_), 0, a); \
        return 0;                                                                                             \
    }
    SQPH_WG_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

}  // namespace sqph
