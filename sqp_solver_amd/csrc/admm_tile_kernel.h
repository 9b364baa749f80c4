// Register-tiled ADMM kernel: ONE 64-lane wavefront per QP, everything on-chip.
//
// Lane grid 8 x 8 (lane = 8*c + r).  Lane (r,c) keeps the A tile
//        a[s][t] = A[8*s + r][TC*c + t]      s < TR, t < TC          (rows cyclic, columns blocked)
// in VGPRs for the whole solve, so BOTH products of an ADMM iteration use the same registers:
//        A' w : pb[t] = sum_s a[s][t] * w[8s+r]   -> reduce-scatter over the 8 lanes sharing c
//        A x~ : pz[s] = sum_t a[s][t] * x~[TC c+t] -> reduce-scatter over the 8 lanes sharing r
// The Schur factor W (n x n lower triangular, S^-1 = W'W; applied as x~ = W'(W b)) is tiled the same way (lane (r,c) holds rows TC*c'+r, c'=0..7, and
// columns TC*c+t): in VGPRs for small tiles, in LDS ([element][lane], conflict-free) for large ones.
// Vectors live "scattered" (each element owned by exactly one lane: m-index i = lane + 64k,
// n-index j = TC*c + r for r < TC) and are all-gathered / reduce-scattered with whole-register
// butterflies (wave_ops.h) — no LDS traffic and no barriers inside the iteration.
//
// Numerics: identical update formulas to admm_generic.h (reference src/qp.cpp:84-144 on the
// Schur-ordered system), fp64 arithmetic; inputs may be fp32 or fp64 (TIN).
#pragma once
#include "block_ops.h"
#include "kargs.h"
#include "wave_ops.h"

// Opaque copies of loop-invariant values. The factor block, the tile load and the residual check
// sit inside the solve loops but run rarely; without this LICM hoists all their index/address/mask
// arithmetic (hundreds of VGPRs for a 8 x TC tile) to the kernel prologue and keeps it alive across
// the hot iteration loop. A value that went through an empty asm is not provably invariant.
#ifdef SQPH_SIM
#define SQPH_OPAQUE_S(x) (void)0
#define SQPH_OPAQUE_V(x) (void)0
#else
#define SQPH_OPAQUE_S(x) asm volatile("" : "+s"(x))
#define SQPH_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif

namespace sqph {

// one lane's view of a scattered m-vector: NS slots, slot k <-> index lane + 64k
template <typename T, int NS>
struct MVec {
    T v[NS];
};

// ---------------------------------------------------------------------------------------------
// Collectives through LDS (XLDS = true).  Measured on gfx950 (tools/ubench): a lone wave issues one
// VALU instruction per ~5 cycles whatever its type, so the register butterflies of wave_ops.h
// (~45-50 VALU instructions per 8-value all-gather / reduce-scatter of doubles) dominated the
// iteration.  The same exchanges cost 5-20 DS instructions when staged through a few hundred bytes
// of LDS.  One wave owns its scratch, DS operations of a wave execute in order, so no s_barrier is
// involved: __syncthreads() in a 64-thread workgroup is only a compiler-level fence here.
// Layout strides are chosen so that ds_write_b64 / ds_read_b128 are bank-conflict free.
// ---------------------------------------------------------------------------------------------
typedef double sqph_d2 __attribute__((vector_size(16)));

template <int TR>
struct XlLayout {
    static constexpr int ROWS_STRIDE = 18;  // per r; 2*18 dwords == 4 (mod 32)
    static constexpr int O_ROWS = 0;        // [8][18]
    static constexpr int O_COLS = 144;      // [64]
    static constexpr int O_BLK = 208;       // [8][10]
    static constexpr int O_RS = 288;        // reduce-scatter staging, shared
    static constexpr int RS_COLS_STRIDE = 72;  // per c:  [8 t][8 r']
    static constexpr int RS_BLK_STRIDE = 66;   // per r:  [8 c'][8 c'']
    static constexpr int RS_ROWS_STRIDE = ((8 * TR + 13) / 16) * 16 + 2;  // per r: [TR s][8 c''], == 2 (mod 16)
    static constexpr int RS_SIZE = (8 * RS_ROWS_STRIDE > 8 * RS_COLS_STRIDE) ? 8 * RS_ROWS_STRIDE : 8 * RS_COLS_STRIDE;
    static constexpr int TOTAL = O_RS + RS_SIZE;
};

// 8 contiguous doubles (16-byte aligned) from LDS
__device__ __forceinline__ void lds_read8(const double *p, double (&v)[8]) {
    const sqph_d2 *q = reinterpret_cast<const sqph_d2 *>(__builtin_assume_aligned(p, 16));
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const sqph_d2 t = q[k];
        v[2 * k] = t[0];
        v[2 * k + 1] = t[1];
    }
}
__device__ __forceinline__ double lds_sum8(const double *p) {
    double v[8];
    lds_read8(p, v);
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

template <typename T, typename TIN, int TR, int TC, bool SI_LDS, bool XLDS>
struct TileKernel {
    using XL = XlLayout<TR>;
    static constexpr int NS = (TR + 7) / 8;  // m-slots per lane (m <= 64*NS)
    static constexpr int MP = 8 * TR;        // padded m
    static constexpr int NP = 8 * TC;        // padded n
    static_assert(TC >= 1 && TC <= 8 && TR >= 1 && TR <= 16, "tile shape");

    // ---------------------------------------------------------------- products on the A tile
    // scattered m-vector -> "row form": wr[s] = w[8s + r], s < TR (all-gather over lanes sharing r)
    static __device__ __forceinline__ void gather_rows(const T (&w)[NS], int r, int c, T *xl, T (&wr)[8 * NS]) {
        if constexpr (XLDS) {
            T *buf = xl + XL::O_ROWS + XL::ROWS_STRIDE * r;
#pragma unroll
            for (int k = 0; k < NS; k++) buf[c + 8 * k] = w[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NS; k++) {
                T tmp[8];
                lds_read8(buf + 8 * k, tmp);
#pragma unroll
                for (int e = 0; e < 8; e++) wr[8 * k + e] = tmp[e];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NS; k++) {
                T tmp[8];
                ag8<8, 16, 32>(w[k], c, tmp);
#pragma unroll
                for (int e = 0; e < 8; e++) wr[8 * k + e] = tmp[e];
            }
        }
    }
    // scattered n-vector -> "column form": xc[t] = x[TC*c + t] (all-gather over lanes sharing c)
    static __device__ __forceinline__ void gather_cols(T x, int r, int c, T *xl, T (&xc)[8]) {
        if constexpr (XLDS) {
            xl[XL::O_COLS + 8 * c + r] = x;
            __syncthreads();
            lds_read8(xl + XL::O_COLS + 8 * c, xc);
        } else {
            ag8<1, 2, 4>(x, r, xc);
        }
    }
    // scattered n-vector -> "block form": yr[c'] = y[TC*c' + r] (all-gather over lanes sharing r)
    static __device__ __forceinline__ void gather_blk(T y, int r, int c, T *xl, T (&yr)[8]) {
        if constexpr (XLDS) {
            xl[XL::O_BLK + 10 * r + c] = y;
            __syncthreads();
            lds_read8(xl + XL::O_BLK + 10 * r, yr);
        } else {
            ag8<8, 16, 32>(y, c, yr);
        }
    }
    // reduce-scatter over the lanes sharing c: lane r receives sum_r' pb_{r'}[r]
    static __device__ __forceinline__ T reduce_cols(const T (&pb)[8], int r, int c, T *xl) {
        if constexpr (XLDS) {
            T *buf = xl + XL::O_RS + XL::RS_COLS_STRIDE * c;
            __syncthreads();  // the staging area is shared by all reduce_*: order after the previous readers
#pragma unroll
            for (int t = 0; t < TC; t++) buf[8 * t + r] = pb[t];
            __syncthreads();
            return lds_sum8(buf + 8 * (r < TC ? r : 0));
        } else {
            return rs8<1, 2, 4>(pb, r);
        }
    }
    // reduce-scatter over the lanes sharing r: lane c receives sum_c'' px_{c''}[c]
    static __device__ __forceinline__ T reduce_blk(const T (&px)[8], int r, int c, T *xl) {
        if constexpr (XLDS) {
            T *buf = xl + XL::O_RS + XL::RS_BLK_STRIDE * r;
            __syncthreads();
#pragma unroll
            for (int cp = 0; cp < 8; cp++) buf[8 * cp + c] = px[cp];
            __syncthreads();
            return lds_sum8(buf + 8 * c);
        } else {
            return rs8<8, 16, 32>(px, c);
        }
    }
    // reduce-scatter of row partials over the lanes sharing r: slot k of lane c receives row s = c + 8k
    static __device__ __forceinline__ void reduce_rows(const T (&pz)[8 * NS], int r, int c, T *xl, T (&out)[NS]) {
        if constexpr (XLDS) {
            T *buf = xl + XL::O_RS + XL::RS_ROWS_STRIDE * r;
            __syncthreads();
#pragma unroll
            for (int s_ = 0; s_ < TR; s_++) buf[8 * s_ + c] = pz[s_];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NS; k++) out[k] = lds_sum8(buf + 8 * ((c + 8 * k) < TR ? (c + 8 * k) : 0));
        } else {
#pragma unroll
            for (int k = 0; k < NS; k++) {
                T tmp[8];
#pragma unroll
                for (int e = 0; e < 8; e++) tmp[e] = pz[8 * k + e];
                out[k] = rs8<8, 16, 32>(tmp, c);
            }
        }
    }

    // A' w  -> scattered n-vector
    static __device__ __forceinline__ T mul_AT(const T (&a)[TR][TC], const T (&wr)[8 * NS], int r, int c, T *xl) {
        T pb[8];
#pragma unroll
        for (int t = 0; t < 8; t++) pb[t] = 0;
#pragma unroll
        for (int s = 0; s < TR; s++)
#pragma unroll
            for (int t = 0; t < TC; t++) pb[t] = tfma(a[s][t], wr[s], pb[t]);
        return reduce_cols(pb, r, c, xl);
    }
    // A x -> scattered m-vector
    static __device__ __forceinline__ void mul_A(const T (&a)[TR][TC], const T (&xc)[8], int r, int c, T *xl, T (&out)[NS]) {
        T pz[8 * NS];
#pragma unroll
        for (int s = 0; s < 8 * NS; s++) pz[s] = 0;
#pragma unroll
        for (int t = 0; t < TC; t++)
#pragma unroll
            for (int s = 0; s < TR; s++) pz[s] = tfma(a[s][t], xc[t], pz[s]);
        reduce_rows(pz, r, c, xl, out);
    }
    // (n x n matrix tiled as si[c'][t]) * column-form vector -> scattered n-vector
    static __device__ __forceinline__ T mul_sq(const T (&si)[8][TC], const T (&bc)[8], int r, int c, T *xl) {
        T px[8];
#pragma unroll
        for (int cp = 0; cp < 8; cp++) {
            T acc = 0;
#pragma unroll
            for (int t = 0; t < TC; t++) acc = tfma(si[cp][t], bc[t], acc);
            px[cp] = acc;
        }
        return reduce_blk(px, r, c, xl);
    }
    // transpose product with the same tile: scattered vector y -> (M' y) scattered
    static __device__ __forceinline__ T mul_sqT(const T (&si)[8][TC], T y, int r, int c, T *xl) {
        T yr[8];
        gather_blk(y, r, c, xl, yr);  // yr[c'] = y[TC*c' + r]
        T px[8];
#pragma unroll
        for (int t = 0; t < 8; t++) px[t] = 0;
#pragma unroll
        for (int cp = 0; cp < 8; cp++)
#pragma unroll
            for (int t = 0; t < TC; t++) px[t] = tfma(si[cp][t], yr[cp], px[t]);
        return reduce_cols(px, r, c, xl);
    }
    static __device__ __forceinline__ T mul_sqT_lds(const T *si_lds, int lane, T y, int r, int c, T *xl) {
        T yr[8];
        gather_blk(y, r, c, xl, yr);
        T px[8];
#pragma unroll
        for (int t = 0; t < 8; t++) px[t] = 0;
#pragma unroll
        for (int cp = 0; cp < 8; cp++)
#pragma unroll
            for (int t = 0; t < TC; t++) px[t] = tfma(si_lds[(cp * TC + t) * 64 + lane], yr[cp], px[t]);
        return reduce_cols(px, r, c, xl);
    }
    // same product with the matrix streamed from global memory (used for P x at the rare checks)
    template <typename TM>
    static __device__ __forceinline__ T mul_sq_gmem(const TM *__restrict__ M, int n, int r, const T (&bc)[8], int c, T *xl) {
        T px[8];
#pragma unroll
        for (int cp = 0; cp < 8; cp++) {
            const int i = TC * cp + r;
            T acc = 0;
#pragma unroll
            for (int t = 0; t < TC; t++) {
                const int j = TC * c + t;
                const T mv = (r < TC && i < n && j < n) ? (T)M[(long)j * n + i] : T(0);
                acc = tfma(mv, bc[t], acc);
            }
            px[cp] = acc;
        }
        return reduce_blk(px, r, c, xl);
    }
    static __device__ __forceinline__ T mul_sq_lds(const T *si_lds, int lane, const T (&bc)[8], int r, int c, T *xl) {
        T px[8];
#pragma unroll
        for (int cp = 0; cp < 8; cp++) {
            T acc = 0;
#pragma unroll
            for (int t = 0; t < TC; t++) acc = tfma(si_lds[(cp * TC + t) * 64 + lane], bc[t], acc);
            px[cp] = acc;
        }
        return reduce_blk(px, r, c, xl);
    }

    // ---------------------------------------------------------------- tile loads
    static __device__ __forceinline__ void load_A_tile(const TIN *__restrict__ gA, int n, int m, int r, int c, T (&a)[TR][TC]) {
#pragma unroll
        for (int t = 0; t < TC; t++) {
            const int j = TC * c + t;
#pragma unroll
            for (int s = 0; s < TR; s++) {
                const int i = 8 * s + r;
                a[s][t] = (j < n && i < m) ? (T)gA[(long)j * m + i] : T(0);
            }
        }
    }
    // tile of a col-major n x n matrix: si[c'][t] = M[row TC*c'+r][col TC*c+t]
    template <typename TM>
    static __device__ __forceinline__ void load_sq_tile(const TM *__restrict__ M, int n, int r, int c, T (&si)[8][TC]) {
#pragma unroll
        for (int cp = 0; cp < 8; cp++) {
            const int i = TC * cp + r;
#pragma unroll
            for (int t = 0; t < TC; t++) {
                const int j = TC * c + t;
                si[cp][t] = (r < TC && i < n && j < n) ? (T)M[(long)j * n + i] : T(0);
            }
        }
    }

    // ---------------------------------------------------------------- Schur factor
    // S = Psym + sigma I + A' diag(rho) A as a register tile, factored in place (no pivoting: S is SPD).
    // On return si = W (lower triangular), S^-1 = W' W.
    // rho_lds: [MP] rho per row (0 in the padding). rowbuf: [NP + 8] LDS scratch. Returns false on a
    // non-positive / non-finite pivot (=> NUMERICAL_ISSUES), wave-uniform.
    static __device__ __forceinline__ bool factor(const TIN *__restrict__ gP, const TIN *__restrict__ gA, int n, int m, T sigma,
                                  const T *rho_lds, T *rowbuf, int lane, int r, int c, T (&si)[8][TC]) {
        const bool rvalid = r < TC;
        int jr[8], jc[TC];  // my row / column indices, clamped for addressing
#pragma unroll
        for (int cp = 0; cp < 8; cp++) {
            const int i = TC * cp + r;
            jr[cp] = (rvalid && i < n) ? i : 0;
        }
#pragma unroll
        for (int t = 0; t < TC; t++) {
            const int j = TC * c + t;
            jc[t] = j < n ? j : 0;
        }
#pragma unroll
        for (int cp = 0; cp < 8; cp++)
#pragma unroll
            for (int t = 0; t < TC; t++) si[cp][t] = 0;
        // A' diag(rho) A : walk the rows of A, columns come straight from global memory (L1/L2 hits)
        for (int i = 0; i < m; i++) {
            const T ri = rho_lds[i];
            T a1[8], a2[TC];
#pragma unroll
            for (int cp = 0; cp < 8; cp++) a1[cp] = (T)gA[(long)jr[cp] * m + i] * ri;
#pragma unroll
            for (int t = 0; t < TC; t++) a2[t] = (T)gA[(long)jc[t] * m + i];
#pragma unroll
            for (int cp = 0; cp < 8; cp++)
#pragma unroll
                for (int t = 0; t < TC; t++) si[cp][t] = tfma(a1[cp], a2[t], si[cp][t]);
        }
#pragma unroll
        for (int cp = 0; cp < 8; cp++) {
            const int i = TC * cp + r;
#pragma unroll
            for (int t = 0; t < TC; t++) {
                const int j = TC * c + t;
                const bool ok = rvalid && i < n && j < n;
                const int lo = i > j ? i : j, hi = i > j ? j : i;
                // only the lower triangle of P reaches the reference's factor (Eigen::LDLT<.,Lower>)
                const T p = ok ? (T)gP[(long)hi * n + lo] : T(0);
                si[cp][t] = ok ? (si[cp][t] + p + (i == j ? sigma : T(0))) : T(0);
            }
        }
        // Jacobi scaling to unit diagonal: S~ = D^-1/2 S D^-1/2.  Every pivot of the sweep is then <= 1,
        // which keeps the row-k update  g - (1 - 1/d) g  free of cancellation (see below), and it is
        // the near-optimal diagonal preconditioning of an SPD matrix anyway.
        T dg = T(1);
#pragma unroll
        for (int cp = 0; cp < 8; cp++)
#pragma unroll
            for (int t = 0; t < TC; t++) dg = (cp == c && t == r && rvalid && TC * c + r < n) ? si[cp][t] : dg;
        const bool dg_bad = !(dg > T(0)) || !(dg * T(0) == T(0));
        const T sc_own = dg_bad ? T(1) : T(1) / (T)sqrt((double)dg);
        T scol[8], srow[8];
        ag8<1, 2, 4>(sc_own, r, scol);    // scol[t]  = scale of column TC*c + t
        ag8<8, 16, 32>(sc_own, c, srow);  // srow[c'] = scale of row TC*c' + r
#pragma unroll
        for (int cp = 0; cp < 8; cp++)
#pragma unroll
            for (int t = 0; t < TC; t++) si[cp][t] = si[cp][t] * srow[cp] * scol[t];
        // Forward elimination of [S~ | I] in place (see admm_generic.h factor_schur for the algebra): after
        // step k the strictly-lower part of column k holds the unit-lower L^-1 entries, the rest U = D L'.
        // Row k lives in the lanes with r == rk (register row ck, static because ck is unrolled); it is
        // broadcast through LDS with "d + 1" on the pivot position so the plain rank-1 update
        //        a_ij -= (a_ki / d) * g_j ,  i > k          (a_ik == a_ki: the trailing block is symmetric)
        // also writes a_ik = -l_ik.  Pivots are <= 1 after the Jacobi scaling, so that is cancellation-free.
        for (int e = lane; e < NP + 8; e += 64) rowbuf[e] = 0;
        __syncthreads();
        if (wave_nanmax(dg_bad ? T(1) : T(0)) != T(0)) return false;  // non-positive diagonal: not SPD
        bool ok_all = true;
        T dsave[8];
#pragma unroll
        for (int cp = 0; cp < 8; cp++) dsave[cp] = T(1);
#pragma unroll
        for (int ck = 0; ck < 8; ck++) {
#pragma unroll 1
            for (int rk = 0; rk < TC; rk++) {
                const int k = TC * ck + rk;
                if (k >= n || !ok_all) break;
                if (r == rk) {
#pragma unroll
                    for (int t = 0; t < TC; t++) rowbuf[TC * c + t] = si[ck][t];
                }
                __syncthreads();
                const T d = rowbuf[k];
                if (!(d > T(0)) || !(d * T(0) == T(0))) {
                    ok_all = false;
                    break;
                }
                const T dinv = T(1) / d;
                T g[TC], f[8];
#pragma unroll
                for (int t = 0; t < TC; t++) {
                    const T gt = rowbuf[TC * c + t];
                    g[t] = (c == ck && t == rk) ? d + T(1) : gt;
                }
#pragma unroll
                for (int cp = 0; cp < 8; cp++) {
                    const T gi = rowbuf[rvalid ? TC * cp + r : 0];
                    f[cp] = (rvalid && TC * cp + r > k) ? gi * dinv : T(0);
                }
                __syncthreads();  // everyone has read rowbuf before the next step overwrites it
#pragma unroll
                for (int cp = 0; cp < 8; cp++)
#pragma unroll
                    for (int t = 0; t < TC; t++) si[cp][t] = tfma(-f[cp], g[t], si[cp][t]);
                dsave[ck] = (r == rk) ? d : dsave[ck];
            }
        }
        // W = D^-1/2 L^-1 D_J^-1/2 : lower triangular, S^-1 = W' W
#pragma unroll
        for (int cp = 0; cp < 8; cp++) {
            const int i = TC * cp + r;
            const T rs = T(1) / (T)sqrt((double)dsave[cp]);
#pragma unroll
            for (int t = 0; t < TC; t++) {
                const int j = TC * c + t;
                const bool ok = rvalid && i < n && j < n;
                const T v = i > j ? si[cp][t] * rs : (i == j ? rs : T(0));
                si[cp][t] = ok ? v * scol[t] : T(0);
            }
        }
        return ok_all;
    }

    static __device__ __forceinline__ void store_sq_tile(T *__restrict__ M, int n, int r, int c, const T (&si)[8][TC]) {
#pragma unroll
        for (int cp = 0; cp < 8; cp++) {
            const int i = TC * cp + r;
#pragma unroll
            for (int t = 0; t < TC; t++) {
                const int j = TC * c + t;
                if (r < TC && i < n && j < n) M[(long)j * n + i] = si[cp][t];
            }
        }
    }

    // ---------------------------------------------------------------- the kernel body
    static __device__ void run(const KArgs<T, TIN> &a, T *rho_lds, T *rowbuf, T *si_lds, T *xl) {
        const int lane = threadIdx.x & 63;
        const int r = lane & 7, c = lane >> 3;
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
        T *gSinv = a.Sinv + (long)qp * n * n;

        sqph_info info = a.info[qp];
        T rho_s = a.rho[qp];
        const int mode = a.mode;
        if (!(mode & (MODE_SETUP | MODE_UPDATE)) &&
            (info.status == SQPH_UNINITIALIZED || info.status == SQPH_NUMERICAL_ISSUES))
            return;  // qp.cpp:68-71

        // ---- scattered state: n-index jn = TC*c + r (lanes r < TC), m-index lane + 64k
        const int jn = TC * c + r;
        const bool nvalid = (r < TC) && (jn < n);
        bool mvalid[NS];
        int ctype[NS];
        T q = nvalid ? (T)gq[jn] : T(0);
        T x = 0;
        T z[NS], y[NS], lo[NS], up[NS], rho[NS], rinv[NS];
        const T INF = T(1) / T(0);
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const int i = lane + 64 * k;
            mvalid[k] = i < m;
            lo[k] = mvalid[k] ? (T)gl[i] : -INF;
            up[k] = mvalid[k] ? (T)gu[i] : INF;
            z[k] = y[k] = 0;
            rho[k] = rinv[k] = T(1);
            ctype[k] = SQPH_INEQUALITY_CONSTRAINT;
        }

        T si[8][TC];  // S^-1 tile (registers; mirrored into LDS when SI_LDS)
        const T lt = a.loose_thresh, et = a.eq_tol;

        if (mode & (MODE_SETUP | MODE_UPDATE)) {
            rho_s = a.rho0;
#pragma unroll
            for (int k = 0; k < NS; k++) {
                if (mvalid[k]) {
                    int ct;
                    if (lo[k] < -lt && up[k] > lt)
                        ct = SQPH_LOOSE_BOUNDS;
                    else if (up[k] - lo[k] < et)
                        ct = SQPH_EQUALITY_CONSTRAINT;
                    else
                        ct = SQPH_INEQUALITY_CONSTRAINT;
                    ctype[k] = ct;
                    rho[k] = rho_for_type<T>(ct, rho_s, a.rho_min, a.rho_eq_factor);
                    rinv[k] = T(1) / rho[k];
                    sct[lane + 64 * k] = ct;
                    srho[lane + 64 * k] = rho[k];
                }
            }
            info.rho_updates += 1;
            if (!(mode & MODE_SETUP)) {
                x = nvalid ? sx[jn] : T(0);
#pragma unroll
                for (int k = 0; k < NS; k++)
                    if (mvalid[k]) {
                        z[k] = sz[lane + 64 * k];
                        y[k] = sy[lane + 64 * k];
                    }
            }
        } else {
            x = nvalid ? sx[jn] : T(0);
#pragma unroll
            for (int k = 0; k < NS; k++)
                if (mvalid[k]) {
                    const int i = lane + 64 * k;
                    z[k] = sz[i];
                    y[k] = sy[i];
                    rho[k] = srho[i];
                    rinv[k] = T(1) / rho[k];
                    ctype[k] = sct[i];
                }
        }

        auto publish_rho = [&]() {
            // rho per constraint row for the factor routine (0 in the padding)
            __syncthreads();  // rho_lds aliases the reduce staging area
#pragma unroll
            for (int k = 0; k < NS; k++) rho_lds[lane + 64 * k] = mvalid[k] ? rho[k] : T(0);
            __syncthreads();
        };
        auto stash_si = [&]() {
            if constexpr (SI_LDS) {
#pragma unroll
                for (int cp = 0; cp < 8; cp++)
#pragma unroll
                    for (int t = 0; t < TC; t++) si_lds[(cp * TC + t) * 64 + lane] = si[cp][t];
                __syncthreads();
            }
        };

        // (the comment block above SQPH_OPAQUE_* explains the laundering of invariants below)
        // One call site for the (large, fully unrolled) factor routine: the solve is a small state
        // machine  [factor] -> [iterate until done or until adaptive rho asks for a new factor] -> ...
        bool need_factor = (mode & (MODE_SETUP | MODE_UPDATE)) != 0;
        bool solving = false;
        bool state_dirty = (mode & MODE_SETUP) != 0;
        if (!need_factor) load_sq_tile<T>(gSinv, n, r, c, si);
        const T alpha = a.alpha, sigma = a.sigma, oma = T(1) - a.alpha;
        int iter = 1;
        for (;;) {
            if (need_factor) {
                publish_rho();
                int n_f = n, m_f = m, r_f = r, c_f = c, lane_f = lane;
                const TIN *gA_f = gA, *gP_f = gP;
                SQPH_OPAQUE_S(n_f); SQPH_OPAQUE_S(m_f); SQPH_OPAQUE_V(r_f); SQPH_OPAQUE_V(c_f); SQPH_OPAQUE_V(lane_f);
                SQPH_OPAQUE_S(gA_f); SQPH_OPAQUE_S(gP_f);
                const bool ok = factor(gP_f, gA_f, n_f, m_f, sigma, rho_lds, rowbuf, lane_f, r_f, c_f, si);
                store_sq_tile(gSinv, n_f, r_f, c_f, si);
                need_factor = false;
                if (!solving) {
                    info.status = ok ? SQPH_UNSOLVED : SQPH_NUMERICAL_ISSUES;  // qp.cpp:39-43, 57-61
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
                    x = 0;
#pragma unroll
                    for (int k = 0; k < NS; k++) z[k] = y[k] = 0;
                }
            }
            stash_si();
            T at[TR][TC];
            {
                int n_t = n, m_t = m, r_t = r, c_t = c;
                const TIN *gA_t = gA;
                SQPH_OPAQUE_S(n_t); SQPH_OPAQUE_S(m_t); SQPH_OPAQUE_V(r_t); SQPH_OPAQUE_V(c_t); SQPH_OPAQUE_S(gA_t);
                load_A_tile(gA_t, n_t, m_t, r_t, c_t, at);  // (re)loaded after every factor: its registers were free meanwhile
            }
            for (; iter <= a.max_iter; iter++) {
                // w = R (z - R^-1 y)   [rhs tail of qp.cpp:275 pre-multiplied by R]
                T w[NS];
#pragma unroll
                for (int k = 0; k < NS; k++) w[k] = mvalid[k] ? rho[k] * (z[k] - rinv[k] * y[k]) : T(0);
                T wr[8 * NS];
                gather_rows(w, r, c, xl, wr);
                T b = mul_AT(at, wr, r, c, xl);
                b = nvalid ? (sigma * x - q) + b : T(0);
                T bc[8];
                gather_cols(b, r, c, xl, bc);
                T xt;  // x~ = W' (W b)
                if constexpr (SI_LDS) {
                    T wb = mul_sq_lds(si_lds, lane, bc, r, c, xl);
                    wb = nvalid ? wb : T(0);
                    xt = mul_sqT_lds(si_lds, lane, wb, r, c, xl);
                } else {
                    T wb = mul_sq(si, bc, r, c, xl);
                    wb = nvalid ? wb : T(0);
                    xt = mul_sqT(si, wb, r, c, xl);
                }
                xt = nvalid ? xt : T(0);
                T xc[8];
                gather_cols(xt, r, c, xl, xc);
                T zt[NS];
                mul_A(at, xc, r, c, xl, zt);
                x = alpha * xt + oma * x;
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    const T zr = alpha * zt[k] + oma * z[k];
                    T zn = zr + rinv[k] * y[k];
                    zn = zn < lo[k] ? lo[k] : zn;  // cwiseMax(l) then cwiseMin(u), qp.cpp:278-281
                    zn = zn > up[k] ? up[k] : zn;
                    y[k] = mvalid[k] ? y[k] + rho[k] * (zr - zn) : T(0);
                    z[k] = mvalid[k] ? zn : T(0);
                }

                const bool check = a.check_termination != 0 && (iter % a.check_termination == 0);
                const bool adapt = a.adaptive_rho && (iter % a.adaptive_rho_interval == 0);
                if (check || adapt) {
                    // update_state + residuals, qp.cpp:316-331, 353-361
                    T xcur[8];
                    gather_cols(x, r, c, xl, xcur);
                    T Ax[NS];
                    mul_A(at, xcur, r, c, xl, Ax);
                    T yr[8 * NS];
                    gather_rows(y, r, c, xl, yr);
                    const T ATy = mul_AT(at, yr, r, c, xl);
                    int n_c = n, r_c = r, c_c = c;
                    const TIN *gP_c = gP;
                    SQPH_OPAQUE_S(n_c); SQPH_OPAQUE_V(r_c); SQPH_OPAQUE_V(c_c); SQPH_OPAQUE_S(gP_c);
                    const T Px = mul_sq_gmem<TIN>(gP_c, n_c, r_c, xcur, c_c, xl);  // full P (both triangles), as qp.cpp:324
                    T v[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int k = 0; k < NS; k++)
                        if (mvalid[k]) {
                            v[0] = nanmax(v[0], tabs(Ax[k]));
                            v[1] = nanmax(v[1], tabs(z[k]));
                            v[2] = nanmax(v[2], tabs(Ax[k] - z[k]));
                        }
                    if (nvalid) {
                        v[3] = tabs(Px);
                        v[4] = tabs(ATy);
                        v[5] = tabs(q);
                        v[6] = tabs(Px + q + ATy);
                    }
#pragma unroll
                    for (int e = 0; e < 7; e++) v[e] = wave_nanmax(v[e]);
                    const T nrm_prim = nanmax(v[0], v[1]);
                    const T nrm_dual = nanmax(v[3], nanmax(v[4], v[5]));
                    info.res_prim = (double)v[2];
                    info.res_dual = (double)v[6];
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
                        info.rho_estimate = (double)new_rho;
                        if (new_rho < rho_s / a.rho_tol || new_rho > rho_s * a.rho_tol) {
                            rho_s = new_rho;
#pragma unroll
                            for (int k = 0; k < NS; k++)
                                if (mvalid[k]) {
                                    rho[k] = rho_for_type<T>(ctype[k], rho_s, a.rho_min, a.rho_eq_factor);
                                    rinv[k] = T(1) / rho[k];
                                }
                            info.rho_updates += 1;
                            need_factor = true;
                            break;  // leave the iteration loop WITHOUT advancing iter; the factor block does it
                        }
                    }
                }
            }
            if (!need_factor) break;  // converged, exhausted, or no refactor pending
        }
        if (solving) {
            if (iter > a.max_iter) info.status = SQPH_MAX_ITER_EXCEEDED;
            info.iter = iter;
        }

        if (state_dirty) {
            if (nvalid) sx[jn] = x;
#pragma unroll
            for (int k = 0; k < NS; k++)
                if (mvalid[k]) {
                    const int i = lane + 64 * k;
                    sz[i] = z[k];
                    sy[i] = y[k];
                    srho[i] = rho[k];
                }
        }
        if (lane == 0) {
            a.info[qp] = info;
            a.rho[qp] = rho_s;
        }
    }
};

// WPE = waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument)
template <typename T, typename TIN, int TR, int TC, bool SI_LDS, int WPE, bool XLDS = true>
__global__ __launch_bounds__(64, WPE) void admm_tile_kernel(KArgs<T, TIN> a) {
    // factor-time scratch (rho per row, broadcast row) aliases the reduce-scatter staging area
    __shared__ __attribute__((aligned(16))) T xl[XlLayout<TR>::TOTAL];
    __shared__ __attribute__((aligned(16))) T si_lds[SI_LDS ? 8 * TC * 64 : 2];
    static_assert(XlLayout<TR>::RS_SIZE >= 64 * ((TR + 7) / 8) + 8 * TC + 8, "factor scratch must fit the staging area");
    T *rho_lds = xl + XlLayout<TR>::O_RS;
    T *rowbuf = rho_lds + 64 * ((TR + 7) / 8);
    TileKernel<T, TIN, TR, TC, SI_LDS, XLDS>::run(a, rho_lds, rowbuf, si_lds, xl);
}

// tile shapes compiled into the library: {TR, TC, SI_LDS, WPE}; first fit wins
#define SQPH_TILE_SHAPES(X) \
    X(1, 1, false, 3)       \
    X(3, 2, false, 2)       \
    X(5, 3, false, 2)       \
    X(8, 4, false, 2)       \
    X(8, 8, true, 1)        \
    X(13, 7, true, 1)

#ifdef SQPH_SIM
template <typename T, typename TIN>
inline int sim_run_tile(const KArgs<T, TIN> &a) {
#define SQPH_SIM_CASE(TR_, TC_, L_, W_)                                                                 \
    if (a.m <= 8 * TR_ && a.n <= 8 * TC_) {                                                             \
        ::sqph_sim::launch(admm_tile_kernel<T, TIN, TR_, TC_, L_, W_>, dim3(a.batch), dim3(64), 0, a);  \
        return 0;                                                                                       \
    }
    SQPH_TILE_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

}  // namespace sqph
