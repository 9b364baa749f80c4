// Blocked set-up of the generic kernel on the f64 MFMA, every n x n matrix as 16 x 16 blocks in GLOBAL memory (the factor workspace).
//
// The generic kernel (admm_generic.h) serves what no on-chip kernel takes: dense problems with n > 256 or m > 512.  Its set-up was the
// textbook form — every entry of S = P + sigma I + A' R A as a dot product over m, then n sequential pivots, each a sweep over the whole
// n x n matrix in global memory with three dependent memory round trips and three barriers: 15.7 of the 26 ms of a 64 x (300, 600) call
// with 100 iterations.  This is the set-up of admm_wg_msetup.h (same phases, same block algebra, same in-wavefront elimination of a
// diagonal block) with the blocks where a matrix of this size has to live:
//     S blocks   = sum over k of (rho_k At[k][16 I ..])' At[k][16 K ..]        one wavefront per group of four blocks of a block row
//     S~ = D_J S D_J (Jacobi), blocked elimination  S~ = L L',  W = L^-1 D_J   step J: diagonal block in one wavefront (through LDS),
//                                                                              panel and trailing update as block products, operands from L2
// Reference: construct_KKT_mat / update_KKT_rho and Eigen::LDLT::compute, src/qp.cpp:159-259 (on the Schur complement).
// Workspace: the lower blocks live in the second half of the factor workspace (Wt), the step's panel L_IJ in the first (Wm); the
// canonical W (column-major) and its row-major copy are written at the end.  Needs: T = double, >= 4 wavefronts, n >= 48 (the blocks
// must fit the n^2 doubles of a half).  Results differ from the unblocked set-up in the last bits (another summation order).
#pragma once
#include "admm_wg_kernel.h"  // MSetup (admm_wg_msetup.h), mfma16 (wave_ops.h)

namespace sqph {

struct GenericBlocked {
    using T = double;
    using MS = MSetup<2, 16, 8, 7, 7, 4>;  // diag_block and its padded LDS block layout (BS = 272, ix(i, j) = 17 i + j)
    static_assert(!MS::SWZ && MS::BS == 272, "padded LDS blocks");
    static constexpr int GB = 256;  // doubles of a block in global memory: plain 16 x 16, row-major
    static __host__ __device__ int blk(int I, int K) { return I * (I + 1) / 2 + K; }
    static __host__ __device__ bool fits(int n, int nt) {
        const int NB = (n + 15) / 16;
        return nt >= 256 && (nt & 63) == 0 && n >= 48 && (long)NB * (NB + 1) / 2 * GB <= (long)n * n && (long)NB * GB <= (long)n * n &&
               16 * NB + 8 + 3 * MS::BS <= 8 * nt;
    }
    static __device__ __forceinline__ T gN(const T *b, int kq, int lr, int lq) { return b[lr * 16 + 4 * kq + lq]; }
    static __device__ __forceinline__ T gT(const T *b, int kq, int lr, int lq) { return b[(4 * kq + lq) * 16 + lr]; }
    static __device__ __forceinline__ void gld(const T *b, int lr, int lq, sqph_acc4 &a) {
#pragma unroll
        for (int e = 0; e < 4; e++) a.v[e] = b[(lq + 4 * e) * 16 + lr];
    }
    static __device__ __forceinline__ void gst(T *b, int lr, int lq, const sqph_acc4 &a) {
#pragma unroll
        for (int e = 0; e < 4; e++) b[(lq + 4 * e) * 16 + lr] = a.v[e];
    }

    // returns false (block-uniform) when S is not positive definite / not finite.  stage: LDS, 8 nt doubles; rho: LDS [m].
    template <typename TIN>
    static __device__ SQPH_GENERIC_NOINLINE bool factor(int n_, int m_, const TIN *__restrict__ P_, const T *__restrict__ At_, const T *rho, T sigma,
                                                        T *__restrict__ Wm, T *__restrict__ Wt, T *stage) {
#ifdef SQPH_SIM
        const int n = n_, m = m_;
#else
        const int n = __builtin_amdgcn_readfirstlane(n_), m = __builtin_amdgcn_readfirstlane(m_);
#endif
        const TIN *__restrict__ P = P_;
        const T *__restrict__ At = At_;
        const int t = threadIdx.x, nt = blockDim.x, NW = nt >> 6;
        const int wave = MS::wave_of(t), l = t & 63, lr = l & 15, lq = l >> 4;
        const int NB = (n + 15) >> 4;
        T *SB = Wt, *XS = Wm;
        T *sj = stage, *flag = stage + 16 * NB, *TB = stage + ((16 * NB + 8 + 1) & ~1), *DL = TB + 2 * MS::BS;
        if (t < 2) flag[t] = T(0);

        // ---- S, lower blocks, four blocks of a block row per job
        {
            int job = 0;
            for (int I = 0; I < NB; I++) {
                for (int K0 = 0; K0 <= I; K0 += 4, job++) {
                    if (job % NW != wave) continue;
                    sqph_acc4 acc[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[c] = sqph_acc4{{0, 0, 0, 0}};
                    const int ci = 16 * I + lr;
                    const bool iok = ci < n;
                    for (int k0 = 0; k0 < m; k0 += 4) {
                        const int kk = k0 + lq;
                        const bool kok = kk < m;
                        const T *row = At + (long)(kok ? kk : 0) * n;
                        const T a = (kok && iok) ? row[ci] * rho[kk] : T(0);
                        T b[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const int cj = 16 * (K0 + c) + lr;
                            b[c] = (kok && cj < n) ? row[cj] : T(0);
                        }
#pragma unroll
                        for (int c = 0; c < 4; c++)
                            if (K0 + c <= I) mfma16(a, b[c], acc[c]);
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int K = K0 + c;
                        if (K > I) break;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int i = 16 * I + lq + 4 * e, j = 16 * K + lr;
                            const bool ok = i < n && j < n;
                            const int lo = i > j ? i : j, hi = i > j ? j : i;
                            // only the lower triangle of P reaches the reference's factor (Eigen::LDLT<.,Lower>, qp.hpp:129); the padding
                            // (rows / columns >= n of the 16 NB square) is an identity block: its factor and inverse are identities
                            const T pv = ok ? (T)P[(long)hi * n + lo] : T(0);
                            const T v = acc[c].v[e] + pv + (i == j ? sigma : T(0));
                            acc[c].v[e] = ok ? v : (i == j ? T(1) : T(0));
                        }
                        gst(SB + (long)blk(I, K) * GB, lr, lq, acc[c]);
                    }
                }
            }
        }
        __syncthreads();
        // ---- Jacobi scaling
        for (int i = t; i < 16 * NB; i += nt) {
            const T d = SB[(long)blk(i >> 4, i >> 4) * GB + (i & 15) * 17];
            const bool bad = !(d > T(0)) || !(d * T(0) == T(0));  // non-positive / non-finite diagonal => not SPD
            sj[i] = bad ? T(1) : T(1) / (T)sqrt((double)d);
            if (bad) flag[0] = T(1);
        }
        __syncthreads();
        if (flag[0] != T(0)) return false;
        {
            int job = 0;
            for (int I = 0; I < NB; I++)
                for (int K = 0; K <= I; K++, job++) {
                    if (job % NW != wave) continue;
                    T *b = SB + (long)blk(I, K) * GB;
                    sqph_acc4 a;
                    gld(b, lr, lq, a);
                    const T dc = sj[16 * K + lr];
#pragma unroll
                    for (int e = 0; e < 4; e++) a.v[e] = a.v[e] * sj[16 * I + lq + 4 * e] * dc;
                    gst(b, lr, lq, a);
                }
        }
        // ---- blocked elimination (admm_wg_msetup.h phase 3 without the look-ahead): step J
        //   (1) wavefront 0: M_JJ -> Winv_JJ (unscaled, TB in LDS) and W_JJ = Winv_JJ D_J
        //   (2) panel L_IJ = M_IJ Winv_JJ' (I > J) into XS[I]; the finished row W_JK = Winv_JJ E_JK (K < J)
        //   (3) E_IJ = -L_IJ Winv_JJ D_J, E_IK -= L_IJ W_JK (K < J), M_IK -= L_IJ L_KJ' (J < K <= I)
        for (int J = 0; J < NB; J++) {
            __syncthreads();
            if (wave == 0) {
                T *Mjj = SB + (long)blk(J, J) * GB;
#pragma unroll
                for (int e = 0; e < 4; e++) DL[MS::ix(lq + 4 * e, lr)] = Mjj[(lq + 4 * e) * 16 + lr];
                MS::wave_fence();
                MS::diag_block(DL, TB, sj + 16 * J, flag, l);
                MS::wave_fence();
#pragma unroll
                for (int e = 0; e < 4; e++) Mjj[(lq + 4 * e) * 16 + lr] = DL[MS::ix(lq + 4 * e, lr)];
            }
            __syncthreads();
            int job = 0;
            for (int I = J + 1; I < NB; I++, job++) {
                if (job % NW != wave) continue;
                sqph_acc4 a = {{0, 0, 0, 0}};
                const T *Mij = SB + (long)blk(I, J) * GB;
#pragma unroll
                for (int kq = 0; kq < 4; kq++) mfma16(gN(Mij, kq, lr, lq), MS::opN(TB, kq, lr, lq), a);
                gst(XS + (long)I * GB, lr, lq, a);
            }
            for (int K = 0; K < J; K++, job++) {
                if (job % NW != wave) continue;
                sqph_acc4 a = {{0, 0, 0, 0}};
                T *Ejk = SB + (long)blk(J, K) * GB;
                T bv[4];
#pragma unroll
                for (int kq = 0; kq < 4; kq++) bv[kq] = gT(Ejk, kq, lr, lq);
#pragma unroll
                for (int kq = 0; kq < 4; kq++) mfma16(MS::opN(TB, kq, lr, lq), bv[kq], a);
                gst(Ejk, lr, lq, a);
            }
            if (J == NB - 1) break;
            __syncthreads();
            job = 0;
            const T dc = sj[16 * J + lr];
            for (int I = J + 1; I < NB; I++) {
                const T *Li = XS + (long)I * GB;
                if (job++ % NW == wave) {
                    sqph_acc4 a = {{0, 0, 0, 0}};
#pragma unroll
                    for (int kq = 0; kq < 4; kq++) mfma16(-gN(Li, kq, lr, lq), MS::opT(TB, kq, lr, lq), a);
#pragma unroll
                    for (int e = 0; e < 4; e++) a.v[e] *= dc;
                    gst(SB + (long)blk(I, J) * GB, lr, lq, a);
                }
                for (int K = 0; K < J; K++) {
                    if (job++ % NW != wave) continue;
                    sqph_acc4 a;
                    T *Eik = SB + (long)blk(I, K) * GB;
                    const T *Wjk = SB + (long)blk(J, K) * GB;
                    gld(Eik, lr, lq, a);
#pragma unroll
                    for (int kq = 0; kq < 4; kq++) mfma16(-gN(Li, kq, lr, lq), gT(Wjk, kq, lr, lq), a);
                    gst(Eik, lr, lq, a);
                }
                for (int K = J + 1; K <= I; K++) {
                    if (job++ % NW != wave) continue;
                    sqph_acc4 a;
                    T *Mik = SB + (long)blk(I, K) * GB;
                    const T *Lk = XS + (long)K * GB;
                    gld(Mik, lr, lq, a);
#pragma unroll
                    for (int kq = 0; kq < 4; kq++) mfma16(-gN(Li, kq, lr, lq), gN(Lk, kq, lr, lq), a);
                    gst(Mik, lr, lq, a);
                }
            }
        }
        __syncthreads();
        if (flag[1] != T(0)) return false;
        // ---- canonical W (column-major, zero above the diagonal), then its row-major copy over the blocks' place
        for (long e = t; e < (long)n * n; e += nt) {
            const int j = (int)(e / n), i = (int)(e - (long)j * n);
            Wm[e] = i >= j ? SB[(long)blk(i >> 4, j >> 4) * GB + (i & 15) * 16 + (j & 15)] : T(0);
        }
        __syncthreads();
        for (long e = t; e < (long)n * n; e += nt) {
            const int i = (int)(e / n), j = (int)(e - (long)i * n);
            Wt[e] = Wm[(long)j * n + i];
        }
        __syncthreads();
        return true;
    }
};

}  // namespace sqph
