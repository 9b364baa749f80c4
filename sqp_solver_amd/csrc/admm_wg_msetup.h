// MFMA set-up of the register-tiled kernels (included by admm_wg_kernel.h): the three matrix phases of a factorisation —
//     S = P_lower + sigma I + A' diag(rho) A          (reference: construct_KKT_mat / update_KKT_rho, src/qp.cpp:159-235)
//     S = D_J^1/2 (L L') D_J^1/2 ,  W = L^-1 D_J^-1/2   (reference: Eigen::LDLT::compute, src/qp.cpp:237-259)
//     B = A W'                                          (the operator the iteration runs on, admm_wg_kernel.h)
// on v_mfma_f64_16x16x4_f64 with every n x n matrix held in LDS as 16 x 16 blocks.
//
// Why: the VALU forms of these phases (WgKernel::factor, build_B_inplace) run at ~10 % of the fp64 peak — per multiply-add they move
// 4-5 bytes through LDS, and the elimination is n sequential pivots of two workgroup barriers each (53 k of a C3 set-up's 161 k
// cycles).  The f64 MFMA has the vector pipe's rate (measured: 64 cycles per 16x16x4 on a SIMD = 16 multiply-adds per cycle,
// tools/ubench/mfma_f64_rate.hip) but takes ONE instruction and two 8-byte operands per lane for 1,024 multiply-adds, and it turns
// the elimination into ceil(n/16) block steps: a 16 x 16 diagonal block is eliminated inside one wavefront (every lane a column,
// pivot row and multipliers by DPP row broadcast: no LDS, no barrier), everything off the diagonal is 16 x 16 x 16 block products.
//
// Layouts.  A 16 x 16 block is stored row-major with rows padded to 17 doubles (BS = 272; unpadded and XOR-swizzled where NB > 4, see
// SWZ): the four access patterns below are (nearly) bank-conflict free with it.  Lower blocks (I >= K) only, block (I, K) at index I (I + 1) / 2 + K.
//     opN(blk, kq): element (lr, 4 kq + lq)   — A-operand X[i][k];  B-operand of X Y' (Y[j][k])
//     opT(blk, kq): element (4 kq + lq, lr)   — B-operand of X Y  (Y[k][j]);  A-operand of X' Y
//     D layout    : lane holds (lq + 4 e, lr), e < 4            with lr = lane & 15, lq = lane >> 4
// The lane grid of the tiled kernels (R = 16: r = t % 16 = lr, c = t / 16 = 4 wave + lq) makes the D layout of B' = W A' — rows
// permuted so that D row lq + 4 e is W row C (4 Jq + e) + c — exactly the B register tile: no staging of the result.
#pragma once

namespace sqph {

template <int NW, int R, int C, int TR, int TC, int TW>
struct MSetup {
    using T = double;
    using L = WgLayout<NW, R, C, TR, TC, TW>;
    static constexpr int NT = L::NT;
    static constexpr int NB = (L::NP + 15) / 16;      // 16-column blocks of an n x n matrix
    static constexpr int NBLK = NB * (NB + 1) / 2;    // lower blocks
    // seven block columns (56 < n <= 112, the 16 x 16 / 7 x 7 grids): the 28 + 7 + 2 padded blocks would take 10,064 doubles and one of
    // the two workgroups a CU holds; there the rows are 16 doubles and column j of row i sits at j ^ i (the same four access patterns
    // are conflict-free: a group of lanes reads one or two rows, the XOR permutes inside a row and rows of different parity lie in
    // different bank halves) — a few address instructions more per access, 9,720 doubles
    static constexpr bool SWZ = NB > 4;
    static constexpr int BS = SWZ ? 256 : 16 * 17;    // doubles per block
    static __device__ __forceinline__ int ix(int i, int j) { return SWZ ? i * 16 + (j ^ i) : i * 17 + j; }  // element (i, j) of a block
    static constexpr int NQ = (NBLK + (NW > 0 ? NW : 1) - 1) / (NW > 0 ? NW : 1);  // S blocks accumulated per wavefront
    static constexpr int NJQ = (TC + 3) / 4;          // 4-column groups of the B tile = accumulators of build_B
    // LDS map (doubles), all of it inside the set-up scratch [0, L::O_QV)
    static constexpr int O_RHO = L::O_RHO;            // [MP]    rho per constraint row (0 beyond m), written by the caller
    static constexpr int O_SJ = O_RHO + L::MP;        // [16 NB] diagonal of S, then the Jacobi scale 1 / sqrt(S_ii) (1 on the padding)
    static constexpr int O_FLAG = O_SJ + 16 * NB;     // [8]     0: not SPD at the diagonal test, 1: a non-positive pivot
    static constexpr int O_XS = L::ev(O_FLAG + 8);    // [NB][16][17]  a block of 16 rows of A by 16-column blocks; the L panel in the factorisation
    static constexpr int O_SB = O_XS + NB * BS;       // [NBLK][16][17] S, then L / E, finally W;  P is staged here first (natural n x n)
    static constexpr int O_TB = O_SB + NBLK * BS;     // [2][16][17]   the unscaled inverses of the current and the next diagonal block
    // P is staged (LDS-DMA) where its n x n block fits behind O_SB; where only the blocks fit (the four-wave 16 x 16 grid with n <= 64)
    // the P phase reads it from global memory
    static constexpr bool PST = O_SB + L::NP * L::NP <= L::O_QV;
    // 32-row grids (RH = 2 half-blocks of 16 rows per tile row): the D layout of B' = W A' is not the register tile there, the result
    // blocks go through LDS (build_B_rows), one block per 16 columns of B
    static constexpr int RH = R / 16;
    static constexpr int O_DS = O_TB + 2 * BS;
    static constexpr int END = L::mx(O_DS + (RH > 1 ? NB * BS : 0), PST ? O_SB + L::NP * L::NP : 0);
#ifdef SQPH_XP_NO_MSET4  // (experiment builds: the scalar set-up for the four-wave grid)
    static constexpr bool ENABLED = NW == 2 && R == 16 && C == 4 * NW && L::NP <= 64 && END <= L::O_QV && (O_SB % 2) == 0;
#else
    static constexpr bool ENABLED = NW >= 1 && ((R == 16 && C == 4 * NW && (L::NP <= 64 || L::MSX)) || L::MSR) && END <= L::O_QV && (O_SB % 2) == 0;
#endif
    static_assert(!L::MSR || ENABLED, "the 32 x 16 grid's set-up scratch has room for the MFMA set-up");
    static_assert(!L::MSX || ENABLED, "WgLayout sized the set-up scratch of this grid for the MFMA set-up");

    static __device__ __forceinline__ int blk(int I, int K) { return I * (I + 1) / 2 + K; }
    // wavefront index as a scalar (the compiler cannot tell that t >> 6 is uniform: without this every per-wave decision below
    // becomes an exec-mask region and every block offset a vector multiply)
    static __device__ __forceinline__ int wave_of(int t) {
#ifdef SQPH_SIM
        return t >> 6;
#else
        return __builtin_amdgcn_readfirstlane(t >> 6);
#endif
    }
    // orders the LDS operations of ONE wavefront (they execute in program order; this only pins the compiler)
    static __device__ __forceinline__ void wave_fence() {
#ifdef SQPH_SIM
        ::sqph_sim::yield_wait(2);
#else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
    }
    // 1 / sqrt(d) for a positive finite normal d: v_rsq_f64 refined by two Newton steps (within an ulp or two; only ever used as a
    // scaling that is applied and removed with the same value, or where an ulp does not matter)
    static __device__ __forceinline__ T fast_rsqrt(T d) {
#ifdef SQPH_SIM
        return T(1) / (T)sqrt((double)d);
#else
        T y = __builtin_amdgcn_rsq(d);
        const T h = T(0.5) * d;
        y = __builtin_fma(y, __builtin_fma(-h * y, y, T(0.5)), y);
        y = __builtin_fma(y, __builtin_fma(-h * y, y, T(0.5)), y);
        return y;
#endif
    }
    static __device__ __forceinline__ T opN(const T *b, int kq, int lr, int lq) { return b[ix(lr, 4 * kq + lq)]; }
    static __device__ __forceinline__ T opT(const T *b, int kq, int lr, int lq) { return b[ix(4 * kq + lq, lr)]; }
    static __device__ __forceinline__ void ldD(const T *b, int lr, int lq, sqph_acc4 &a) {
#pragma unroll
        for (int e = 0; e < 4; e++) a.v[e] = b[ix(lq + 4 * e, lr)];
    }
    static __device__ __forceinline__ void stD(T *b, int lr, int lq, const sqph_acc4 &a) {
#pragma unroll
        for (int e = 0; e < 4; e++) b[ix(lq + 4 * e, lr)] = a.v[e];
    }
    // element (i, j) of the lower triangular W held in SB (zero above the diagonal and beyond n)
    static __device__ __forceinline__ T Wget(const T *SB, int i, int j, int n) {
        const bool in = i < n && j < n && (i >> 4) >= (j >> 4);
        const int a = blk(i >> 4, in ? (j >> 4) : 0) * BS + ix(i & 15, j & 15);
        const T v = SB[in ? a : 0];
        return in ? v : T(0);
    }

    // ------------------------------------------------------------------ one 16 x 16 diagonal block, one wavefront
    // In-place Gauss-Jordan of [M | I] without pivoting (M SPD): lane l works on column j = l & 15 — the four 16-lane rows of the
    // wavefront hold copies, so nothing is masked — and keeps it in 16 registers.  Step k: pivot d = M[k][k] and the multipliers
    // M[i][k] come from lane k by DPP row broadcast (v_fmac_f64 with row_newbcast: no LDS, no barrier), lane j scales its own
    // entry of the pivot row.  Result: Winv = D^-1/2 L~^-1 (M = L~ D L~'), i.e. the inverse of M's Cholesky factor.
    // the multiply-adds of elimination step K as ONE statement: M[i] += M[i](lane K of my row) * tk for i > K.  One block per step so
    // that the two wait states a DPP read needs behind the VALU write of its source (the previous step's block) are provable:
    // the compiler neither sees into an asm statement nor reorders inside one; the trailing wait states cover a DPP read of the
    // last result by whatever follows.  (F(i, t, K): operand i, tk = operand t.)
#define SQPH_GJ_F(i, t, K) "v_fmac_f64_dpp %" #i ", %" #i ", %" #t " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
    template <int K>
    static __device__ __forceinline__ void gj_fmacs(T (&M)[16], T tk) {
#ifdef SQPH_SIM
#pragma unroll
        for (int i = K + 1; i < 16; i++) M[i] = __builtin_fma(bcast16<K>(M[i]), tk, M[i]);
#else
        if constexpr (K == 0)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 15, 0) SQPH_GJ_F(1, 15, 0) SQPH_GJ_F(2, 15, 0) SQPH_GJ_F(3, 15, 0) SQPH_GJ_F(4, 15, 0)
                SQPH_GJ_F(5, 15, 0) SQPH_GJ_F(6, 15, 0) SQPH_GJ_F(7, 15, 0) SQPH_GJ_F(8, 15, 0) SQPH_GJ_F(9, 15, 0)
                SQPH_GJ_F(10, 15, 0) SQPH_GJ_F(11, 15, 0) SQPH_GJ_F(12, 15, 0) SQPH_GJ_F(13, 15, 0) SQPH_GJ_F(14, 15, 0) "s_nop 1"
                : "+v"(M[1]), "+v"(M[2]), "+v"(M[3]), "+v"(M[4]), "+v"(M[5]), "+v"(M[6]), "+v"(M[7]), "+v"(M[8]), "+v"(M[9]),
                  "+v"(M[10]), "+v"(M[11]), "+v"(M[12]), "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 1)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 14, 1) SQPH_GJ_F(1, 14, 1) SQPH_GJ_F(2, 14, 1) SQPH_GJ_F(3, 14, 1) SQPH_GJ_F(4, 14, 1)
                SQPH_GJ_F(5, 14, 1) SQPH_GJ_F(6, 14, 1) SQPH_GJ_F(7, 14, 1) SQPH_GJ_F(8, 14, 1) SQPH_GJ_F(9, 14, 1)
                SQPH_GJ_F(10, 14, 1) SQPH_GJ_F(11, 14, 1) SQPH_GJ_F(12, 14, 1) SQPH_GJ_F(13, 14, 1) "s_nop 1"
                : "+v"(M[2]), "+v"(M[3]), "+v"(M[4]), "+v"(M[5]), "+v"(M[6]), "+v"(M[7]), "+v"(M[8]), "+v"(M[9]), "+v"(M[10]),
                  "+v"(M[11]), "+v"(M[12]), "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 2)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 13, 2) SQPH_GJ_F(1, 13, 2) SQPH_GJ_F(2, 13, 2) SQPH_GJ_F(3, 13, 2) SQPH_GJ_F(4, 13, 2)
                SQPH_GJ_F(5, 13, 2) SQPH_GJ_F(6, 13, 2) SQPH_GJ_F(7, 13, 2) SQPH_GJ_F(8, 13, 2) SQPH_GJ_F(9, 13, 2)
                SQPH_GJ_F(10, 13, 2) SQPH_GJ_F(11, 13, 2) SQPH_GJ_F(12, 13, 2) "s_nop 1"
                : "+v"(M[3]), "+v"(M[4]), "+v"(M[5]), "+v"(M[6]), "+v"(M[7]), "+v"(M[8]), "+v"(M[9]), "+v"(M[10]), "+v"(M[11]),
                  "+v"(M[12]), "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 3)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 12, 3) SQPH_GJ_F(1, 12, 3) SQPH_GJ_F(2, 12, 3) SQPH_GJ_F(3, 12, 3) SQPH_GJ_F(4, 12, 3)
                SQPH_GJ_F(5, 12, 3) SQPH_GJ_F(6, 12, 3) SQPH_GJ_F(7, 12, 3) SQPH_GJ_F(8, 12, 3) SQPH_GJ_F(9, 12, 3)
                SQPH_GJ_F(10, 12, 3) SQPH_GJ_F(11, 12, 3) "s_nop 1"
                : "+v"(M[4]), "+v"(M[5]), "+v"(M[6]), "+v"(M[7]), "+v"(M[8]), "+v"(M[9]), "+v"(M[10]), "+v"(M[11]), "+v"(M[12]),
                  "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 4)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 11, 4) SQPH_GJ_F(1, 11, 4) SQPH_GJ_F(2, 11, 4) SQPH_GJ_F(3, 11, 4) SQPH_GJ_F(4, 11, 4)
                SQPH_GJ_F(5, 11, 4) SQPH_GJ_F(6, 11, 4) SQPH_GJ_F(7, 11, 4) SQPH_GJ_F(8, 11, 4) SQPH_GJ_F(9, 11, 4)
                SQPH_GJ_F(10, 11, 4) "s_nop 1"
                : "+v"(M[5]), "+v"(M[6]), "+v"(M[7]), "+v"(M[8]), "+v"(M[9]), "+v"(M[10]), "+v"(M[11]), "+v"(M[12]), "+v"(M[13]),
                  "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 5)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 10, 5) SQPH_GJ_F(1, 10, 5) SQPH_GJ_F(2, 10, 5) SQPH_GJ_F(3, 10, 5) SQPH_GJ_F(4, 10, 5)
                SQPH_GJ_F(5, 10, 5) SQPH_GJ_F(6, 10, 5) SQPH_GJ_F(7, 10, 5) SQPH_GJ_F(8, 10, 5) SQPH_GJ_F(9, 10, 5) "s_nop 1"
                : "+v"(M[6]), "+v"(M[7]), "+v"(M[8]), "+v"(M[9]), "+v"(M[10]), "+v"(M[11]), "+v"(M[12]), "+v"(M[13]),
                  "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 6)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 9, 6) SQPH_GJ_F(1, 9, 6) SQPH_GJ_F(2, 9, 6) SQPH_GJ_F(3, 9, 6) SQPH_GJ_F(4, 9, 6)
                SQPH_GJ_F(5, 9, 6) SQPH_GJ_F(6, 9, 6) SQPH_GJ_F(7, 9, 6) SQPH_GJ_F(8, 9, 6) "s_nop 1"
                : "+v"(M[7]), "+v"(M[8]), "+v"(M[9]), "+v"(M[10]), "+v"(M[11]), "+v"(M[12]), "+v"(M[13]), "+v"(M[14]),
                  "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 7)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 8, 7) SQPH_GJ_F(1, 8, 7) SQPH_GJ_F(2, 8, 7) SQPH_GJ_F(3, 8, 7) SQPH_GJ_F(4, 8, 7)
                SQPH_GJ_F(5, 8, 7) SQPH_GJ_F(6, 8, 7) SQPH_GJ_F(7, 8, 7) "s_nop 1"
                : "+v"(M[8]), "+v"(M[9]), "+v"(M[10]), "+v"(M[11]), "+v"(M[12]), "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 8)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 7, 8) SQPH_GJ_F(1, 7, 8) SQPH_GJ_F(2, 7, 8) SQPH_GJ_F(3, 7, 8) SQPH_GJ_F(4, 7, 8)
                SQPH_GJ_F(5, 7, 8) SQPH_GJ_F(6, 7, 8) "s_nop 1"
                : "+v"(M[9]), "+v"(M[10]), "+v"(M[11]), "+v"(M[12]), "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 9)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 6, 9) SQPH_GJ_F(1, 6, 9) SQPH_GJ_F(2, 6, 9) SQPH_GJ_F(3, 6, 9) SQPH_GJ_F(4, 6, 9)
                SQPH_GJ_F(5, 6, 9) "s_nop 1"
                : "+v"(M[10]), "+v"(M[11]), "+v"(M[12]), "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 10)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 5, 10) SQPH_GJ_F(1, 5, 10) SQPH_GJ_F(2, 5, 10) SQPH_GJ_F(3, 5, 10) SQPH_GJ_F(4, 5, 10) "s_nop 1"
                : "+v"(M[11]), "+v"(M[12]), "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 11)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 4, 11) SQPH_GJ_F(1, 4, 11) SQPH_GJ_F(2, 4, 11) SQPH_GJ_F(3, 4, 11) "s_nop 1"
                : "+v"(M[12]), "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 12)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 3, 12) SQPH_GJ_F(1, 3, 12) SQPH_GJ_F(2, 3, 12) "s_nop 1"
                : "+v"(M[13]), "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 13)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 2, 13) SQPH_GJ_F(1, 2, 13) "s_nop 1"
                : "+v"(M[14]), "+v"(M[15])
                : "v"(tk));
        else if constexpr (K == 14)
            asm("s_nop 1\n\t" SQPH_GJ_F(0, 1, 14) "s_nop 1"
                : "+v"(M[15])
                : "v"(tk));
#endif
    }
#undef SQPH_GJ_F
    template <int K>
    static __device__ __forceinline__ void gj_steps(T (&M)[16], int j, T &mypiv, bool &bad) {
        const T d = bcast16<K>(M[K]);
        bad = bad || !(d > T(0)) || !(d * T(0) == T(0));
        const T dinv = fast_rcp(d);
        T tk = (j == K) ? d + T(1) : M[K];
        tk = -tk * dinv;
        mypiv = (j == K) ? d : mypiv;
        if constexpr (K + 1 < 16) {
            gj_fmacs<K>(M, tk);
            gj_steps<K + 1>(M, j, mypiv, bad);
        }
    }
    template <int I>
    static __device__ __forceinline__ void scale_rows(T (&M)[16], int j, T rs) {
        // row I of the result: rs_I * (I > j ? M[I][j] : I == j ? 1 : 0), rs_I from lane I
        const T v = fmac_bcast16<I>(T(0), rs, M[I]);
        M[I] = I > j ? v : (I == j ? rs : T(0));
        if constexpr (I + 1 < 16) scale_rows<I + 1>(M, j, rs);
    }
    // BRANCHFREE: the failure flag is raised behind a wave-uniform condition (a ballot) instead of a lane-divergent branch: inside the
    // block-row kernel's elimination loop one divergent branch made the compiler structurize the whole loop (admm_csrb_kernel.h)
    template <bool BRANCHFREE = false>
    static __device__ __forceinline__ void diag_block(T *Sjj, T *TB, const T *djp, T *flag, int l) {
        const int j = l & 15;
        T M[16];
#pragma unroll
        for (int i = 0; i < 16; i++) M[i] = Sjj[ix(i, j)];
        T mypiv = T(1);
        bool bad = false;
        gj_steps<0>(M, j, mypiv, bad);
        const T rs = T(1) / (T)sqrt((double)(mypiv > T(0) ? mypiv : T(1)));
        scale_rows<0>(M, j, rs);
        const T dj = djp[j];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            TB[ix(i, j)] = M[i];           // unscaled: operand of this step's panel products
            Sjj[ix(i, j)] = M[i] * dj;     // W_JJ = Winv_JJ D_J^-1/2 (columns)
        }
        if constexpr (BRANCHFREE) {
#ifdef SQPH_SIM
            if (bad) flag[1] = T(1);
#else
            if (__builtin_amdgcn_ballot_w64(bad) != 0) flag[1] = T(1);  // (wave-uniform condition: every lane stores)
#endif
        } else {
            if (bad) flag[1] = T(1);
        }
    }

    // ------------------------------------------------------------------ factorisation
    // `at` is the A register tile (rows R s + r, columns C k + c); rho_l = lds + O_RHO holds rho per row (0 beyond m).
    // Returns false (block-uniform) when S is not positive definite / not finite.  Leaves W in SB.
    template <typename TIN, int O_PST, bool PST>
    static __device__ __forceinline__ bool factor(const TIN *__restrict__ gP, const T (&at)[TR][TC], int n, int m, T sigma, T *lds, int t,
                                                  bool p_staged SQPH_STICK_ARGS) {
        const int wave = wave_of(t), l = t & 63, lr = l & 15, lq = l >> 4;
        const int r = t % R, c = t / R;
        const T *rho_l = lds + O_RHO;
        T *sj = lds + O_SJ, *flag = lds + O_FLAG, *XS = lds + O_XS, *SB = lds + O_SB, *TB = lds + O_TB;
        if (t < 2) flag[t] = T(0);
        // ---- phase 1: S = A' diag(rho) A, lower blocks; wavefront w accumulates blocks w, w + NW, ... in the D layout
        sqph_acc4 acc[NQ];
        int bI[NQ], bJ[NQ];
        bool bv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int b = wave + NW * q;
            bv[q] = b < NBLK;
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= b && I + 1 < NB) I++;
            bI[q] = I;
            bJ[q] = bv[q] ? b - I * (I + 1) / 2 : 0;
#pragma unroll
            for (int e = 0; e < 4; e++) acc[q].v[e] = T(0);
        }
        const T *xa[NQ], *xb[NQ];  // my element of the k-step's first row in the A-operand / B-operand column block of accumulator q
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            xa[q] = XS + bI[q] * BS + (SWZ ? 0 : lq * 17 + lr);  // (swizzled blocks: the element is found per k-step)
            xb[q] = XS + bJ[q] * BS + (SWZ ? 0 : lq * 17 + lr);
        }
        T *xw[TC];  // where my TC entries of a tile row go in the staged block
#pragma unroll
        for (int k = 0; k < TC; k++) {
            const int j = L::col(c, k);
            xw[k] = XS + (j >> 4) * BS + ix(r & 15, j & 15);
        }
        constexpr bool ALLV = NBLK % NW == 0;  // every wavefront has NQ blocks: no per-block test
#pragma unroll
        for (int sh = 0; sh < TR * RH; sh++) {
            const int s = sh / RH, h = sh % RH;  // tile row and its 16-row half (RH = 1: the whole tile row)
            if (R * s + 16 * h < m) {
                __syncthreads();
                if (RH == 1 || (r >> 4) == h) {
#pragma unroll
                    for (int k = 0; k < TC; k++) *xw[k] = at[s][k];
                }
                __syncthreads();
                // all four k-steps of the block, straight-line (rows beyond m are zero rows of the tile with rho = 0): the operands of
                // a k-step are requested before the products of the one before it wait for theirs
#pragma unroll
                for (int kq = 0; kq < 4; kq++) {
                    T av[NQ], bw[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        av[q] = xa[q][SWZ ? ix(lq + 4 * kq, lr) : 4 * kq * 17];
                        bw[q] = xb[q][SWZ ? ix(lq + 4 * kq, lr) : 4 * kq * 17];
                    }
                    const T rk = rho_l[R * s + 16 * h + 4 * kq + lq];
#pragma unroll
                    for (int q = 0; q < NQ; q++)
                        if (ALLV || bv[q]) mfma16(av[q], bw[q] * rk, acc[q]);
                }
            }
        }
        SQPH_STICK(1)
        // ---- phase 2: + P_lower + sigma I, Jacobi scaling, blocks to LDS
#ifndef SQPH_SIM
        if (PST && p_staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my share of the P block has landed in LDS
#endif
        __syncthreads();
        {
#ifdef SQPH_SIM
            const TIN *Pst = reinterpret_cast<const TIN *>(lds + O_PST);
#else
            // (an LDS-qualified pointer: through a generic one the compiler merges the two paths below into flat loads)
            const __attribute__((address_space(3))) TIN *Pst = (const __attribute__((address_space(3))) TIN *)reinterpret_cast<const TIN *>(lds + O_PST);
#endif
            const int nn = n * n;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                if (ALLV || bv[q]) {
                    const int j = 16 * bJ[q] + lr;
                    int ix[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int i = 16 * bI[q] + lq + 4 * e;
                        const int lo = i > j ? i : j, hi = i > j ? j : i;
                        const int x = hi * n + lo;  // only the lower triangle of P reaches the reference's factor (LDLT<.,Lower>)
                        ix[e] = x < nn ? x : 0;     // (padding: any valid element, the value is not used)
                    }
                    T pv[4];
                    if (PST && p_staged) {
#pragma unroll
                        for (int e = 0; e < 4; e++) pv[e] = (T)Pst[ix[e]];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) pv[e] = (T)gP[ix[e]];
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int i = 16 * bI[q] + lq + 4 * e;
                        const bool ok = i < n && j < n;
                        // the padding (rows / columns >= n of the 16 NB square) is an identity block: its factor and inverse are identities
                        const T v = acc[q].v[e] + pv[e] + (i == j ? sigma : T(0));
                        acc[q].v[e] = ok ? v : (i == j ? T(1) : T(0));
                    }
                    if (bI[q] == bJ[q]) {  // a diagonal block (scalar test): the lanes with lr % 4 == lq hold its diagonal, element lr / 4
                        // (values first, selects second: a select between two loads of the accumulator array becomes a load through a
                        // selected address, which sends the whole array to scratch)
                        const T a0 = acc[q].v[0], a1 = acc[q].v[1], a2 = acc[q].v[2], a3 = acc[q].v[3];
                        const T d01 = (lr & 4) ? a1 : a0, d23 = (lr & 4) ? a3 : a2;
                        const T dv = (lr & 8) ? d23 : d01;
                        if ((lr & 3) == lq) sj[16 * bI[q] + lr] = dv;
                    }
                }
            }
        }
        __syncthreads();  // every read of P is done (SB aliases it); the diagonal is published
        if (t < 16 * NB) {
            const T d = sj[t];
            const bool bad = !(d > T(0)) || !(d * T(0) == T(0));  // non-positive / non-finite diagonal => not SPD
            sj[t] = bad ? T(1) : fast_rsqrt(d);
            if (bad) flag[0] = T(1);
        }
        __syncthreads();
        if (flag[0] != T(0)) return false;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if (ALLV || bv[q]) {
                const T dc = sj[16 * bJ[q] + lr];
                T *b = SB + (wave + NW * q) * BS;
#pragma unroll
                for (int e = 0; e < 4; e++) b[ix(lq + 4 * e, lr)] = acc[q].v[e] * sj[16 * bI[q] + lq + 4 * e] * dc;
            }
        }
        SQPH_STICK(2)
        // ---- phase 3: blocked elimination.  Step J: (1) the diagonal block M_JJ -> Winv_JJ inside one wavefront; (2) panel
        // L_IJ = M_IJ Winv_JJ' (I > J) and the finished rows W_JK = Winv_JJ E_JK (K < J); (3) E_IJ = -L_IJ Winv_JJ,
        // E_IK -= L_IJ W_JK (K < J), M_IK -= L_IJ L_KJ' (J < K <= I).  E is the unit-block-lower inverse in the making (its columns
        // carry the Jacobi scale from their creation on), stored in place of the eliminated blocks; L_IJ lives in XS for its step.
        // Look-ahead: in part (3) of step J one wavefront updates M_J+1,J+1 first and eliminates it right away (into the other of
        // the two Winv buffers) while the others do the rest of (3) — two barriers per step, the diagonal blocks off the critical path.
        __syncthreads();
        if (wave == 0) diag_block(SB + blk(0, 0) * BS, TB, sj, flag, l);
#pragma unroll 1
        for (int J = 0; J < NB; J++) {
            const T *Wd = TB + (J & 1) * BS;  // Winv_JJ, unscaled
            __syncthreads();
            int job = 0;
#pragma unroll 1
            for (int I = J + 1; I < NB; I++, job++) {
                if (job % NW == wave) {
                    sqph_acc4 a = {{0, 0, 0, 0}};
                    const T *Mij = SB + blk(I, J) * BS;
#pragma unroll
                    for (int kq = 0; kq < 4; kq++) mfma16(opN(Mij, kq, lr, lq), opN(Wd, kq, lr, lq), a);
                    stD(XS + I * BS, lr, lq, a);
                }
            }
#pragma unroll 1
            for (int K = 0; K < J; K++, job++) {
                if (job % NW == wave) {
                    sqph_acc4 a = {{0, 0, 0, 0}};
                    T *Ejk = SB + blk(J, K) * BS;
#pragma unroll
                    for (int kq = 0; kq < 4; kq++) mfma16(opN(Wd, kq, lr, lq), opT(Ejk, kq, lr, lq), a);
                    stD(Ejk, lr, lq, a);
                }
            }
            if (J == NB - 1) break;
            __syncthreads();
            const int la = (J + 1) % NW;  // the look-ahead wavefront of this step
            if (wave == la) {
                sqph_acc4 a;
                T *Mik = SB + blk(J + 1, J + 1) * BS;
                const T *Li = XS + (J + 1) * BS;
                ldD(Mik, lr, lq, a);
#pragma unroll
                for (int kq = 0; kq < 4; kq++) mfma16(-opN(Li, kq, lr, lq), opN(Li, kq, lr, lq), a);
                stD(Mik, lr, lq, a);
                wave_fence();  // the block just written is read back by this wavefront only
                diag_block(Mik, TB + ((J + 1) & 1) * BS, sj + 16 * (J + 1), flag, l);
            }
            if (NW == 1 || wave != la) {
                // the other wavefronts share the rest of part (3); `me` counts them 0 .. NW - 2
                const int nother = NW > 1 ? NW - 1 : 1;
                const int me = NW > 1 ? (wave - la - 1 + NW) % NW : 0;
                job = 0;
                const T dc = sj[16 * J + lr];
#pragma unroll 1
                for (int I = J + 1; I < NB; I++) {
                    const T *Li = XS + I * BS;
                    if (job++ % nother == me) {
                        sqph_acc4 a = {{0, 0, 0, 0}};
#pragma unroll
                        for (int kq = 0; kq < 4; kq++) mfma16(-opN(Li, kq, lr, lq), opT(Wd, kq, lr, lq), a);
#pragma unroll
                        for (int e = 0; e < 4; e++) a.v[e] *= dc;
                        stD(SB + blk(I, J) * BS, lr, lq, a);
                    }
#pragma unroll 1
                    for (int K = 0; K < J; K++) {
                        if (job++ % nother == me) {
                            sqph_acc4 a;
                            T *Eik = SB + blk(I, K) * BS;
                            const T *Wjk = SB + blk(J, K) * BS;
                            ldD(Eik, lr, lq, a);
#pragma unroll
                            for (int kq = 0; kq < 4; kq++) mfma16(-opN(Li, kq, lr, lq), opT(Wjk, kq, lr, lq), a);
                            stD(Eik, lr, lq, a);
                        }
                    }
#pragma unroll 1
                    for (int K = J + 1; K <= I; K++) {
                        if (I == J + 1 && K == J + 1) continue;  // (the look-ahead wavefront's block)
                        if (job++ % nother == me) {
                            sqph_acc4 a;
                            T *Mik = SB + blk(I, K) * BS;
                            const T *Lk = XS + K * BS;
                            ldD(Mik, lr, lq, a);
#pragma unroll
                            for (int kq = 0; kq < 4; kq++) mfma16(-opN(Li, kq, lr, lq), opN(Lk, kq, lr, lq), a);
                            stD(Mik, lr, lq, a);
                        }
                    }
                }
            }
        }
        __syncthreads();
        SQPH_STICK(3)
        return flag[1] == T(0);
    }

    // ------------------------------------------------------------------ B = A W' in place over the A tile
    // B' = W A' one block of 16 rows of A at a time: the A-operand rows are W rows permuted so that the D layout is the register
    // tile (D row lq + 4 e of column group Jq  <->  W row C (4 Jq + e) + c), the B-operand is the staged block of A; the k-steps
    // beyond a column group's last column (W lower triangular) and beyond n are not run.
    // STACK: the rows of W' that share the last tile rows of B (stacked row R s + r = SOFF + j) are added from SB.
    template <bool STACK, int SOFF>
    static __device__ __forceinline__ void build_B(T (&at)[TR][TC], int n, int m, T *lds, int t SQPH_STICK_ARGS) {
        const int wave = wave_of(t), l = t & 63, lr = l & 15, lq = l >> 4;
        const int r = t % R, c = t / R;
        T *XS = lds + O_XS;
        const T *SB = lds + O_SB;
        if constexpr (RH > 1) {
            // 32-row grids: B' = W A' per 16-row half of a tile row; wavefront w multiplies block row w of W (W rows 16 w ..) into the
            // staged half-block of A — D[i][j] = B[A row j][W row 16 w + i] — and parks D in LDS, where lane (r, c) of that half picks
            // up B[R s + r][16 k + c] = D_k[c][r & 15]
            static_assert(!STACK && (C == 16 || C == 8), "tile column k is column 8 (k & 1) + c or c of result block k / 2 or k");
            T *DS = lds + O_DS;
            // the k-loop below runs whole 16-column blocks: the columns of the staged block between the tile's width and the block
            // boundary are never written by the tiles, and a solve() that LOADED its factor finds whatever the CU's previous workgroup
            // left there (after a factorisation it is the finite L panel) — W is zero in those columns, but 0 x NaN is NaN
            if constexpr (16 * NB > C * TC) {
                constexpr int PADC = 16 * NB - C * TC;
                for (int e = t; e < 16 * PADC; e += NT) {
                    const int j = C * TC + e % PADC;
                    XS[(j >> 4) * BS + ix(e / PADC, j & 15)] = T(0);
                }
            }
#pragma unroll
            for (int sh = 0; sh < TR * RH; sh++) {
                const int s = sh / RH, h = sh % RH;
                if (R * s + 16 * h < m) {
                    __syncthreads();
                    if ((r >> 4) == h) {
#pragma unroll
                        for (int k = 0; k < TC; k++) {
                            const int j = L::col(c, k);
                            XS[(j >> 4) * BS + ix(r & 15, j & 15)] = at[s][k];
                        }
                    }
                    __syncthreads();
#pragma unroll 1
                    for (int w = wave; w < NB; w += NW) {
                        sqph_acc4 a = {{0, 0, 0, 0}};
#pragma unroll 1
                        for (int jb = 0; jb <= w; jb++) {
                            const T *Wb = SB + blk(w, jb) * BS, *Ab = XS + jb * BS;
#pragma unroll
                            for (int kq = 0; kq < 4; kq++) mfma16(opN(Wb, kq, lr, lq), opN(Ab, kq, lr, lq), a);
                        }
                        stD(DS + w * BS, lr, lq, a);
                    }
                    __syncthreads();
                    if ((r >> 4) == h) {
#pragma unroll
                        for (int k = 0; k < TC; k++) at[s][k] = SQPH_TILE_QUANT(DS[(L::col(c, k) >> 4) * BS + ix(L::col(c, k) & 15, r & 15)]);
                    }
                }
            }
            return;
        }
        // my element of W in the A-operand of accumulator q and column block jb: W[cw][16 jb + ko + lq] with cw the W row of my D
        // row; where that block lies above the diagonal (or cw is not a column of the problem) the address is that of a zero row
        const T *wa[NJQ][NB];
        T *zrow = lds + O_TB;  // (the diagonal-block scratch is free here)
        __syncthreads();
        if (t < 32) zrow[t] = T(0);
#pragma unroll
        for (int q = 0; q < NJQ; q++) {
            const int cw = C * (4 * q + (lr >> 2)) + 4 * wave + (lr & 3);
            const bool in = 4 * q + (lr >> 2) < TC && cw < n;
#pragma unroll
            for (int jb = 0; jb < NB; jb++)
                wa[q][jb] = (in && (cw >> 4) >= jb) ? SB + blk(cw >> 4, jb) * BS + (SWZ ? (cw & 15) * 16 : (cw & 15) * 17 + lq) : zrow + (SWZ ? 0 : lq);
        }
        const int cwlo = (C * (lr >> 2) + 4 * wave + (lr & 3)) & 15;  // my W row inside its block (the same for every q: C is a multiple of 16 or the rows of a q differ by C * 4)
        const T *xv = XS + (SWZ ? 0 : lr * 17 + lq);
#pragma unroll
        for (int s = 0; s < TR; s++) {
            if (R * s < m) {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < TC; k++) {
                    const int j = L::col(c, k);
                    XS[(j >> 4) * BS + ix(r, j & 15)] = at[s][k];
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NJQ; q++) {
                    sqph_acc4 a = {{0, 0, 0, 0}};
                    // columns of W beyond the group's last row index are zero (lower triangular); the count is a compile-time one
                    // (the columns between n and the padded size hold zeros in A and in W)
                    constexpr int KEND = C * (4 * NJQ) < L::NP ? C * (4 * NJQ) : L::NP;
                    const int kend = C * (4 * q + 4) < KEND ? C * (4 * q + 4) : KEND;
                    T wv[KEND / 4], av[KEND / 4];
#pragma unroll
                    for (int ks = 0; ks < KEND / 4; ks++) {
                        if (4 * ks < kend) {
                            wv[ks] = wa[q][ks >> 2][SWZ ? ((((4 * ks) & 15) + lq) ^ cwlo) : ((4 * ks) & 15)];
                            av[ks] = xv[(ks >> 2) * BS + (SWZ ? ix(lr, ((4 * ks) & 15) + lq) : ((4 * ks) & 15))];
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < KEND / 4; ks++)
                        if (4 * ks < kend) mfma16(wv[ks], av[ks], a);
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (4 * q + e < TC) at[s][4 * q + e] = SQPH_TILE_QUANT(a.v[e]);
                }
            }
            if constexpr (STACK) {
                if (R * s + R - 1 >= SOFF) {  // (compile-time: tile rows below the W' rows skip this)
                    const int jp = R * s + r - SOFF;  // row of W' at stacked row R s + r
#pragma unroll
                    for (int k = 0; k < TC; k++) {
                        const T w = (jp >= 0) ? Wget(SB, L::col(c, k), jp >= 0 ? jp : 0, n) : T(0);
                        at[s][k] += SQPH_TILE_QUANT(w);
                    }
                }
            }
        }
    }
    // the W' tile of the iteration: vt[u][k] = W[C k + c][jp(u)] with jp(u) = R u + r (padded operator) or R (TR + u) + r - SOFF (stacked)
    template <int TX, bool STACK, int SOFF>
    static __device__ __forceinline__ void load_vt(const T *lds, int n, int r, int c, T (&vt)[TX][TC]) {
        const T *SB = lds + O_SB;
#pragma unroll
        for (int u = 0; u < TX; u++) {
            const int jp = STACK ? R * (TR + u) + r - SOFF : R * u + r;
#pragma unroll
            for (int k = 0; k < TC; k++) {
                const bool zero = STACK ? (C * k + C - 1 < R * (TR + u) - SOFF) : L::vt_zero(u, k);
                const T w = (!zero && jp >= 0) ? Wget(SB, L::col(c, k), jp >= 0 ? jp : 0, n) : T(0);
                vt[u][k] = zero ? T(0) : SQPH_TILE_QUANT(w);
            }
        }
    }
    // factor residency: W (canonical n x n column-major, zeros above the diagonal) <-> the blocks in LDS
    static __device__ __forceinline__ void store_W(T *__restrict__ gW, int n, const T *lds, int t) {
        const T *SB = lds + O_SB;
        for (int e = t; e < n * n; e += NT) {
            const int i = e % n, j = e / n;
            gW[e] = i >= j ? Wget(SB, i, j, n) : T(0);
        }
    }
    static __device__ __forceinline__ void load_W(const T *__restrict__ gW, int n, T *lds, int t) {
        T *SB = lds + O_SB;
        for (int e = t; e < NBLK * 256; e += NT) {
            const int b = e >> 8, ii = (e >> 4) & 15, jj = e & 15;
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= b) I++;
            const int K = b - I * (I + 1) / 2;
            const int i = 16 * I + ii, j = 16 * K + jj;
            SB[b * BS + ix(ii, jj)] = (i < n && j < n && i >= j) ? gW[(long)j * n + i] : T(0);
        }
    }
};

}  // namespace sqph
