// Sparse-A ADMM kernel, block-row form (BASELINE config 5: n = 200, m = 400, CSR A): ONE 512-lane workgroup (8 wavefronts) per QP.
//
// The factor W (S^-1 = W'W, lower triangular, n <= 16 NB) lives in the CU's register file as 16 x 16 blocks in the accumulator
// layout of v_mfma_f64_16x16x4_f64 (lane l of a wavefront holds rows (l >> 4) + 4 e, e < 4, of column l & 15).  Wavefront w owns
// the block rows I1 = NB - 1 - w and I0 = w (NB + 1 blocks, 4 doubles per lane each); the last wavefront owns none and
// eliminates the diagonal blocks.  What this layout buys over the 32 x 32 lane grid of admm_csr_kernel.h:
//   * the set-up runs on the matrix pipe with the blocks in registers: blocked elimination, only the current panel column / block
//     row (NB blocks) and two diagonal blocks go through LDS (the 2-D cyclic tile of the other kernel took n pivots of ~80
//     instructions and a workgroup barrier each);
//   * y1 = W t is reduced inside the wavefront (a block row's 16 row sums need the 16 lanes of a DPP row: 8 partial sums per lane
//     through a private LDS slice, no workgroup barrier) and handed to x~ = W' y1 by DPP row broadcast inside the multiply-add;
//     the column sums of W' y1 are reduced over the four DPP rows by v_permlane32_swap / v_permlane16_swap and cross the
//     wavefronts as 8 partial sums per column (14 KB instead of 2 x 57 KB of staging per iteration);
//   * with 256 VGPRs per lane the entries of A a lane sums — its slice of a row for A x~, of a column for A' w — stay in
//     registers (values and packed 16-bit indices, at most KR per lane and orientation): a sparse product is one LDS gather
//     per entry instead of an index load and two gathers.
// Iteration (reference src/qp.cpp:84-103 on the Schur-ordered system; four workgroup barriers):
//     t = (sigma x - q) + A' w  |  y1 = W t ; x~ partials = W' y1  |  x~, x  |  z~ = A x~ ; z, y ; w = rho z - y
// Code generation (round 6): csrb.hip / csrb_sp.hip are compiled with -mllvm -structurizecfg-skip-uniform-regions (build.py: CSB_FLAGS).
// All control flow of the set-up is wave-uniform; structurized, every conditional update of a block kept an old and a new copy of all
// NB + 1 blocks live (224 of 256 VGPRs, 160 spilled registers, 56 v_mov_b64 per slot and elimination step).  A block's column is a
// compile-time function of its slot (Own / slot_of), and a step reaches its blocks through dispatch() / static_while() — a taken branch
// per skipped slot body was ~100 cycles.  The no-check instantiation has no scratch access; profiles/r06_ab.txt has the steps.
// Numerics: the formulas of admm_csr_kernel.h / admm_generic.h, fp64 arithmetic, TIN inputs.
// Requirements checked by the host: n <= 16 NB, m <= 512, CSR rows sorted by column without duplicates, everything within LDS.
// A QP whose rows (columns) need more than KR entries per lane takes the LDS form of that sparse product (block-uniform branch).
#pragma once
#include <type_traits>

#include "admm_csr_kernel.h"

#ifndef SQPH_CSB_KR
#define SQPH_CSB_KR 14
#endif

#ifndef SQPH_CSB_EB
#define SQPH_CSB_EB 3  // CSC entries of a column accumulated per batch in the S phase (round 6: 3 / 4 / 6 / 8 measured 19.43 / 19.42 / 19.58 / 19.55 ms)
#endif
#ifndef SQPH_CSB_PLACE_MIN_ITERS
#define SQPH_CSB_PLACE_MIN_ITERS 150  // settings.max_iter from which the register-resident slices are placed (CsbKernel::run)
#endif
#ifndef SQPH_CSB_EB_SP
#define SQPH_CSB_EB_SP 4  // ... in the sparse-P instantiations (no prefetch array of P there: the wider batch pays)
#endif

// keeps the instruction scheduler from moving work across (the unrolled slot loops of the set-up: hoisting every slot's LDS operand
// loads to the top of the loop needs 8 registers per slot next to the 112 of the blocks)
#if defined(SQPH_SIM) || !defined(SQPH_CSB_SCHED_FENCE)
#define SQPH_SLOT_FENCE() (void)0
#else
#define SQPH_SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// -DSQPH_PHASE_TIMING (debug builds, tools/phase_timing_csb.py): s_memtime ticks per phase of one wavefront, returned in x[0..16)
#ifdef SQPH_PHASE_TIMING
#define SQPH_FTICK_ARGS , unsigned long long (&tacc)[16], unsigned long long &tprev
#define SQPH_FTICK_PASS , tacc, tprev
#define SQPH_FTICK(k) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tprev; tprev = tn_; }
#define SQPH_BTICK(k) SQPH_FTICK(k)
#else
#define SQPH_BTICK(k)
#define SQPH_FTICK_ARGS
#define SQPH_FTICK_PASS
#define SQPH_FTICK(k)
#endif
// -DSQPH_PT_SETUP=1 / 2 (with -DSQPH_PHASE_TIMING): the per-iteration slots 3-6, 8 count sub-phases of the S phase / of the sparse
// loading and lane maps instead (tools/phase_timing_csb.py --setup)
#if defined(SQPH_PHASE_TIMING) && defined(SQPH_PT_SETUP)
#define SQPH_ITICK(k)
#define SQPH_STICK1(k) { if (SQPH_PT_SETUP == 1) SQPH_FTICK(k) }
#define SQPH_STICK2(k) { if (SQPH_PT_SETUP == 2) SQPH_FTICK(k) }
#else
#define SQPH_ITICK(k) SQPH_BTICK(k)
#define SQPH_STICK1(k)
#define SQPH_STICK2(k)
#endif

namespace sqph {

template <int NB>
struct CsbLayout {
    static constexpr int NT = 512, NW = 8;
    static constexpr int NP = 16 * NB;
    static constexpr int BS = 16 * 17;  // a 16 x 16 block, rows padded to 17 doubles (admm_wg_msetup.h: opN / opT / D accesses conflict-free)
    static constexpr int LDP = NP + 1;  // S panel column stride
    static constexpr int ev(int x) { return (x + 1) & ~1; }
    static constexpr int mx(int a, int b) { return a > b ? a : b; }
    // the work area (offsets in doubles from 0) is shared by: the S panel of the set-up | the elimination's panel / diagonal
    // buffers | the iteration's staging | the lane maps while they are built
    static constexpr int W_PANEL = 32 * LDP;
    static constexpr int O_XS = 0;                  // [NB][BS] step J: L_IJ at I > J, W_JK at K < J
    static constexpr int O_TB = O_XS + NB * BS;     // [2][BS]  unscaled inverse of the current / next diagonal block
    static constexpr int O_MD = O_TB + 2 * BS;      // [2][BS]  the diagonal block handed to the eliminating wavefront, W_JJ on return
    static constexpr int O_LP = O_MD + 2 * BS;      // [NW][BS] per wavefront: -L_IJ Winv_JJ on its way from the D layout to the A-operand layout
    static constexpr int W_ELIM = O_LP + NW * BS;
    // [8 waves][8 values x 4 DPP rows][PWS] partial sums of y1 = W t.  Element j of row (v, lq) sits at (j + 4 v) mod 16 of a row of
    // PWS = 18 doubles: with plain rows of 16 the four ds_read_b128 of the reduction were 4-way bank conflicted (64 instead of 16 LDS
    // cycles; tools/xp/lds_bank_model.py finds this rotation + stride conflict-free for the reads and the eight ds_write_b64)
    static constexpr int PWS = 18, PWW = 32 * PWS;
    static constexpr int O_PW = 0;
    static constexpr int O_XP = O_PW + 8 * PWW;     // [8 waves][NP] partial sums of x~ = W' y1, wavefront-major (lane-consecutive stores and loads)
    static constexpr int W_ITER = O_XP + NP * 8;
    static constexpr int W_MAP = (2 * NT * 2 + 2 * (544 + 32) * 4 + 7) / 8;
    static constexpr int WORK = ev(mx(mx(W_PANEL, W_ELIM), mx(W_ITER, W_MAP)));
    static constexpr int o_t = WORK;                // t = sigma x - q + A'w, plain-indexed
    static constexpr int o_xt = o_t + NP;
    static constexpr int o_ux = o_xt + NP;
    static constexpr int o_xv = o_ux + NP;         // the primal iterate x (LDS, not a register: see run())
    static constexpr int o_qv = o_xv + NP;
    static constexpr int o_sj = o_qv + NP;
    static constexpr int o_flag = o_sj + NP;        // [8]
    static constexpr int o_red = o_flag + 8;        // [8][8]
    static constexpr int o_colptr_d = o_red + 64;   // (NP + 1) ints
    static constexpr int o_ccur_d = o_colptr_d + ev(NP + 2) / 2;
    static constexpr int o_var = o_ccur_d + NP / 2;
    static constexpr int o_colptr = 2 * o_colptr_d, o_ccur = 2 * o_ccur_d;  // 32-bit words
    int o_lo, o_up, o_rinv, o_wv, o_zs, o_ys, o_rho, o_val;  // doubles: l, u, 1/rho, w, z, y, rho [MP each]; CSR values
    int o_rowptr, o_csc, o_col;                     // 32-bit words
    size_t bytes;
    __host__ __device__ static CsbLayout make(int m, int nnz_cap) {
        CsbLayout L;
        const int MP = (m + 1) & ~1;
        int d = o_var;
        L.o_lo = d; d += MP;
        L.o_up = d; d += MP;
        L.o_rinv = d; d += MP;
        L.o_wv = d; d += MP;
        L.o_zs = d; d += MP;
        L.o_ys = d; d += MP;
        L.o_rho = d; d += MP;
        L.o_val = d; d += nnz_cap;
        int w = 2 * d;
        L.o_csc = w; w += nnz_cap;
        L.o_rowptr = w; w += m + 1;
        L.o_col = w; w += (nnz_cap + 1) / 2;
        L.bytes = (size_t)w * 4 + 16;
        return L;
    }
};

// A 16 x 16 block of W in the accumulator layout (lane l: rows (l >> 4) + 4 e of column l & 15).  On the device the four doubles are
// ONE 256-bit vector value — the register tuple the matrix instruction reads and writes in place; as four separate doubles
// (sqph_acc4) every conditional block update of the set-up left the allocator with an old and a new copy of all 14 blocks: 224 of
// the 256 VGPRs, 56 v_mov_b64 per elimination step and four more for every slot whose condition was false.
#ifdef SQPH_SIM
typedef sqph_acc4 csb_blk;
#else
struct csb_blk {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 v;
};
__device__ __forceinline__ void mfma16(double a, double b, csb_blk &c) { c.v = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c.v, 0, 0, 0); }
#endif
// reduce-scatter steps over the four 16-lane rows of a wavefront: one swap instruction per 32-bit half pairs two values
//   swap_reduce32(a, b): lanes 0-31 get a[l] + a[l + 32], lanes 32-63 get b[l - 32] + b[l]
//   swap_reduce16(a, b): even rows get a[row] + a[row + 1], odd rows get b[row - 1] + b[row]
__device__ __forceinline__ double swap_reduce32(double a, double b) {
#ifdef SQPH_SIM
    const double pa = xchg<32>(a), pb = xchg<32>(b);
    return ((int)threadIdx.x & 32) ? pb + b : a + pa;
#else
    auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
#endif
}
__device__ __forceinline__ double swap_reduce16(double a, double b) {
#ifdef SQPH_SIM
    const double pa = xchg<16>(a), pb = xchg<16>(b);
    return ((int)threadIdx.x & 16) ? pb + b : a + pa;
#else
    auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
#endif
}
// acc + y(lane N .. N + 3 of my 16-lane row) * b[0 .. 3]: four v_fmac_f64 with the row_newbcast control behind ONE pair of wait states
template <int N>
__device__ __forceinline__ double fmac4_bcast16(double acc, double y, const csb_blk &b) {
#ifdef SQPH_SIM
    acc = __builtin_fma(bcast16<N>(y), b.v[0], acc);
    acc = __builtin_fma(bcast16<N + 1>(y), b.v[1], acc);
    acc = __builtin_fma(bcast16<N + 2>(y), b.v[2], acc);
    acc = __builtin_fma(bcast16<N + 3>(y), b.v[3], acc);
    return acc;
#else
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %3 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(y), "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "n"(N), "n"(N + 1), "n"(N + 2), "n"(N + 3));
    return acc;
#endif
}

template <typename TIN, int NB>
struct CsbKernel {
    using T = double;
    using Lay = CsbLayout<NB>;
    using MS = MSetup<2, 16, 8, 7, 7, 4>;  // the in-wavefront elimination of a 16 x 16 diagonal block (admm_wg_msetup.h: diag_block)
    static_assert(!MS::SWZ && MS::BS == Lay::BS, "diag_block works on padded blocks");
    static constexpr int NT = Lay::NT, NW = Lay::NW, NP = Lay::NP, BS = Lay::BS, LDP = Lay::LDP;
    static constexpr int NH = (NB + 1) / 2;  // wavefronts that own blocks
    static constexpr int DW = NW - 1;        // the wavefront that eliminates the diagonal blocks
    static constexpr int KR = SQPH_CSB_KR;   // entries of A per lane and orientation held in registers
    static_assert(NH <= DW && NB >= 1 && NB <= 14 && (KR % 2) == 0, "block rows pair up on seven wavefronts");

    // block rows and register slots of wavefront W: slot K <= I1 holds block (I1, K) of row I1 = NB - 1 - W, slot NB - K >= I1 + 1 block
    // (I0, K) of row I0 = W (the second row is stored from the top down: the block column of a slot is then a compile-time function of
    // the slot — K = s or NB - s — whatever the wavefront, and a step of the set-up reaches "block (row, K)" through a static index)
    template <int W>
    struct Own {
        static constexpr bool any = W < NH;
        static constexpr int I1 = NB - 1 - W, I0 = W;
        static constexpr bool has0 = any && I0 < I1;
    };

    // the same map with the wavefront index a run-time scalar: the set-up is ONE code path for all wavefronts (seven specialised
    // copies of the elimination — 14 eight-register accumulator tuples each — were more than the register allocator could hold
    // together: two of the copies kept their blocks in scratch memory); only the iteration's two dense stages are specialised
    struct Slot {
        int I, K;
        bool valid;
    };
    static __device__ __forceinline__ Slot slot_of(int W, int s) {
        const int I1 = NB - 1 - W;
        Slot d;
        d.valid = W < NH && (s <= I1 || W < I1);
        d.I = s <= I1 ? I1 : W;
        d.K = s <= I1 ? s : NB - s;
        return d;
    }
    // f(integral_constant<int, K>) for K = LO, LO + 1, ... HI while it returns true (an unrolled loop over compile-time indices that can
    // be left early: one taken branch ends it)
    template <int LO, int HI, typename F>
    static __device__ __forceinline__ void static_while(F &&f) {
        if constexpr (LO <= HI) {
            if (f(std::integral_constant<int, LO>{})) static_while<LO + 1, HI>(f);
        }
    }
    // f(integral_constant<int, i>) for a wave-uniform run-time i in [LO, HI]: a binary tree of scalar branches (four taken or untaken
    // branches for 14 values; a chain of "if (s == i)" over the unrolled slots cost a taken branch — an instruction-buffer refill,
    // ~100 cycles — for every slot that did not match)
    template <int LO, int HI, typename F>
    static __device__ __forceinline__ void dispatch(int i, F &&f) {
        if constexpr (LO >= HI) {
            f(std::integral_constant<int, LO>{});
        } else {
            constexpr int MID = (LO + HI) / 2;
            if (i <= MID) dispatch<LO, MID>(i, f);
            else dispatch<MID + 1, HI>(i, f);
        }
    }
    static __device__ __forceinline__ int wave_of(int t) { return MS::wave_of(t); }
    // this lane's index in its wavefront, computed on the spot (two VALU instructions the compiler can neither hoist nor merge): the
    // iteration loop derives its lane coordinates from it instead of keeping the thread index live across the loop — spilled, it
    // was reloaded from scratch memory once per iteration, with a full memory wait behind the dense stages
    static __device__ __forceinline__ int fresh_lane() {
#ifdef SQPH_SIM
        return (int)(threadIdx.x & 63);
#else
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
#endif
    }
    template <typename P>
    static __device__ __forceinline__ const P *uniform_ptr(const P *p) {
#ifdef SQPH_SIM
        return p;
#else
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const P *>(((unsigned long long)hi << 32) | lo);
#endif
    }
    // orders the LDS operations of ONE wavefront for the compiler (the hardware executes a wavefront's LDS operations in order).
    // NOT the release / acquire fence pair of admm_wg_msetup.h: at wavefront scope that lowers to s_waitcnt vmcnt(0) lgkmcnt(0), and
    // behind the elimination's spill stores every fence then waited for scratch memory (4.9 k cycles per elimination step)
    static __device__ __forceinline__ void wave_fence() {
#ifdef SQPH_SIM
        MS::wave_fence();
#else
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#endif
    }
    static __device__ __forceinline__ T opN(const T *b, int kq, int lr, int lq) { return b[lr * 17 + 4 * kq + lq]; }
    static __device__ __forceinline__ T opT(const T *b, int kq, int lr, int lq) { return b[(4 * kq + lq) * 17 + lr]; }
    static __device__ __forceinline__ void ldD(const T *b, int lr, int lq, csb_blk &a) {
#pragma unroll
        for (int e = 0; e < 4; e++) a.v[e] = b[(lq + 4 * e) * 17 + lr];
    }
    static __device__ __forceinline__ void stD(T *b, int lr, int lq, const csb_blk &a) {
#pragma unroll
        for (int e = 0; e < 4; e++) b[(lq + 4 * e) * 17 + lr] = a.v[e];
    }

    // ---------------------------------------------------------------- sparse data: CSR -> LDS, CSC index, lane maps
    static constexpr int MAP_VALID = 1 << 15;
    static __device__ __forceinline__ int lanes_for(int len, int K) {
        const int need = (len + K - 1) / K;
        return need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 4096;
    }
#ifdef SQPH_SIM
    static inline int shfl_up_i(int v, int d) {
        const int lane = (int)(threadIdx.x & 63);
        return (int)(uint32_t)::sqph_sim::wave_exchange((uint64_t)(uint32_t)v, lane >= d ? lane - d : lane);
    }
    static inline int shfl_i(int v, int src) { return (int)(uint32_t)::sqph_sim::wave_exchange((uint64_t)(uint32_t)v, src); }
#else
    static __device__ __forceinline__ int shfl_up_i(int v, int d) { return __shfl_up(v, d); }
    static __device__ __forceinline__ int shfl_i(int v, int src) { return __shfl(v, src); }
#endif

    // CSR of this QP -> LDS, CSC index (row | position-in-CSR) by counting sort, entries of a column ordered by row
    static __device__ __forceinline__ void load_sparse(const CsrArgs<TIN> &ca, int qp, int n, int m, const Lay &L, unsigned char *smem) {
        const int t = threadIdx.x;
        T *lds = reinterpret_cast<T *>(smem);
        int *li = reinterpret_cast<int *>(smem);
        int *rowptr = li + L.o_rowptr, *colptr = li + L.o_colptr, *ccur = li + L.o_ccur;
        unsigned *csc = reinterpret_cast<unsigned *>(li + L.o_csc);
        unsigned short *col = reinterpret_cast<unsigned short *>(li + L.o_col);
        T *val = lds + L.o_val;
        const int *grp = ca.rowptr + (long long)qp * ca.s_rowptr;
        const int *gci = ca.colind + (long long)qp * ca.s_colind;
        const TIN *gv = ca.val + (long long)qp * ca.s_val;
        for (int i = t; i <= m; i += NT) rowptr[i] = grp[i];
        for (int j = t; j <= NP; j += NT) colptr[j] = 0;
        __syncthreads();
        const int nnz = rowptr[m];
        // four entries per lane and round, all their loads requested before the first is used: the arrays are read once, cold from HBM,
        // and entry by entry this loop was nine exposed memory round trips (load_sparse 38 k of the set-up's 420 k cycles)
        for (int e0 = t; e0 < nnz; e0 += 4 * NT) {
            int jj[4];
            T vv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = e0 + k * NT;
                jj[k] = e < nnz ? gci[e] : 0;
                vv[k] = e < nnz ? (T)gv[e] : T(0);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = e0 + k * NT;
                if (e < nnz) {
                    col[e] = (unsigned short)jj[k];
                    val[e] = vv[k];
                    lds_atomic_inc(&colptr[jj[k] + 1]);  // integer counts: order-independent
                }
            }
        }
        __syncthreads();
        if (t < 64) {  // exclusive scan of the column counts: wave 0, 4 per lane
            int v[4], s = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = 4 * t + k;
                v[k] = (j < NP) ? colptr[j + 1] : 0;
                s += v[k];
            }
            int incl = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = shfl_up_i(incl, d);
                if ((t & 63) >= d) incl += o;
            }
            int run = incl - s;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = 4 * t + k;
                if (j < NP) {
                    colptr[j + 1] = run + v[k];
                    ccur[j] = run;
                }
                run += v[k];
            }
        }
        __syncthreads();
        // fill in the order the atomics are served, then make canonical by ranking (the keys of a column are distinct: the final slot
        // of an entry is the number of smaller keys in its column — one lane per entry)
        unsigned *tmp = reinterpret_cast<unsigned *>(lds);  // the work area is idle
        const bool ranked = (size_t)nnz * sizeof(unsigned) <= (size_t)Lay::WORK * sizeof(T);
        unsigned *fill = ranked ? tmp : csc;
        for (int i = t; i < m; i += NT) {
            for (int e = rowptr[i]; e < rowptr[i + 1]; e++) {
                const int j = col[e];
                const int slot = lds_atomic_inc(&ccur[j]);
                fill[slot] = ((unsigned)i << 16) | (unsigned)e;
            }
        }
        __syncthreads();
        if (ranked) {
            for (int s0 = t; s0 < nnz; s0 += NT) {
                const unsigned key = tmp[s0];
                const int j = col[key & 0xffffu];
                const int e0 = colptr[j], e1 = colptr[j + 1];
                int rank = 0;
                for (int f = e0; f < e1; f++) rank += tmp[f] < key ? 1 : 0;
                csc[e0 + rank] = key;
            }
        } else {
            for (int j = t; j < n; j += NT) {
                const int e0 = colptr[j], e1 = colptr[j + 1];
                for (int e = e0 + 1; e < e1; e++) {
                    const unsigned key = csc[e];
                    int f = e - 1;
                    while (f >= e0 && csc[f] > key) {
                        csc[f + 1] = csc[f];
                        f--;
                    }
                    csc[f + 1] = key;
                }
            }
        }
        __syncthreads();
    }

    // A row (column) of A gets 1, 2, 4 or 8 lanes by its length such that no lane carries more than K entries, K the smallest bound
    // for which everything fits the NT lanes; groups are laid out by size (aligned for the xor butterfly that adds their partial sums).
    // Map word (16 bits): element in bits 0-8, part in 9-11, log2(lanes of the element) in 12-13, valid in 15.
    // BOTH maps in one pass — rows (ptr0 / cnt0 -> map[0 .. NT), K0) by wavefront 0, columns (ptr1 / cnt1 -> map[NT .. 2 NT), K1) by
    // wavefront 1: the scans and the dealing are single-wavefront work, and built one after the other the two maps were 43 k of the
    // set-up's 420 k cycles with seven wavefronts idle (hist: 2 x (HL + 32) ints).
    static constexpr int MAP_HL = 544, MAP_HS = MAP_HL + 32;  // lengths are <= 512
    static __device__ __forceinline__ void build_lane_maps(const int *ptr0, int cnt0, const int *ptr1, int cnt1, unsigned short *map, int *hist,
                                                            int &K0, int &K1) {
        const int t = threadIdx.x, w = wave_of(t), lane = t & 63;
        constexpr int NC = 15, HL = MAP_HL, HS = MAP_HS;
        constexpr int KC[NC] = {4, 5, 6, 7, 8, 10, 12, 14, 16, 24, 32, 48, 64, 128, 256};
        for (int e = t; e < 2 * HS; e += NT) hist[e] = 0;
        map[t] = 0;
        map[NT + t] = 0;
        __syncthreads();
        if (t < cnt0) lds_atomic_inc(&hist[ptr0[t + 1] - ptr0[t]]);
        if (t < cnt1) lds_atomic_inc(&hist[HS + ptr1[t + 1] - ptr1[t]]);
        __syncthreads();
        if (w < 2) {
            const int *h = hist + w * HS;
            int *need = hist + w * HS + HL;
            int loc[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) loc[c] = 0;
            for (int len = lane; len < HL; len += 64) {
                const int hv = h[len];
                if (hv) {
#pragma unroll
                    for (int c = 0; c < NC; c++) loc[c] += hv * lanes_for(len, KC[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                int v = loc[c];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = shfl_up_i(v, d);
                    if (lane >= d) v += o;
                }
                if (lane == 63) need[c] = v;
            }
        }
        __syncthreads();
        K0 = K1 = KC[NC - 1];
#pragma unroll
        for (int c = NC - 1; c >= 0; c--) {
            if (hist[HL + c] <= NT) K0 = KC[c];
            if (hist[HS + HL + c] <= NT) K1 = KC[c];
        }
        if (w < 2) {  // one wavefront per map: eight consecutive elements per lane, class offsets by a scan over the 64 lanes
            const int *ptr = w ? ptr1 : ptr0;
            const int count = w ? cnt1 : cnt0, K = w ? K1 : K0;
            unsigned short *mapw = map + w * NT;
            int cnt[4] = {0, 0, 0, 0}, cls[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = 8 * lane + k;
                cls[k] = -1;
                if (i < count) {
                    const int p = lanes_for(ptr[i + 1] - ptr[i], K);
                    const int c = p == 1 ? 0 : p == 2 ? 1 : p == 4 ? 2 : 3;
                    cls[k] = c;
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) cnt[cc] += (cc == c) ? 1 : 0;
                }
            }
            int incl[4], tot[4];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                int v = cnt[cc];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = shfl_up_i(v, d);
                    if (lane >= d) v += o;
                }
                incl[cc] = v;
                tot[cc] = shfl_i(v, 63);
            }
            int base[4];
            base[3] = 0;
            base[2] = 8 * tot[3];
            base[1] = base[2] + 4 * tot[2];
            base[0] = base[1] + 2 * tot[1];
            int run[4];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) run[cc] = incl[cc] - cnt[cc];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int c = cls[k];
                if (c >= 0) {
                    int b0 = 0, rn = 0;
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) {
                        if (cc == c) {
                            b0 = base[cc];
                            rn = run[cc];
                            run[cc] += 1;
                        }
                    }
                    const int p = 1 << c, lane0 = b0 + rn * p;
                    for (int part = 0; part < p; part++)
                        mapw[lane0 + part] = (unsigned short)(MAP_VALID | (c << 12) | (part << 9) | (8 * lane + k));
                }
            }
        }
        __syncthreads();
    }
    static __device__ __forceinline__ T group_sum(T s, int lg) {
        T o = xchg<1>(s);
        s += lg >= 1 ? o : T(0);
        o = xchg<2>(s);
        s += lg >= 2 ? o : T(0);
        o = xchg<4>(s);
        s += lg >= 3 ? o : T(0);
        return s;
    }
    // sparse products from LDS (a QP whose map needs more than KR entries per lane): as admm_csr_kernel.h
    static __device__ __forceinline__ T csr_row_dot_lds(const int *rowptr, const unsigned short *col, const T *val, const T *v, int mp) {
        T a0 = 0, a1 = 0;
        const int lg = (mp >> 12) & 3;
        if (mp & MAP_VALID) {
            const int i = mp & 511, p = 1 << lg;
            const int e1 = rowptr[i + 1];
            int e = rowptr[i] + ((mp >> 9) & 7);
            for (; e + p < e1; e += 2 * p) {
                a0 = wg_fma(val[e], v[col[e]], a0);
                a1 = wg_fma(val[e + p], v[col[e + p]], a1);
            }
            if (e < e1) a0 = wg_fma(val[e], v[col[e]], a0);
        }
        return group_sum(a0 + a1, lg);
    }
    static __device__ __forceinline__ T csc_col_dot_lds(const int *colptr, const unsigned *csc, const T *val, const T *v, int mp) {
        T a0 = 0, a1 = 0;
        const int lg = (mp >> 12) & 3;
        if (mp & MAP_VALID) {
            const int j = mp & 511, p = 1 << lg;
            const int e1 = colptr[j + 1];
            int e = colptr[j] + ((mp >> 9) & 7);
            for (; e + p < e1; e += 2 * p) {
                const unsigned p0 = csc[e], p1 = csc[e + p];
                a0 = wg_fma(val[p0 & 0xffffu], v[p0 >> 16], a0);
                a1 = wg_fma(val[p1 & 0xffffu], v[p1 >> 16], a1);
            }
            if (e < e1) {
                const unsigned p0 = csc[e];
                a0 = wg_fma(val[p0 & 0xffffu], v[p0 >> 16], a0);
            }
        }
        return group_sum(a0 + a1, lg);
    }
    // the same with the lane's entries in registers: sv[k] the values, si[k / 2] two packed 16-bit BYTE offsets into v (slots beyond
    // the lane's share hold value 0 and offset 0).  Same summation order as the LDS form (even / odd entries, then the group).
    // The packed words pass through an opaque asm on every call: loop-invariant as they are, the compiler would otherwise keep the
    // 2 KR unpacked gather addresses in registers across the loop (28 VGPRs that the iterates and part of the slices paid for
    // with scratch memory).
    static __device__ __forceinline__ T gather_at(const T *v, int off) {
        return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(v) + off);
    }
    static __device__ __forceinline__ T reg_dot(const T (&sv)[KR], const int (&si)[KR / 2], const T *v, int mp) {
        T g[KR];
#pragma unroll
        for (int k = 0; k < KR; k += 2) {
            int w = si[k >> 1];
            SQPH_OPAQUE_V(w);
            g[k] = gather_at(v, w & 0xffff);
            g[k + 1] = gather_at(v, (int)((unsigned)w >> 16));
        }
        T a0 = 0, a1 = 0;
#pragma unroll
        for (int k = 0; k < KR; k += 2) {
            a0 = wg_fma(sv[k], g[k], a0);
            a1 = wg_fma(sv[k + 1], g[k + 1], a1);
        }
        return group_sum(a0 + a1, (mp >> 12) & 3);
    }
    // ---------------------------------------------------------------- which register slot each entry of a lane's slice goes to
    // The two sparse products gather a vector element per entry: 2 x 14 ds_read_b64 per lane and iteration whose addresses are the
    // matrix's column / row indices, i.e. random over the 32 bank pairs (2.1 - 2.3 LDS passes per half-wavefront and slot instead of
    // 1; the slots a lane does not fill gathered element 0, one more address on bank 0).  A slot's position in a lane does not matter
    // to the product, so the entries are PLACED: a lane claims, for each entry, a slot in which no other lane of its half-wavefront
    // reads the entry's bank.  Rounds in lockstep, wavefront-local: every lane with an entry left proposes one (slot, bank) cell by
    // ds_min_u32 of its lane number, the lowest proposer wins the cell (recorded in the bank's slot mask by ds_or_b32), the others
    // propose again.  Minimum and OR commute and a wavefront's LDS operations execute in order: the outcome does not depend on the
    // order the LDS serves the atomics of one instruction in (repeated solves stay bit-identical).  Unfilled slots gather an element
    // whose bank no lane of the half-wavefront reads in that slot (one address: a broadcast).  tools/xp/gather_model.py: 470 + 510 ->
    // 265 + 275 passes per iteration at config 5 (floor 224 + 224), 16 - 18 rounds.
    // Code words: one byte per slot — the entry's ordinal in the lane's slice, or 0x80 | the element an unfilled slot gathers.
    struct SlotCode {
        unsigned long long lo, hi;  // slots 0 - 7, 8 - 13
    };
    static constexpr int PLACE_WORDS = 32 + KR * 32 + 16;  // per half-wavefront: slot mask per bank | lowest proposer per cell | pad element per slot
    static_assert(16 * PLACE_WORDS * 4 <= Lay::WORK * 8, "the placement tables fit the work area");
    static __device__ __forceinline__ void lds_min_u(unsigned *p, unsigned v) {
#ifdef SQPH_SIM
        *p = *p < v ? *p : v;
#else
        __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    }
    static __device__ __forceinline__ void lds_or_u(unsigned *p, unsigned v) {
#ifdef SQPH_SIM
        *p |= v;
#else
        __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    }
    static __device__ __forceinline__ bool wave_any(bool p) {
#ifdef SQPH_SIM
        int v = p ? 1 : 0;
        const int lane = (int)(threadIdx.x & 63);
        for (int d = 1; d < 64; d <<= 1) v |= shfl_i(v, lane ^ d);
        return v != 0;
#else
        return __builtin_amdgcn_ballot_w64(p) != 0;
#endif
    }
    static __device__ __forceinline__ void code_set(SlotCode &c, int s, unsigned v) {
        const int sh = 8 * (s & 7);
        const unsigned long long m = ~(0xffull << sh), w = (unsigned long long)v << sh;
        if (s < 8) c.lo = (c.lo & m) | w;
        else c.hi = (c.hi & m) | w;
    }
    static __device__ __forceinline__ unsigned code_get(const SlotCode &c, int s) {
        return (unsigned)((s < 8 ? c.lo : c.hi) >> (8 * (s & 7))) & 0xffu;
    }
    // ROWS: the CSR slices (gathering x~, `vlen` = NP elements are defined); otherwise the CSC slices (gathering w, vlen = m)
    template <bool ROWS>
    // place = false (block-uniform): no rounds, the entries keep their storage order (what the loop below leaves is dealt in order)
    static __device__ __forceinline__ SlotCode place_slots(const int *ptr, const unsigned short *col, const unsigned *csc, int mp, int vlen,
                                                          unsigned *area, int t, bool place) {
        constexpr unsigned KMASK = (1u << KR) - 1u;
        const int hw = t >> 5, lh = t & 31;
        unsigned *taken = area + hw * PLACE_WORDS, *cell = taken + 32, *pad = cell + KR * 32;
        for (int i = lh; i < PLACE_WORDS; i += 32) taken[i] = (i >= 32 && i < 32 + KR * 32) ? 0xffffffffu : 0u;
        wave_fence();
        const bool own = (mp & MAP_VALID) != 0;
        const int i = mp & 511, p = 1 << ((mp >> 12) & 3);
        const int e1 = own ? ptr[i + 1] : 0;
        const int e0 = own ? ptr[i] + ((mp >> 9) & 7) : 0;
        const int cnt = e1 > e0 ? (e1 - e0 + p - 1) / p : 0;  // <= KR: the orientation is register resident
        SlotCode c{0x8080808080808080ull, 0x8080808080808080ull};
        unsigned used = 0;
        int e = 0;
#pragma unroll 1
        for (int round = 0; round < (place ? 4 * KR : 0); round++) {
            const bool active = e < cnt;
            if (!wave_any(active)) break;
            int s = -1, b = 0;
            if (active) {
                const int pos = e0 + e * p;
                b = (ROWS ? (int)col[pos] : (int)(csc[pos] >> 16)) & 31;
                const unsigned avail = ~(taken[b] | used) & KMASK;
                if (avail) {
                    const int start = (e + lh) % KR;  // first choices spread over the slots
                    const unsigned from = (avail >> start) << start;
                    s = __builtin_ffs((int)(from ? from : avail)) - 1;
                    lds_min_u(&cell[s * 32 + b], (unsigned)lh);
                } else {  // the bank is read in every slot this lane has left: any of them
                    const int f = __builtin_ffs((int)(~used & KMASK)) - 1;
                    code_set(c, f, (unsigned)e);
                    used |= 1u << f;
                    e++;
                }
            }
            wave_fence();
            if (s >= 0 && cell[s * 32 + b] == (unsigned)lh) {
                lds_or_u(&taken[b], 1u << s);
                code_set(c, s, (unsigned)e);
                used |= 1u << s;
                e++;
            }
            wave_fence();
        }
        while (e < cnt) {  // (what the rounds left)
            const int f = __builtin_ffs((int)(~used & KMASK)) - 1;
            code_set(c, f, (unsigned)e);
            used |= 1u << f;
            e++;
        }
        if (lh < KR) {  // an element (< vlen, < 32) whose bank nobody reads in slot lh
            unsigned freeb = 0;
#pragma unroll
            for (int b = 0; b < 32; b++) freeb |= ((~taken[b] >> lh) & 1u) << b;
            if (vlen < 32) freeb &= (1u << vlen) - 1u;
            pad[lh] = freeb ? (unsigned)(__builtin_ffs((int)freeb) - 1) : 0u;
        }
        wave_fence();
#pragma unroll
        for (int k = 0; k < KR; k++)
            if (!((used >> k) & 1u)) code_set(c, k, 0x80u | pad[k]);
        wave_fence();
        return c;
    }
    static __device__ __forceinline__ void load_row_regs(const int *rowptr, const unsigned short *col, const T *val, int mp, const SlotCode &sc,
                                                         T (&sv)[KR], int (&si)[KR / 2]) {
        const bool own = (mp & MAP_VALID) != 0;
        const int i = mp & 511, p = 1 << ((mp >> 12) & 3);
        const int e0 = own ? rowptr[i] + ((mp >> 9) & 7) : 0;
#pragma unroll
        for (int k = 0; k < KR / 2; k++) si[k] = 0;
#pragma unroll
        for (int k = 0; k < KR; k++) {
            const unsigned code = code_get(sc, k);
            const bool in = !(code & 0x80u);
            const int e = e0 + (int)(code & 15u) * p;
            sv[k] = in ? val[e] : T(0);
            si[k >> 1] |= (in ? 8 * (int)col[e] : 8 * (int)(code & 31u)) << (16 * (k & 1));
        }
    }
    static __device__ __forceinline__ void load_col_regs(const int *colptr, const unsigned *csc, const T *val, int mp, const SlotCode &sc,
                                                         T (&sv)[KR], int (&si)[KR / 2]) {
        const bool own = (mp & MAP_VALID) != 0;
        const int j = mp & 511, p = 1 << ((mp >> 12) & 3);
        const int e0 = own ? colptr[j] + ((mp >> 9) & 7) : 0;
#pragma unroll
        for (int k = 0; k < KR / 2; k++) si[k] = 0;
#pragma unroll
        for (int k = 0; k < KR; k++) {
            const unsigned code = code_get(sc, k);
            const bool in = !(code & 0x80u);
            const unsigned pk = in ? csc[e0 + (int)(code & 15u) * p] : 0u;
            sv[k] = in ? val[pk & 0xffffu] : T(0);
            si[k >> 1] |= (in ? (int)(8 * (pk >> 16)) : 8 * (int)(code & 31u)) << (16 * (k & 1));
        }
    }

    // LDS add without a returned value (ds_add_f64)
    static __device__ __forceinline__ void lds_add_f64(T *p, T v) {
#ifdef SQPH_SIM
        *p += v;
#else
        __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    }
    // ---------------------------------------------------------------- S = P_sym + sigma I + A' diag(rho) A -> blocks
    // One 32-column panel at a time in LDS: the 16-lane group g owns column j = 32 p + g — for every CSC entry (i, pos) of that
    // column, in row order, its lanes add rho_i A_ij * (row i of A) into the panel column (distinct k per lane: rows are
    // duplicate-free; successive entries are ordered by the wavefront's program order).  Only k >= j is formed (+ the lower triangle
    // of P: what reaches the reference's factor, Eigen::LDLT<.,Lower>, qp.hpp:129); diagonal blocks are mirrored on pick-up.
    static __device__ __forceinline__ void pick_one(csb_blk &Bs, int I, int K, int n, const T *Sp, int lr, int lq) {
        const int kk = K & 1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int il = lq + 4 * e, i = 16 * I + il, j = 16 * K + lr;
            const int hi = il > lr ? il : lr, lo = il > lr ? lr : il;
            // a diagonal block is mirrored: (lo, hi) of the lower triangle; elsewhere column lr, row il
            const int a = I == K ? (16 * kk + lo) * LDP + 16 * I + hi : (16 * kk + lr) * LDP + i;
            const T v = Sp[a];
            Bs.v[e] = (i < n && j < n) ? v : (i == j ? T(1) : T(0));  // the padding is an identity block
        }
    }
    // the blocks of panel p (block columns 2 p and 2 p + 1) of my rows, out of the panel in LDS
    static __device__ __forceinline__ void pick_up(int W, int p, int n, const T *Sp, int lr, int lq, csb_blk (&B)[NB + 1]) {
        if (W >= NH) return;
        const int I1 = NB - 1 - W, I0 = W;
        const bool has0 = I0 < I1;
        dispatch<0, (NB - 1) / 2>(p, [&](auto pc) __attribute__((always_inline)) {
            constexpr int K0 = 2 * decltype(pc)::value, K1 = K0 + 1;
            if (K0 <= I1) pick_one(B[K0], I1, K0, n, Sp, lr, lq);
            if constexpr (K1 < NB) {
                if (K1 <= I1) pick_one(B[K1], I1, K1, n, Sp, lr, lq);
            }
            if (has0) {
                if (K0 <= I0) pick_one(B[NB - K0], I0, K0, n, Sp, lr, lq);
                if constexpr (K1 < NB) {
                    if (K1 <= I0) pick_one(B[NB - K1], I0, K1, n, Sp, lr, lq);
                }
            }
        });
    }
    // any lane of this lane's aligned group of 16 (the groups of a wavefront may have diverged)
    static __device__ __forceinline__ bool group16_any(bool p) {
#ifdef SQPH_SIM
        uint64_t v = p ? 1 : 0;
        for (int d = 1; d < 16; d <<= 1) v |= ::sqph_sim::group16_exchange(v, (int)(threadIdx.x & 15) ^ d);
        return v != 0;
#else
        return ((__builtin_amdgcn_ballot_w64(p) >> (threadIdx.x & 48)) & 0xffffull) != 0;
#endif
    }
    // SP: P in compressed-column form (sqph_csc_P: the symmetric matrix, rows ascending inside a column) — gP is then its value
    // array, pcol / prow its column pointers and row indices, read from global memory in place of the dense columns.  The sums are
    // formed in the order of the dense path (an absent entry is a dense zero, whose addition changes nothing): bit-identical S.
    template <bool SP = false>
    static __device__ __forceinline__ void form_S(const TIN *__restrict__ gP, const int *__restrict__ pcol, const int *__restrict__ prow, int n,
                                                  T sigma, const Lay &L, unsigned char *smem, int t_, int wave_, csb_blk (&B)[NB + 1] SQPH_FTICK_ARGS) {
        int t = t_;
        const int c16 = t & 15, g = t >> 4;
        T *lds = reinterpret_cast<T *>(smem);
        const int *li = reinterpret_cast<const int *>(smem);
        const int *rowptr = li + L.o_rowptr, *colptr = li + L.o_colptr;
        const unsigned *csc = reinterpret_cast<const unsigned *>(li + L.o_csc);
        const unsigned short *col = reinterpret_cast<const unsigned short *>(li + L.o_col);
        const T *val = lds + L.o_val, *rho = lds + L.o_rho;
        T *Sp = lds;
        // the group's share of a column of P (lower triangle: rows j .. n - 1, 16 lanes x 14 = 224 rows), requested ONE PANEL AHEAD:
        // P is read once per factorisation, cold from HBM — two exposed round trips per panel were a quarter of this phase
        auto load_P = [&](int pp, T (&pv)[14]) {
            const int j = 32 * pp + g;
            const TIN *pc = gP + (long)(j < n ? j : 0) * n;
#pragma unroll
            for (int q = 0; q < 14; q++) {
                const int k = j + c16 + 16 * q;
                pv[q] = (j < n && k < n) ? (T)pc[k] : T(0);
            }
        };
        T pv[14];
        if constexpr (!SP) load_P(0, pv);
#pragma unroll 1
        for (int p = 0; p < (NB + 1) / 2; p++) {
            __syncthreads();
            SQPH_STICK1(6)
            for (int e = t; e < 32 * LDP; e += NT) Sp[e] = 0;
            __syncthreads();
            SQPH_STICK1(3)

            const int j = 32 * p + g;
            if (j < n) {
                constexpr int EB = SP ? SQPH_CSB_EB_SP : SQPH_CSB_EB;
                // EB CSC entries of the column at a time: their index / coefficient / row loads are all in flight together, the
                // products go to the panel by ds_add_f64 (no returned value, so no entry waits for the LDS round trip of the one
                // before it; additions to one address are issued by this wavefront in program order: deterministic sums).  As a
                // read-modify-write chain, one entry at a time, this loop was ~700 cycles per entry (106 k of the phase's 211 k)
                const int e1 = colptr[j + 1];
                for (int e = colptr[j]; e < e1; e += EB) {
                    T cf[EB], v[EB];
                    int f0[EB], f1[EB], k[EB];
                    bool in[EB];
#pragma unroll
                    for (int bb = 0; bb < EB; bb++) {
                        const bool ok = e + bb < e1;
                        const unsigned pk = csc[ok ? e + bb : e];
                        const int i = (int)(pk >> 16);
                        cf[bb] = rho[i] * val[pk & 0xffffu];
                        f0[bb] = rowptr[i] + c16;
                        f1[bb] = ok ? rowptr[i + 1] : 0;
                    }
#pragma unroll
                    for (int bb = 0; bb < EB; bb++) {
                        in[bb] = f0[bb] < f1[bb];
                        const int f = in[bb] ? f0[bb] : 0;
                        k[bb] = col[f];
                        v[bb] = val[f];
                    }
                    // (entry by entry — a row's first 16 non-zeros, then the rest of a longer row — so that the order of the additions to
                    // one address does not depend on EB: the dense-P and the sparse-P instantiations batch differently and must agree bit for bit)
#pragma unroll
                    for (int bb = 0; bb < EB; bb++) {
                        if (in[bb] && k[bb] >= j) lds_add_f64(&Sp[g * LDP + k[bb]], cf[bb] * v[bb]);
                        for (int f = f0[bb] + 16; f < f1[bb]; f += 16) {  // (rows of more than 16 entries)
                            const int kk = col[f];
                            if (kk >= j) lds_add_f64(&Sp[g * LDP + kk], cf[bb] * val[f]);
                        }
                    }
                }
                SQPH_STICK1(4)
                // + lower triangle of P + sigma I
#ifdef SQPH_SIM
                ::sqph_sim::group16_sync();  // (the emulator runs a lane up to its next rendezvous: the hardware's program order of the ds_add_f64 above and below)
#endif
                if constexpr (SP) {
                    const int pe1 = pcol[j + 1];
                    bool diag = false;
                    for (int e = pcol[j] + c16; e < pe1; e += 16) {
                        const int kk = prow[e];
                        T pe = (T)gP[e];
                        if (kk == j) {
                            pe += sigma;
                            diag = true;
                        }
                        if (kk >= j && kk < n) lds_add_f64(&Sp[g * LDP + kk], pe);
                    }
                    if (!group16_any(diag) && c16 == 0) lds_add_f64(&Sp[g * LDP + j], sigma);  // (no stored diagonal entry: 0 + sigma)
                } else {
#pragma unroll
                for (int q = 0; q < 14; q++) {
                    const int kk = j + c16 + 16 * q;
                    if (kk < n) lds_add_f64(&Sp[g * LDP + kk], pv[q] + (kk == j ? sigma : T(0)));
                }
                }
            }
            // the next panel's share of P, requested as soon as this panel's has been added: in flight across the barrier, the pick-up,
            // the clearing of the panel and the next accumulation loop, in the ONE array (a second array filled at the top of the panel
            // measured the same time, 21.55 ms, with 38 more VGPRs spilled: 8.0 against 6.9 GB of HBM traffic per launch)
            if constexpr (!SP) load_P(p + 1 < (NB + 1) / 2 ? p + 1 : p, pv);
            SQPH_STICK1(5)
            __syncthreads();
            SQPH_STICK1(6)
            {   // (the lane's coordinates derived again per panel: hoisted out of this loop, the pick-up's per-slot addresses, bounds
                // tests and identity-padding values of all 14 slots were spilled and reloaded from scratch memory)
                int tp = t_, wp = wave_;
                SQPH_OPAQUE_V(tp);
                SQPH_OPAQUE_S(wp);
                pick_up(wp, p, n, Sp, tp & 15, (tp >> 4) & 3, B);
                SQPH_STICK1(8)
            }
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- Jacobi scaling + blocked elimination on the matrix pipe
    // S~ = D_J S D_J = L L' in 16 x 16 blocks, W = L^-1 D_J (admm_wg_msetup.h phase 3, with the blocks in registers).  Step J:
    //   A: panel L_IJ = M_IJ Winv_JJ' of my rows I > J (through XS[I]: the block is the A-operand of its own product); the owner of
    //      row J + 1 updates M_J+1,J+1 at once and hands it to the eliminating wavefront; the owner of row J finishes it:
    //      W_JK = Winv_JJ E_JK (K < J, published in XS[K]), W_JJ from the eliminating wavefront
    //   B: E_IK -= L_IJ W_JK (K < J), E_IJ = -L_IJ Winv_JJ D_J, M_IK -= L_IJ L_KJ' (J < K <= I) on my rows I > J, both operands
    //      from LDS, the accumulators in place; meanwhile the last wavefront eliminates M_J+1,J+1 (look-ahead)
    // E is the unit-block-lower inverse in the making, stored in place of the eliminated blocks.
    static __device__ __forceinline__ void diag_to_sj(int W, const csb_blk (&B)[NB + 1], T *sj, int lr, int lq) {
#pragma unroll
        for (int s = 0; s <= NB; s++) {
            const Slot d = slot_of(W, s);
            if (d.valid && d.I == d.K) {
                const T a0 = B[s].v[0], a1 = B[s].v[1], a2 = B[s].v[2], a3 = B[s].v[3];
                const T d01 = (lr & 4) ? a1 : a0, d23 = (lr & 4) ? a3 : a2;
                const T dv = (lr & 8) ? d23 : d01;
                if ((lr & 3) == lq) sj[16 * d.I + lr] = dv;
            }
        }
    }
    static __device__ __forceinline__ void scale_blocks(int W, csb_blk (&B)[NB + 1], const T *sj, T *wk, int lr, int lq) {
#pragma unroll
        for (int s = 0; s <= NB; s++) {
            const Slot d = slot_of(W, s);
            if (d.valid) {
                const T dc = sj[16 * d.K + lr];
#pragma unroll
                for (int e = 0; e < 4; e++) B[s].v[e] = B[s].v[e] * sj[16 * d.I + lq + 4 * e] * dc;
                if (d.I == 0) stD(wk + Lay::O_MD, lr, lq, B[s]);  // M_00 to the eliminating wavefront
            }
        }
    }
    // L_IJ = M_IJ Winv_JJ' of my row I > J, through XS[I] (the block is the A-operand of its own product)
    static __device__ __forceinline__ void panel_block(const csb_blk &Bs, int I, T *XS, const T *Wd, int lr, int lq) {
        T *x = XS + I * BS;
        stD(x, lr, lq, Bs);
        wave_fence();
        csb_blk a = {{0, 0, 0, 0}};
        T av[4], bv[4];
#pragma unroll
        for (int kq = 0; kq < 4; kq++) {
            av[kq] = opN(x, kq, lr, lq);
            bv[kq] = opN(Wd, kq, lr, lq);
        }
#pragma unroll
        for (int kq = 0; kq < 4; kq++) mfma16(av[kq], bv[kq], a);
        wave_fence();
        stD(x, lr, lq, a);
    }
    // W_JK = Winv_JJ E_JK of the row being finished (K < J), in place (the D layout of a block is its B-operand layout)
    static __device__ __forceinline__ void finish_block(csb_blk &Bs, const T *Wd, int lr, int lq) {
        csb_blk a = {{0, 0, 0, 0}};
        T av[4];
#pragma unroll
        for (int kq = 0; kq < 4; kq++) av[kq] = opN(Wd, kq, lr, lq);
#pragma unroll
        for (int kq = 0; kq < 4; kq++) mfma16(av[kq], Bs.v[kq], a);
        Bs = a;
    }
    // look-ahead: the next diagonal block, updated now (with my own L_J+1,J) and handed to the eliminating wavefront
    static __device__ __forceinline__ void lookahead_block(csb_blk &Bs, int J, T *wk, int lr, int lq) {
        const T *x = wk + Lay::O_XS + (J + 1) * BS;
        T av[4];
#pragma unroll
        for (int kq = 0; kq < 4; kq++) av[kq] = opN(x, kq, lr, lq);
#pragma unroll
        for (int kq = 0; kq < 4; kq++) mfma16(-av[kq], av[kq], Bs);
        stD(wk + Lay::O_MD + ((J + 1) & 1) * BS, lr, lq, Bs);
    }
    static __device__ __forceinline__ void elim_A(int W, int J, csb_blk (&B)[NB + 1], T *wk, int lr, int lq) {
        if (W >= NH) return;  // (the eliminating wavefront owns no block)
        T *XS = wk + Lay::O_XS;
        const T *Wd = wk + Lay::O_TB + (J & 1) * BS;
        const int I1 = NB - 1 - W, I0 = W;
        const bool has0 = I0 < I1;
        constexpr int KH = (NB - 1) / 2;  // block columns of a second row: K <= I0 <= KH
        // my panel blocks (I, J), I > J: slot J of row I1, slot NB - J of row I0
        if (I1 > J) dispatch<0, NB - 1>(J, [&](auto kc) __attribute__((always_inline)) { panel_block(B[decltype(kc)::value], I1, XS, Wd, lr, lq); });
        if (has0 && I0 > J) dispatch<0, KH>(J, [&](auto kc) __attribute__((always_inline)) { panel_block(B[NB - decltype(kc)::value], I0, XS, Wd, lr, lq); });
        // the row J itself, if it is mine: its blocks E_JK (K < J) are published AS THEY ARE — the other rows multiply by
        // (L_IJ Winv_JJ) E_JK instead of L_IJ (Winv_JJ E_JK), so that the J products W_JK = Winv_JJ E_JK of this row are not on every
        // wavefront's way to the trailing update: they follow in elim_B — and W_JJ = Winv_JJ D_J comes from the eliminating wavefront
        if (I1 == J) {
            static_while<0, NB - 1>([&](auto kc) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value;
                if (K >= J) return false;
                stD(XS + K * BS, lr, lq, B[K]);
                return true;
            });
            dispatch<0, NB - 1>(J, [&](auto kc) __attribute__((always_inline)) { ldD(wk + Lay::O_MD + (J & 1) * BS, lr, lq, B[decltype(kc)::value]); });
        } else if (has0 && I0 == J) {
            static_while<0, KH>([&](auto kc) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value;
                if (K >= J) return false;
                stD(XS + K * BS, lr, lq, B[NB - K]);
                return true;
            });
            dispatch<0, KH>(J, [&](auto kc) __attribute__((always_inline)) { ldD(wk + Lay::O_MD + (J & 1) * BS, lr, lq, B[NB - decltype(kc)::value]); });
        }
        wave_fence();
        if (I1 == J + 1) dispatch<0, NB - 1>(J + 1, [&](auto kc) __attribute__((always_inline)) { lookahead_block(B[decltype(kc)::value], J, wk, lr, lq); });
        else if (has0 && I0 == J + 1) dispatch<0, KH>(J + 1, [&](auto kc) __attribute__((always_inline)) { lookahead_block(B[NB - decltype(kc)::value], J, wk, lr, lq); });
    }
    // Trailing update of a row I > J.  First its block in column J: a = -L_IJ Winv_JJ, E_IJ = a D_J; `a` also goes through the wavefront's
    // private LDS block from the D layout to the A-operand layout (avp).  Then block by block, K a compile-time constant (the operand's
    // block is an immediate offset; which operands a block takes are selects, no branch between the two common cases):
    //   K < J:  E_IK += (-L_IJ Winv_JJ) E_JK     (A-operand avp, E_JK read transposed from XS[K] where row J's owner left it)
    //   K > J:  M_IK += (-L_IJ) L_KJ'            (A-operand av = -L_IJ)
    static __device__ __forceinline__ void pivot_block(csb_blk &Bs, const T (&av)[4], T (&avp)[4], const T *Wd, T *xp, T dc, int lr, int lq) {
        csb_blk a = {{0, 0, 0, 0}};
        T bv[4];
#pragma unroll
        for (int kq = 0; kq < 4; kq++) bv[kq] = opT(Wd, kq, lr, lq);
#pragma unroll
        for (int kq = 0; kq < 4; kq++) mfma16(av[kq], bv[kq], a);
#pragma unroll
        for (int e = 0; e < 4; e++) Bs.v[e] = a.v[e] * dc;
        wave_fence();
        stD(xp, lr, lq, a);
        wave_fence();
#pragma unroll
        for (int kq = 0; kq < 4; kq++) avp[kq] = opN(xp, kq, lr, lq);
    }
    template <int K>
    static __device__ __forceinline__ void update_block(csb_blk &Bs, int I, int J, const T (&av)[4], const T (&avp)[4], const T *XS, int lr, int lq) {
        if (K != J && !(I == J + 1 && K == J + 1)) {  // (column J was pivot_block's; block (J + 1, J + 1) is the look-ahead's)
            const bool tr = K < J;
            const T *b = XS + K * BS + (tr ? lq * 17 + lr : lr * 17 + lq);
            const int st = tr ? 4 * 17 : 4;
            T bv[4], aa[4];
#pragma unroll
            for (int kq = 0; kq < 4; kq++) {
                bv[kq] = b[kq * st];  // == opT / opN (XS + K BS, kq, lr, lq)
                aa[kq] = tr ? avp[kq] : av[kq];
            }
#pragma unroll
            for (int kq = 0; kq < 4; kq++) mfma16(aa[kq], bv[kq], Bs);
        }
    }
    static __device__ __forceinline__ void elim_B(int W, int J, csb_blk (&B)[NB + 1], T *wk, const T *sj, int lr, int lq) {
        const T *XS = wk + Lay::O_XS;
        const T *Wd = wk + Lay::O_TB + (J & 1) * BS;
        T *xp = wk + Lay::O_LP + W * BS;
        const T dc = sj[16 * J + lr];
        const int I1 = NB - 1 - W, I0 = W;
        constexpr int KH = (NB - 1) / 2;
        if (I1 > J) {
            T av[4], avp[4];
#pragma unroll
            for (int kq = 0; kq < 4; kq++) av[kq] = -opN(XS + I1 * BS, kq, lr, lq);
            dispatch<0, NB - 1>(J, [&](auto kc) __attribute__((always_inline)) { pivot_block(B[decltype(kc)::value], av, avp, Wd, xp, dc, lr, lq); });
            static_while<0, NB - 1>([&](auto kc) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value;
                if (K > I1) return false;
                update_block<K>(B[K], I1, J, av, avp, XS, lr, lq);
                return true;
            });
        } else if (I1 == J) {  // my row J is finished here, off the other rows' way: W_JK = Winv_JJ E_JK
            static_while<0, NB - 1>([&](auto kc) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value;
                if (K >= J) return false;
                finish_block(B[K], Wd, lr, lq);
                return true;
            });
        }
        if (I0 < I1 && I0 > J) {
            T av[4], avp[4];
#pragma unroll
            for (int kq = 0; kq < 4; kq++) av[kq] = -opN(XS + I0 * BS, kq, lr, lq);
            dispatch<0, KH>(J, [&](auto kc) __attribute__((always_inline)) { pivot_block(B[NB - decltype(kc)::value], av, avp, Wd, xp, dc, lr, lq); });
            static_while<0, KH>([&](auto kc) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value;
                if (K > I0) return false;
                update_block<K>(B[NB - K], I0, J, av, avp, XS, lr, lq);
                return true;
            });
        } else if (I0 < I1 && I0 == J) {
            static_while<0, KH>([&](auto kc) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value;
                if (K >= J) return false;
                finish_block(B[NB - K], Wd, lr, lq);
                return true;
            });
        }
    }
#define SQPH_CSB_SWITCH(wave, CALL) \
    switch (wave) { \
        case 0: if constexpr (Own<0>::any) { CALL(0); } break; \
        case 1: if constexpr (Own<1>::any) { CALL(1); } break; \
        case 2: if constexpr (Own<2>::any) { CALL(2); } break; \
        case 3: if constexpr (Own<3>::any) { CALL(3); } break; \
        case 4: if constexpr (Own<4>::any) { CALL(4); } break; \
        case 5: if constexpr (Own<5>::any) { CALL(5); } break; \
        case 6: if constexpr (Own<6>::any) { CALL(6); } break; \
        default: break; \
    }
    // returns false (block-uniform) when S is not positive definite / not finite; leaves W in B
    template <bool SP = false>
    static __device__ __forceinline__ bool factor(const TIN *__restrict__ gP, const int *__restrict__ pcol, const int *__restrict__ prow, int n,
                                                  T sigma, const Lay &L, unsigned char *smem, int t, csb_blk (&B)[NB + 1] SQPH_FTICK_ARGS) {
        T *lds = reinterpret_cast<T *>(smem);
        T *sj = lds + Lay::o_sj, *flag = lds + Lay::o_flag, *wk = lds;
        const int wave = wave_of(t), l = t & 63, lr = l & 15, lq = l >> 4;
        form_S<SP>(gP, pcol, prow, n, sigma, L, smem, t, wave, B SQPH_FTICK_PASS);
        SQPH_FTICK(1)
        if (t < 2) flag[t] = T(0);
        if (t < NP) sj[t] = T(1);
        __syncthreads();
        diag_to_sj(wave, B, sj, lr, lq);
        __syncthreads();
        if (t < NP) {
            const T d = sj[t];
            const bool bad = !(d > T(0)) || !(d * T(0) == T(0));  // non-positive / non-finite diagonal => not SPD
            sj[t] = bad ? T(1) : T(1) / (T)sqrt((double)d);
            if (bad) flag[0] = T(1);
        }
        __syncthreads();
        if (flag[0] != T(0)) return false;
        scale_blocks(wave, B, sj, wk, lr, lq);
        __syncthreads();
        SQPH_FTICK(2)
        if (wave == DW) MS::diag_block(wk + Lay::O_MD, wk + Lay::O_TB, sj, flag, l);
#pragma unroll 1
        for (int J = 0; J < NB - 1; J++) {
            // the lane's coordinates are derived again in every step: as loop invariants the LDS addresses of all 14 slots' operands
            // (~40 words) were hoisted out of this loop, spilled, and reloaded from scratch memory in front of every block product
            int lj = l, wj = wave;
            SQPH_OPAQUE_V(lj);
            SQPH_OPAQUE_S(wj);
            const int lrj = lj & 15, lqj = lj >> 4;
            __syncthreads();
            SQPH_FTICK(12)
            elim_A(wj, J, B, wk, lrj, lqj);
            SQPH_FTICK(13)
            __syncthreads();
            SQPH_FTICK(14)
            if (wj == DW)
                MS::template diag_block<true>(wk + Lay::O_MD + ((J + 1) & 1) * BS, wk + Lay::O_TB + ((J + 1) & 1) * BS, sj + 16 * (J + 1), flag, lj);
            else
                elim_B(wj, J, B, wk, sj, lrj, lqj);
            SQPH_FTICK(15)
        }
        {   // the last step has no trailing update (peeled: a loop left from its middle kept an old and a new copy of every block)
            int lj = l, wj = wave;
            SQPH_OPAQUE_V(lj);
            SQPH_OPAQUE_S(wj);
            __syncthreads();
            SQPH_FTICK(12)
            elim_A(wj, NB - 1, B, wk, lj & 15, lj >> 4);
            {   // (no trailing update behind the last step: the last row's W_JK = Winv_JJ E_JK here)
                const T *Wd = wk + Lay::O_TB + ((NB - 1) & 1) * BS;
                if (wj == 0) {  // row NB - 1 is wavefront 0's first row
                    static_while<0, NB - 2>([&](auto kc) __attribute__((always_inline)) {
                        finish_block(B[decltype(kc)::value], Wd, lj & 15, lj >> 4);
                        return true;
                    });
                }
            }
            SQPH_FTICK(13)
        }
        __syncthreads();
        SQPH_FTICK(7)
        return flag[1] == T(0);
    }

    // ---------------------------------------------------------------- the two dense stages of an iteration, one wavefront
    // y1 = W t for my block rows (partial sums over the 16 lanes of a DPP row through my slice of PW, summed by two lanes per
    // value), then the partial column sums of W' y1 over my rows, reduced over the four DPP rows and written to XP[column][wave].
    template <int W>
    static __device__ __forceinline__ void stages(const csb_blk (&B)[NB + 1], const T *tv, T *pw, T *xp, int n, int wave, int lr, int lq) {
        using O = Own<W>;
        T acc1[4] = {0, 0, 0, 0}, acc0[4] = {0, 0, 0, 0};
#pragma unroll
        for (int K = 0; K <= O::I1; K++) {
            const T tk = tv[16 * K + lr];
#pragma unroll
            for (int e = 0; e < 4; e++) acc1[e] = wg_fma(B[K].v[e], tk, acc1[e]);
            if (O::has0 && K <= O::I0) {
#pragma unroll
                for (int e = 0; e < 4; e++) acc0[e] = wg_fma(B[O::has0 ? NB - K : 0].v[e], tk, acc0[e]);
            }
        }
        T *mine = pw + wave * Lay::PWW;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            mine[(e * 4 + lq) * Lay::PWS + ((lr + 4 * e) & 15)] = acc1[e];
            mine[((4 + e) * 4 + lq) * Lay::PWS + ((lr + 4 * (4 + e)) & 15)] = acc0[e];
        }
        wave_fence();
        T ytot;
        {
            const int v = lr & 7, h = lr >> 3;
            T p[8];
            const T *row = mine + (v * 4 + lq) * Lay::PWS;
#pragma unroll
            for (int q = 0; q < 4; q++) {  // elements 8 h + 2 q, + 1 (an aligned pair stays one: the rotation is a multiple of 2)
                T two[2];
                wg_read<2>(row + ((8 * h + 2 * q + 4 * v) & 15), two);
                p[2 * q] = two[0];
                p[2 * q + 1] = two[1];
            }
            T s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
            s += xchg16<8>(s);
            const int i = 16 * (v < 4 ? O::I1 : O::I0) + lq + 4 * (v & 3);
            ytot = i < n ? s : T(0);  // lanes v and v + 8 of DPP row lq: y1 of row 16 I(v) + lq + 4 (v & 3)
        }
        T cacc[14];
#pragma unroll
        for (int K = 0; K < 14; K++) cacc[K] = T(0);
#pragma unroll
        for (int K = 0; K <= O::I1; K++) {
            cacc[K] = fmac4_bcast16<0>(cacc[K], ytot, B[K]);
            if (O::has0 && K <= O::I0) cacc[K] = fmac4_bcast16<4>(cacc[K], ytot, B[O::has0 ? NB - K : 0]);
        }
        T r[8];
#pragma unroll
        for (int i = 0; i < 7; i++) r[i] = swap_reduce32(cacc[i], cacc[i + 7]);
        r[7] = T(0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const T q = swap_reduce16(r[i], r[i + 4]);
            // row lq holds the total of column block K = i (lq 0), i + 4 (lq 1), i + 7 (lq 2), i + 11 (lq 3)
            const int K = i + ((lq & 1) ? 4 : 0) + ((lq & 2) ? 7 : 0);
            const bool valid = ((lq & 1) ? i < 3 : true) && K < NB;
            if (valid) xp[wave * NP + 16 * K + lr] = q;
        }
    }

    // tile <-> global workspace (the factor survives between setup() and solve() calls there), canonical col-major n x n, lower part
    static __device__ __forceinline__ void store_blocks(int W, T *__restrict__ gW, int n, int lr, int lq, const csb_blk (&B)[NB + 1]) {
#pragma unroll
        for (int s = 0; s <= NB; s++) {
            const Slot d = slot_of(W, s);
            if (!d.valid) continue;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 16 * d.I + lq + 4 * e, j = 16 * d.K + lr;
                if (i < n && j < n && i >= j) gW[(long)j * n + i] = B[s].v[e];
            }
        }
    }
    static __device__ __forceinline__ void load_blocks(int W, const T *__restrict__ gW, int n, int lr, int lq, csb_blk (&B)[NB + 1]) {
#pragma unroll
        for (int s = 0; s <= NB; s++) {
            const Slot d = slot_of(W, s);
            if (!d.valid) continue;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 16 * d.I + lq + 4 * e, j = 16 * d.K + lr;
                B[s].v[e] = (i < n && j < n && i >= j) ? gW[(long)j * n + i] : T(0);
            }
        }
    }

    // ---------------------------------------------------------------- a check-free segment of `seg` iterations
    // What an iteration needs besides the blocks (block-uniform values and this lane's maps), handed over as one record
    struct IterCtx {
        int n, rmap, cmap, im, flags;  // flags: 1 lead, 2 rreg, 4 creg
        T alpha, oma, sigma;
        int o_lo, o_up, o_rinv, o_wv, o_zs, o_ys, o_rho, o_val, o_rowptr, o_csc, o_col;
        SlotCode rsc, csc_;  // this lane's slot codes (place_slots)
    };
    static __device__ __forceinline__ void segment(const csb_blk (&B)[NB + 1], int seg, const IterCtx &c SQPH_FTICK_ARGS) {
        SQPH_DYN_SMEM(smem);
        T *lds = reinterpret_cast<T *>(smem);
        const int *li = reinterpret_cast<const int *>(smem);
        const int t = threadIdx.x, wave = wave_of(t), l = t & 63;
        const int n = c.n, rmap = c.rmap, cmap = c.cmap, im = c.im;
        const bool lead = c.flags & 1, rreg = c.flags & 2, creg = c.flags & 4;
        const T alpha = c.alpha, oma = c.oma, sigma = c.sigma;
        const int *rowptr = li + c.o_rowptr, *colptr = li + Lay::o_colptr;
        const unsigned *csc = reinterpret_cast<const unsigned *>(li + c.o_csc);
        const unsigned short *col = reinterpret_cast<const unsigned short *>(li + c.o_col);
        const T *val = lds + c.o_val;
        T *tv = lds + Lay::o_t, *xt = lds + Lay::o_xt, *ux = lds + Lay::o_ux, *qv = lds + Lay::o_qv, *wv = lds + c.o_wv, *xv = lds + Lay::o_xv;
        T *lov = lds + c.o_lo, *upv = lds + c.o_up, *rinvv = lds + c.o_rinv, *zs = lds + c.o_zs, *ys = lds + c.o_ys, *rhov = lds + c.o_rho;
        T *pw = lds + Lay::O_PW, *xp = lds + Lay::O_XP;
        // this lane's slices of A (CSR row group, CSC column group) in registers for the segment
        T rv[KR], cv[KR];
        int ri[KR / 2], ci[KR / 2];
        if (rreg) load_row_regs(rowptr, col, val, rmap, c.rsc, rv, ri);
        if (creg) load_col_regs(colptr, csc, val, cmap, c.csc_, cv, ci);
#pragma unroll 1
        for (int seg_i = 0; seg_i < seg; seg_i++) {
                __syncthreads();
                SQPH_ITICK(8)
                {   // t = (sigma x - q) + A' w
                    const T s = creg ? reg_dot(cv, ci, wv, cmap) : csc_col_dot_lds(colptr, csc, val, wv, cmap);
                    const int j = cmap & 511;
                    if ((cmap & MAP_VALID) && ((cmap >> 9) & 7) == 0) tv[j] = ux[j] + s;
                }
                __syncthreads();
                SQPH_ITICK(3)
                const int li = fresh_lane();
                const int lr_i = li & 15, lq_i = li >> 4;
#define SQPH_CSB_CALL(W_) stages<W_>(B, tv, pw, xp, n, wave, lr_i, lq_i)
                SQPH_CSB_SWITCH(wave, SQPH_CSB_CALL)
#undef SQPH_CSB_CALL
                __syncthreads();
                SQPH_ITICK(4)
                {   // x~ = W' y1; x relaxation (qp.cpp:96)
                    const int ti = (wave << 6) | fresh_lane();  // (addresses derived from the lane index are recomputed per phase: hoisted out of the loop they were spilled)
                    if (ti < NP) {
                        T p[8];
#pragma unroll
                        for (int w = 0; w < 8; w++) p[w] = xp[w * NP + ti];
                        const T xo = xv[ti], qj = qv[ti];
                        const bool own = ti < n;
                        const T xtj = own ? ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])) : T(0);
                        xt[ti] = xtj;
                        const T xn = alpha * xtj + oma * xo;
                        xv[ti] = xn;
                        ux[ti] = own ? sigma * xn - qj : T(0);  // next iteration's u
                    }
                }
                __syncthreads();
                SQPH_ITICK(5)
                {   // z~ = A x~ ; z, y updates (qp.cpp:99-103, 278-281)
                    const T zt = rreg ? reg_dot(rv, ri, xt, rmap) : csr_row_dot_lds(rowptr, col, val, xt, rmap);
                    if (lead) {
                        const T z = zs[im], y = ys[im], rho = rhov[im], rinv = rinvv[im];
                        const T zr = alpha * zt + oma * z;
                        T zn = zr + rinv * y;
                        const T lo = lov[im], up = upv[im];
                        zn = zn < lo ? lo : zn;
                        zn = zn > up ? up : zn;
                        const T yn = y + rho * (zr - zn);
                        zs[im] = zn;
                        ys[im] = yn;
                        wv[im] = rho * (zn - rinv * yn);  // next iteration's w (read after the loop-top barrier)
                    }
                }
                SQPH_ITICK(6)
                }
    }
    // (Until round 6 the checking instantiation ran its segments as a REAL CALL with the blocks handed over through memory: inlined, the
    // allocator had kept this lane's slices of A in scratch.  That was the structurized control flow of this kernel — see build.py's
    // CSB_FLAGS; with the uniform branches left alone the inlined segment is the faster form: config 5 under the SQP driver's settings
    // 21.1 -> 19.9 ms, under the reference defaults 48.6 either way.)
    static __device__ __forceinline__ void segment_call(const csb_blk (&Bm)[NB + 1], int seg, const IterCtx &c SQPH_FTICK_ARGS) { segment(Bm, seg, c SQPH_FTICK_PASS); }

    // the seven block-wide maxima of a termination check (|Ax|, |z|, |Ax - z|, |Px|, |A'y|, |q|, |Px + q + A'y|: qp.cpp:316-331, 353-361)
    // with the sparse products read from LDS and P streamed from global memory; every lane of the workgroup calls this.
    // PRIMAL FIRST, like the dense kernels: the dual half — A'y and above all P x, 8 n^2 bytes streamed per QP (2.6 GB per batch-wide
    // check at config 5) — only runs when the primal test passes or the caller needs it (rho adaptation; the last check a solve can
    // reach, whose residuals a MAX_ITER_EXCEEDED solve reports): primal_part() then dual_part().
    // SP (P in compressed columns, see form_S): gP_ is the value array and px... = (column pointers, row indices) — a trailing pack.
    // The primal half, inline in the solve loop (a sparse product from LDS and three maxima: a dozen registers): most checks of a solve fail
    // the primal test, and the call of the dual half costs its caller the registers that are live across it (~0.9 GB of scratch traffic
    // per batch-wide check at config 5).  Leaves x in xt and y in wv for the dual half; v[0..2] = |Ax|, |z|, |Ax - z|.
    static __device__ __forceinline__ void primal_part(const Lay &L, unsigned char *smem, int rmap, bool lead, int im, bool nown, T (&v)[7]) {
        const int t = threadIdx.x, wave = wave_of(t), l = t & 63;
        T *lds = reinterpret_cast<T *>(smem);
        const int *li = reinterpret_cast<const int *>(smem);
        const int *rowptr = li + L.o_rowptr;
        const unsigned short *col = reinterpret_cast<const unsigned short *>(li + L.o_col);
        const T *val = lds + L.o_val;
        T *xt = lds + Lay::o_xt, *wv = lds + L.o_wv, *xv = lds + Lay::o_xv;
        T *zs = lds + L.o_zs, *ys = lds + L.o_ys, *red = lds + Lay::o_red;
        __syncthreads();
        if (t < NP) xt[t] = nown ? xv[t] : T(0);
        if (lead) wv[im] = ys[im];
        __syncthreads();
        const T Ax = csr_row_dot_lds(rowptr, col, val, xt, rmap);
#pragma unroll
        for (int e = 0; e < 7; e++) v[e] = T(0);
        if (lead) {
            const T z = zs[im];
            v[0] = tabs(Ax);
            v[1] = tabs(z);
            v[2] = tabs(Ax - z);
        }
#pragma unroll
        for (int e = 0; e < 3; e++) v[e] = wave_nanmax(v[e]);
        if (l == 0) {
#pragma unroll
            for (int e = 0; e < 3; e++) red[e * 8 + wave] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 3; e++) {
            T mval = red[e * 8];
#pragma unroll
            for (int k = 1; k < 8; k++) mval = nanmax(mval, red[e * 8 + k]);
            v[e] = mval;
        }
    }
    // The dual half (a real call): v[3..6] = |Px|, |A'y|, |q|, |Px + q + A'y|
#ifdef SQPH_SIM
    template <bool SP = false, typename... PX>
    static inline void dual_part(const TIN *__restrict__ gP_, int n_, const Lay &L, unsigned char *smem, int cmap, bool nown, T (&v)[7], PX... px) {
#else
    template <bool SP = false, typename... PX>
    // (stays a real call: inlined, config 5 under the reference defaults gains 3 % and under the SQP driver's settings loses 35 %)
    static __device__ __attribute__((noinline)) void dual_part(const TIN *__restrict__ gP_, int n_, const Lay &L, unsigned char *smem, int cmap,
                                                                bool nown, T (&v)[7], PX... px) {
#endif
        const int n = uniform_int(n_);
        const TIN *__restrict__ gP = uniform_ptr(gP_);
        const int t = threadIdx.x, wave = wave_of(t), l = t & 63;
        (void)smem;
        SQPH_DYN_SMEM(smem_l);  // (the LDS-qualified base: through the pointer argument the accesses would be flat ones)
        T *lds = reinterpret_cast<T *>(smem_l);
        const int *li = reinterpret_cast<const int *>(smem_l);
        const int *colptr = li + Lay::o_colptr;
        const unsigned *csc = reinterpret_cast<const unsigned *>(li + uniform_int(L.o_csc));
        const T *val = lds + uniform_int(L.o_val);
        T *tv = lds + Lay::o_t, *xt = lds + Lay::o_xt, *qv = lds + Lay::o_qv, *wv = lds + uniform_int(L.o_wv);
        T *pw = lds + Lay::O_PW;
        T *red = lds + Lay::o_red;
        {   // A' y by the column map's lanes, handed to the lanes that track x through tv
            const T sATy = csc_col_dot_lds(colptr, csc, val, wv, cmap);
            if ((cmap & MAP_VALID) && ((cmap >> 9) & 7) == 0) tv[cmap & 511] = sATy;
        }
        {   // P x with the full P (both triangles, qp.cpp:324), streamed: lane t takes row t & 255 and every second column
            const int i = t & 255, h = t >> 8;
            T acc = 0;
            if constexpr (SP) {
                // column i of the symmetric P is its row i; the entries in ascending order, the two lanes of a row taking the even /
                // odd columns: the dense path's chains without their zero terms
                const int *const pp[2] = {px...};
                const int *__restrict__ pcol = uniform_ptr(pp[0]), *__restrict__ prow = uniform_ptr(pp[1]);
                if (i < n) {
                    const int e1 = pcol[i + 1];
                    for (int e = pcol[i]; e < e1; e++) {
                        const int k = prow[e];
                        if ((k & 1) == h && k < n) acc = wg_fma((T)gP[e], xt[k], acc);
                    }
                }
            } else
            if (i < n) {
                const TIN *pr = gP + i;
                int j = h;
                for (; j + 6 < n; j += 8) {
                    const T p0 = (T)pr[(long)j * n], p1 = (T)pr[(long)(j + 2) * n], p2 = (T)pr[(long)(j + 4) * n],
                            p3 = (T)pr[(long)(j + 6) * n];
                    acc = wg_fma(p0, xt[j], acc);
                    acc = wg_fma(p1, xt[j + 2], acc);
                    acc = wg_fma(p2, xt[j + 4], acc);
                    acc = wg_fma(p3, xt[j + 6], acc);
                }
                for (; j < n; j += 2) acc = wg_fma((T)pr[(long)j * n], xt[j], acc);
            }
            pw[h * 256 + i] = acc;
        }
        __syncthreads();
        if (nown) {
            const T Px = pw[t] + pw[256 + t], ATy = tv[t], q = qv[t];
            v[3] = tabs(Px);
            v[4] = tabs(ATy);
            v[5] = tabs(q);
            v[6] = tabs(Px + q + ATy);
        }
#pragma unroll
        for (int e = 3; e < 7; e++) v[e] = wave_nanmax(v[e]);
        if (l == 0) {
#pragma unroll
            for (int e = 3; e < 7; e++) red[e * 8 + wave] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 3; e < 7; e++) {
            T mval = red[e * 8];
#pragma unroll
            for (int k = 1; k < 8; k++) mval = nanmax(mval, red[e * 8 + k]);
            v[e] = mval;
        }
        __syncthreads();
    }

    // CHECKS = false: the instantiation for calls that never look at the residuals (check_termination == 0, no adaptive rho)
    // SP = true: P comes in compressed columns (CsrArgs::p_*: the sqph_*_csr_sp entry points) instead of the dense a.P
    template <bool CHECKS = true, bool SP = false>
    static __device__ __forceinline__ void run(const KArgs<T, TIN> &a, const CsrArgs<TIN> &ca, unsigned char *smem) {
        const int qp = blockIdx.x;
        if (qp >= a.batch) return;
        const int n = a.n, m = a.m;
        const Lay L = Lay::make(m, ca.nnz_cap);
        T *lds = reinterpret_cast<T *>(smem);
        int *li = reinterpret_cast<int *>(smem);
        const int *rowptr = li + L.o_rowptr, *colptr = li + L.o_colptr;
        const unsigned *csc = reinterpret_cast<const unsigned *>(li + L.o_csc);
        const unsigned short *col = reinterpret_cast<const unsigned short *>(li + L.o_col);
        const T *val = lds + L.o_val;
        T *tv = lds + Lay::o_t, *xt = lds + Lay::o_xt, *ux = lds + Lay::o_ux, *qv = lds + Lay::o_qv, *wv = lds + L.o_wv;
        T *lov = lds + L.o_lo, *upv = lds + L.o_up, *rinvv = lds + L.o_rinv;
        T *zs = lds + L.o_zs, *ys = lds + L.o_ys, *rhov = lds + L.o_rho;  // z, y, rho of the constraint rows live in LDS (written by a row's lead lane)
        T *pw = lds + Lay::O_PW, *xp = lds + Lay::O_XP, *xv = lds + Lay::o_xv;

        const TIN *gP;
        const int *pcol = nullptr, *prow = nullptr;
        if constexpr (SP) {
            gP = ca.p_val + (long)qp * ca.s_pval;
            pcol = ca.p_colptr + (long)qp * ca.s_pcolptr;
            prow = ca.p_rowind + (long)qp * ca.s_prowind;
        } else {
            gP = a.P + (long)qp * a.sP;
        }
        const TIN *gq = a.q + (long)qp * a.sq;
        const TIN *gl = a.l + (long)qp * a.sl;
        const TIN *gu = a.u + (long)qp * a.su;
        T *sx = a.x + (long)qp * n;
        T *sz = a.z + (long)qp * m;
        T *sy = a.y + (long)qp * m;
        T *srho = a.rho_vec + (long)qp * m;
        int *sct = a.ctype + (long)qp * m;
        T *gW = a.Sinv + (long)qp * 2 * n * n;

        sqph_info info = a.info[qp];
        T rho_s = a.rho[qp];
        const int mode = a.mode;
        if (!(mode & (MODE_SETUP | MODE_UPDATE)) && (info.status == SQPH_UNINITIALIZED || info.status == SQPH_NUMERICAL_ISSUES))
            return;  // qp.cpp:68-71 (block-uniform)

        int t = threadIdx.x;
        const int wave = wave_of(t), l = t & 63, lr = l & 15, lq = l >> 4;
#ifdef SQPH_PHASE_TIMING
        unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
        const unsigned long long tstart = tprev;
#endif
        // element owners: lane j < n tracks x_j; the lanes the ROW MAP gives constraint row i track z_i, y_i, rho_i (all of them keep a
        // copy, the one with part 0 writes); the COLUMN MAP's lanes sum the columns of A' w
        // q, l, u: requested now (plain-indexed, every lane its share), stored behind the sparse loading — their memory latency hides
        // behind it instead of following the lane maps (l and u are kept per row, whoever leads the row later)
        T q_early = T(0), l_early[2] = {T(0), T(0)}, u_early[2] = {T(0), T(0)};
        if (t < n) q_early = (T)gq[t];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = t + k * NT;
            if (i < m) {
                l_early[k] = (T)gl[i];
                u_early[k] = (T)gu[i];
            }
        }
        load_sparse(ca, qp, n, m, L, smem);
        SQPH_STICK2(3)
        int rmap, cmap;
        bool rreg, creg;
        {
            unsigned short *tmap = reinterpret_cast<unsigned short *>(lds);  // the work area is idle
            int *scratch = reinterpret_cast<int *>(tmap + 2 * NT);
            int Kr, Kc;
            build_lane_maps(rowptr, m, colptr, n, tmap, scratch, Kr, Kc);
            SQPH_STICK2(4)
            rmap = (int)tmap[t];
            cmap = (int)tmap[NT + t];
            rreg = Kr <= KR;
            creg = Kc <= KR;
            __syncthreads();
        }
        SQPH_BTICK(0)
        const int im = rmap & 511;
        const bool mown = (rmap & MAP_VALID) != 0, lead = mown && ((rmap >> 9) & 7) == 0;
        const bool nown = t < n;

        // x lives in LDS: as a register it was spilled inside the iteration loop (a scratch round trip on the chain between two
        // barriers, twice per iteration); each lane touches its own element only
        if (t < NP) xv[t] = T(0);
        if (t < NP) {
            qv[t] = q_early;
            tv[t] = T(0);
            xt[t] = T(0);
            ux[t] = T(0);
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = t + k * NT;
            if (i < m) {
                lov[i] = l_early[k];
                upv[i] = u_early[k];
            }
        }
        __syncthreads();
        bool rho_differs = false;  // against the vector the resident factor was built with (MODE_SAME_MATRICES)
        if (mode & (MODE_SETUP | MODE_UPDATE)) {
            rho_s = a.rho0;
            if (lead) {
                const T lo = lov[im], up = upv[im];
                int ctype = SQPH_INEQUALITY_CONSTRAINT;
                if (lo < -a.loose_thresh && up > a.loose_thresh)
                    ctype = SQPH_LOOSE_BOUNDS;
                else if (up - lo < a.eq_tol)
                    ctype = SQPH_EQUALITY_CONSTRAINT;
                const T rho = rho_for_type<T>(ctype, rho_s, a.rho_min, a.rho_eq_factor);
                rho_differs = !(rho == srho[im]);
                rhov[im] = rho;
                rinvv[im] = T(1) / rho;
                sct[im] = ctype;
                srho[im] = rho;
                const bool keep = !(mode & MODE_SETUP);
                zs[im] = keep ? sz[im] : T(0);
                ys[im] = keep ? sy[im] : T(0);
            }
            info.rho_updates += 1;
            if (!(mode & MODE_SETUP) && nown) xv[t] = sx[t];
        } else {
            if (nown) xv[t] = sx[t];
            if (lead) {
                const T rho = srho[im];
                zs[im] = sz[im];
                ys[im] = sy[im];
                rhov[im] = rho;
                rinvv[im] = T(1) / rho;
            }
        }

        csb_blk B[NB + 1];
#pragma unroll
        for (int s = 0; s <= NB; s++) B[s] = csb_blk{{0, 0, 0, 0}};  // (defined on every path: the slots a wavefront does not own are never touched)
        bool need_factor = (mode & (MODE_SETUP | MODE_UPDATE | MODE_REFACTOR)) != 0;
        if ((mode & MODE_SAME_MATRICES) && (mode & (MODE_SETUP | MODE_UPDATE)) && !(mode & MODE_REFACTOR) &&
            info.status != SQPH_NUMERICAL_ISSUES && info.status != SQPH_UNINITIALIZED) {  // (a failed set-up left no valid factor)
            // sqph_setup_solve_reuse_csr (the SQP second-order correction re-solves with new bounds only, src/sqp.cpp:244-276, TODO
            // :273): P and A are those of the resident factor — which is also the factor this set-up would build unless some row's
            // freshly classified rho differs from the vector it was built with (workgroup-wide OR through one LDS word)
            T *same = lds + Lay::o_flag + 2;
            if (t == 0) *same = T(0);
            __syncthreads();
            if (rho_differs) *same = T(1);
            __syncthreads();
            if (*same == T(0)) {
                need_factor = false;
                info.status = SQPH_UNSOLVED;  // qp.cpp:39-43
            }
            __syncthreads();
        }
        bool solving = false;
        bool state_dirty = (mode & MODE_SETUP) != 0;
        if (!need_factor) {
            load_blocks(wave, gW, n, lr, lq, B);
        }
        const T alpha = a.alpha, sigma = a.sigma, oma = T(1) - a.alpha;
        int iter = 1;
        int next_check = a.check_termination > 0 ? a.check_termination : -1;
        int next_adapt = (a.adaptive_rho && a.adaptive_rho_interval > 0) ? a.adaptive_rho_interval : -1;
        for (;;) {
            if (need_factor) {
                __syncthreads();
                SQPH_BTICK(10)
                bool ok;
                ok = factor<SP>(gP, pcol, prow, n, sigma, L, smem, t, B SQPH_FTICK_PASS);
                if (!(mode & MODE_NO_FACTOR_STORE)) store_blocks(wave, gW, n, lr, lq, B);  // kept for later solve() calls
                SQPH_BTICK(11)
                __syncthreads();
                need_factor = false;
                if (!solving) {
                    if (mode & (MODE_SETUP | MODE_UPDATE)) info.status = ok ? SQPH_UNSOLVED : SQPH_NUMERICAL_ISSUES;  // qp.cpp:39-43, 57-61
                    else if (!ok) info.status = SQPH_NUMERICAL_ISSUES;  // solve() rebuilding a factor that was not kept
                } else if (!ok) {
                    info.status = SQPH_NUMERICAL_ISSUES;  // qp.cpp:139-142
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
                    if (t < NP) xv[t] = T(0);
                    if (lead) zs[im] = ys[im] = T(0);
                }
            }
            // (re)load the sparse slices: the work area they border on was used by the set-up; w, u of the first iteration
            __syncthreads();
            SlotCode rsc{0, 0}, csc_{0, 0};
            {   // slots of the register-resident slices (the work area is idle; per half-wavefront tables, no workgroup barrier inside)
                unsigned *area = reinterpret_cast<unsigned *>(lds);
                // ... where the solve is long enough to pay for it: same-box at config 5, placement against storage order — 200 fixed
                // iterations 21.42 / 22.01 ms, the reference defaults (max_iter 1000) 55.4 / 55.5, the SQP driver's settings (max_iter 100,
                // a check every 10) 27.4 / 25.7.  The choice depends on settings.max_iter alone (block-uniform, the same for every QP
                // and every call with these settings: results stay reproducible); the summation order inside a lane follows the slots.
                const bool place = a.max_iter >= SQPH_CSB_PLACE_MIN_ITERS;
                SQPH_STICK2(6)
                if (rreg) rsc = place_slots<true>(rowptr, col, csc, rmap, NP, area, t, place);
                if (creg) csc_ = place_slots<false>(colptr, col, csc, cmap, m, area, t, place);
                __syncthreads();
                SQPH_STICK2(8)
            }
            for (int e = t; e < NP * 8; e += NT) xp[e] = T(0);
            if (lead) wv[im] = rhov[im] * (zs[im] - rinvv[im] * ys[im]);
            if (t < NP) ux[t] = nown ? sigma * xv[t] - qv[t] : T(0);
            SQPH_BTICK(10)
            // Segments: the iterations up to the next residual check run in an inner loop that contains no check code — with the check
            // (it streams P and runs both sparse products again) inside the iteration loop the allocator reloaded the blocks from scratch
            // in EVERY iteration (153 scratch loads per iteration; config 5 under the default settings 245 ms against the tile kernel's 97)
            const IterCtx ic{n, rmap, cmap, im, (lead ? 1 : 0) | (rreg ? 2 : 0) | (creg ? 4 : 0), alpha, oma, sigma,
                             L.o_lo, L.o_up, L.o_rinv, L.o_wv, L.o_zs, L.o_ys, L.o_rho, L.o_val, L.o_rowptr, L.o_csc, L.o_col, rsc, csc_};
            while (iter <= a.max_iter) {
                int seg = a.max_iter - iter + 1;
                if constexpr (CHECKS) {
                    if (next_check > 0 && next_check < seg) seg = next_check;
                    if (next_adapt > 0 && next_adapt < seg) seg = next_adapt;
                }
                if constexpr (CHECKS) {
                    // what only the checks need (the info record, the scalar rho) waits in LDS while a segment runs: every lane
                    // writes the same values (they are block-uniform), every lane reads them back behind the segment
                    T *park = lds + Lay::o_red + 56;
                    park[0] = (T)info.status; park[1] = (T)info.rho_updates; park[2] = (T)info.rho_estimate;
                    park[3] = (T)info.res_prim; park[4] = (T)info.res_dual; park[5] = rho_s;
                }
                if constexpr (CHECKS) segment_call(B, seg, ic SQPH_FTICK_PASS);
                else segment(B, seg, ic SQPH_FTICK_PASS);
                iter += seg - 1;  // the iteration the checks below belong to (qp.cpp:105: iter % check_termination == 0)
                if constexpr (CHECKS) {
                    const T *park = lds + Lay::o_red + 56;
                    info.status = (int)park[0]; info.rho_updates = (int)park[1]; info.rho_estimate = (double)park[2];
                    info.res_prim = (double)park[3]; info.res_dual = (double)park[4]; rho_s = park[5];
                }
                bool check = false, adapt = false;
                if constexpr (CHECKS) {
                    if (next_check > 0 && (next_check -= seg) == 0) {
                        check = true;
                        next_check = a.check_termination;
                    }
                    if (next_adapt > 0 && (next_adapt -= seg) == 0) {
                        adapt = true;
                        next_adapt = a.adaptive_rho_interval;
                    }
                }
                if (CHECKS && (check || adapt)) {
                    // update_state + residuals, qp.cpp:316-331, 353-361: the primal half inline, the dual half (P streamed eight loads
                    // deep, A'y, four more reductions) OUT OF LINE — a real call, only when the primal test passes or the caller needs it
                    T v[7];
                    const bool last_check = check && iter + a.check_termination > a.max_iter;
                    primal_part(L, smem, rmap, lead, im, nown, v);
                    const bool dual = adapt || last_check || (v[2] <= (T)a.eps_abs + (T)a.eps_rel * nanmax(v[0], v[1]));  // (block-uniform)
                    if (dual) {
                        if constexpr (SP) dual_part<true>(gP, n, L, smem, cmap, nown, v, pcol, prow);
                        else dual_part(gP, n, L, smem, cmap, nown, v);
                    } else {
                        __syncthreads();
                    }
                    const T nrm_prim = nanmax(v[0], v[1]);
                    const T nrm_dual = nanmax(v[3], nanmax(v[4], v[5]));
                    info.res_prim = (double)v[2];
                    if (dual) info.res_dual = (double)v[6];  // (otherwise the primal test failed: the solve goes on, and the last check it can reach is a full one)
                    if (check && dual) {
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
                            if (lead) {
                                const T rho = rho_for_type<T>(sct[im], rho_s, a.rho_min, a.rho_eq_factor);
                                rhov[im] = rho;
                                rinvv[im] = T(1) / rho;
                            }
                            info.rho_updates += 1;
                            need_factor = true;
                            break;  // leave WITHOUT advancing iter; the factor block does it
                        }
                    }
                    // the check borrowed xp's neighbours: PW held the P x partial sums, wv held y
                    __syncthreads();
                    if (lead) wv[im] = rhov[im] * (zs[im] - rinvv[im] * ys[im]);
                }
                iter++;
            }
            // the thread index starts a new life here: live across the segments it was the allocator's spill victim, reloaded from
            // scratch memory inside the iteration loop (the write-back below is its next use)
            t = (wave << 6) | fresh_lane();
            if (!need_factor) break;
        }
        if (solving) {
            if (iter > a.max_iter) info.status = SQPH_MAX_ITER_EXCEEDED;
            info.iter = iter;
        }
#ifdef SQPH_PHASE_TIMING
        tacc[9] = __builtin_amdgcn_s_memtime() - tstart;
#ifndef SQPH_PT_WAVE
#define SQPH_PT_WAVE 0
#endif
        __syncthreads();
        if (t >= 64 * SQPH_PT_WAVE && t < 64 * SQPH_PT_WAVE + 16) xv[t - 64 * SQPH_PT_WAVE] = (T)tacc[t - 64 * SQPH_PT_WAVE];  // debug build only: one wavefront's phase ticks instead of x[0..16)
        __syncthreads();
#endif
        if (state_dirty) {
            if (nown) sx[t] = xv[t];
            if (lead) {
                sz[im] = zs[im];
                sy[im] = ys[im];
                srho[im] = rhov[im];
            }
        }
        if (t == 0) {
            a.info[qp] = info;
            a.rho[qp] = rho_s;
        }
    }
#undef SQPH_CSB_SWITCH
};

template <typename TIN, int NB>
__global__ __launch_bounds__(512) void admm_csrb_kernel(CsrLaunch<TIN> p) {
    SQPH_DYN_SMEM(smem_raw);
    CsbKernel<TIN, NB>::template run<true>(p.a, p.ca, smem_raw);
}
// the same without the residual-check block
template <typename TIN, int NB>
__global__ __launch_bounds__(512) void admm_csrb_nocheck_kernel(CsrLaunch<TIN> p) {
    SQPH_DYN_SMEM(smem_raw);
    CsbKernel<TIN, NB>::template run<false>(p.a, p.ca, smem_raw);
}
// ... and the two with P in compressed columns (sqph_*_csr_sp; instantiated in csrb_sp.hip only)
template <typename TIN, int NB>
__global__ __launch_bounds__(512) void admm_csrb_sp_kernel(CsrLaunch<TIN> p) {
    SQPH_DYN_SMEM(smem_raw);
    CsbKernel<TIN, NB>::template run<true, true>(p.a, p.ca, smem_raw);
}
template <typename TIN, int NB>
__global__ __launch_bounds__(512) void admm_csrb_sp_nocheck_kernel(CsrLaunch<TIN> p) {
    SQPH_DYN_SMEM(smem_raw);
    CsbKernel<TIN, NB>::template run<false, true>(p.a, p.ca, smem_raw);
}

// block-row counts compiled into the library (n <= 16 NB): first fit wins
#if defined(SQPH_SLIM) && defined(SQPH_SLIM_CSR)
#define SQPH_CSB_SHAPES(X) X(13)
#elif defined(SQPH_SLIM)
#define SQPH_CSB_SHAPES(X)
#else
#define SQPH_CSB_SHAPES(X) X(5) X(9) X(13) X(14)
#endif
// additional small counts for the host SIMT emulation in the CPU test-suite
#define SQPH_CSB_SIM_SHAPES(X) X(1) X(2) X(3) X(5) X(13)

// launches the kernel (nocheck: the instantiation without the residual-check block) where a block-row count NB is compiled in:
// > 0 launched, 0 no such count, < 0 HIP error.  Defined in csrb.hip, the only translation unit that instantiates these kernels.
template <typename TIN>
int csrb_launch(int NB, bool nocheck, int m, int nnz_cap, int batch, hipStream_t stream, const CsrLaunch<TIN> &p);
extern template int csrb_launch<double>(int, bool, int, int, int, hipStream_t, const CsrLaunch<double> &);
extern template int csrb_launch<float>(int, bool, int, int, int, hipStream_t, const CsrLaunch<float> &);
// the same for the sparse-P instantiations (csrb_sp.hip)
template <typename TIN>
int csrb_sp_launch(int NB, bool nocheck, int m, int nnz_cap, int batch, hipStream_t stream, const CsrLaunch<TIN> &p);
extern template int csrb_sp_launch<double>(int, bool, int, int, int, hipStream_t, const CsrLaunch<double> &);
extern template int csrb_sp_launch<float>(int, bool, int, int, int, hipStream_t, const CsrLaunch<float> &);

#ifdef SQPH_SIM
template <typename TIN>
inline int sim_run_csrb(const KArgs<double, TIN> &a, const CsrArgs<TIN> &ca) {
    if (a.m > 512) return -1;
#define SQPH_SIM_CASE(NB_)                                                                                               \
    if (a.n <= 16 * NB_) {                                                                                               \
        const CsbLayout<NB_> L = CsbLayout<NB_>::make(a.m, ca.nnz_cap);                                                  \
        if (a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0))                                \
            ::sqph_sim::launch(admm_csrb_nocheck_kernel<TIN, NB_>, dim3(a.batch), dim3(512), L.bytes, CsrLaunch<TIN>{a, ca}); \
        else                                                                                                             \
            ::sqph_sim::launch(admm_csrb_kernel<TIN, NB_>, dim3(a.batch), dim3(512), L.bytes, CsrLaunch<TIN>{a, ca});    \
        return 0;                                                                                                        \
    }
    SQPH_CSB_SIM_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
template <typename TIN>
inline int sim_run_csrb_sp(const KArgs<double, TIN> &a, const CsrArgs<TIN> &ca) {
    if (a.m > 512) return -1;
#define SQPH_SIM_CASE(NB_)                                                                                               \
    if (a.n <= 16 * NB_) {                                                                                               \
        const CsbLayout<NB_> L = CsbLayout<NB_>::make(a.m, ca.nnz_cap);                                                  \
        if (a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0))                                \
            ::sqph_sim::launch(admm_csrb_sp_nocheck_kernel<TIN, NB_>, dim3(a.batch), dim3(512), L.bytes, CsrLaunch<TIN>{a, ca}); \
        else                                                                                                             \
            ::sqph_sim::launch(admm_csrb_sp_kernel<TIN, NB_>, dim3(a.batch), dim3(512), L.bytes, CsrLaunch<TIN>{a, ca}); \
        return 0;                                                                                                        \
    }
    SQPH_CSB_SIM_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

}  // namespace sqph
