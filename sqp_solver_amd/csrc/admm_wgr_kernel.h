// Row-split workgroup kernel: the register-tiled ADMM iteration of admm_wg_kernel.h with ONE workgroup barrier per iteration.
//
// Same R x C lane grid and the same set-up (WgKernel::factor / build_B_inplace are reused unchanged), but
//   (1) lanes are numbered c-fastest (t = r * C + c), so a wavefront holds R / NW whole ROWS of the grid instead of whole columns.
//       The iteration's two exchanges then become:   stage 1  y1 = M' v   (reduced over r: crosses the waves -> the one barrier)
//                                                     stage 2  [z~; x~] = M y1  (reduced over c: inside a wave)
//       and the owner of a row lives in the wave that produced the row's partial sums, so the z / y / x updates and the publication
//       of the next operand need no barrier either.  (The column-split kernel needs two barriers: partials -> owners -> operands both
//       cross the waves; measured there: 13 % / 28 % of an iteration spent waiting at them.)
//   (2) the stacked operator M = [B; W'] ((m + n) x n) is tiled as ONE matrix: TS = ceil((m + n) / R) tile rows per lane instead of
//       ceil(m / R) + ceil(n / R): the rows of W' first fill the R * (TS - TR) slots of the tile rows behind B, then the slots that B
//       leaves free in its last tile row.  n = 50, m = 100, R = 16: 10 instead of 7 + 4 tile rows (70 FMAs per stage instead of 77,
//       14 VGPRs fewer).
// A "slot" is (r, s): row r of the lane grid, tile row s < TS.  Slot (r, s) is owned by lane (r, c = s mod C) in pass s / C; it is a
// z-slot (constraint row i = R s + r < m), an x-slot (variable j, a row of W') or empty.
//
// Numerics: the formulas of admm_wg_kernel.h (reference src/qp.cpp:84-144 on the Schur-ordered system); the summation order of the
// partial sums differs (over r first / over c first are unchanged, the tile-row order of W' rows is), i.e. results agree with the
// column-split kernel to rounding, not bit for bit.
#pragma once
#include "admm_wg_kernel.h"

// what-if timing builds (wrong results): -DSQPH_WGR_WHATIF=<bit mask> replaces the LDS addresses of one access class of the iteration by
// lane-linear, conflict-free ones (1 operand read, 2 stage-1 store, 4 stage-1 reduce, 8 y1, 16 stage-2 store, 32 owner read, 64 publish)
#ifndef SQPH_WGR_WHATIF
#define SQPH_WGR_WHATIF 0
#endif

namespace sqph {

template <int NW, int R, int C, int TR, int TC, int TW, int TS>
struct WgrLayout {
    using L0 = WgLayout<NW, R, C, TR, TC, TW>;  // the set-up's scratch map (factor, build_B) and the residual check's staging
    static_assert(NW >= 1 && R % NW == 0 && (R / NW) * C == 64, "a wavefront holds R / NW whole rows of the lane grid");
    static_assert(TS >= TR && TS <= TR + TW, "stacked tile rows");
    static constexpr int NT = R * C;
    static constexpr int RW = R / NW;            // grid rows per wavefront
    static constexpr int NP = C * TC, MP = R * TR;
    static constexpr int XS = TS - TR;           // tile rows that hold rows of W' only
    static constexpr int NX0 = R * XS;           // rows of W' in those; rows j >= NX0 sit in the free slots of tile row TR - 1
    static constexpr int NPASS = (TS + C - 1) / C;
    static constexpr int ev(int x) { return (x + 1) & ~1; }
    static constexpr int mx(int a, int b) { return a > b ? a : b; }
    // iteration map (doubles); every region starts 16-byte aligned
    // Strides chosen with the bank model of tools/xp/lds_bank_model.py (MI355X_MICROARCH.md, LDS): every access of the iteration
    // is conflict-free except the stage-2 stores (176 instead of 144 LDS cycles per iteration and QP).
    //   operands  [R][TSp]                         TSp = 24: rows of two consecutive r in different bank halves for the b64 stores
    //   y1        [NW][C][TCp]                     TCp = 10
    //   stage 1   [2][c CS1 + k KS1 + r]           one 16-entry row per column TC c + k  (double-buffered: one barrier per iteration)
    //   stage 2   [r RS2 + s SS2 + c]              one C-entry row per slot (r, s)
#ifndef SQPH_WGR_TSP
#define SQPH_WGR_TSP 24
#define SQPH_WGR_TCP 10
#define SQPH_WGR_CS1 114
#define SQPH_WGR_KS1 16
#define SQPH_WGR_RS2 82
#define SQPH_WGR_SS2 8
#endif
    static_assert(R == 16 && C == 8 && TC <= 7 && TS <= 10, "the strides above were searched for the 16 x 8 grid");
    static constexpr int TSp = SQPH_WGR_TSP, TCp = SQPH_WGR_TCP, CS1 = SQPH_WGR_CS1, KS1 = SQPH_WGR_KS1, RS2 = SQPH_WGR_RS2, SS2 = SQPH_WGR_SS2;
    static constexpr int S1 = ev((C - 1) * CS1 + (TC - 1) * KS1 + R);  // one stage-1 buffer
    static constexpr int O_ROWV = 0;
    static constexpr int O_Y1 = O_ROWV + R * TSp;
    static constexpr int O_ST1 = ev(O_Y1 + NW * C * TCp);
    static constexpr int O_ST2 = O_ST1 + 2 * S1;
    static constexpr int LOOP_END = ev(O_ST2 + (R - 1) * RS2 + mx(TS, C * NPASS) * SS2 + C);  // (over-reads of slot-less lanes included)
    // per-slot constants of the owners (lo, up, 1/rho) for the passes that can hold z-slots, behind everything the set-up and the
    // residual check alias
    static constexpr int NZC = R * C * ((TR + C - 1) / C);
    static constexpr int O_C0 = ev(mx(LOOP_END, L0::O_QV));
    static constexpr int O_C1 = O_C0 + NZC;
    static constexpr int O_C2 = O_C1 + NZC;
    static constexpr int TOTAL = ev(O_C2 + NZC);
    static_assert(TSp % 2 == 0 && TCp % 2 == 0 && CS1 % 2 == 0 && KS1 % 2 == 0 && RS2 % 2 == 0 && SS2 % 2 == 0, "16-byte aligned rows");
};

template <typename TIN, int NW, int R, int C, int TR, int TC, int TW, int TS>
struct WgrKernel {
    using T = double;
    using K0 = WgKernel<TIN, NW, R, C, TR, TC, TW>;
    using L0 = typename K0::L;
    using L = WgrLayout<NW, R, C, TR, TC, TW, TS>;
    static constexpr int XS = L::XS, NPASS = L::NPASS, RW = L::RW;
    enum : int { K_NONE = 0, K_Z = 1, K_X = 2 };

    static __device__ __forceinline__ void wave_sync() { K0::wave_sync(); }

    // pass p of the owners can hold z-slots / x-slots at all (compile time)
    static constexpr bool pass_has_z(int p) { return C * p < TR; }
    static constexpr bool pass_has_x(int p) { return (C * p + C - 1 < TS - 1 ? C * p + C - 1 : TS - 1) >= TR - 1; }

    //   stage 1:  y1[TC c + k] partial over my TS slots  (reduced over r, across the waves)
    static __device__ __forceinline__ void stage1(const T (&bt)[TR][TC], const T (&xt)[XS > 0 ? XS : 1][TC], const T (&v)[TS], T *st1, int r, int c) {
        T pb[TC];
#pragma unroll
        for (int k = 0; k < TC; k++) pb[k] = 0;
#pragma unroll
        for (int s = 0; s < TR; s++)
#pragma unroll
            for (int k = 0; k < TC; k++) pb[k] = wg_fma(bt[s][k], v[s], pb[k]);
#pragma unroll
        for (int u = 0; u < XS; u++)
#pragma unroll
            for (int k = 0; k < TC; k++) pb[k] = wg_fma(xt[u][k], v[TR + u], pb[k]);
#if defined(SQPH_WGR_ASM_ST1) && !defined(SQPH_SIM)
        // single ds_write_b64 stores: merged into ds_write2_b64 pairs (8-lane groups, two addresses per lane) the two halves of a pair
        // share their banks in every layout with 16-byte aligned rows; plain 8-byte stores are conflict-free with CS1 = 114, KS1 = 16
        typedef __attribute__((address_space(3))) T lds_t;
        const unsigned a0 = (unsigned)(unsigned long)((lds_t *)(st1 + c * L::CS1 + r));
#pragma unroll
        for (int k = 0; k < TC; k++) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a0), "v"(pb[k]), "n"(k * L::KS1 * 8) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#pragma unroll
        for (int k = 0; k < TC; k++) st1[(SQPH_WGR_WHATIF & 2) ? (int)(threadIdx.x & 63) + 64 * k : c * L::CS1 + k * L::KS1 + r] = pb[k];
#endif
    }
    //   stage 2:  [z~; x~] of my TS slots, partial over my TC columns  (reduced over c, inside the wave)
    static __device__ __forceinline__ void stage2(const T (&bt)[TR][TC], const T (&xt)[XS > 0 ? XS : 1][TC], const T (&y1)[TC], T *st2, int r, int c) {
        T pz[TS];
#pragma unroll
        for (int s = 0; s < TS; s++) pz[s] = 0;
#pragma unroll
        for (int k = 0; k < TC; k++) {
#pragma unroll
            for (int s = 0; s < TR; s++) pz[s] = wg_fma(bt[s][k], y1[k], pz[s]);
#pragma unroll
            for (int u = 0; u < XS; u++) pz[TR + u] = wg_fma(xt[u][k], y1[k], pz[TR + u]);
        }
#pragma unroll
        for (int s = 0; s < TS; s++) st2[(SQPH_WGR_WHATIF & 16) ? (int)(threadIdx.x & 63) + 64 * s : s * L::SS2 + r * L::RS2 + c] = pz[s];
    }

    template <bool CHECKS = true>
    static __device__ __forceinline__ void run(const KArgs<T, TIN> &a, T *lds) {
        const int t = threadIdx.x;
        const int r = t / C, c = t % C;
        const int wv = t >> 6, rl = r % RW;
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

        sqph_info info = a.info[qp];
        T rho_s = a.rho[qp];
        const int mode = a.mode;
        if (!(mode & (MODE_SETUP | MODE_UPDATE)) && (info.status == SQPH_UNINITIALIZED || info.status == SQPH_NUMERICAL_ISSUES))
            return;  // qp.cpp:68-71 (block-uniform)

        // ---- my slots: pass p owns slot (r, s = c + C p)
        const int nx_free = n > L::NX0 ? n - L::NX0 : 0;  // rows of W' that live in free slots of tile row TR - 1 (lanes r >= R - nx_free)
        int kind[NPASS], el[NPASS];
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            const int s = c + C * p;
            kind[p] = K_NONE;
            el[p] = 0;
            if (s < TR && R * s + r < m) {
                kind[p] = K_Z;
                el[p] = R * s + r;
            } else if (s == TR - 1 && R - 1 - r < nx_free) {
                kind[p] = K_X;
                el[p] = L::NX0 + (R - 1 - r);
            } else if (s >= TR && s < TS && R * (s - TR) + r < n && R * (s - TR) + r < L::NX0) {
                kind[p] = K_X;
                el[p] = R * (s - TR) + r;
            }
        }
        const T INF = T(1) / T(0);
        // per-slot constants in LDS (read once per iteration by the owner): lo, up, 1/rho of a z-slot; an x-slot (and an empty one) is
        // carried through the SAME update as a z-slot with lo = -inf, up = +inf, 1/rho = 0, y = 0 and "rho" = sigma: then
        //     zn = clip(zt + 0 y) = zt = x_new ,   y += sigma (zt - zn) = 0 ,   operand = sigma (zn - 0 y) - q
        // so the owners' code has no branch on the kind of slot.  q of an x-slot stays in a register (0 for the others).
        T *c0 = lds + L::O_C0 + R * c + r, *c1 = lds + L::O_C1 + R * c + r, *c2 = lds + L::O_C2 + R * c + r;  // + R C p for pass p
        T qx[NPASS];
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            qx[p] = kind[p] == K_X ? (T)gq[el[p]] : T(0);
            if (pass_has_z(p)) {
                c0[R * C * p] = kind[p] == K_Z ? (T)gl[el[p]] : -INF;
                c1[R * C * p] = kind[p] == K_Z ? (T)gu[el[p]] : INF;
                c2[R * C * p] = kind[p] == K_Z ? T(1) : T(0);
            }
        }
        // iterates of my slots: sa = z (z-slot) or x (x-slot), sb = y, srh = rho of the constraint
        T sa[NPASS], sb[NPASS], srh[NPASS];
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            sa[p] = sb[p] = 0;
            srh[p] = kind[p] == K_Z ? T(1) : (kind[p] == K_X ? (T)a.sigma : T(0));  // the multiplier of the published operand
        }

        bool rho_differs = false;
        if (mode & (MODE_SETUP | MODE_UPDATE)) {
            rho_s = a.rho0;
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                if (kind[p] == K_Z) {
                    const T lo = c0[R * C * p], up = c1[R * C * p];
                    int ctype = SQPH_INEQUALITY_CONSTRAINT;
                    if (lo < -a.loose_thresh && up > a.loose_thresh)
                        ctype = SQPH_LOOSE_BOUNDS;
                    else if (up - lo < a.eq_tol)
                        ctype = SQPH_EQUALITY_CONSTRAINT;
                    srh[p] = rho_for_type<T>(ctype, rho_s, a.rho_min, a.rho_eq_factor);
                    c2[R * C * p] = T(1) / srh[p];
                    rho_differs = rho_differs || !(srh[p] == srho[el[p]]);  // against the vector the resident factor was built with
                    sct[el[p]] = ctype;
                    srho[el[p]] = srh[p];
                }
            }
            info.rho_updates += 1;
        }
        if (!(mode & MODE_SETUP)) {
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                if (kind[p] == K_X) sa[p] = sx[el[p]];
                if (kind[p] == K_Z) {
                    sa[p] = sz[el[p]];
                    sb[p] = sy[el[p]];
                    if (!(mode & MODE_UPDATE)) {
                        srh[p] = srho[el[p]];
                        c2[R * C * p] = T(1) / srh[p];
                    }
                }
            }
        }

        T at[TR][TC];               // the A tile; turned into B = A W' in place once the factor is known
        T xt[XS > 0 ? XS : 1][TC];  // rows of W' in the tile rows behind B
        bool need_factor = (mode & (MODE_SETUP | MODE_UPDATE | MODE_REFACTOR)) != 0;
        if ((mode & MODE_SAME_MATRICES) && (mode & (MODE_SETUP | MODE_UPDATE)) && !(mode & MODE_REFACTOR) &&
            info.status != SQPH_NUMERICAL_ISSUES && info.status != SQPH_UNINITIALIZED) {
            // sqph_setup_solve_reuse: same P and A as the resident factor, which is also the one setup() would build if no slot's rho differs
            T *flag = lds + L::O_ROWV;
            __syncthreads();
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
        bool have_A = false;
        const T alpha = a.alpha, sigma = a.sigma, oma = T(1) - a.alpha;
        int iter = 1;
        int next_check = a.check_termination > 0 ? a.check_termination : -1;
        int next_adapt = (a.adaptive_rho && a.adaptive_rho_interval > 0) ? a.adaptive_rho_interval : -1;
        T *rowv = lds + L::O_ROWV + r * L::TSp + c;   // + C p: operand of my slot in pass p
        const T *st2o = lds + L::O_ST2 + c * L::SS2 + r * L::RS2;  // + C SS2 p: partial sums of my slot in pass p
#ifdef SQPH_PHASE_TIMING
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
        for (;;) {
            {  // ---- set-up part of a pass (the W tile is local to it)
            T wt[TW][TC];
            if (!need_factor) K0::template load_sq_tile<T>(gW, n, r, c, wt);  // solve() on a previously set-up instance
            if (need_factor) {
                __syncthreads();
                // rho of every constraint row for S = A' R A (zeros beyond m)
                for (int e = t; e < L::MP; e += L::NT) lds[L0::O_RHO + e] = T(0);
                __syncthreads();
#pragma unroll
                for (int p = 0; p < NPASS; p++)
                    if (kind[p] == K_Z) lds[L0::O_RHO + el[p]] = srh[p];
                __syncthreads();
                int n_f = n, m_f = m, r_f = r, c_f = c, t_f = t;
                const TIN *gA_f = gA, *gP_f = gP;
                SQPH_OPAQUE_S(n_f); SQPH_OPAQUE_S(m_f); SQPH_OPAQUE_V(r_f); SQPH_OPAQUE_V(c_f); SQPH_OPAQUE_V(t_f);
                SQPH_OPAQUE_S(gA_f); SQPH_OPAQUE_S(gP_f);
                K0::load_A_tile(gA_f, n_f, m_f, r_f, c_f, at);
                const bool ok = K0::factor(gP_f, at, n_f, m_f, sigma, lds, t_f, r_f, c_f, wt);
                if (!(mode & MODE_NO_FACTOR_STORE)) K0::store_sq_tile(gW, n_f, r_f, c_f, wt);
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
#pragma unroll
                    for (int p = 0; p < NPASS; p++) sa[p] = sb[p] = 0;
                }
            }
            if (!have_A) {
                int n_t = n, m_t = m, r_t = r, c_t = c;
                const TIN *gA_t = gA;
                SQPH_OPAQUE_S(n_t); SQPH_OPAQUE_S(m_t); SQPH_OPAQUE_V(r_t); SQPH_OPAQUE_V(c_t); SQPH_OPAQUE_S(gA_t);
                K0::load_A_tile(gA_t, n_t, m_t, r_t, c_t, at);
            }
            have_A = false;
            {
                int n_t = n, r_t = r, c_t = c;
                SQPH_OPAQUE_S(n_t); SQPH_OPAQUE_V(r_t); SQPH_OPAQUE_V(c_t);
                K0::build_B_inplace(at, wt, n_t, lds, r_t, c_t);
            }
            }  // ---- end of the set-up part
            {
                // rows of W' into their slots, from the transposed copy build_B staged in [0, NP * WSTR):  Wf[j][SLOT c + k] = W[TC c + k][j]
                __syncthreads();
#pragma unroll
                for (int u = 0; u < XS; u++) {
                    const int j = R * u + r;
                    T tmp[L0::SLOT];
                    wg_read<L0::SLOT>(lds + (j < L0::NP ? j : 0) * L0::WSTR + L0::SLOT * c, tmp);
#pragma unroll
                    for (int k = 0; k < TC; k++) xt[u][k] = (j < n && j < L::NX0 && TC * c + k < n) ? tmp[k] : T(0);
                }
                {   // the free slots of B's last tile row (rows R (TR - 1) + r >= m hold zeros there)
                    const int j = L::NX0 + (R - 1 - r);
                    const bool mine = R - 1 - r < nx_free;
                    T tmp[L0::SLOT];
                    wg_read<L0::SLOT>(lds + (mine && j < L0::NP ? j : 0) * L0::WSTR + L0::SLOT * c, tmp);
#pragma unroll
                    for (int k = 0; k < TC; k++) at[TR - 1][k] = (mine && TC * c + k < n) ? tmp[k] : at[TR - 1][k];
                }
                __syncthreads();
            }
            T (&bt)[TR][TC] = at;
            // operands of the first iteration: w = R (z - R^-1 y) [rhs tail of qp.cpp:275 pre-multiplied by R] | u = sigma x - q | 0
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                if (c + C * p < TS) {
                    T v = T(0);
                    if (kind[p] == K_Z) v = srh[p] * (sa[p] - c2[R * C * p] * sb[p]);
                    if (kind[p] == K_X) v = sigma * sa[p] - qx[p];
                    rowv[C * p] = v;
                }
            }
            int par = 0;
#ifdef SQPH_PHASE_TIMING
            tprev = __builtin_amdgcn_s_memtime();
#define SQPH_RTICK(k) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tprev; tprev = tn_; }
#else
#define SQPH_RTICK(k)
#endif
            while (iter <= a.max_iter) {
                int seg = a.max_iter - iter + 1;
                if constexpr (CHECKS) {
                    if (next_check > 0 && next_check < seg) seg = next_check;
                    if (next_adapt > 0 && next_adapt < seg) seg = next_adapt;
                }
                for (int seg_i = 0; seg_i < seg; seg_i++) {
                    wave_sync();  // my wave's operands are published (LDS operations of a wave execute in program order)
                    {
                        T v[TS];
                        wg_read<TS>((SQPH_WGR_WHATIF & 1) ? lds + L::O_ST2 + 2 * (t & 63) : lds + L::O_ROWV + r * L::TSp, v);
                        stage1(bt, xt, v, lds + L::O_ST1 + par * L::S1, r, c);
                    }
                    SQPH_RTICK(0)
                    __syncthreads();  // the one barrier of the iteration: stage-1 partials of both waves are in place
                    SQPH_RTICK(1)
                    // y1 = sum over r; every wave reduces the columns it consumes into its private copy
#pragma unroll
                    for (int q = 0; q < (TC + RW - 1) / RW; q++) {
                        const int kq = rl + RW * q;
                        if (kq < TC) lds[(SQPH_WGR_WHATIF & 8) ? L::O_Y1 + (t & 63) : L::O_Y1 + (wv * C + c) * L::TCp + kq] = wg_sum<R>((SQPH_WGR_WHATIF & 4) ? lds + L::O_ST1 + 2 * (t & 63) : lds + L::O_ST1 + par * L::S1 + c * L::CS1 + kq * L::KS1);
                    }
                    par ^= 1;
                    wave_sync();
                    SQPH_RTICK(2)
                    {
                        T y1c[TC];
                        wg_read<TC>((SQPH_WGR_WHATIF & 8) ? lds + L::O_ST2 + 2 * (t & 63) : lds + L::O_Y1 + (wv * C + c) * L::TCp, y1c);
                        stage2(bt, xt, y1c, lds + L::O_ST2, r, c);
                    }
                    SQPH_RTICK(3)
                    // the owners' constants do not depend on the partial sums: fetched ahead of them
                    T k0[NPASS], k1[NPASS], k2[NPASS];
#pragma unroll
                    for (int p = 0; p < NPASS; p++) {
                        if (pass_has_z(p)) {
                            k0[p] = c0[R * C * p];
                            k1[p] = c1[R * C * p];
                            k2[p] = c2[R * C * p];
                        }
                    }
                    wave_sync();
                    // sums over c of my slots' partials (z~ of a z-slot, x~ of an x-slot), all passes in flight together
                    T zt[NPASS];
#pragma unroll
                    for (int p = 0; p < NPASS; p++)
                        zt[p] = wg_sum<C>((SQPH_WGR_WHATIF & 32) ? lds + L::O_ST2 + 2 * (t & 63) + 128 * p : st2o + C * L::SS2 * p);
#pragma unroll
                    for (int p = 0; p < NPASS; p++) {
                        const T zr = alpha * zt[p] + oma * sa[p];
                        if (pass_has_z(p)) {  // z-, x- and empty slots through one branch-free update (see the constants above)
                            T zn = zr + k2[p] * sb[p];
                            zn = zn < k0[p] ? k0[p] : zn;  // cwiseMax(l) then cwiseMin(u), qp.cpp:278-281
                            zn = zn > k1[p] ? k1[p] : zn;
                            sb[p] = sb[p] + srh[p] * (zr - zn);
                            sa[p] = zn;
                            rowv[(SQPH_WGR_WHATIF & 64) ? (int)(t & 63) - r * L::TSp - c : C * p] = srh[p] * (zn - k2[p] * sb[p]) - qx[p];
                        } else {              // a pass of x-slots only
                            sa[p] = zr;
                            rowv[(SQPH_WGR_WHATIF & 64) ? (int)(t & 63) - r * L::TSp - c : C * p] = srh[p] * zr - qx[p];
                        }
                    }
                    SQPH_RTICK(4)
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
                    if (check || adapt) {
                        // update_state + residuals, qp.cpp:316-331, 353-361, on the column-split kernel's staging map (L0): A and P are
                        // streamed from global memory (the register tiles hold B and W')
                        __syncthreads();
#pragma unroll
                        for (int p = 0; p < NPASS; p++) {
                            const int s = c + C * p;
                            if (s < TR) K0::put_rowv(lds, r, s, kind[p] == K_Z ? sb[p] : T(0));   // y in row-gather order
                            if (kind[p] == K_X) K0::put_colv(lds, el[p], sa[p]);                  // x in column-gather order
                        }
                        for (int e = n + t; e < L0::NP; e += L::NT) K0::put_colv(lds, e, T(0));
                        __syncthreads();
                        {
                            T yr[TR];
                            K0::get_rowv(lds, r, yr);
                            int n_c = n, m_c = m, r_c = r, c_c = c;
                            const TIN *gA_c = m > 0 ? gA : gP;
                            SQPH_OPAQUE_S(n_c); SQPH_OPAQUE_S(m_c); SQPH_OPAQUE_V(r_c); SQPH_OPAQUE_V(c_c); SQPH_OPAQUE_S(gA_c);
                            K0::stage_A_AT_gmem(gA_c, n_c, m_c, r_c, c_c, yr, lds);  // A x (over c) and A' y (over r)
                        }
                        __syncthreads();
                        T prod[NPASS];  // A x of a z-slot, A' y of an x-slot
#pragma unroll
                        for (int p = 0; p < NPASS; p++) {
                            prod[p] = T(0);
                            if (kind[p] == K_Z) prod[p] = K0::reduce_over_c(lds, el[p]);
                            if (kind[p] == K_X) prod[p] = K0::reduce_over_r(lds, el[p]);
                        }
                        __syncthreads();
                        {
                            int n_c = n, r_c = r, c_c = c;
                            const TIN *gP_c = gP;
                            SQPH_OPAQUE_S(n_c); SQPH_OPAQUE_V(r_c); SQPH_OPAQUE_V(c_c); SQPH_OPAQUE_S(gP_c);
                            K0::stage_P_gmem(gP_c, n_c, r_c, c_c, lds);  // full P (both triangles), as qp.cpp:324
                        }
                        __syncthreads();
                        T v[4] = {0, 0, 0, 0};  // nrm_prim | res_prim | nrm_dual | res_dual
#pragma unroll
                        for (int p = 0; p < NPASS; p++) {
                            if (kind[p] == K_Z) {
                                v[0] = nanmax(v[0], nanmax(tabs(prod[p]), tabs(sa[p])));
                                v[1] = nanmax(v[1], tabs(prod[p] - sa[p]));
                            }
                            if (kind[p] == K_X) {
                                const T Px = K0::reduce_over_c(lds, el[p]);
                                const T q = qx[p];
                                v[2] = nanmax(v[2], nanmax(tabs(Px), nanmax(tabs(prod[p]), tabs(q))));
                                v[3] = nanmax(v[3], tabs(Px + q + prod[p]));
                            }
                        }
                        __syncthreads();
                        {   // workgroup-wide NaN-propagating max: butterfly inside each wave, NW values through LDS
                            T *red = lds + L0::O_RED;
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
                        bool leave = false;
                        if (check) {
                            if (v[1] <= a.eps_abs + a.eps_rel * nrm_prim && v[3] <= a.eps_abs + a.eps_rel * nrm_dual) {
                                info.status = SQPH_SOLVED;
                                iter--;  // the iteration the test passed at (the segment loop has already counted past it)
                                leave = true;
                            }
                        }
                        if (adapt && !leave) {
                            const T eps = a.regul;
                            const T rp_norm = v[1] / (nrm_prim + eps);
                            const T rd_norm = v[3] / (nrm_dual + eps);
                            T new_rho = rho_s * (T)sqrt((double)(rp_norm / (rd_norm + eps)));
                            new_rho = new_rho < a.rho_max ? new_rho : a.rho_max;
                            new_rho = new_rho > a.rho_min ? new_rho : a.rho_min;
                            info.rho_estimate = (double)new_rho;
                            if (new_rho < rho_s / a.rho_tol || new_rho > rho_s * a.rho_tol) {
                                rho_s = new_rho;
#pragma unroll
                                for (int p = 0; p < NPASS; p++) {
                                    if (kind[p] == K_Z) {
                                        srh[p] = rho_for_type<T>(sct[el[p]], rho_s, a.rho_min, a.rho_eq_factor);  // type re-read from the state array (rare)
                                        c2[R * C * p] = T(1) / srh[p];
                                    }
                                }
                                info.rho_updates += 1;
                                need_factor = true;
                                iter--;
                                leave = true;  // leave the iteration loop WITHOUT advancing iter; the factor block does it
                            }
                        }
                        if (leave) break;
                        // the check used the staging map of the column-split kernel: publish every operand again
                        __syncthreads();
#pragma unroll
                        for (int p = 0; p < NPASS; p++) {
                            if (c + C * p < TS) {
                                T vv = T(0);
                                if (kind[p] == K_Z) vv = srh[p] * (sa[p] - c2[R * C * p] * sb[p]);
                                if (kind[p] == K_X) vv = sigma * sa[p] - qx[p];
                                rowv[C * p] = vv;
                            }
                        }
                        __syncthreads();
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
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                if (kind[p] == K_X) sx[el[p]] = sa[p];
                if (kind[p] == K_Z) {
                    sz[el[p]] = sa[p];
                    sy[el[p]] = sb[p];
                    srho[el[p]] = srh[p];
                }
            }
        }
#ifdef SQPH_PHASE_TIMING
        __syncthreads();
        if (t < 8) sx[t] = (T)tacc[t];            // debug build only: wave 0's phase ticks in x[0..8), wave 1's in y[64..72)
        if (t >= 64 && t < 72) sy[t] = (T)tacc[t - 64];
#endif
        if (t == 0) {
            a.info[qp] = info;
            a.rho[qp] = rho_s;
        }
    }
};

template <typename TIN, int NW, int R, int C, int TR, int TC, int TW, int TS, int WPE>
__global__ __launch_bounds__(64 * NW, WPE) void admm_wgr_kernel(KArgs<double, TIN> a) {
    __shared__ __attribute__((aligned(16))) double lds[WgrLayout<NW, R, C, TR, TC, TW, TS>::TOTAL];
#ifdef SQPH_SIM
    ::sqph_sim::poison_static_lds(lds, sizeof(lds));
#endif
    WgrKernel<TIN, NW, R, C, TR, TC, TW, TS>::template run<true>(a, lds);
}
template <typename TIN, int NW, int R, int C, int TR, int TC, int TW, int TS, int WPE>
__global__ __launch_bounds__(64 * NW, WPE) void admm_wgr_nocheck_kernel(KArgs<double, TIN> a) {
    __shared__ __attribute__((aligned(16))) double lds[WgrLayout<NW, R, C, TR, TC, TW, TS>::TOTAL];
#ifdef SQPH_SIM
    ::sqph_sim::poison_static_lds(lds, sizeof(lds));
#endif
    WgrKernel<TIN, NW, R, C, TR, TC, TW, TS>::template run<false>(a, lds);
}

// shapes {NW, R, C, TR, TC, TW, TS, WPE}: m <= R TR, n <= C TC, n <= R TW, m + n <= R TS; first fit wins.  Problems that fit a
// one-wavefront shape of admm_wg_kernel.h (m <= 64, n <= 32) stay there.
#define SQPH_WGR_SHAPES(X) X(2, 16, 8, 7, 7, 4, 10, 2)
#define SQPH_WGR_FITS(a, R_, C_, TR_, TC_, TW_, TS_) \
    ((a).m <= R_ * TR_ && (a).n <= C_ * TC_ && (a).n <= R_ * TW_ && (a).m + (a).n <= R_ * TS_ && ((a).m > 64 || (a).n > 32))

#ifdef SQPH_SIM
template <typename TIN>
inline int sim_run_wgr(const KArgs<double, TIN> &a) {
#define SQPH_SIM_CASE(NW_, R_, C_, TR_, TC_, TW_, TS_, W_)                                                                            \
    if (SQPH_WGR_FITS(a, R_, C_, TR_, TC_, TW_, TS_)) {                                                                              \
        if (a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0))                                             \
            ::sqph_sim::launch(admm_wgr_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, TS_, W_>, dim3(a.batch), dim3(64 * NW_), 0, a); \
        else                                                                                                                          \
            ::sqph_sim::launch(admm_wgr_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, TS_, W_>, dim3(a.batch), dim3(64 * NW_), 0, a);        \
        return 0;                                                                                                                     \
    }
    SQPH_WGR_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

}  // namespace sqph
