// Lane-per-QP ADMM kernel for tiny problems (n <= NMAX, m <= MMAX; the SQP driver's subproblems, BASELINE config 4: n = 2, m = 3).
//
// For a 2 x 3 QP the workgroup- and group-tiled kernels spend their time on LDS round trips and wave-level hand-offs that a problem
// of 17 numbers does not need: here ONE LANE owns one QP — P, A, the factor W and every vector live in that lane's registers, there
// is no LDS, no cross-lane operation and no barrier; 64 QPs per wavefront, lanes diverge freely (different iteration counts,
// refactorisations).  A solve that is latency-bound at ~1 us per iteration in the 16-lanes-per-QP kernel runs at the FMA dependency
// chain of a handful of operations instead.
//
// Numerics: the formulas of admm_generic.h (reference src/qp.cpp:84-144 on the Schur-ordered system, factor W of factor_schur:
// S = P_sym + sigma I + A'RA = D_J^1/2 (L D L') D_J^1/2, W = D^-1/2 L^-1 D_J^-1/2, x~ = W'(W (sigma x - q + A'w)), z~ = A x~), fp64
// arithmetic, TIN inputs.  Modes, status / iteration bookkeeping and the factor's layout in the workspace are those of the other
// kernels (kargs.h), so setup() with this kernel and solve() with another — or the reverse — compose.
#pragma once
#include "block_ops.h"
#include "kargs.h"
#include "wave_ops.h"

namespace sqph {

// products of the iteration: fused multiply-add, like every other kernel of the library (round 3; -DSQPH_LANE_NO_FMA gives separate
// multiplies and adds back — NOT the exact statement order of rounds 1-2: the A'w accumulation keeps its two interleaved chains,
// see the iteration below — for callers who want the oracle's unfused products).  Measured on the MI355X:
// 0.083 -> 0.062 ms per 65,536 x 200 iterations, 72 -> 55 us per SQP-style launch, BatchSQP 19.9 -> 17.8 ms per 1,024 SimpleNLP
// instances; iterates move by ~1e-8 relative on some adaptive-rho QPs (inside the 1e-6 bar).  The SQP parity suite
// (tests/cpp/sqp_batch_test.cpp) is unchanged by it: 936 / 869 / 149 / 231 strict instances against 937 / 874 / 150 / 231 before,
// ~50,000 subproblems re-solved by the oracle on identical inputs, 0 mismatches (profiles/r03_sqp_parity_log.txt).
#ifndef SQPH_LANE_NO_FMA
// in the arithmetic type itself (the true-fp32 instantiation through doubles cost three conversions per product: it ran slower than fp64)
__device__ __forceinline__ double lane_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float lane_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#define LFMA(a, b, c) lane_fma((T)(a), (T)(b), (T)(c))
#else
#define LFMA(a, b, c) ((a) * (b) + (c))
#endif

// An ADMM iteration of a QP this small is ONE chain of dependent fp64 operations (w -> A'w -> W -> W' -> A x~ -> relax -> clip -> y), and a
// lone wavefront pays ~25 cycles per link (measured, tools/xp/quad_lat.sh): the three helpers below are the same formulas with fewer
// links (round 4: 20 -> 17 per iteration).  -DSQPH_LANE_NO_FMA keeps the unfused forms of these three helpers.
// w = R (z - R^-1 y), the rhs tail of qp.cpp:275 pre-multiplied by R:  rho z - y in one fused operation
template <typename T>
__device__ __forceinline__ T lane_w(T rho, T rinv, T y, T z) {
#ifndef SQPH_LANE_NO_FMA
    (void)rinv;
    return lane_fma(rho, z, -y);
#else
    return rho * (z - rinv * y);
#endif
}
// z = min(max(zr + R^-1 y, l), u) (cwiseMax(l) then cwiseMin(u), qp.cpp:278-281), y += R (zr - z) (qp.cpp:103).  lmin = min(l, u):
// what "max with l, then min with u" returns below l — with it the two comparisons no longer depend on each other (identical results,
// NaN included)
template <typename T>
__device__ __forceinline__ void lane_tail(T zr, T rho, T rinv, T l, T lmin, T u, T &y, T &z) {
    const T zn0 = LFMA(rinv, y, zr);
#ifndef SQPH_LANE_NO_FMA
    const T hi = zn0 > u ? u : zn0;
    const T zn = zn0 < l ? lmin : hi;
    y = lane_fma(rho, zr - zn, y);  // (not (y + rho zr) - rho z: a row that never binds must keep y == 0 exactly)
#else
    T zn = zn0 < l ? l : zn0;
    zn = zn > u ? u : zn;
    (void)lmin;
    y = y + rho * (zr - zn);
#endif
    z = zn;
}


// EXACT: n == NMAX and m == MMAX are compile-time constants (every bound check folds away; the SQP driver's shapes);
// otherwise the arrays are padded to NMAX x MMAX and the run-time n, m guard every row and column.
// TA = arithmetic type: double (default: fp64 whatever the interface Scalar is) or float (SQPH_FLAG_F32_ARITH with a float
// interface: a true single-precision solve — iterates, factor and residuals in fp32; the resident state arrays stay fp64 and are
// converted at the kernel's boundary).
// LPQ = lanes per QP: 1 (throughput: 64 QPs per wavefront), or 4 — the QUAD variant for small batches (the SQP driver's 1,024
// instances are 16 wavefronts on a chip of 1,024 SIMDs; round 4, asked for by the round-3 review on the premise that a call takes
// as long as the instruction stream of ONE lane's iterations, ~110 instructions per ADMM iteration — see SQPH_QUAD_MAX_BATCH below for
// what the measurement said).  The four lanes of a quad hold
// the same QP and run set-up, factorisation and the residual checks redundantly; in the iteration lane k of the quad owns constraint
// row k (m <= 4): it forms w_k, its row's contribution to A'w (summed over the quad by two DPP quad_perm exchanges per entry) and its
// row of z~, z, y — ~45 instructions per iteration.  Lane 0 of the quad writes the results.
template <typename TA, typename TIN, int NMAX, int MMAX, bool EXACT, int LPQ = 1>
struct LaneKernel {
    using T = TA;
    static constexpr int MM = MMAX > 0 ? MMAX : 1;
    static_assert(LPQ == 1 || (LPQ == 4 && MMAX <= 4), "the quad variant gives every constraint row a lane");
    // element k of a small register array by a chain of selects (k is a lane index)
    template <int N>
    static __device__ __forceinline__ T pick(const T (&v)[N], int k, T pad) {
        T r = pad;
#pragma unroll
        for (int i = 0; i < N; i++) r = (k == i) ? v[i] : r;
        return r;
    }
    // every lane of the quad gets the m-vector whose element k lane k holds
    static __device__ __forceinline__ void quad_gather(T vk, T (&v)[MM]) {
        if constexpr (MM > 0) v[0] = quad_bcast<0>(vk);
        if constexpr (MM > 1) v[1] = quad_bcast<1>(vk);
        if constexpr (MM > 2) v[2] = quad_bcast<2>(vk);
        if constexpr (MM > 3) v[3] = quad_bcast<3>(vk);
    }

    // S = P_sym + sigma I + A' diag(rho) A ; Jacobi scaling ; forward elimination of [S~ | I] ; W = D^-1/2 L^-1 D_J^-1/2.
    // false on a non-positive / non-finite pivot (this QP only).
    static __device__ __forceinline__ bool factor(const T (&P)[NMAX][NMAX], const T (&A)[MM][NMAX], const T (&rho)[MM], int n, int m, T sigma,
                                                  T (&W)[NMAX][NMAX]) {
        T S[NMAX][NMAX], sj[NMAX], dsv[NMAX];
#pragma unroll
        for (int i = 0; i < NMAX; i++)
#pragma unroll
            for (int j = 0; j < NMAX; j++) {
                S[i][j] = 0;
                W[i][j] = 0;
            }
#pragma unroll
        for (int i = 0; i < NMAX; i++) {
#pragma unroll
            for (int j = 0; j <= i; j++) {
                if (i < n) {
                    // only the lower triangle of P reaches the reference's factor (Eigen::LDLT<.,Lower>, qp.hpp:129)
                    const T acc = P[i][j] + (i == j ? sigma : T(0));
                    T s = 0;
#pragma unroll
                    for (int k = 0; k < MMAX; k++)
                        if (k < m) s += A[k][i] * rho[k] * A[k][j];
                    S[i][j] = acc + s;
                    S[j][i] = S[i][j];
                }
            }
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NMAX; j++) {
            sj[j] = T(1);
            dsv[j] = T(1);
            if (j < n) {
                const T d = S[j][j];
                if (!(d > T(0)) || !(d * T(0) == T(0))) ok = false;
                sj[j] = T(1) / (T)sqrt((double)d);
            }
        }
        if (!ok) return false;
#pragma unroll
        for (int i = 0; i < NMAX; i++)
#pragma unroll
            for (int j = 0; j < NMAX; j++) S[i][j] = S[i][j] * sj[i] * sj[j];
#pragma unroll
        for (int k = 0; k < NMAX; k++) {
            if (k < n && ok) {
                const T d = S[k][k];
                if (!(d > T(0)) || !(d * T(0) == T(0))) {
                    ok = false;
                } else {
                    const T dinv = T(1) / d;
                    dsv[k] = d;
                    T row[NMAX];
#pragma unroll
                    for (int j = 0; j < NMAX; j++) row[j] = S[k][j];
#pragma unroll
                    for (int i = 0; i < NMAX; i++) {
                        if (i > k && i < n) {
                            const T f = row[i] * dinv;
#pragma unroll
                            for (int j = 0; j < NMAX; j++) {
                                const T g = (j == k) ? d + T(1) : row[j];
                                S[i][j] = S[i][j] - f * g;
                            }
                        }
                    }
                }
            }
        }
        if (!ok) return false;
#pragma unroll
        for (int i = 0; i < NMAX; i++) {
            const T rs = T(1) / (T)sqrt((double)dsv[i]);
#pragma unroll
            for (int j = 0; j < NMAX; j++) {
                const T v = i > j ? S[i][j] * rs : (i == j ? rs : T(0));
                W[i][j] = (i < n && j < n) ? v * sj[j] : T(0);
            }
        }
        return true;
    }

    static __device__ __forceinline__ void run(const KArgs<double, TIN> &a) {
        const int gt = blockIdx.x * blockDim.x + threadIdx.x;
        const int qp = gt / LPQ, kq = gt % LPQ;  // QP and (quad variant) my lane of its quad
        const bool writer = LPQ == 1 || kq == 0;
        if (qp >= a.batch) return;
        const int n = EXACT ? NMAX : a.n, m = EXACT ? MMAX : a.m;
        const TIN *gP = a.P + (long)qp * a.sP;
        const TIN *gq = a.q + (long)qp * a.sq;
        const TIN *gA = a.A + (long)qp * a.sA;
        const TIN *gl = a.l + (long)qp * a.sl;
        const TIN *gu = a.u + (long)qp * a.su;
        double *sx = a.x + (long)qp * n;
        double *sz = a.z + (long)qp * m;
        double *sy = a.y + (long)qp * m;
        double *srho = a.rho_vec + (long)qp * m;
        int *sct = a.ctype + (long)qp * m;
        double *gW = a.Sinv + (long)qp * 2 * n * n;

        sqph_info info = a.info[qp];
        T rho_s = (T)a.rho[qp];
        // settings and class constants in the arithmetic type
        const T a_rho0 = (T)a.rho0, a_loose = (T)a.loose_thresh, a_eq_tol = (T)a.eq_tol, a_rho_min = (T)a.rho_min, a_rho_max = (T)a.rho_max;
        const T a_eqf = (T)a.rho_eq_factor, a_eps_abs = (T)a.eps_abs, a_eps_rel = (T)a.eps_rel, a_regul = (T)a.regul, a_rho_tol = (T)a.rho_tol;
        const int mode = a.mode;
        if (!(mode & (MODE_SETUP | MODE_UPDATE)) && (info.status == SQPH_UNINITIALIZED || info.status == SQPH_NUMERICAL_ISSUES))
            return;  // qp.cpp:68-71

        const T INF = T(1) / T(0);
        T P[NMAX][NMAX], A[MM][NMAX], W[NMAX][NMAX];
        T q[NMAX], x[NMAX], l[MM], u[MM], lmin[MM], z[MM], y[MM], rho[MM], rinv[MM];
        int ct[MM];
#pragma unroll
        for (int j = 0; j < NMAX; j++) {
            q[j] = j < n ? (T)gq[j] : T(0);
            x[j] = 0;
#pragma unroll
            for (int i = 0; i < NMAX; i++) P[i][j] = (i < n && j < n) ? (T)gP[(long)j * n + i] : T(0);  // column-major
#pragma unroll
            for (int i = 0; i < MMAX; i++) A[i][j] = (i < m && j < n) ? (T)gA[(long)j * m + i] : T(0);
        }
#pragma unroll
        for (int i = 0; i < MMAX; i++) {
            l[i] = i < m ? (T)gl[i] : -INF;
            u[i] = i < m ? (T)gu[i] : INF;
            lmin[i] = l[i] > u[i] ? u[i] : l[i];
            z[i] = y[i] = 0;
            rho[i] = rinv[i] = T(1);
            ct[i] = SQPH_INEQUALITY_CONSTRAINT;
        }
        // the reference's factor reads the lower triangle only: mirror it so that S is built from P_sym; the residual check uses
        // the full P as given (qp.cpp:324)
        T Pl[NMAX][NMAX];
#pragma unroll
        for (int i = 0; i < NMAX; i++)
#pragma unroll
            for (int j = 0; j < NMAX; j++) Pl[i][j] = i >= j ? P[i][j] : P[j][i];

        // the resident factor was built with exactly this rho vector — and is a valid factor: after a set-up that ended in
        // NUMERICAL_ISSUES (or on an UNINITIALIZED instance) the reference's re-solve runs setup() again (sqp.cpp:274 -> 221-229)
        bool rho_unchanged = (mode & MODE_SAME_MATRICES) != 0 && info.status != SQPH_NUMERICAL_ISSUES && info.status != SQPH_UNINITIALIZED;
        if (mode & (MODE_SETUP | MODE_UPDATE)) {
            rho_s = a_rho0;
#pragma unroll
            for (int i = 0; i < MMAX; i++) {
                if (i < m) {
                    int c = SQPH_INEQUALITY_CONSTRAINT;
                    if (l[i] < -a_loose && u[i] > a_loose)
                        c = SQPH_LOOSE_BOUNDS;
                    else if (u[i] - l[i] < a_eq_tol)
                        c = SQPH_EQUALITY_CONSTRAINT;
                    ct[i] = c;
                    rho[i] = rho_for_type<T>(c, rho_s, a_rho_min, a_eqf);
                    rinv[i] = T(1) / rho[i];
                    if (rho_unchanged && !((double)rho[i] == srho[i])) rho_unchanged = false;
                }
            }
            if constexpr (LPQ > 1) quad_sync();  // every lane of the quad has compared before lane 0 overwrites
            if (writer) {
#pragma unroll
                for (int i = 0; i < MMAX; i++) {
                    if (i < m) {
                        sct[i] = ct[i];
                        srho[i] = (double)rho[i];
                    }
                }
            }
            info.rho_updates += 1;
            if (!(mode & MODE_SETUP)) {
#pragma unroll
                for (int j = 0; j < NMAX; j++)
                    if (j < n) x[j] = (T)sx[j];
#pragma unroll
                for (int i = 0; i < MMAX; i++)
                    if (i < m) {
                        z[i] = (T)sz[i];
                        y[i] = (T)sy[i];
                    }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NMAX; j++)
                if (j < n) x[j] = (T)sx[j];
#pragma unroll
            for (int i = 0; i < MMAX; i++)
                if (i < m) {
                    z[i] = (T)sz[i];
                    y[i] = (T)sy[i];
                    rho[i] = (T)srho[i];
                    rinv[i] = T(1) / rho[i];
                    ct[i] = sct[i];
                }
        }

        bool need_factor = (mode & (MODE_SETUP | MODE_UPDATE | MODE_REFACTOR)) != 0;
        if ((mode & (MODE_SETUP | MODE_UPDATE)) && !(mode & MODE_REFACTOR) && rho_unchanged) {
            need_factor = false;  // sqph_setup_solve_reuse: same P, A, rho vector => the resident factor is the one setup() would build
            info.status = SQPH_UNSOLVED;
        }
        bool solving = false;
        bool state_dirty = (mode & MODE_SETUP) != 0;
        const T alpha = (T)a.alpha, sigma = (T)a.sigma, oma = T(1) - (T)a.alpha;
        int iter = 1;
        int next_check = a.check_termination > 0 ? a.check_termination : -1;
        int next_adapt = (a.adaptive_rho && a.adaptive_rho_interval > 0) ? a.adaptive_rho_interval : -1;
        if (!need_factor) {
#pragma unroll
            for (int i = 0; i < NMAX; i++)
#pragma unroll
                for (int j = 0; j < NMAX; j++) W[i][j] = (i < n && j < n) ? (T)gW[(long)j * n + i] : T(0);
        }
        for (;;) {
            if (need_factor) {
                const bool ok = factor(Pl, A, rho, n, m, sigma, W);
                if (!(mode & MODE_NO_FACTOR_STORE) && writer) {
#pragma unroll
                    for (int i = 0; i < NMAX; i++)
#pragma unroll
                        for (int j = 0; j < NMAX; j++)
                            if (i < n && j < n) gW[(long)j * n + i] = (double)W[i][j];
                }
                need_factor = false;
                if (!solving) {
                    if (mode & (MODE_SETUP | MODE_UPDATE)) info.status = ok ? SQPH_UNSOLVED : SQPH_NUMERICAL_ISSUES;  // qp.cpp:39-43, 57-61
                    else if (!ok) info.status = SQPH_NUMERICAL_ISSUES;
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
                    for (int j = 0; j < NMAX; j++) x[j] = 0;
#pragma unroll
                    for (int i = 0; i < MMAX; i++) z[i] = y[i] = 0;
                }
            }
            // quad variant: my constraint row (lanes beyond m carry a padding row: A = 0, rho = 1, no bounds)
            T Ak[NMAX], rk = T(1), rik = T(1), lk = -INF, lmk = -INF, uk = INF, zk = 0, yk = 0;
            if constexpr (LPQ > 1) {
#pragma unroll
                for (int j = 0; j < NMAX; j++) {
                    T col[MM];
#pragma unroll
                    for (int i = 0; i < MM; i++) col[i] = A[i][j];
                    Ak[j] = pick<MM>(col, kq, T(0));
                }
                rk = pick<MM>(rho, kq, T(1));
                rik = pick<MM>(rinv, kq, T(1));
                lk = pick<MM>(l, kq, -INF);
                lmk = pick<MM>(lmin, kq, -INF);
                uk = pick<MM>(u, kq, INF);
                zk = pick<MM>(z, kq, T(0));
                yk = pick<MM>(y, kq, T(0));
            }
            for (; iter <= a.max_iter; iter++) {
                T b[NMAX], t[NMAX], xt[NMAX];
                if constexpr (LPQ > 1) {
                    const T w = lane_w<T>(rk, rik, yk, zk);
#pragma unroll
                    for (int j = 0; j < NMAX; j++) {
                        T cj = Ak[j] * w;
                        cj += quad_xor<1>(cj);
                        cj += quad_xor<2>(cj);
                        b[j] = LFMA(sigma, x[j], -q[j]) + cj;
                    }
                } else {
                // b = sigma x - q + A'w as two interleaved accumulation chains (even rows onto sigma x - q, odd rows apart)
                T b1[NMAX];
#pragma unroll
                for (int j = 0; j < NMAX; j++) {
                    b[j] = LFMA(sigma, x[j], -q[j]);
                    b1[j] = 0;
                }
#pragma unroll
                for (int i = 0; i < MMAX; i++) {
                    const T w = lane_w<T>(rho[i], rinv[i], y[i], z[i]);
#pragma unroll
                    for (int j = 0; j < NMAX; j++) {
                        if (i & 1) b1[j] = (i == 1) ? A[i][j] * w : LFMA(A[i][j], w, b1[j]);
                        else b[j] = LFMA(A[i][j], w, b[j]);
                    }
                }
                if constexpr (MMAX > 1) {
#pragma unroll
                    for (int j = 0; j < NMAX; j++) b[j] = b[j] + b1[j];
                }
                }
#pragma unroll
                for (int i = 0; i < NMAX; i++) {
                    T s = 0;
#pragma unroll
                    for (int j = 0; j <= i; j++) s = LFMA(W[i][j], b[j], s);
                    t[i] = s;
                }
#pragma unroll
                for (int j = 0; j < NMAX; j++) {
                    T s = 0;
#pragma unroll
                    for (int i = j; i < NMAX; i++) s = LFMA(W[i][j], t[i], s);
                    xt[j] = s;
                }
#pragma unroll
                for (int j = 0; j < NMAX; j++) x[j] = LFMA(alpha, xt[j], oma * x[j]);
                if constexpr (LPQ > 1) {
                    T zt = 0;
#pragma unroll
                    for (int j = 0; j < NMAX; j++) zt = LFMA(Ak[j], xt[j], zt);
                    const T zr = LFMA(alpha, zt, oma * zk);
                    lane_tail<T>(zr, rk, rik, lk, lmk, uk, yk, zk);
                } else {
#pragma unroll
                for (int i = 0; i < MMAX; i++) {
                    T zt = 0;
#pragma unroll
                    for (int j = 0; j < NMAX; j++) zt = LFMA(A[i][j], xt[j], zt);
                    const T zr = LFMA(alpha, zt, oma * z[i]);
                    lane_tail<T>(zr, rho[i], rinv[i], l[i], lmin[i], u[i], y[i], z[i]);
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
                    if constexpr (LPQ > 1) {  // the checks run on the whole vectors, in every lane of the quad
                        quad_gather(zk, z);
                        quad_gather(yk, y);
                    }
                    // update_state + residuals, qp.cpp:316-331, 353-361
                    T v[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int i = 0; i < MMAX; i++) {
                        if (i < m) {
                            T Ax = 0;
#pragma unroll
                            for (int j = 0; j < NMAX; j++) Ax += A[i][j] * x[j];
                            v[0] = nanmax(v[0], tabs(Ax));
                            v[1] = nanmax(v[1], tabs(z[i]));
                            v[2] = nanmax(v[2], tabs(Ax - z[i]));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NMAX; j++) {
                        if (j < n) {
                            T Px = 0, ATy = 0;
#pragma unroll
                            for (int k = 0; k < NMAX; k++) Px += P[j][k] * x[k];
#pragma unroll
                            for (int i = 0; i < MMAX; i++) ATy += A[i][j] * y[i];
                            v[3] = nanmax(v[3], tabs(Px));
                            v[4] = nanmax(v[4], tabs(ATy));
                            v[5] = nanmax(v[5], tabs(q[j]));
                            v[6] = nanmax(v[6], tabs(Px + q[j] + ATy));
                        }
                    }
                    const T nrm_prim = nanmax(v[0], v[1]);
                    const T nrm_dual = nanmax(v[3], nanmax(v[4], v[5]));
                    info.res_prim = (double)v[2];
                    info.res_dual = (double)v[6];
                    if (check && a.trace && qp == a.trace_qp && writer) {  // print_status, qp.cpp:373-383 (recorded; the host prints)
                        T obj = 0;
#pragma unroll
                        for (int j = 0; j < NMAX; j++) {
                            T Px = 0;
#pragma unroll
                            for (int k = 0; k < NMAX; k++) Px += P[j][k] * x[k];
                            obj += x[j] * (T(0.5) * Px + q[j]);
                        }
                        const int k = (int)a.trace[0];
                        if (k < a.trace_cap) {
                            a.trace[1 + 4 * k] = (double)iter;
                            a.trace[2 + 4 * k] = (double)obj;
                            a.trace[3 + 4 * k] = (double)v[2];
                            a.trace[4 + 4 * k] = (double)v[6];
                            a.trace[0] = (double)(k + 1);
                        }
                    }
                    if (check) {
                        if (v[2] <= a_eps_abs + a_eps_rel * nrm_prim && v[6] <= a_eps_abs + a_eps_rel * nrm_dual) {
                            info.status = SQPH_SOLVED;
                            break;
                        }
                    }
                    if (adapt) {
                        const T eps = a_regul;
                        const T rp_norm = v[2] / (nrm_prim + eps);
                        const T rd_norm = v[6] / (nrm_dual + eps);
                        T new_rho = rho_s * (T)sqrt((double)(rp_norm / (rd_norm + eps)));
                        new_rho = new_rho < a_rho_max ? new_rho : a_rho_max;
                        new_rho = new_rho > a_rho_min ? new_rho : a_rho_min;
                        info.rho_estimate = (double)new_rho;
                        if (new_rho < rho_s / a_rho_tol || new_rho > rho_s * a_rho_tol) {
                            rho_s = new_rho;
#pragma unroll
                            for (int i = 0; i < MMAX; i++) {
                                rho[i] = rho_for_type<T>(ct[i], rho_s, a_rho_min, a_eqf);
                                rinv[i] = T(1) / rho[i];
                            }
                            info.rho_updates += 1;
                            need_factor = true;
                            break;  // leave the iteration loop WITHOUT advancing iter; the factor block does it
                        }
                    }
                }
            }
            if constexpr (LPQ > 1) {  // (after a check the vectors are current already; after an exhausted loop they are not)
                quad_gather(zk, z);
                quad_gather(yk, y);
            }
            if (!need_factor) break;
        }
        if (solving) {
            if (iter > a.max_iter) info.status = SQPH_MAX_ITER_EXCEEDED;
            info.iter = iter;
        }
        if (!writer) return;
        if (state_dirty) {
#pragma unroll
            for (int j = 0; j < NMAX; j++)
                if (j < n) sx[j] = (double)x[j];
#pragma unroll
            for (int i = 0; i < MMAX; i++)
                if (i < m) {
                    sz[i] = (double)z[i];
                    sy[i] = (double)y[i];
                    srho[i] = (double)rho[i];
                }
        }
        a.info[qp] = info;
        a.rho[qp] = (double)rho_s;
    }
};

template <typename TA, typename TIN, int NMAX, int MMAX, bool EXACT, int LPQ = 1>
__global__ __launch_bounds__(64) void admm_lane_kernel(KArgs<double, TIN> a) {
    LaneKernel<TA, TIN, NMAX, MMAX, EXACT, LPQ>::run(a);
}
// batches up to this size take the quad variant where the shape has one (m <= 4).  Measured on the MI355X (tools/xp/quad_lat.sh,
// gpurun_out/r04_quad_lat.txt; (2,3), device-resident): 1,024 QPs x 200 iterations 43 us against 47 us for one QP per lane, the SQP
// driver's settings 47 against 52 us — an ADMM iteration of a QP this small is a chain of ~20 DEPENDENT fp64 operations (~25 cycles
// each for a lone wavefront), not an instruction-issue problem, so spreading the rows over lanes shortens it only by the length of
// the A'w accumulation; from ~4,096 QPs on the extra wavefronts cost more than that (16,384: 0.357 against 0.279 ms, default settings)
#define SQPH_QUAD_MAX_BATCH 2048

// shapes compiled into the library: {NMAX, MMAX, EXACT}; first match wins (exact: n == NMAX && m == MMAX; else n <= NMAX && m <= MMAX).
// The exact ones are the shapes of the reference's SQP test problems (tests/sqp_test.cpp, tests/sqp_test_autodiff.cpp).
#if defined(SQPH_SLIM) && !defined(SQPH_SLIM_LANE)
#define SQPH_LANE_SHAPES(X)
#else
#define SQPH_LANE_SHAPES(X) \
    X(2, 3, true)           \
    X(2, 2, true)           \
    X(3, 3, true)           \
    X(2, 1, true)           \
    X(4, 4, false)          \
    X(4, 6, false)
#endif
#define SQPH_LANE_MATCH(a, N_, M_, E_) ((E_) ? ((a).n == N_ && (a).m == M_) : ((a).n <= N_ && (a).m <= M_))

#ifdef SQPH_SIM
template <typename TIN, typename TA = double>
inline int sim_run_lane(const KArgs<double, TIN> &a, bool quad = false) {
#define SQPH_SIM_CASE(N_, M_, E_)                                                                          \
    if (SQPH_LANE_MATCH(a, N_, M_, E_)) {                                                                  \
        if constexpr (M_ <= 4 && sizeof(TA) == 8) {                                                        \
            if (quad) {                                                                                    \
                ::sqph_sim::launch(admm_lane_kernel<TA, TIN, N_, M_, E_, 4>, dim3((a.batch + 15) / 16), dim3(64), 0, a); \
                return 0;                                                                                  \
            }                                                                                              \
        }                                                                                                  \
        ::sqph_sim::launch(admm_lane_kernel<TA, TIN, N_, M_, E_>, dim3((a.batch + 63) / 64), dim3(64), 0, a);  \
        return 0;                                                                                          \
    }
    SQPH_LANE_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

}  // namespace sqph
