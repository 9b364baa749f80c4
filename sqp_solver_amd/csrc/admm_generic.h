// Generic ADMM kernel: one workgroup (1..4 wavefronts) per QP, runtime (n, m).
//
// This is the shape-agnostic fallback: matrices stay in global memory (L1/L2 resident while a
// QP is being iterated), vectors live in LDS.  The register-tiled kernels in admm_wg_kernel.h are
// the fast path for the shapes they cover.  Algorithm = the reference ADMM
// (/root/reference/src/qp.cpp:64-157) on the Schur-ordered KKT system: with R = diag(rho_vec)
//     S = P + sigma I + A' R A = (W'W)^-1 (n x n, SPD)          [replaces the (n+m)^2 LDL^T, qp.cpp:159-259]
//     x~ = W' W ( sigma x - q + A' R (z - R^-1 y) )             [== head(n) of K^-1 rhs, qp.cpp:89-92]
//     z~ = A x~                                                 [== z_prev + R^-1 (nu - y),  qp.cpp:93]
// followed by the verbatim x / z / y updates (qp.cpp:96-103), the residual/termination block
// (qp.cpp:105-123, 316-371) and adaptive rho (qp.cpp:125-144, 296-314, 333-341).
#pragma once
#include "block_ops.h"
#include "kargs.h"

namespace sqph {

template <typename T>
__host__ __device__ inline size_t generic_lds_elems(int n, int m, int nt) {
    // n-vectors: x xt b q Px row col ; m-vectors: z y w l u rho rinv zt Ax ; part[nt] ; red[7*nt]
    return (size_t)7 * n + (size_t)9 * m + (size_t)8 * nt + 8;
}

// out[r] = sum_k M[k*ld + r] * v[k]   (outputs contiguous in memory => coalesced across lanes).
// v, out, part in LDS. Ends with a barrier; callers must have synchronised v beforehand.
// tri (block-uniform): 0 = full; 1 = M[k ld + r] is zero for k > r (W, column-major), 2 = zero for k < r (W' from the row-major copy):
// the structural zeros are not streamed (a quarter of an iteration's bytes at m = 2 n).  The terms that remain keep their chains (the
// skipped ones added exact zeros): results bit for bit those of the full product.
template <typename T, typename TM>
__device__ __forceinline__ void matvec_cm(const TM *__restrict__ M, long ld, int R, int K_, const T *v, T *out, T *part, int tri = 0) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (R <= 0) return;  // m == 0 (unconstrained QP): block-uniform
    if (R >= nt) {
        for (int r = tid; r < R; r += nt) {
            T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            const int K = (tri == 1 && r + 1 < K_) ? r + 1 : K_;
            int k = tri == 2 ? (r & ~3) : 0;
            // eight loads requested before the first is used (the matrices of these shapes stream from L2 / the Infinity Cache: with
            // four in flight per lane a 1,024-lane workgroup moved 15 GB/s); the chains a0 .. a3 and their order are unchanged
            for (; k + 7 < K; k += 8) {
                T mv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) mv[e] = (T)M[(long)(k + e) * ld + r];
                a0 += mv[0] * v[k];
                a1 += mv[1] * v[k + 1];
                a2 += mv[2] * v[k + 2];
                a3 += mv[3] * v[k + 3];
                a0 += mv[4] * v[k + 4];
                a1 += mv[5] * v[k + 5];
                a2 += mv[6] * v[k + 6];
                a3 += mv[7] * v[k + 7];
            }
            for (; k + 3 < K; k += 4) {
                a0 += (T)M[(long)k * ld + r] * v[k];
                a1 += (T)M[(long)(k + 1) * ld + r] * v[k + 1];
                a2 += (T)M[(long)(k + 2) * ld + r] * v[k + 2];
                a3 += (T)M[(long)(k + 3) * ld + r] * v[k + 3];
            }
            for (; k < K; k++) a0 += (T)M[(long)k * ld + r] * v[k];
            out[r] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        return;
    }
    // fewer outputs than threads: split the reduction range over G lane groups
    const int G = nt / R;
    const int g = tid / R;
    const int r = tid - g * R;
    T a0 = 0, a1 = 0;
    if (g < G) {
        const int K = (tri == 1 && r + 1 < K_) ? r + 1 : K_;
        int k = g;
        if (tri == 2 && r > g) {  // the first k >= r of this group's sequence g, g + G, ... whose position in it is even
            const int idx = (r - g + G - 1) / G;
            k = g + (idx & ~1) * G;
        }
        for (; k + 7 * G < K; k += 8 * G) {  // (eight loads in flight; the two chains keep their terms and order)
            T mv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) mv[e] = (T)M[(long)(k + e * G) * ld + r];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                a0 += mv[e] * v[k + e * G];
                a1 += mv[e + 1] * v[k + (e + 1) * G];
            }
        }
        for (; k + G < K; k += 2 * G) {
            a0 += (T)M[(long)k * ld + r] * v[k];
            a1 += (T)M[(long)(k + G) * ld + r] * v[k + G];
        }
        for (; k < K; k += G) a0 += (T)M[(long)k * ld + r] * v[k];
    }
    part[tid] = a0 + a1;
    __syncthreads();
    if (tid < R) {
        T s = 0;
        for (int gg = 0; gg < G; gg++) s += part[gg * R + tid];
        out[tid] = s;
    }
    __syncthreads();
}

// Factor of the Schur matrix S = Psym + sigma I + A' diag(rho) A (SPD):
//     S = D_J^1/2 (L D L') D_J^1/2      (Jacobi scaling to unit diagonal, then LDL' without pivoting)
//     W = D^-1/2 L^-1 D_J^-1/2          (lower triangular)      =>   S^-1 = W' W
// The solve x~ = W'(W b) is two mat-vecs with cond(W) = sqrt(cond(S)); multiplying by an explicit
// S^-1 instead costs ~2 more digits in the ADMM iterates (measured: 1.6e-11 vs 2e-13 after 50
// iterations at cond(S) = 4e3), which matters once adaptive rho amplifies differences.
// L^-1 is formed in place by forward elimination of [S | I]: after step k the strictly-lower part of
// column k holds -l_ik, later steps apply their row operations to it as well; the "+1" on the pivot
// position of the broadcast row makes the generic rank-1 update write that entry (pivots are <= 1
// after the scaling, so there is no cancellation).
// Wm: n*n col-major result W; Wt: n*n row-major copy (W' with coalesced rows). row/sj/dsv: LDS [n]; stage: LDS [8 nt].
// Returns false (block-uniform) on a non-positive / non-finite pivot  => NUMERICAL_ISSUES.
#ifdef SQPH_SIM
#define SQPH_GENERIC_NOINLINE
#else
#define SQPH_GENERIC_NOINLINE __attribute__((noinline))  // (three call sites; a real call keeps the set-up's registers out of the iteration loop's allocation)
#endif
}  // namespace sqph
#include "admm_generic_msetup.h"
namespace sqph {
template <typename T, typename TIN>
__device__ SQPH_GENERIC_NOINLINE bool factor_schur(int n, int m, const TIN *__restrict__ P, const T *__restrict__ At, const T *rho, T sigma,
                             T *__restrict__ Wm, T *__restrict__ Wt, T *row, T *sj, T *dsv, T *stage) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nn = n * n;
    if constexpr (sizeof(T) == 8) {  // the blocked set-up on the matrix pipe where the blocks fit the workspace (admm_generic_msetup.h)
        if (GenericBlocked::fits(n, nt)) return GenericBlocked::factor<TIN>(n, m, P, At, rho, sigma, Wm, Wt, stage);
    }
    // S through LDS: KB rows of At at a time in `stage` (the idle reduction scratch, 8 nt elements); a thread owns row i of a block of four
    // columns (a "unit": one LDS read of At[k][i] and four broadcast reads of At[k][j..j+3] per four products) and up to U = 2 units per
    // pass, all of a pass's sums in registers across the whole k range — every entry is still the two chains s0 (even k) + s1 (odd k) of
    // (At[k][i] rho_k) At[k][j], bit for bit what the entry-by-entry loop over global memory computed (13 of this set-up's 18 ms at n = 300)
    int KB = (8 * nt) / (n > 0 ? n : 1);
    KB = KB > 16 ? 16 : (KB & ~1);
    if (KB >= 2) {
        constexpr int U = 2;  // (two units = 16 sums per pass: the kernel is compiled for 1,024 lanes, 128 VGPRs)
        const int JB = (n + 3) >> 2;
        const long units = (long)n * JB;
        for (long u0 = 0; u0 < units; u0 += (long)nt * U) {
            int ui[U], uj[U];
            T s0[U][4], s1[U][4];
#pragma unroll
            for (int q = 0; q < U; q++) {
                const long u = u0 + (long)q * nt + tid;
                const bool ok = u < units;
                const int jb = ok ? (int)(u / n) : 0;
                ui[q] = ok ? (int)(u - (long)jb * n) : -1;
                uj[q] = 4 * jb;
#pragma unroll
                for (int c = 0; c < 4; c++) s0[q][c] = s1[q][c] = T(0);
            }
            for (int k0 = 0; k0 < m; k0 += KB) {
                __syncthreads();
                for (int e = tid; e < KB * n; e += nt) {
                    const int kk = e / n, c = e - kk * n;
                    stage[e] = (k0 + kk < m) ? At[(long)(k0 + kk) * n + c] : T(0);
                }
                __syncthreads();
                const int kb = (m - k0 < KB) ? m - k0 : KB;
                for (int kk = 0; kk < kb; kk += 2) {  // (a row beyond m is staged as zeros: its products add +0 to s1)
                    const T *r0 = stage + kk * n, *r1 = r0 + n;
                    const T rh0 = rho[k0 + kk];
                    const T rh1 = kk + 1 < kb ? rho[k0 + kk + 1] : T(0);
#pragma unroll
                    for (int q = 0; q < U; q++) {
                        const int iq = ui[q] >= 0 ? ui[q] : 0;
                        const T a0 = r0[iq] * rh0, a1 = r1[iq] * rh1;
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const int jj = uj[q] + c < n ? uj[q] + c : n - 1;
                            s0[q][c] += a0 * r0[jj];
                            s1[q][c] += a1 * r1[jj];
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < U; q++) {
                if (ui[q] >= 0) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int i = ui[q], jj = uj[q] + c;
                        if (jj < n) {
                            const int lo = i > jj ? i : jj, hi = i > jj ? jj : i;
                            // only the lower triangle of P reaches the reference's factor (Eigen::LDLT<.,Lower>, qp.hpp:129)
                            const T acc = (T)P[(long)hi * n + lo] + (i == jj ? sigma : T(0));
                            Wm[(long)jj * n + i] = acc + (s0[q][c] + s1[q][c]);
                        }
                    }
                }
            }
        }
    } else
    for (int e = tid; e < nn; e += nt) {
        const int j = e / n, i = e - j * n;
        const int lo = i > j ? i : j, hi = i > j ? j : i;
        // only the lower triangle of P reaches the reference's factor (Eigen::LDLT<.,Lower>, qp.hpp:129)
        T acc = (T)P[(long)hi * n + lo] + (i == j ? sigma : T(0));
        T s0 = 0, s1 = 0;
        int k = 0;
        for (; k + 1 < m; k += 2) {
            s0 += At[(long)k * n + i] * rho[k] * At[(long)k * n + j];
            s1 += At[(long)(k + 1) * n + i] * rho[k + 1] * At[(long)(k + 1) * n + j];
        }
        if (k < m) s0 += At[(long)k * n + i] * rho[k] * At[(long)k * n + j];
        Wm[e] = acc + (s0 + s1);
    }
    __syncthreads();
    for (int j = tid; j < n; j += nt) sj[j] = Wm[(long)j * n + j];
    __syncthreads();
    bool bad = false;
    for (int j = 0; j < n; j++) {
        const T d = sj[j];
        if (!(d > T(0)) || !(d * T(0) == T(0))) bad = true;  // uniform: every thread scans the same LDS words
    }
    __syncthreads();
    if (bad) return false;
    for (int j = tid; j < n; j += nt) sj[j] = T(1) / (T)sqrt((double)sj[j]);
    __syncthreads();
    for (int e = tid; e < nn; e += nt) {
        const int j = e / n, i = e - j * n;
        Wm[e] = Wm[e] * sj[i] * sj[j];
    }
    __syncthreads();
    for (int k = 0; k < n; k++) {
        for (int j = tid; j < n; j += nt) row[j] = Wm[(long)j * n + k];
        __syncthreads();
        const T d = row[k];
        if (!(d > T(0)) || !(d * T(0) == T(0))) return false;
        const T dinv = T(1) / d;
        if (tid == 0) dsv[k] = d;
        // rows i > k of every column, 64 lanes down a column (the same update entry by entry as a sweep over all n^2 entries with a
        // division and a test each — which was most of this kernel's set-up: 20 ms at n = 300)
        {
            const int jl = tid >> 6, JS = nt >> 6;
            for (int i0 = k + 1; i0 < n; i0 += 64) {
                const int i = i0 + (tid & 63);
                const bool iok = i < n;
                const T li = iok ? row[i] * dinv : T(0);
                for (int j0 = jl; j0 < n; j0 += 4 * JS) {  // four columns' entries requested before the first is updated
                    T v[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int j = j0 + c * JS;
                        v[c] = (iok && j < n) ? Wm[(long)j * n + i] : T(0);
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int j = j0 + c * JS;
                        if (iok && j < n) Wm[(long)j * n + i] = v[c] - li * ((j == k) ? d + T(1) : row[j]);
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < nn; e += nt) {
        const int j = e / n, i = e - j * n;
        const T rs = T(1) / (T)sqrt((double)dsv[i]);
        const T v = (i > j ? Wm[e] * rs : (i == j ? rs : T(0))) * sj[j];
        Wm[e] = v;
        Wt[(long)i * n + j] = v;
    }
    __syncthreads();
    return true;
}

#ifdef SQPH_SIM
#define SQPH_DYN_SMEM(name) unsigned char *name = ::sqph_sim::dyn_smem()
#else
#define SQPH_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

template <typename T, typename TIN>
__global__ void admm_generic_kernel(KArgs<T, TIN> a) {
    SQPH_DYN_SMEM(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int qp = blockIdx.x;
    if (qp >= a.batch) return;
    const int n = a.n, m = a.m;

    T *lds = reinterpret_cast<T *>(smem_raw);
    T *x = lds;          lds += n;
    T *xt = lds;         lds += n;
    T *b = lds;          lds += n;
    T *q = lds;          lds += n;
    T *Px = lds;         lds += n;
    T *gjrow = lds;      lds += n;
    T *gjcol = lds;      lds += n;
    T *z = lds;          lds += m;
    T *y = lds;          lds += m;
    T *w = lds;          lds += m;
    T *l = lds;          lds += m;
    T *u = lds;          lds += m;
    T *rho = lds;        lds += m;
    T *rinv = lds;       lds += m;
    T *zt = lds;         lds += m;
    T *Ax = lds;         lds += m;
    T *part = lds;       lds += nt;
    T *red = lds;

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
    T *Wm = a.Sinv + (long)qp * 2 * n * n;  // factor W (col-major) followed by its row-major copy
    T *Wt = Wm + (long)n * n;
    T *At = a.At + (long)qp * m * n;

    sqph_info info = a.info[qp];
    T rho_s = a.rho[qp];
    const int mode = a.mode;

    if (!(mode & (MODE_SETUP | MODE_UPDATE)) &&
        (info.status == SQPH_UNINITIALIZED || info.status == SQPH_NUMERICAL_ISSUES))
        return;  // qp.cpp:68-71

    for (int j = tid; j < n; j += nt) q[j] = (T)gq[j];
    for (int i = tid; i < m; i += nt) {
        l[i] = (T)gl[i];
        u[i] = (T)gu[i];
    }

    if (mode & (MODE_SETUP | MODE_UPDATE)) {
        // constr_type_init (qp.cpp:283-294) + rho_vec_update(settings.rho) (qp.cpp:296-314)
        rho_s = a.rho0;
        for (int i = tid; i < m; i += nt) {
            const T li = l[i], ui = u[i];
            int ct;
            if (li < -a.loose_thresh && ui > a.loose_thresh)
                ct = SQPH_LOOSE_BOUNDS;
            else if (ui - li < a.eq_tol)
                ct = SQPH_EQUALITY_CONSTRAINT;
            else
                ct = SQPH_INEQUALITY_CONSTRAINT;
            sct[i] = ct;
            const T r = rho_for_type<T>(ct, rho_s, a.rho_min, a.rho_eq_factor);
            rho[i] = r;
            rinv[i] = T(1) / r;
            srho[i] = r;
        }
        info.rho_updates += 1;
        if (mode & MODE_SETUP) {
            for (int j = tid; j < n; j += nt) x[j] = 0;
            for (int i = tid; i < m; i += nt) z[i] = y[i] = 0;
        } else {
            for (int j = tid; j < n; j += nt) x[j] = sx[j];
            for (int i = tid; i < m; i += nt) {
                z[i] = sz[i];
                y[i] = sy[i];
            }
        }
        // row-major copy of A: At[i*n + j] = A[j*m + i]
        for (int e = tid; e < m * n; e += nt) {
            const int i = e / n, j = e - i * n;
            At[e] = (T)gA[(long)j * m + i];
        }
        __syncthreads();
        const bool ok = factor_schur<T, TIN>(n, m, gP, At, rho, a.sigma, Wm, Wt, gjrow, gjcol, Px, part);
        __syncthreads();
        info.status = ok ? SQPH_UNSOLVED : SQPH_NUMERICAL_ISSUES;
    } else {
        for (int j = tid; j < n; j += nt) x[j] = sx[j];
        for (int i = tid; i < m; i += nt) {
            z[i] = sz[i];
            y[i] = sy[i];
            const T r = srho[i];
            rho[i] = r;
            rinv[i] = T(1) / r;
        }
        if (mode & MODE_REFACTOR) {
            // solve() on an instance whose factor this kernel cannot use (not kept, or built by another kernel family):
            // rebuild the row-major copy of A and the factor from the current rho vector
            for (int e = tid; e < m * n; e += nt) {
                const int i = e / n, j = e - i * n;
                At[e] = (T)gA[(long)j * m + i];
            }
            __syncthreads();
            const bool ok = factor_schur<T, TIN>(n, m, gP, At, rho, a.sigma, Wm, Wt, gjrow, gjcol, Px, part);
            __syncthreads();
            if (!ok) info.status = SQPH_NUMERICAL_ISSUES;
        }
    }
    __syncthreads();

    bool state_dirty = (mode & MODE_SETUP) != 0;

    if ((mode & MODE_SOLVE) && info.status != SQPH_NUMERICAL_ISSUES && info.status != SQPH_UNINITIALIZED) {
        if ((mode & MODE_COLD_RESET) && !a.warm_start) {
            for (int j = tid; j < n; j += nt) x[j] = 0;
            for (int i = tid; i < m; i += nt) z[i] = y[i] = 0;
            __syncthreads();
        }
        state_dirty = true;
        const T alpha = a.alpha, sigma = a.sigma;
        const T one_m_alpha = T(1) - alpha;
        T nrm_prim = 0, nrm_dual = 0;  // max_Ax_z_norm_, max_Px_ATy_q_norm_
        int iter;
        for (iter = 1; iter <= a.max_iter; iter++) {
            // rhs tail (qp.cpp:275) pre-multiplied by R: w = R (z - R^-1 y)
            for (int i = tid; i < m; i += nt) w[i] = rho[i] * (z[i] - rinv[i] * y[i]);
            __syncthreads();
            matvec_cm<T>(At, n, n, m, w, b, part);  // b = A' w
            for (int j = tid; j < n; j += nt) b[j] = (sigma * x[j] - q[j]) + b[j];
            __syncthreads();
            matvec_cm<T>(Wm, n, n, n, b, Px, part, 1);   // W b      (Px is scratch outside the checks)
            matvec_cm<T>(Wt, n, n, n, Px, xt, part, 2);  // x~ = W' (W b)
            matvec_cm<T>(gA, m, m, n, xt, zt, part);   // z~ = A x~
            for (int j = tid; j < n; j += nt) x[j] = alpha * xt[j] + one_m_alpha * x[j];
            for (int i = tid; i < m; i += nt) {
                const T zr = alpha * zt[i] + one_m_alpha * z[i];
                T zn = zr + rinv[i] * y[i];
                zn = zn < l[i] ? l[i] : zn;  // cwiseMax(l) then cwiseMin(u), qp.cpp:278-281
                zn = zn > u[i] ? u[i] : zn;
                y[i] = y[i] + rho[i] * (zr - zn);
                z[i] = zn;
            }
            __syncthreads();

            const bool check = a.check_termination != 0 && (iter % a.check_termination == 0);
            const bool adapt = a.adaptive_rho && (iter % a.adaptive_rho_interval == 0);
            if (check || adapt) {
                // update_state (qp.cpp:316-331) + residuals (qp.cpp:353-361)
                matvec_cm<T>(gA, m, m, n, x, Ax, part);
                matvec_cm<T>(gP, n, n, n, x, Px, part);
                matvec_cm<T>(At, n, n, m, y, b, part);  // b <- A' y (b is dead here)
                T v[7] = {0, 0, 0, 0, 0, 0, 0};
                for (int i = tid; i < m; i += nt) {
                    v[0] = nanmax(v[0], tabs(Ax[i]));
                    v[1] = nanmax(v[1], tabs(z[i]));
                    v[2] = nanmax(v[2], tabs(Ax[i] - z[i]));
                }
                for (int j = tid; j < n; j += nt) {
                    v[3] = nanmax(v[3], tabs(Px[j]));
                    v[4] = nanmax(v[4], tabs(b[j]));
                    v[5] = nanmax(v[5], tabs(q[j]));
                    v[6] = nanmax(v[6], tabs(Px[j] + q[j] + b[j]));
                }
                block_nanmax<T, 7>(v, red);
                nrm_prim = nanmax(v[0], v[1]);
                nrm_dual = nanmax(v[3], nanmax(v[4], v[5]));
                info.res_prim = (double)v[2];
                info.res_dual = (double)v[6];
                if (check && a.trace && qp == a.trace_qp) {  // print_status, qp.cpp:373-383 (recorded; the host prints)
                    if (tid == 0) {
                        T obj = 0;
                        for (int j = 0; j < n; j++) obj += x[j] * (T(0.5) * Px[j] + q[j]);
                        const int k = (int)a.trace[0];
                        if (k < a.trace_cap) {
                            a.trace[1 + 4 * k] = (double)iter;
                            a.trace[2 + 4 * k] = (double)obj;
                            a.trace[3 + 4 * k] = (double)v[2];
                            a.trace[4 + 4 * k] = (double)v[6];
                            a.trace[0] = (double)(k + 1);
                        }
                    }
                }
                if (check) {
                    // termination_criteria, qp.cpp:343-351, 363-371
                    if (v[2] <= a.eps_abs + a.eps_rel * nrm_prim && v[6] <= a.eps_abs + a.eps_rel * nrm_dual) {
                        info.status = SQPH_SOLVED;
                        break;
                    }
                }
                if (adapt) {
                    // rho_estimate, qp.cpp:333-341 ; clamp + tolerance test, qp.cpp:130-136
                    const T eps = a.regul;
                    const T rp_norm = v[2] / (nrm_prim + eps);
                    const T rd_norm = v[6] / (nrm_dual + eps);
                    T new_rho = rho_s * (T)sqrt((double)(rp_norm / (rd_norm + eps)));
                    new_rho = new_rho < a.rho_max ? new_rho : a.rho_max;  // fmax(RHO_MIN, fmin(new_rho, RHO_MAX))
                    new_rho = new_rho > a.rho_min ? new_rho : a.rho_min;
                    info.rho_estimate = (double)new_rho;
                    if (new_rho < rho_s / a.rho_tol || new_rho > rho_s * a.rho_tol) {
                        rho_s = new_rho;
                        for (int i = tid; i < m; i += nt) {
                            const T r = rho_for_type<T>(sct[i], rho_s, a.rho_min, a.rho_eq_factor);
                            rho[i] = r;
                            rinv[i] = T(1) / r;
                        }
                        info.rho_updates += 1;
                        __syncthreads();
                        const bool ok = factor_schur<T, TIN>(n, m, gP, At, rho, sigma, Wm, Wt, gjrow, gjcol, Px, part);
                        __syncthreads();
                        if (!ok) {
                            info.status = SQPH_NUMERICAL_ISSUES;
                            break;
                        }
                    }
                }
            }
        }
        if (iter > a.max_iter) info.status = SQPH_MAX_ITER_EXCEEDED;
        info.iter = iter;
    }

    if (state_dirty) {
        for (int j = tid; j < n; j += nt) sx[j] = x[j];
        for (int i = tid; i < m; i += nt) {
            sz[i] = z[i];
            sy[i] = y[i];
            srho[i] = rho[i];
        }
    }
    if (tid == 0) {
        a.info[qp] = info;
        a.rho[qp] = rho_s;
    }
}

}  // namespace sqph
