// Host-side dispatch of the register-tiled kernels (admm_wg_kernel.h): first compiled shape that fits wins.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "admm_lane_kernel.h"
#include "admm_wg_kernel.h"

namespace sqph {

// one QP per lane for tiny shapes (admm_lane_kernel.h): >0 launched, 0 not covered, <0 error
template <typename TIN, typename TA = double>
inline int lane_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name) {
#ifdef SQPH_EXPERIMENTS  // environment knobs exist in experiment builds only (tools/slim_build.sh), never in the shipped library
    static const bool off = getenv("SQPH_NO_LANE") != nullptr;
    if (off) return 0;
    static const bool no_quad = getenv("SQPH_NO_QUAD") != nullptr;
#else
    constexpr bool no_quad = false;
#endif
#define SQPH_LANE_CASE(N_, M_, E_)                                                                                    \
    if (SQPH_LANE_MATCH(a, N_, M_, E_)) {                                                                             \
        if constexpr (M_ <= 4 && sizeof(TA) == 8) {   /* small batch: four lanes per QP (admm_lane_kernel.h, LPQ = 4) */ \
            if (a.batch <= SQPH_QUAD_MAX_BATCH && !no_quad) {                                                          \
                hipLaunchKernelGGL((admm_lane_kernel<TA, TIN, N_, M_, E_, 4>), dim3((a.batch + 15) / 16), dim3(64), 0, stream, a); \
                *name = (E_) ? "quad_" #N_ "x" #M_ "_exact" : "quad_" #N_ "x" #M_;                                       \
                return hipGetLastError() == hipSuccess ? 1 : -1;                                                      \
            }                                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((admm_lane_kernel<TA, TIN, N_, M_, E_>), dim3((a.batch + 63) / 64), dim3(64), 0, stream, a);   \
        *name = sizeof(TA) == 4 ? ((E_) ? "lane_" #N_ "x" #M_ "_exact_f32" : "lane_" #N_ "x" #M_ "_f32")                \
                                : ((E_) ? "lane_" #N_ "x" #M_ "_exact" : "lane_" #N_ "x" #M_);                          \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                              \
    }
    SQPH_LANE_SHAPES(SQPH_LANE_CASE)
#undef SQPH_LANE_CASE
    return 0;
}

// four-QPs-per-wavefront kernels for small shapes (admm_wg_kernel.h, run_group): >0 launched, 0 not covered, <0 error
template <typename TIN>
inline int g16_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name) {
#ifdef SQPH_EXPERIMENTS
    static const bool off = getenv("SQPH_NO_G16") != nullptr;
    if (off) return 0;
#endif
#define SQPH_G16_CASE(TR_, TC_, W_)                                                                                            \
    if (a.m <= 4 * TR_ && a.n <= 4 * TC_) {                                                                                    \
        hipLaunchKernelGGL((admm_g16_kernel<TIN, TR_, TC_, W_>), dim3((a.batch + 3) / 4), dim3(64), 0, stream, a);             \
        *name = "g16_" #TR_ "x" #TC_ "_w" #W_;                                                                                 \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                       \
    }
    SQPH_G16_SHAPES(SQPH_G16_CASE)
#undef SQPH_G16_CASE
    return 0;
}

// the same shapes without the residual-check block, for calls that never check (wg_nocheck.hip: a translation unit of its own)
template <typename TIN>
int wg_nocheck_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name, int skip);
extern template int wg_nocheck_try_launch<double>(const KArgs<double, double> &, hipStream_t, const char **, int);
extern template int wg_nocheck_try_launch<float>(const KArgs<double, float> &, hipStream_t, const char **, int);

// the fp32-product variant of the register-tiled kernels (wg_f32.hip): QPSolver<float> with SQPH_FLAG_F32_ARITH
int wgf_try_launch(const KArgs<double, float> &a, hipStream_t stream, const char **name);

// the stacked-operator variant of the register-tiled kernels (wg_stack.hip)
template <typename TIN>
int wgs_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name);
extern template int wgs_try_launch<double>(const KArgs<double, double> &, hipStream_t, const char **);
extern template int wgs_try_launch<float>(const KArgs<double, float> &, hipStream_t, const char **);

// workgroup-tiled kernels (admm_wg_kernel.h): >0 launched, 0 not covered, <0 launch error
template <typename TIN>
inline int wg_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name) {
#ifdef SQPH_EXPERIMENTS
    // SQPH_WG_SKIP=k: skip the first k shapes that would fit; SQPH_WG_ALWAYS_CHECKS: never take the no-check instantiation
    static const int skip_env = getenv("SQPH_WG_SKIP") ? atoi(getenv("SQPH_WG_SKIP")) : 0;
    static const bool always_checks = getenv("SQPH_WG_ALWAYS_CHECKS") != nullptr;
#else
    constexpr int skip_env = 0;
    constexpr bool always_checks = false;
#endif
    int skip = skip_env;
#ifdef SQPH_EXPERIMENTS
    static const bool no_stack = getenv("SQPH_NO_STACK") != nullptr;
#else
    constexpr bool no_stack = false;
#endif
    // problems whose first fitting shape is the two-wave 16 x 8 grid and whose m leaves room for W' inside ten stacked tile rows (wg_stack.hip)
    if (!no_stack && skip == 0 && !always_checks && !(a.m <= 128 && a.n <= 32)) {  // (n <= 32: the one-wave grids and the 16 x 8 / 8 x 4 grid come first)
        const int rc = wgs_try_launch<TIN>(a, stream, name);
        if (rc != 0) return rc;
    }
    if (a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0) && !always_checks)
        return wg_nocheck_try_launch<TIN>(a, stream, name, skip);
#define SQPH_WG_CASE(NW_, R_, C_, TR_, TC_, TW_, W_)                                                                            \
    if (a.m <= R_ * TR_ && a.n <= C_ * TC_ && skip-- <= 0) {                                                                    \
        hipLaunchKernelGGL((admm_wg_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_>), dim3(a.batch), dim3(64 * NW_), 0, stream, a); \
        *name = "wg" #NW_ "_" #R_ "x" #C_ "_" #TR_ "x" #TC_ "_w" #W_;                                                                    \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                        \
    }
    SQPH_WG_SHAPES(SQPH_WG_CASE)
#undef SQPH_WG_CASE
    return 0;
}

}  // namespace sqph
