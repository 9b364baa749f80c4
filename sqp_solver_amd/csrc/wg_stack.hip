// The register-tiled kernels on the STACKED operator (WgKernel::run<CHECKS, false, STACK = true>, admm_wg_kernel.h): problems whose
// m leaves room for the C TC rows of W' inside R (TR + TW - 1) stacked rows — the C3 shape with m <= 104, BASELINE config 3 among
// them — run the iteration on TR + TW - 1 tile rows instead of TR + TW.  A translation unit of its own (the code generated for the other kernels does not depend
// on these being instantiated next to them).
#include <hip/hip_runtime.h>

#include "admm_wg_kernel.h"

namespace sqph {

// > 0 launched, 0 no stacked shape for this problem, < 0 launch error
template <typename TIN>
int wgs_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name) {
    const bool checks = !(a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0));
#define SQPH_WGS_CASE(NW_, R_, C_, TR_, TC_, TW_, W_)                                                                                      \
    if (a.m <= R_ * TR_ && a.n <= C_ * TC_ && a.m <= R_ * (TR_ + TW_ - 1) - C_ * TC_) {                                                         \
        if (checks)                                                                                                                        \
            hipLaunchKernelGGL((admm_wgs_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_>), dim3(a.batch), dim3(64 * NW_), 0, stream, a);       \
        else                                                                                                                               \
            hipLaunchKernelGGL((admm_wgs_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, W_>), dim3(a.batch), dim3(64 * NW_), 0, stream, a); \
        *name = "wg" #NW_ "_" #R_ "x" #C_ "_" #TR_ "x" #TC_ "s_w" #W_;                                                                      \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                                   \
    }
    SQPH_WGS_SHAPES(SQPH_WGS_CASE)
#undef SQPH_WGS_CASE
    return 0;
}
template int wgs_try_launch<double>(const KArgs<double, double> &, hipStream_t, const char **);
template int wgs_try_launch<float>(const KArgs<double, float> &, hipStream_t, const char **);

}  // namespace sqph
