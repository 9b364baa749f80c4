// The CU-wide kernel (admm_csr_kernel.h) in its dense-A mode, CsrKernel::run<CHECKS, DENSE = true>: dense problems beyond the
// register-tiled kernels' shapes — 112 < n <= 256 (tile edge 8 since round 6: 224 < n <= 256), or m beyond their row counts, m <= 512 — keep the factor W in the CU's register file
// and stream the column-major A from global memory twice per iteration.  A translation unit of its own: the code generated for the
// sparse kernels does not depend on these being instantiated next to them.
#include <hip/hip_runtime.h>

#include "admm_csr_kernel.h"

namespace sqph {

template <typename TIN>
int csrd_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name) {
    if (a.m > 512 || a.m < 1) return 0;
    const bool checks = !(a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0));
    CsrLaunch<TIN> p;
    p.a = a;
    p.ca = CsrArgs<TIN>{nullptr, nullptr, nullptr, 0, 0, 0, 0};
#define SQPH_CSRD_CASE(TT_)                                                                                                               \
    if (a.n <= 32 * TT_) {                                                                                                                \
        const CsrLayout<TT_> L = CsrLayout<TT_>::make(a.m, 0);                                                                            \
        const void *k = checks ? (const void *)admm_csrd_kernel<TIN, TT_> : (const void *)admm_csrd_nocheck_kernel<TIN, TT_>;             \
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes) != hipSuccess) return -1;                    \
        if (checks) hipLaunchKernelGGL((admm_csrd_kernel<TIN, TT_>), dim3(a.batch), dim3(1024), L.bytes, stream, p);                      \
        else hipLaunchKernelGGL((admm_csrd_nocheck_kernel<TIN, TT_>), dim3(a.batch), dim3(1024), L.bytes, stream, p);                     \
        *name = "cud_t" #TT_;                                                                                                             \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                                  \
    }
    SQPH_CSRD_SHAPES(SQPH_CSRD_CASE)
#undef SQPH_CSRD_CASE
    return 0;
}
template int csrd_try_launch<double>(const KArgs<double, double> &, hipStream_t, const char **);
template int csrd_try_launch<float>(const KArgs<double, float> &, hipStream_t, const char **);

}  // namespace sqph
