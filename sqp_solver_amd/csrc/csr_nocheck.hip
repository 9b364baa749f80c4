// The sparse kernel without the residual-check block (CsrKernel::run<false>, admm_csr_kernel.h): what a call runs that never looks
// at the residuals.  A translation unit of its own, like wg_nocheck.hip.
#include <hip/hip_runtime.h>

#include "admm_csr_kernel.h"

namespace sqph {

template <typename TIN>
int csr_nocheck_launch(int TT, int m, int nnz_cap, int batch, hipStream_t stream, const CsrLaunch<TIN> &p) {
#define SQPH_CSR_CASE(TT_)                                                                                                      \
    if (TT == TT_) {                                                                                                            \
        const CsrLayout<TT_> L = CsrLayout<TT_>::make(m, nnz_cap);                                                              \
        if (hipFuncSetAttribute((const void *)admm_csr_nocheck_kernel<TIN, TT_>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                (int)L.bytes) != hipSuccess)                                                                    \
            return -1;                                                                                                          \
        hipLaunchKernelGGL((admm_csr_nocheck_kernel<TIN, TT_>), dim3(batch), dim3(1024), L.bytes, stream, p);                   \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                        \
    }
    SQPH_CSR_SHAPES(SQPH_CSR_CASE)
#undef SQPH_CSR_CASE
    return 0;
}
template int csr_nocheck_launch<double>(int, int, int, int, hipStream_t, const CsrLaunch<double> &);
template int csr_nocheck_launch<float>(int, int, int, int, hipStream_t, const CsrLaunch<float> &);

}  // namespace sqph
