// The block-row sparse kernels with P in compressed columns (CsbKernel::run<CHECKS, SP = true>, admm_csrb_kernel.h) and their launcher:
// the sqph_*_csr_sp entry points.  A translation unit of its own so that csrb.hip — the dense-P kernels config 5 is measured on — is
// what it was; the same -simplifycfg-sink-common=false (csrb.hip says why; sqp_solver_amd/build.py: UNIT_FLAGS).
#include <hip/hip_runtime.h>

#include "admm_csrb_kernel.h"

namespace sqph {

template <typename TIN>
int csrb_sp_launch(int NB, bool nocheck, int m, int nnz_cap, int batch, hipStream_t stream, const CsrLaunch<TIN> &p) {
#define SQPH_CSB_CASE(NB_)                                                                                                         \
    if (NB == NB_) {                                                                                                               \
        const CsbLayout<NB_> L = CsbLayout<NB_>::make(m, nnz_cap);                                                                 \
        const void *fn = nocheck ? (const void *)admm_csrb_sp_nocheck_kernel<TIN, NB_> : (const void *)admm_csrb_sp_kernel<TIN, NB_>; \
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes) != hipSuccess) return -1;            \
        if (nocheck)                                                                                                               \
            hipLaunchKernelGGL((admm_csrb_sp_nocheck_kernel<TIN, NB_>), dim3(batch), dim3(512), L.bytes, stream, p);               \
        else                                                                                                                       \
            hipLaunchKernelGGL((admm_csrb_sp_kernel<TIN, NB_>), dim3(batch), dim3(512), L.bytes, stream, p);                       \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                           \
    }
    SQPH_CSB_SHAPES(SQPH_CSB_CASE)
#undef SQPH_CSB_CASE
    return 0;
}
template int csrb_sp_launch<double>(int, bool, int, int, int, hipStream_t, const CsrLaunch<double> &);
template int csrb_sp_launch<float>(int, bool, int, int, int, hipStream_t, const CsrLaunch<float> &);

}  // namespace sqph
