// The block-row sparse kernels (CsbKernel::run<CHECKS>, admm_csrb_kernel.h) and their launcher.  A translation unit of its own,
// compiled with -mllvm -simplifycfg-sink-common=false (sqp_solver_amd/build.py: UNIT_FLAGS): the kernel switches on its wavefront index
// into per-wavefront specialisations that update different slots of the block register array; sinking the specialisations' last
// stores into their common successor turns them into stores through a selected ADDRESS, which keeps those slots in scratch memory
// for the whole kernel (ten doubles per lane, reloaded twice per iteration: 62 instead of 30 ms at BASELINE config 5).  And with
// -mllvm -structurizecfg-skip-uniform-regions: the wave-uniform branches of the set-up stay scalar branches (admm_csrb_kernel.h).
#include <hip/hip_runtime.h>

#include "admm_csrb_kernel.h"

namespace sqph {

template <typename TIN>
int csrb_launch(int NB, bool nocheck, int m, int nnz_cap, int batch, hipStream_t stream, const CsrLaunch<TIN> &p) {
#define SQPH_CSB_CASE(NB_)                                                                                                   \
    if (NB == NB_) {                                                                                                         \
        const CsbLayout<NB_> L = CsbLayout<NB_>::make(m, nnz_cap);                                                           \
        const void *fn = nocheck ? (const void *)admm_csrb_nocheck_kernel<TIN, NB_> : (const void *)admm_csrb_kernel<TIN, NB_>; \
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes) != hipSuccess) return -1;      \
        if (nocheck)                                                                                                         \
            hipLaunchKernelGGL((admm_csrb_nocheck_kernel<TIN, NB_>), dim3(batch), dim3(512), L.bytes, stream, p);            \
        else                                                                                                                 \
            hipLaunchKernelGGL((admm_csrb_kernel<TIN, NB_>), dim3(batch), dim3(512), L.bytes, stream, p);                    \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                     \
    }
    SQPH_CSB_SHAPES(SQPH_CSB_CASE)
#undef SQPH_CSB_CASE
    return 0;
}
template int csrb_launch<double>(int, bool, int, int, int, hipStream_t, const CsrLaunch<double> &);
template int csrb_launch<float>(int, bool, int, int, int, hipStream_t, const CsrLaunch<float> &);

}  // namespace sqph
