// The row-split workgroup kernels without the residual-check block (WgrKernel::run<false>, admm_wgr_kernel.h): what a call runs that
// never looks at the residuals (check_termination == 0, no adaptive rho).  A translation unit of its own, like wg_nocheck.hip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "admm_wgr_kernel.h"

namespace sqph {

#ifdef SQPH_EXPERIMENTS
#define SQPH_OCC(K, NT_)                                                                          \
    {                                                                                             \
        static bool once_ = false;                                                                \
        if (!once_) {                                                                             \
            once_ = true;                                                                         \
            int nb_ = 0;                                                                          \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, K, NT_, 0);                  \
            hipFuncAttributes fa_;                                                                \
            (void)hipFuncGetAttributes(&fa_, (const void *)K);                                    \
            fprintf(stderr, "[xp] occupancy %d blocks/CU, %d regs, %zu B LDS, %zu B scratch\n", nb_, fa_.numRegs, fa_.sharedSizeBytes, fa_.localSizeBytes); \
        }                                                                                         \
    }
#else
#define SQPH_OCC(K, NT_)
#endif

template <typename TIN>
int wgr_nocheck_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name) {
#define SQPH_WGR_CASE(NW_, R_, C_, TR_, TC_, TW_, TS_, W_)                                                                                      \
    if (SQPH_WGR_FITS(a, R_, C_, TR_, TC_, TW_, TS_)) {                                                                                        \
        SQPH_OCC((admm_wgr_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, TS_, W_>), 64 * NW_);          \
        hipLaunchKernelGGL((admm_wgr_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, TS_, W_>), dim3(a.batch), dim3(64 * NW_), 0, stream, a);   \
        *name = "wgr" #NW_ "_" #R_ "x" #C_ "_" #TS_ "x" #TC_ "_w" #W_;                                                                          \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                                        \
    }
    SQPH_WGR_SHAPES(SQPH_WGR_CASE)
#undef SQPH_WGR_CASE
    return 0;
}
template int wgr_nocheck_try_launch<double>(const KArgs<double, double> &, hipStream_t, const char **);
template int wgr_nocheck_try_launch<float>(const KArgs<double, float> &, hipStream_t, const char **);

}  // namespace sqph
