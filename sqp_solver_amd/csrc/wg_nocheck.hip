// The workgroup-tiled kernels without the residual-check block (WgKernel::run<false>, admm_wg_kernel.h): what a call runs that never
// looks at the residuals (check_termination == 0, no adaptive rho).  A translation unit of its own so that the code generated for the
// checking kernels in capi.hip does not depend on these being instantiated next to them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "admm_wg_kernel.h"

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
int wg_nocheck_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name, int skip) {
#define SQPH_WG_CASE(NW_, R_, C_, TR_, TC_, TW_, W_)                                                                                     \
    if (a.m <= R_ * TR_ && a.n <= C_ * TC_ && skip-- <= 0) {                                                                             \
        SQPH_OCC((admm_wg_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, wg_nocheck_wpe(NW_, R_, C_, TR_, TC_, W_)>), 64 * NW_);          \
        hipLaunchKernelGGL((admm_wg_nocheck_kernel<TIN, NW_, R_, C_, TR_, TC_, TW_, wg_nocheck_wpe(NW_, R_, C_, TR_, TC_, W_)>), dim3(a.batch), dim3(64 * NW_), 0, stream, a); \
        *name = "wg" #NW_ "_" #R_ "x" #C_ "_" #TR_ "x" #TC_ "_w" #W_;                                                                    \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                                 \
    }
    SQPH_WG_SHAPES(SQPH_WG_CASE)
#undef SQPH_WG_CASE
    return 0;
}
template int wg_nocheck_try_launch<double>(const KArgs<double, double> &, hipStream_t, const char **, int);
template int wg_nocheck_try_launch<float>(const KArgs<double, float> &, hipStream_t, const char **, int);

}  // namespace sqph
