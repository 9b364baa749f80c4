// The register-tiled kernels with fp32 products (WgKernel::run<CHECKS, F32 = true>, admm_wg_kernel.h): what a QPSolver<float> created
// with SQPH_FLAG_F32_ARITH runs at the BASELINE dense shapes — SURVEY section 8 row f4, reference src/qp.cpp:385-386 (the float
// instantiation).  The B / W' tiles, the operand vectors and the partial sums of the iteration's two stages are single precision
// (v_pk_fma_f32, two columns per instruction); the Schur factorisation that builds the tiles, the iterates and the residual checks
// stay double (the Schur form factored in fp32 is 30-60x less accurate than the reference's float KKT path, profiles/r03_f32_tiles.json).
// A translation unit of its own: the code generated for the fp64 kernels does not depend on these being instantiated next to them.
#include <hip/hip_runtime.h>

#include "admm_wg_kernel.h"

namespace sqph {

// >0 launched, 0 no fp32-product kernel for this shape, <0 launch error
int wgf_try_launch(const KArgs<double, float> &a, hipStream_t stream, const char **name) {
    const bool checks = !(a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0));
#define SQPH_WGF_CASE(NW_, R_, C_, TR_, TC_, TW_, W_)                                                                                       \
    if (a.m <= R_ * TR_ && a.n <= C_ * TC_) {                                                                                               \
        /* the two-wave shape runs the stacked operator where m <= 104 leaves room for W' in ten tile rows (as the fp64 kernels do, wg_stack.hip) */      \
        constexpr bool CAN_STACK = NW_ == 2;                                                                                                \
        const bool stack = CAN_STACK && a.m <= R_ * (TR_ + TW_ - 1) - C_ * TC_;                                                                  \
        if (stack && checks)                                                                                                                \
            hipLaunchKernelGGL((admm_wgf_kernel<float, NW_, R_, C_, TR_, TC_, TW_, W_, CAN_STACK>), dim3(a.batch), dim3(64 * NW_), 0, stream, a); \
        else if (stack)                                                                                                                     \
            hipLaunchKernelGGL((admm_wgf_nocheck_kernel<float, NW_, R_, C_, TR_, TC_, TW_, wg_nocheck_wpe(NW_, R_, C_, TR_, TC_, W_), CAN_STACK>), dim3(a.batch), dim3(64 * NW_), 0, stream, a); \
        else if (checks)                                                                                                                    \
            hipLaunchKernelGGL((admm_wgf_kernel<float, NW_, R_, C_, TR_, TC_, TW_, W_>), dim3(a.batch), dim3(64 * NW_), 0, stream, a);       \
        else                                                                                                                                \
            hipLaunchKernelGGL((admm_wgf_nocheck_kernel<float, NW_, R_, C_, TR_, TC_, TW_, wg_nocheck_wpe(NW_, R_, C_, TR_, TC_, W_)>), dim3(a.batch), dim3(64 * NW_), 0, stream, a); \
        *name = stack ? "wg" #NW_ "_" #R_ "x" #C_ "_" #TR_ "x" #TC_ "s_w" #W_ "_f32" : "wg" #NW_ "_" #R_ "x" #C_ "_" #TR_ "x" #TC_ "_w" #W_ "_f32"; \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                                    \
    }
    SQPH_WGF_SHAPES(SQPH_WGF_CASE)
#undef SQPH_WGF_CASE
    return 0;
}

}  // namespace sqph
