// Resource-usage probe: instantiates ONE kernel shape so that hipcc -Rpass-analysis=kernel-resource-usage answers in seconds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c -Rpass-analysis=kernel-resource-usage \
//         -DKRES_SHAPE="2,16,8,7,7,4,2" tools/kres/kres.hip -o /dev/null
#include <hip/hip_runtime.h>
#include "../../sqp_solver_amd/csrc/admm_wg_kernel.h"
#ifndef KRES_TIN
#define KRES_TIN double
#endif
template __global__ void sqph::admm_wg_kernel<KRES_TIN, KRES_SHAPE>(sqph::KArgs<double, KRES_TIN>);
