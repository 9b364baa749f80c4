// placeholder: device code of the register-tiled kernels (see admm_tile.h)
#pragma once
#include "kargs.h"
namespace sqph {
#ifdef SQPH_SIM
template <typename T> inline int sim_run_tile(const KArgs<T> &) { return -1; }
#endif
}
