// Register-tiled wave-per-QP ADMM kernels (fast path). Placeholder until the tiled kernels land:
// tile_try_launch returns 0 ("shape not covered") so every shape takes the generic kernel.
#pragma once
#include "kargs.h"

namespace sqph {

// >0: launched (name set), 0: shape not covered, <0: launch error
template <typename T>
inline int tile_try_launch(const KArgs<T> &, hipStream_t, const char **) {
    return 0;
}

}  // namespace sqph
