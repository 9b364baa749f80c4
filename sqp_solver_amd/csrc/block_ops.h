// Small device helpers shared by the kernels. Compiles for gfx950 with hipcc and, under
// SQPH_SIM, for the host-side SIMT emulator used by the CPU unit tests (tests/sim/).
#pragma once
#ifndef SQPH_SIM
#include <hip/hip_runtime.h>
#endif

namespace sqph {

template <typename T> struct Num;
template <> struct Num<double> {
    static __host__ __device__ constexpr double eps() { return 2.220446049250313e-16; }
};
template <> struct Num<float> {
    static __host__ __device__ constexpr float eps() { return 1.1920929e-07f; }
};

// NaN-propagating max of non-negative values (|.|_inf must not hide a NaN iterate)
template <typename T>
__device__ __forceinline__ T nanmax(T a, T b) {
    return (b > a || b != b) ? b : a;
}
template <typename T>
__device__ __forceinline__ T tabs(T a) {
    return a < T(0) ? -a : a;
}

// rho_vec_update, reference src/qp.cpp:296-314 (ConstraintType order of qp.hpp:134: 0 ineq, 1 eq, 2 loose)
template <typename T>
__device__ __forceinline__ T rho_for_type(int ctype, T rho0, T rho_min, T eq_factor) {
    return ctype == 2 ? rho_min : (ctype == 1 ? eq_factor * rho0 : rho0);
}

// Block-wide NaN-propagating max of K values per thread; result replicated to every thread.
// red: LDS scratch of K*blockDim.x elements. Ends with a barrier.
template <typename T, int K>
__device__ __forceinline__ void block_nanmax(T (&v)[K], T *red) {
    const int tid = threadIdx.x, nt = blockDim.x;
#pragma unroll
    for (int k = 0; k < K; k++) red[k * nt + tid] = v[k];
    __syncthreads();
    for (int s = nt >> 1; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int k = 0; k < K; k++) red[k * nt + tid] = nanmax(red[k * nt + tid], red[k * nt + tid + s]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = red[k * nt];
    __syncthreads();
}

}  // namespace sqph
