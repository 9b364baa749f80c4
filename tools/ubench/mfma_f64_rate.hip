// v_mfma_f64_16x16x4_f64 on gfx950: cycles per MFMA for NACC round-robin accumulators (1 = dependent chain), operands from LDS or
// registers, one or two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f64_rate.hip -o tools/ubench/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDSOP>
__global__ __launch_bounds__(512) void rate(unsigned long long *out, double *sink, int iters) {
    __shared__ double buf[4096];
    const int l = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) buf[e] = 1e-3 * e;
    __syncthreads();
    d4 c[NACC];
    for (int k = 0; k < NACC; k++) c[k] = d4{0, 0, 0, 0};
    double a = l * 1e-3, b = 1.0 - l * 1e-4;
    const double *p = buf + l;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < NACC; k++) {
            if constexpr (LDSOP) {
                a = p[(64 * k + 128 * (it & 7)) & 4095];
                b = p[(64 * k + 128 * (it & 7) + 2048) & 4095];
            }
            c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < NACC; k++) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (l == 0) { out[2 * (blockIdx.x * 8 + (threadIdx.x >> 6))] = t0; out[2 * (blockIdx.x * 8 + (threadIdx.x >> 6)) + 1] = t1; }
}
template <int NACC, bool LDSOP>
void run(const char *name, int threads) {
    unsigned long long *out; double *sink;
    hipMalloc(&out, 8 * 256 * 16); hipMalloc(&sink, 8 * 1024 * 512);
    const int iters = 2000;
    rate<NACC, LDSOP><<<256, threads>>>(out, sink, iters);
    rate<NACC, LDSOP><<<256, threads>>>(out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), out, 8 * 256 * 16, hipMemcpyDeviceToHost);
    double t = 0;   // per block: last end - first start over its waves
    for (int bq = 0; bq < 256; bq++) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < threads / 64; w++) { lo = std::min(lo, h[2 * (bq * 8 + w)]); hi = std::max(hi, h[2 * (bq * 8 + w) + 1]); }
        t += (double)(hi - lo);
    }
    t /= 256;
    printf("%-28s waves/CU %d: %.1f cycles per MFMA per wave (%.2f FMA/clk/SIMD aggregate)\n", name, threads / 64, t / (iters * NACC),
           1024.0 * iters * NACC * (threads / 256.0 < 1 ? 1 : threads / 256.0) / t);
    hipFree(out); hipFree(sink);
}
int main() {
    for (int threads : {64, 256, 512}) {
        run<1, false>("1 acc (dependent), reg ops", threads);
        run<2, false>("2 acc, reg ops", threads);
        run<4, false>("4 acc, reg ops", threads);
        run<5, false>("5 acc, reg ops", threads);
        run<5, true>("5 acc, LDS ops", threads);
        run<10, true>("10 acc, LDS ops", threads);
    }
    return 0;
}
