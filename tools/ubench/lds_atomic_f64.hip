// LDS fp64 atomic add (ds_add_f64, no return) rate on gfx950: cycles per wave-instruction, one workgroup of NW wavefronts per CU, as a function
// of the number of ACTIVE lanes per instruction and of the address pattern.  (Round 6: the block-row kernel's S phase issues ~1,400 of
// them per QP in ~85 k cycles — is that the unit's rate, and does it scale with the active lanes?)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic_f64.hip -o /tmp/lds_atomic && /tmp/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void lds_add(double *p, double v) {
    asm volatile("ds_add_f64 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v) : "memory");
}
// mode 0: ds_add_f64; mode 1: ds_write_b64 (for comparison); mode 2: ds_read_b64
template <int MODE>
__global__ void k(int active, int stride, int iters, unsigned long long *out, double *sink) {
    extern __shared__ double sm[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 8192; i += blockDim.x) sm[i] = 0;
    __syncthreads();
    // address: a private 1,024-double slice per wavefront; lane l -> l * stride (mod 1,024)
    double *base = sm + 1024 * wave;
    const bool on = lane < active;
    double acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            double *p = base + ((lane * stride + 37 * u + it) & 1023);
            if (on) {
                if (MODE == 0) lds_add(p, 1.0);
                else if (MODE == 1) *(volatile double *)p = 1.0;
                else acc += *(volatile double *)p;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 123.456) sink[0] = acc;
}
int main() {
    unsigned long long *out;
    double *sink;
    hipMalloc(&out, 1024 * 8);
    hipMalloc(&sink, 64);
    const int iters = 200;
    const char *names[3] = {"ds_add_f64", "ds_write_b64", "ds_read_b64"};
    for (int mode = 0; mode < 3; mode++)
        for (int nw : {1, 4, 8})
            for (int stride : {1, 17})
                for (int active : {64, 32, 16, 5, 1}) {
                    void (*fn)(int, int, int, unsigned long long *, double *) = mode == 0 ? k<0> : mode == 1 ? k<1> : k<2>;
                    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(fn, dim3(256), dim3(64 * nw), 65536, 0, active, stride, iters, out, sink);
                    hipDeviceSynchronize();
                    std::vector<unsigned long long> h(256);
                    hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
                    double s = 0;
                    for (auto v : h) s += (double)v;
                    s /= 256;
                    // (s_memtime counts shader clocks on gfx9: 1.58 M ticks for a 0.72 ms QP in tools/phase_timing_csb.py)
                    const double instr = (double)iters * 8 * nw;
                    printf("%-13s waves %d stride %2d active %2d: %9.0f cycles, %.1f per wave-instruction CU-wide, %.2f per active lane\n",
                           names[mode], nw, stride, active, s, s / instr, s / instr / active);
                }
    return 0;
}
