// v_fmac_f64 with the DPP row_newbcast control on gfx950: semantics (which operand is broadcast, from which lane) and issue rate
// against the plain v_fmac_f64.   hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_f64.hip -o tools/ubench/dpp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void sem(const double *a, const double *b, double *out) {
    const int l = threadIdx.x;
    double x = a[l], y = b[l], acc = 1000.0;
    // expectation: acc += x(lane 16*(l/16) + 5) * y(lane l)
    asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
    out[l] = acc;
}

template <int MODE>
__global__ __launch_bounds__(64) void rate(unsigned long long *out, double *sink, int iters) {
    const int l = threadIdx.x;
    double f[8], g = 0.999999 + l * 1e-9, h = 1e-9 * (l + 1);
    for (int k = 0; k < 8; k++) f[k] = 1.0 + 0.1 * k;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define FM(K) "v_fmac_f64_e32 %" #K ", %8, %9\n"
#define FD(K, N) "v_fmac_f64_dpp %" #K ", %8, %9 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n"
        if constexpr (MODE == 0) {
            asm volatile(FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7)
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(g), "v"(h));
        } else {
            asm volatile(FD(0, 0) FD(1, 1) FD(2, 2) FD(3, 3) FD(4, 4) FD(5, 5) FD(6, 6) FD(7, 7) FD(0, 8) FD(1, 9) FD(2, 10) FD(3, 11) FD(4, 12) FD(5, 13) FD(6, 14) FD(7, 15)
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(g), "v"(h));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < 8; k++) s += f[k];
    if (l == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + l] = s;
}

int main() {
    double *a, *b, *o;
    hipMalloc(&a, 64 * 8); hipMalloc(&b, 64 * 8); hipMalloc(&o, 64 * 8);
    std::vector<double> ha(64), hb(64), ho(64);
    for (int l = 0; l < 64; l++) { ha[l] = l + 1; hb[l] = 0.5 * (l + 1); }
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, a, b, o);
    hipMemcpy(ho.data(), o, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const double want = 1000.0 + ha[16 * (l / 16) + 5] * hb[l];
        if (ho[l] != want) { if (bad < 4) printf("lane %d got %g want %g\n", l, ho[l], want); bad++; }
    }
    printf("semantics acc += src0[lane 16*(l/16)+N] * src1[l]: %s\n", bad ? "MISMATCH" : "OK");
    const int iters = 2000;
    for (int waves : {1, 1024, 2048, 4096}) {
        unsigned long long *t; double *sink;
        hipMalloc(&t, waves * 8); hipMalloc(&sink, (size_t)waves * 64 * 8);
        std::vector<unsigned long long> ht(waves);
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(waves), dim3(64), 0, 0, t, sink, iters);
            else hipLaunchKernelGGL(rate<1>, dim3(waves), dim3(64), 0, 0, t, sink, iters);
            hipDeviceSynchronize();
            hipMemcpy(ht.data(), t, waves * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : ht) s += (double)v;
            printf("%-22s waves %5d: %.1f ticks per 16 instructions (%.2f per instruction)\n", mode ? "v_fmac_f64_dpp newbcast" : "v_fmac_f64 plain", waves, s / waves / iters, s / waves / iters / 16);
        }
        hipFree(t); hipFree(sink);
    }
    return bad != 0;
}
