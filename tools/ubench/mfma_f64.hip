// v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 on gfx950: operand layout check, issue rate, dependent latency, and whether
// MFMAs overlap with f64 VALU FMAs of the same wave.   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f64.hip -o tools/ubench/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// D = A(16x4) B(4x16): expected lane l holds A[l&15][l>>4], B[l>>4][l&15]; D reg q of lane l = D[(l>>4) + 4 q][l & 15]
__global__ void layout16(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int q = 0; q < 4; q++) D[((l >> 4) + 4 * q) * 16 + (l & 15)] = c[q];
}
// 4 blocks of 4x4x4: raw dump — lane l passes a = l+1, b = 100+l
__global__ void layout4(double *Dout) {
    const int l = threadIdx.x;
    double c = 0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64((double)(l + 1), (double)(1 << (l & 15)) , c, 0, 0, 0);
    Dout[l] = c;
}
template <int MODE>
__global__ __launch_bounds__(64) void rate(unsigned long long *out, double *sink, int iters) {
    const int l = threadIdx.x;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = l * 1e-3, b = 1.0 - l * 1e-4, f0 = 1.0, f1 = 1.1, f2 = 1.2, f3 = 1.3, g = 0.999999, h = 1e-9, e = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {  // 4 independent accumulators
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else if constexpr (MODE == 1) {  // dependent chain
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        } else if constexpr (MODE == 2) {  // 4 MFMAs + 16 independent f64 VALU FMAs
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));
        } else if constexpr (MODE == 3) {  // 16 VALU FMAs alone (reference for MODE 2)
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));
        } else if constexpr (MODE == 4) {  // 4x4x4 4-block, 4 independent
            e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e, 0, 0, 0);
            f0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f1, 0, 0, 0);
            f2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f2, 0, 0, 0);
        } else if constexpr (MODE == 5) {  // 4x4x4 dependent
            e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + e;
    if (l == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    // layout check with asymmetric A, B
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i * 0.37 + k * 1.91 + i * k * 0.013;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 - k * 0.53 + j * 0.29 + k * j * 0.007;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 256 * 8);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    layout16<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int e = 0; e < 256; e++) err = fmax(err, fabs(D[e] - R[e]));
    printf("16x16x4 layout (A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4q][l&15]): max err %.3e %s\n", err, err < 1e-12 ? "OK" : "MISMATCH");
    layout4<<<1, 64>>>(dD);
    hipMemcpy(D.data(), dD, 64 * 8, hipMemcpyDeviceToHost);
    printf("4x4x4_4b raw: lane l passes a=l+1, b=2^(l&15); results per lane:\n");
    for (int l = 0; l < 64; l++) printf("%s%8.0f", l % 8 ? " " : "\n  ", D[l]);
    printf("\n");
    unsigned long long *dout; double *sink;
    const int NB = 1024;  // one wave per SIMD when NB == #SIMDs
    hipMalloc(&dout, NB * 8); hipMalloc(&sink, NB * 64 * 8);
    std::vector<unsigned long long> out(NB);
    const int iters = 4096;
    const char *names[] = {"16x16x4 x4 independent", "16x16x4 x4 dependent", "4 MFMA + 16 VALU f64 FMA", "16 VALU f64 FMA alone", "4x4x4_4b x4 independent", "4x4x4_4b x4 dependent"};
    for (int mode = 0; mode < 6; mode++) {
        for (int nb : {1, 1024, 2048}) {
            switch (mode) {
                case 0: rate<0><<<nb, 64>>>(dout, sink, iters); break;
                case 1: rate<1><<<nb, 64>>>(dout, sink, iters); break;
                case 2: rate<2><<<nb, 64>>>(dout, sink, iters); break;
                case 3: rate<3><<<nb, 64>>>(dout, sink, iters); break;
                case 4: rate<4><<<nb, 64>>>(dout, sink, iters); break;
                case 5: rate<5><<<nb, 64>>>(dout, sink, iters); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(out.data(), dout, nb * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (int b = 0; b < nb; b++) s += out[b];
            printf("%-28s waves %5d: %.1f ticks per loop body (s_memtime ticks; 100 MHz clock => x24 shader cycles?)\n", names[mode], nb, s / nb / iters);
        }
    }
    return 0;
}
