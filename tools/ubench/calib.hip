// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for THIS project's access patterns
// (MI355X_MICROARCH.md: FETCH_SIZE halves wide 16 B/lane streams; other widths are uncalibrated).
// Streams 1 GiB (> 256 MiB Infinity Cache) with 8 B/lane loads, 16 B/lane loads and 8 B/lane stores.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const double *p, double *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 12345.678) out[0] = acc;
}
__global__ void read16(const double2 *p, double *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i < n; i += stride) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 12345.678) out[0] = acc;
}
__global__ void write8(double *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 1.0;
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 8;
    double *a, *o;
    hipMalloc(&a, bytes); hipMalloc(&o, 64);
    hipMemset(a, 0, bytes);
    for (int rep = 0; rep < 2; rep++) {
        read8<<<2048, 256>>>(a, o, n);
        read16<<<2048, 256>>>((const double2 *)a, o, n / 2);
        write8<<<2048, 256>>>(a, n);
    }
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes\n", bytes);
    return 0;
}
