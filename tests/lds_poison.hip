// Test tool: fills the LDS of every CU with a byte pattern (0xFF..: NaN as doubles, -1 as ints).  LDS is not cleared between kernels
// (nor between processes), so a kernel that reads a word it never wrote sees what the CU's previous workgroup left there; with this
// launched before every solver call of the GPU suite (tests/conftest.py) such a read shows up as NaN instead of depending on the
// test order.  (Round 6: the 32- and 64-row grids of admm_wg_msetup.h::build_B read the pad columns of the staged A block.)
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(1024) lds_poison_kernel(unsigned pat, int words, unsigned *sink) {
    extern __shared__ unsigned sm[];
    // pat == 1: a different pseudo-random word everywhere (random finite doubles, infinities, NaNs and integers of every size — what
    // another tenant's kernel may have left), re-seeded per call by `words >> 20`
    const unsigned seed = (unsigned)words >> 20;
    words &= 0xFFFFF;
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ (blockIdx.x * 40503u + seed * 2246822519u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        sm[i] = pat == 1u ? h : pat;
    }
    __syncthreads();
    if (sink && sm[(threadIdx.x * 7) % words] == 12345u) *sink = 1;
}
static unsigned g_calls = 0;
extern "C" int lds_poison(unsigned pat, int bytes) {
    hipError_t e = hipFuncSetAttribute((const void *)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    // 160 KB per workgroup = one workgroup per CU at a time; four rounds over the 256 CUs so that every CU takes at least one
    hipLaunchKernelGGL(lds_poison_kernel, dim3(256 * 4), dim3(1024), bytes, 0, pat, (bytes / 4) | (int)((g_calls++ & 0x7FF) << 20), (unsigned *)nullptr);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceSynchronize();
}
