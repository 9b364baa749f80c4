// Micro-benchmarks of per-instruction issue cost on gfx950 for the instruction mix of the ADMM tile kernel.
// One wave per SIMD (grid = #CUs*4 blocks of 64 threads would need placement control; we simply launch
// `waves` blocks of 64 threads and time inside the wave with s_memtime).  Prints cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int TEST>
__global__ __launch_bounds__(64) void k(unsigned long long *out, double *sink, int iters) {
    double a0 = threadIdx.x * 1.0, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
    double b = 0.999999, c = 1e-9;
    int i0 = threadIdx.x, i1 = threadIdx.x * 3, i2 = 5, i3 = 7;
    __shared__ double lds[64 * 8];
    for (int e = threadIdx.x; e < 64 * 8; e += 64) lds[e] = e;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (TEST == 0) {  // 8 independent f64 FMA chains
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                               "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (TEST == 1) {  // dependent f64 FMA chain
            REP16(asm volatile(REP4("v_fma_f64 %0, %0, %1, %2\n") REP4("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));)
        } else if constexpr (TEST == 2) {  // v_cndmask_b32 e64 with SGPR mask, independent
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                               "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0) : "vcc");)
        } else if constexpr (TEST == 3) {  // dpp quad_perm mov, independent
            REP16(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));)
        } else if constexpr (TEST == 4) {  // dpp row_ror:8
            REP16(asm volatile(REP4("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n")
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));)
        } else if constexpr (TEST == 5) {  // permlane32_swap
            REP16(asm volatile(REP4("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));)
        } else if constexpr (TEST == 6) {  // permlane16_swap
            REP16(asm volatile(REP4("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));)
        } else if constexpr (TEST == 7) {  // accvgpr read/write pair
            REP16(asm volatile(REP4("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %1, a0\n") : "+v"(i0), "+v"(i1) : : "a0");)
        } else if constexpr (TEST == 8) {  // v_add_f64 independent
            REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                               "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
        } else if constexpr (TEST == 9) {  // ds_bpermute + wait (latency-bound pairs)
            REP16(asm volatile(REP4("ds_bpermute_b32 %0, %2, %0\n ds_bpermute_b32 %1, %2, %1\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(i0), "+v"(i1) : "v"(i2));)
        } else if constexpr (TEST == 10) {  // v_mov_b32
            REP16(asm volatile(REP4("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));)
        } else if constexpr (TEST == 11) {  // f64 FMA interleaved 1:1 with cndmask (do they overlap?)
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_cndmask_b32 %4, %4, %5, vcc\n v_fma_f64 %1, %1, %8, %9\n v_cndmask_b32 %5, %5, %6, vcc\n"
                               "v_fma_f64 %2, %2, %8, %9\n v_cndmask_b32 %6, %6, %7, vcc\n v_fma_f64 %3, %3, %8, %9\n v_cndmask_b32 %7, %7, %4, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(b), "v"(c) : "vcc");)
        } else if constexpr (TEST == 12) {  // ds_read_b64 stride-1, 8 in flight then wait
            double r0, r1, r2, r3, r4, r5, r6, r7;
            int addr = threadIdx.x * 8;
            REP16(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                               "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr)); a0 += r0 + r7;)
        } else if constexpr (TEST == 13) {  // v_readlane_b32
            int s;
            REP16(asm volatile(REP4("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %2, 5\n") : "=s"(s) : "v"(i0), "v"(i1)); i2 += s;)
        } else if constexpr (TEST == 14) {  // v_mul_f64
            REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                               "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if constexpr (TEST == 15) {  // dependent chain: cndmask -> dpp -> add_f64 (the rs8 pattern, 1 value)
            REP16(asm volatile(REP4("v_cndmask_b32 %1, %0, %1, vcc\n s_nop 1\n v_mov_b32_dpp %2, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32 %0, %2, %1\n")
                               : "+v"(i0), "+v"(i1), "+v"(i2) : : "vcc");)
        } else if constexpr (TEST == 17) {  // v_cndmask_b32_e64 with SGPR-pair mask, 4 independent chains
            unsigned long long msk = 0x5555555555555555ull + it;
            REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5\n"
                               "v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5\n"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0), "s"(msk));)
        } else if constexpr (TEST == 18) {  // cndmask e64 writing fresh outputs (no RAW on dst)
            unsigned long long msk = 0x5555555555555555ull + it;
            int o0, o1, o2, o3;
            REP16(asm volatile("v_cndmask_b32_e64 %0, %4, %5, %8\n v_cndmask_b32_e64 %1, %5, %6, %8\n v_cndmask_b32_e64 %2, %6, %7, %8\n v_cndmask_b32_e64 %3, %7, %4, %8\n"
                               "v_cndmask_b32_e64 %0, %4, %6, %8\n v_cndmask_b32_e64 %1, %5, %7, %8\n v_cndmask_b32_e64 %2, %6, %4, %8\n v_cndmask_b32_e64 %3, %7, %5, %8\n"
                               : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "s"(msk)); i2 += o0 + o3;)
        } else if constexpr (TEST == 19) {  // chained operands
            unsigned long long msk = 0x5555555555555555ull + it;
            REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %4\n v_cndmask_b32_e64 %1, %1, %2, %4\n v_cndmask_b32_e64 %2, %2, %3, %4\n v_cndmask_b32_e64 %3, %3, %0, %4\n"
                               "v_cndmask_b32_e64 %0, %0, %1, %4\n v_cndmask_b32_e64 %1, %1, %2, %4\n v_cndmask_b32_e64 %2, %2, %3, %4\n v_cndmask_b32_e64 %3, %3, %0, %4\n"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "s"(msk));)
        } else if constexpr (TEST == 20) {  // v_add_u32 dependent chain
            REP16(asm volatile(REP4("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n") : "+v"(i0) : "v"(i1));)
        } else if constexpr (TEST == 21) {  // v_mov_b64
            REP16(asm volatile(REP4("v_mov_b64 %0, %1\n v_mov_b64 %2, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if constexpr (TEST == 16) {  // v_fmac_f64 with one operand from ds_read_b64 each (LDS-fed FMA, 8 indep)
            int addr = threadIdx.x * 8;
            double r0, r1, r2, r3;
            REP16(asm volatile("ds_read_b64 %4, %8\n ds_read_b64 %5, %8 offset:512\n ds_read_b64 %6, %8 offset:1024\n ds_read_b64 %7, %8 offset:1536\n s_waitcnt lgkmcnt(0)\n"
                               "v_fma_f64 %0, %4, %9, %0\n v_fma_f64 %1, %5, %9, %1\n v_fma_f64 %2, %6, %9, %2\n v_fma_f64 %3, %7, %9, %3\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr), "v"(b));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + lds[threadIdx.x];
}

template <int TEST>
void run(const char *name, int ninstr_per_iter, int blocks) {
    unsigned long long *out;
    double *sink;
    hipMalloc(&out, blocks * 8);
    hipMalloc(&sink, blocks * 64 * 8);
    const int iters = 200;
    k<TEST><<<blocks, 64>>>(out, sink, iters);
    k<TEST><<<blocks, 64>>>(out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    // s_memtime ticks at 100 MHz constant on gfx9; convert using measured shader clock ~ (report raw too)
    printf("%-44s blocks=%5d  ticks/instr=%.4f\n", name, blocks, avg / ((double)iters * ninstr_per_iter));
    hipFree(out);
    hipFree(sink);
}

int main() {
    for (int blocks : {1024, 2048, 4096}) {  // 1, 2, 4 waves per SIMD
        run<0>("v_fma_f64 x8 independent", 128, blocks);
        run<1>("v_fma_f64 dependent chain", 128, blocks);
        run<8>("v_add_f64 x4 independent", 128, blocks);
        run<14>("v_mul_f64 x4 independent", 128, blocks);
        run<2>("v_cndmask_b32 (vcc) x4 indep", 128, blocks);
        run<17>("v_cndmask_b32_e64 sgpr mask x4 indep", 128, blocks);
        run<18>("v_cndmask_b32_e64 fresh dst", 128, blocks);
        run<19>("v_cndmask_b32_e64 chained operands", 128, blocks);
        run<20>("v_add_u32 dependent", 128, blocks);
        run<21>("v_mov_b64", 128, blocks);
        run<10>("v_mov_b32", 128, blocks);
        run<3>("v_mov_b32_dpp quad_perm", 128, blocks);
        run<4>("v_mov_b32_dpp row_ror:8", 128, blocks);
        run<5>("v_permlane32_swap", 128, blocks);
        run<6>("v_permlane16_swap", 128, blocks);
        run<7>("v_accvgpr write+read", 128, blocks);
        run<13>("v_readlane_b32", 128, blocks);
        run<9>("ds_bpermute_b32 (8 then wait)", 128, blocks);
        run<12>("ds_read_b64 (8 then wait)", 128, blocks);
        run<11>("fma_f64 + cndmask 1:1 (pair count)", 128, blocks);
        run<15>("dep chain cndmask->dpp->add (x3 instr)", 192, blocks);
        run<16>("4x(ds_read_b64)+wait+4xfma (8 instr)", 128, blocks);
    }
    return 0;
}
