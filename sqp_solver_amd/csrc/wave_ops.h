// Cross-lane primitives for the wave-per-QP kernels (64-wide wavefront, 8 x 8 lane grid).
//
// lane = 8*c + r :  r = lane & 7 (row group, contiguous lanes), c = lane >> 3 (column group,
// lanes 8 apart).  Two 8-lane communicators exist per lane:
//     "row-of-grid" group  = lanes sharing c (contiguous, index r): xor masks 1, 2, 4
//     "col-of-grid" group  = lanes sharing r (stride 8,  index c): xor masks 8, 16, 32
// rs8  = reduce-scatter of 8 values inside one group (lane with index g ends with sum of v[g]),
// ag8  = all-gather (every lane of the group ends with the 8 values, ordered by index).
// Both are 3-stage butterflies on whole VGPRs: no LDS storage is touched.
//
// Under SQPH_SIM the exchange is routed through the host SIMT emulator (tests/sim/hip_sim.h).
#pragma once
#ifndef SQPH_SIM
#include <hip/hip_runtime.h>
#endif

namespace sqph {

#ifdef SQPH_SIM
template <typename T>
inline T lane_xor(T v, int mask) {
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const int lane = (int)(threadIdx.x & 63);
    uint64_t r = ::sqph_sim::wave_exchange(bits, lane ^ mask);
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
inline double tfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
inline float tfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#else
// ---- gfx950 implementations -----------------------------------------------------------------
// DPP controls (VOP_DPP): quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E, row_ror:n = 0x120+n,
// row_half_mirror = 0x141.
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int MASK>
__device__ __forceinline__ int xor_i32(int v) {
    if constexpr (MASK == 1) {
        return dpp_mov_i32<0xB1>(v);
    } else if constexpr (MASK == 2) {
        return dpp_mov_i32<0x4E>(v);
    } else if constexpr (MASK == 4) {
        // lane^4 == quad-reverse( half-mirror(lane) ): i^7^3
        return dpp_mov_i32<0x1B>(dpp_mov_i32<0x141>(v));
    } else if constexpr (MASK == 8) {
        return dpp_mov_i32<0x128>(v);  // row_ror:8 inside each row of 16
    } else if constexpr (MASK == 16) {
        // v_permlane16_swap: swaps odd rows of the first operand with even rows of the second
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        // r[0] = {row0: v.row0, row1: v.row0, row2: v.row2, row3: v.row2}; r[1] = {v.row1, v.row1, v.row3, v.row3}
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        return (lane & 16) ? (int)r[0] : (int)r[1];
    } else if constexpr (MASK == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        return (lane & 32) ? (int)r[0] : (int)r[1];
    } else {
        return __shfl_xor(v, MASK);
    }
}
template <int MASK>
__device__ __forceinline__ double lane_xor_c(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = xor_i32<MASK>(lo);
    hi = xor_i32<MASK>(hi);
    return __hiloint2double(hi, lo);
}
template <int MASK>
__device__ __forceinline__ float lane_xor_c(float v) {
    return __int_as_float(xor_i32<MASK>(__float_as_int(v)));
}
template <typename T>
__device__ __forceinline__ T lane_xor(T v, int mask) {
    return __shfl_xor(v, mask);
}
__device__ __forceinline__ double tfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float tfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#endif

// partner exchange with a compile-time xor mask
template <int MASK, typename T>
__device__ __forceinline__ T xchg(T v) {
#if defined(SQPH_SIM) || defined(SQPH_XCHG_BPERMUTE)
    return lane_xor<T>(v, MASK);
#else
    return lane_xor_c<MASK>(v);
#endif
}

// partner exchange inside an aligned group of 16 lanes (masks 1, 2, 4, 8 never leave the group on the device; the
// emulator needs to know, because the four groups of a wave may have diverged)
template <int MASK, typename T>
__device__ __forceinline__ T xchg16(T v) {
    static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8, "stays inside 16 lanes");
#ifdef SQPH_SIM
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const uint64_t r = ::sqph_sim::group16_exchange(bits, (int)(threadIdx.x & 15) ^ MASK);
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
#else
    return xchg<MASK>(v);
#endif
}
// NaN-propagating max over the 16 lanes of a group (all of them end with the result)
template <typename T>
__device__ __forceinline__ T group16_nanmax(T v) {
#define SQPH_GM_STEP(M)                       \
    {                                         \
        const T o = xchg16<M>(v);             \
        v = (o > v || o != o) ? o : v;        \
    }
    SQPH_GM_STEP(1) SQPH_GM_STEP(2) SQPH_GM_STEP(4) SQPH_GM_STEP(8)
#undef SQPH_GM_STEP
    return v;
}

// the same for an aligned group of 32 lanes (two QPs per wavefront)
template <typename T>
__device__ __forceinline__ T group32_nanmax(T v) {
#ifdef SQPH_SIM
    for (int M = 1; M < 32; M <<= 1) {
        uint64_t bits = 0;
        memcpy(&bits, &v, sizeof(T));
        const uint64_t r = ::sqph_sim::group_exchange<32>(bits, (int)(threadIdx.x & 31) ^ M);
        T o;
        memcpy(&o, &r, sizeof(T));
        v = (o > v || o != o) ? o : v;
    }
    return v;
#else
#define SQPH_GM_STEP(M)                       \
    {                                         \
        const T o = xchg<M>(v);               \
        v = (o > v || o != o) ? o : v;        \
    }
    SQPH_GM_STEP(1) SQPH_GM_STEP(2) SQPH_GM_STEP(4) SQPH_GM_STEP(8) SQPH_GM_STEP(16)
#undef SQPH_GM_STEP
    return v;
#endif
}

// reduce-scatter of v[0..7] over the 8 lanes {g ^ k*M0-ish}; g = my index in the group (bits -> masks M0,M1,M2)
template <int M0, int M1, int M2, typename T>
__device__ __forceinline__ T rs8(const T (&v)[8], int g) {
    const bool b2 = g & 4, b1 = g & 2, b0 = g & 1;
    T u[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const T send = b2 ? v[k] : v[k + 4];
        const T keep = b2 ? v[k + 4] : v[k];
        u[k] = keep + xchg<M2>(send);
    }
    T w[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const T send = b1 ? u[k] : u[k + 2];
        const T keep = b1 ? u[k + 2] : u[k];
        w[k] = keep + xchg<M1>(send);
    }
    const T send = b0 ? w[0] : w[1];
    const T keep = b0 ? w[1] : w[0];
    return keep + xchg<M0>(send);
}

// all-gather: v[k] = x of the group lane with index k
template <int M0, int M1, int M2, typename T>
__device__ __forceinline__ void ag8(T x, int g, T (&v)[8]) {
    const bool b2 = g & 4, b1 = g & 2, b0 = g & 1;
    const T p = xchg<M0>(x);
    const T a0 = b0 ? p : x, a1 = b0 ? x : p;
    const T p0 = xchg<M1>(a0), p1 = xchg<M1>(a1);
    T q[4];
    q[0] = b1 ? p0 : a0;
    q[1] = b1 ? p1 : a1;
    q[2] = b1 ? a0 : p0;
    q[3] = b1 ? a1 : p1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const T rk = xchg<M2>(q[k]);
        v[k] = b2 ? rk : q[k];
        v[k + 4] = b2 ? q[k] : rk;
    }
}

// wave-wide NaN-propagating max (all 64 lanes end with the result)
template <typename T>
__device__ __forceinline__ T wave_nanmax(T v) {
#define SQPH_WM_STEP(M)                       \
    {                                         \
        const T o = xchg<M>(v);               \
        v = (o > v || o != o) ? o : v;        \
    }
    SQPH_WM_STEP(1) SQPH_WM_STEP(2) SQPH_WM_STEP(4) SQPH_WM_STEP(8) SQPH_WM_STEP(16) SQPH_WM_STEP(32)
#undef SQPH_WM_STEP
    return v;
}


// ---- matrix instruction and 16-lane row broadcasts (MFMA set-up of the register-tiled kernels, admm_wg_msetup.h) -------------------
// four doubles of a 16 x 16 f64 MFMA accumulator: lane l holds D[(l >> 4) + 4 q][l & 15], q < 4
struct sqph_acc4 {
    double v[4];
};
// D = C + A(16 x 4) B(4 x 16): lane l passes a = A[l & 15][l >> 4], b = B[l >> 4][l & 15]   (v_mfma_f64_16x16x4_f64; measured on the
// MI355X: 64 cycles per instruction and SIMD whatever the accumulator dependence — 16 multiply-adds per cycle, the vector pipe's peak,
// from ONE instruction and two 8-byte operands per lane: tools/ubench/mfma_f64_rate.hip)
__device__ __forceinline__ void mfma16(double a, double b, sqph_acc4 &c) {
#ifdef SQPH_SIM
    ::sqph_sim::mfma_f64_16x16x4(a, b, c.v);
#else
    typedef double d4_t __attribute__((ext_vector_type(4)));
    d4_t r = {c.v[0], c.v[1], c.v[2], c.v[3]};
    r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, r, 0, 0, 0);
    c.v[0] = r[0]; c.v[1] = r[1]; c.v[2] = r[2]; c.v[3] = r[3];
#endif
}
// value of lane N of my aligned 16-lane row (DPP row_newbcast)
template <int N>
__device__ __forceinline__ double bcast16(double x) {
#ifdef SQPH_SIM
    uint64_t bits = 0;
    memcpy(&bits, &x, 8);
    const uint64_t r = ::sqph_sim::group16_exchange(bits, N);
    double out;
    memcpy(&out, &r, 8);
    return out;
#else
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x150 + N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x150 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
#endif
}
// acc + x(lane N of my 16-lane row) * y in one instruction (v_fmac_f64 with the row_newbcast control on its first source; the two wait
// states a DPP read needs behind the VALU write of its source are part of the statement: the compiler does not see into inline asm)
template <int N>
__device__ __forceinline__ double fmac_bcast16(double acc, double x, double y) {
#ifdef SQPH_SIM
    return __builtin_fma(bcast16<N>(x), y, acc);
#else
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(N));
    return acc;
#endif
}


// ---- four lanes per QP (the quad variant of the one-QP-per-lane kernel, admm_lane_kernel.h): exchanges inside an aligned quad by DPP
// quad_perm — quads of a wavefront may have diverged (different iteration counts), the lanes of one quad never do
template <int MASK, typename T>
__device__ __forceinline__ T quad_xor(T v) {  // value of lane (l ^ MASK) of my quad, MASK = 1 or 2
    static_assert(MASK == 1 || MASK == 2, "stays inside the quad");
#ifdef SQPH_SIM
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const uint64_t r = ::sqph_sim::group_exchange<4>(bits, (int)(threadIdx.x & 3) ^ MASK);
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
#else
    return xchg<MASK>(v);
#endif
}
template <int I, typename T>
__device__ __forceinline__ T quad_bcast(T v) {  // value of lane I of my quad
#ifdef SQPH_SIM
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const uint64_t r = ::sqph_sim::group_exchange<4>(bits, I);
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
#else
    constexpr int CTRL = I | (I << 2) | (I << 4) | (I << 6);  // quad_perm:[I,I,I,I]
    if constexpr (sizeof(T) == 8) {
        double d = (double)v;
        const int lo = dpp_mov_i32<CTRL>(__double2loint(d)), hi = dpp_mov_i32<CTRL>(__double2hiint(d));
        return (T)__hiloint2double(hi, lo);
    } else {
        return (T)__int_as_float(dpp_mov_i32<CTRL>(__float_as_int((float)v)));
    }
#endif
}
// the lanes of a quad run in lockstep on the device; the emulator's fibers do not — a rendezvous there, nothing here
__device__ __forceinline__ void quad_sync() {
#ifdef SQPH_SIM
    ::sqph_sim::group_sync<4>();
#endif
}

}  // namespace sqph
