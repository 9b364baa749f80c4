// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Minimal host-side SIMT emulator so the HIP kernels under sqp_solver_amd/csrc/ can be
// executed lane-for-lane on a machine without a GPU (the build container has none).
// Every work-item of a workgroup is a ucontext fiber; __syncthreads() and the cross-lane
// primitives of wave_ops.h are rendezvous points handled by a round-robin scheduler, so
// the emulated kernels see exactly the data-flow they see on a 64-wide wavefront.
// Compiled only into tests/sim/libsqph_sim.so (see tests/sim/Makefile); the product
// library never links or includes this file.
#pragma once
#define SQPH_SIM 1

#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void *hipStream_t;

namespace sqph_sim {

struct Lane {
    ucontext_t ctx;
    std::vector<unsigned char> stack;
    bool done = false;
    int wait = 0;  // 0 runnable, 1 block barrier, 2 wave barrier, 3 / 4 / 5 barrier of an aligned 16- / 32- / 4-lane group
};

struct Block {
    std::vector<Lane> lanes;
    ucontext_t sched;
    int cur = -1;
    std::vector<unsigned char> smem;
    std::function<void()> body;
    uint64_t xchg[1024];   // per-lane exchange slots for cross-lane ops
    uint64_t xchg2[1024];  // second operand of the matrix instruction (mfma_f64_16x16x4)
};

inline Block *&cur_block() {
    static Block *b = nullptr;
    return b;
}

inline bool &tile_quant() { static bool q = false; return q; }  // see SQPH_TILE_QUANT (admm_wg_kernel.h)

inline dim3 &tls_threadIdx() { static dim3 v; return v; }
inline dim3 &tls_blockIdx() { static dim3 v; return v; }
inline dim3 &tls_blockDim() { static dim3 v; return v; }
inline dim3 &tls_gridDim() { static dim3 v; return v; }

inline unsigned char *dyn_smem() { return cur_block()->smem.data(); }

inline void yield_wait(int kind) {
    Block *b = cur_block();
    Lane &l = b->lanes[b->cur];
    l.wait = kind;
    swapcontext(&l.ctx, &b->sched);
}

inline void fiber_entry() {
    Block *b = cur_block();
    b->body();
    b->lanes[b->cur].done = true;
    swapcontext(&b->lanes[b->cur].ctx, &b->sched);
}

// Run one workgroup of nthreads lanes.
inline void run_block(unsigned nthreads, size_t smem_bytes, const std::function<void()> &body) {
    Block blk;
    blk.lanes.resize(nthreads);
    blk.smem.assign(smem_bytes + 64, 0xFF);  // poison: uninitialised LDS reads as NaN (doubles) / -1 (ints), like stale LDS might
    blk.body = body;
    cur_block() = &blk;
    const size_t STACK = 256 * 1024;
    for (unsigned t = 0; t < nthreads; t++) {
        Lane &l = blk.lanes[t];
        l.stack.resize(STACK);
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack.data();
        l.ctx.uc_stack.ss_size = STACK;
        l.ctx.uc_link = &blk.sched;
        makecontext(&l.ctx, (void (*)())fiber_entry, 0);
    }
    for (;;) {
        bool any_alive = false, ran = false;
        for (unsigned t = 0; t < nthreads; t++) {
            Lane &l = blk.lanes[t];
            if (l.done) continue;
            any_alive = true;
            if (l.wait) continue;
            blk.cur = (int)t;
            tls_threadIdx() = dim3(t);
            swapcontext(&blk.sched, &l.ctx);
            ran = true;
        }
        if (!any_alive) break;
        // release barriers whose participants (all not-done lanes of the group) have arrived
        bool released = false;
        bool all_block = true;
        for (unsigned t = 0; t < nthreads; t++)
            if (!blk.lanes[t].done && blk.lanes[t].wait != 1) all_block = false;
        if (all_block) {
            for (unsigned t = 0; t < nthreads; t++) blk.lanes[t].wait = 0;
            released = true;
        } else {
            for (unsigned w = 0; w * 64 < nthreads; w++) {
                bool all_wave = true, any = false;
                for (unsigned t = w * 64; t < nthreads && t < (w + 1) * 64; t++) {
                    if (blk.lanes[t].done) continue;
                    any = true;
                    if (blk.lanes[t].wait != 2) all_wave = false;
                }
                if (any && all_wave) {
                    for (unsigned t = w * 64; t < nthreads && t < (w + 1) * 64; t++) blk.lanes[t].wait = 0;
                    released = true;
                }
            }
            // 16- and 32-lane groups (kernels that run several independent QPs per wave; the groups may diverge)
            for (unsigned gs = 4; gs <= 32; gs = gs == 4 ? 16 : gs * 2) {
                const int kind = gs == 16 ? 3 : gs == 32 ? 4 : 5;
                for (unsigned g = 0; g * gs < nthreads; g++) {
                    bool all_grp = true, any = false;
                    for (unsigned t = g * gs; t < nthreads && t < (g + 1) * gs; t++) {
                        if (blk.lanes[t].done) continue;
                        any = true;
                        if (blk.lanes[t].wait != kind) all_grp = false;
                    }
                    if (any && all_grp) {
                        for (unsigned t = g * gs; t < nthreads && t < (g + 1) * gs; t++) blk.lanes[t].wait = 0;
                        released = true;
                    }
                }
            }
        }
        if (!ran && !released) {
            fprintf(stderr, "hip_sim: deadlock (divergent barrier / cross-lane op)\n");
            abort();
        }
    }
    cur_block() = nullptr;
}

template <typename K, typename A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t smem_bytes, A args) {
    tls_gridDim() = grid;
    tls_blockDim() = block;
    for (unsigned bx = 0; bx < grid.x; bx++) {
        tls_blockIdx() = dim3(bx);
        run_block(block.x, smem_bytes, [&]() { kernel(args); });
    }
}

// ---- cross-lane rendezvous: every live lane of the wave publishes v, then reads lane src ----
inline uint64_t wave_exchange(uint64_t v, int src_lane_in_wave) {
    Block *b = cur_block();
    const int t = b->cur;
    const int base = t & ~63;
    b->xchg[t] = v;
    yield_wait(2);
    int s = base + (src_lane_in_wave & 63);
    uint64_t r = (s < (int)b->lanes.size()) ? b->xchg[s] : 0;
    yield_wait(2);
    return r;
}

// v_mfma_f64_16x16x4_f64: D(16x16) = C + A(16x4) B(4x16) over one wavefront; lane l passes A[l & 15][l >> 4] and B[l >> 4][l & 15] and
// holds D[(l >> 4) + 4 q][l & 15], q < 4 (layout verified on the MI355X, tools/ubench/mfma_f64.hip).  Every lane of the wave takes part.
inline void mfma_f64_16x16x4(double a, double b, double (&c)[4]) {
    Block *blk = cur_block();
    const int t = blk->cur, base = t & ~63, l = t & 63;
    memcpy(&blk->xchg[t], &a, 8);
    memcpy(&blk->xchg2[t], &b, 8);
    yield_wait(2);
    for (int q = 0; q < 4; q++) {
        const int i = (l >> 4) + 4 * q, j = l & 15;
        double acc = c[q];
        for (int k = 0; k < 4; k++) {
            double av, bv;
            memcpy(&av, &blk->xchg[base + i + 16 * k], 8);
            memcpy(&bv, &blk->xchg2[base + 16 * k + j], 8);
            acc = __builtin_fma(av, bv, acc);
        }
        c[q] = acc;
    }
    yield_wait(2);
}

// statically allocated __shared__ arrays are plain statics here: kernels poison them on entry (thread 0 runs first)
inline void poison_static_lds(void *p, size_t bytes) {
    if (cur_block()->cur == 0) memset(p, 0xFF, bytes);
}

// the same inside an aligned group of 16 lanes (groups of one wave may have diverged)
template <int GS>
inline void group_sync() {
    static_assert(GS == 4 || GS == 16 || GS == 32, "group size");
    yield_wait(GS == 16 ? 3 : GS == 32 ? 4 : 5);
}
template <int GS>
inline uint64_t group_exchange(uint64_t v, int src_lane_in_group) {
    Block *b = cur_block();
    const int t = b->cur;
    b->xchg[t] = v;
    group_sync<GS>();
    const uint64_t r = b->xchg[(t & ~(GS - 1)) + (src_lane_in_group & (GS - 1))];
    group_sync<GS>();
    return r;
}
inline void group16_sync() { group_sync<16>(); }
inline uint64_t group16_exchange(uint64_t v, int src) { return group_exchange<16>(v, src); }

}  // namespace sqph_sim

#define threadIdx (::sqph_sim::tls_threadIdx())
#define blockIdx (::sqph_sim::tls_blockIdx())
#define blockDim (::sqph_sim::tls_blockDim())
#define gridDim (::sqph_sim::tls_gridDim())

inline void __syncthreads() { ::sqph_sim::yield_wait(1); }
