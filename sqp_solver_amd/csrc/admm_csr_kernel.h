// Sparse-A ADMM kernel (BASELINE config 5: n = 200, m = 400, CSR A): ONE 1024-lane workgroup (16 wavefronts) per QP.
//
// At n = 200 the Schur factor W (S^-1 = W'W, lower triangular, 20,100 doubles) no longer fits LDS next to anything
// else, but it fits the CU's register file: the 1024 lanes form a 32 x 32 grid, lane (r,c) keeps the entries
//     w[a][b] = W[r + 32a][c + 32b]      b <= a < TT       (2-D cyclic, lower tile-triangle only: 28 doubles at TT = 7)
// in VGPRs for the whole solve.  A stays sparse and lives in LDS twice: CSR (values + 16-bit columns) for A x and a
// CSC index (row | position-in-CSR, 32 bit) for A'w, so the iteration is the four-product chain of the reference
//     t = (sigma x - q) + A' w        CSC, 1-8 lanes per column by its length (build_lane_map: no lane carries more than K entries)
//     y1 = W t                        register tile, partial sums over c staged in LDS, summed by 4 lanes per output inside the wavefront
//     x~ = W' y1                      same tile, partial sums over r
//     z~ = A x~                       CSR, 1-8 lanes per row by its length
// (the B = A W' trick of the dense kernels would densify A).  Set-up in the same launch: CSC index by counting sort,
// S = P_sym + sigma I + A'RA accumulated column panel by column panel in LDS from the sparse rows (cost ~ nnz * row
// length, not m n^2), moved into the register tile, Jacobi-scaled and eliminated in registers ([S | I] -> W in place,
// one LDS broadcast vector and one barrier per pivot).  Numerics: identical formulas to admm_generic.h
// (reference src/qp.cpp:84-144 on the Schur-ordered system), fp64 arithmetic, TIN inputs.
//
// Requirements checked by the host (csr_try_launch): n <= 32*TT, m <= 512, CSR rows sorted by column without
// duplicates, nnz small enough for LDS; anything else takes the expand-to-dense path.
#pragma once
#include "admm_generic.h"  // SQPH_DYN_SMEM
#include "admm_wg_kernel.h"  // wg_read, SQPH_OPAQUE_*
#include "block_ops.h"
#include "kargs.h"
#include "wave_ops.h"

namespace sqph {

template <typename TIN>
struct CsrArgs {
    const int *rowptr, *colind;
    const TIN *val;
    long long s_rowptr, s_colind, s_val;
    int nnz_cap;  // per-QP capacity of colind/val == LDS slots reserved
    // P in compressed-column form (sqph_csc_P), read by the block-row kernel's sparse-P instantiations only (admm_csrb_kernel.h)
    const int *p_colptr = nullptr, *p_rowind = nullptr;
    const TIN *p_val = nullptr;
    long long s_pcolptr = 0, s_prowind = 0, s_pval = 0;
};

// LDS map.  Everything whose size depends only on the tile edge TT sits first at compile-time offsets; the arrays
// sized by m and by the nnz capacity follow (five run-time offsets).  Used by the host (launch size) and the kernel.
template <int TT>
struct CsrLayout {
    static constexpr int NP = 32 * TT;
    static constexpr int CS = ((TT + 1) & ~1) + 2;  // stride of one lane-group's slice of a gathered vector (16-B aligned)
    static constexpr int SP = 34;                   // stride of one output's 32 partial sums (even: 16-B aligned quarters for b128 reads)
    static constexpr int LDP = NP + 1;              // S panel column stride
    static constexpr int ev(int x) { return (x + 1) & ~1; }
    // offsets in doubles
    static constexpr int o_stage = 0;
    static constexpr int o_tcol = ev(NP * SP > 32 * LDP ? NP * SP : 32 * LDP);
    static constexpr int o_yrow = o_tcol + 32 * CS;
    static constexpr int o_xt = o_yrow + 32 * CS;
    static constexpr int o_ux = o_xt + NP;  // sigma x - q, plain-indexed (for the lanes that sum a column of A' w)
    static constexpr int o_g = o_ux + NP;
    static constexpr int o_sj = o_g + 2 * NP + 2;
    static constexpr int o_ds = o_sj + NP;
    static constexpr int o_red = o_ds + NP;
    static constexpr int o_colptr_d = o_red + 8 * 16;          // (NP+1) ints
    static constexpr int o_ccur_d = o_colptr_d + ev(NP + 2) / 2;  // NP ints
    static constexpr int o_w = o_ccur_d + NP / 2;
    // offsets in 32-bit words
    static constexpr int o_colptr = 2 * o_colptr_d, o_ccur = 2 * o_ccur_d;
    static constexpr int o_qv = o_w;  // owner constants q [NP] ...
    int o_lo, o_up, o_rinv, o_wv, o_rho, o_val;  // ... l, u, 1/rho [MP each]; w [MP] (= rho during a factorisation); CSR values
    int o_rowptr, o_csc, o_col;                  // 32-bit words
    size_t bytes;
    __host__ __device__ static CsrLayout make(int m, int nnz_cap) {
        CsrLayout L;
        const int MP = (m + 1) & ~1;
        int d = o_qv + NP;
        L.o_lo = d; d += MP;
        L.o_up = d; d += MP;
        L.o_rinv = d; d += MP;
        L.o_wv = d; L.o_rho = d; d += MP;
        L.o_val = d; d += nnz_cap;
        int w = 2 * d;
        L.o_csc = w; w += nnz_cap;
        L.o_rowptr = w; w += m + 1;
        L.o_col = w; w += (nnz_cap + 1) / 2;
        L.bytes = (size_t)w * 4 + 16;
        return L;
    }
};

#ifdef SQPH_SIM
#define SQPH_LANE(tl) const int tl = (int)threadIdx.x
#else
#define SQPH_LANE(tl) int tl = (int)threadIdx.x; SQPH_OPAQUE_V(tl)
#endif

#ifdef SQPH_SIM
inline int lds_atomic_inc(int *p) { return (*p)++; }
inline void lds_atomic_add(int *p, int v) { *p += v; }
inline int uniform_int(int v) { return v; }
#else
__device__ __forceinline__ int lds_atomic_inc(int *p) { return atomicAdd(p, 1); }
__device__ __forceinline__ void lds_atomic_add(int *p, int v) { atomicAdd(p, v); }
// a value known to be workgroup-uniform that the compiler holds in a VGPR (e.g. returned by an out-of-line function)
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

template <typename TIN, int TT>
struct CsrKernel {
    using T = double;
    static constexpr int NT = 1024, NP = 32 * TT, NE = TT * (TT + 1) / 2;
    static constexpr int idx(int a, int b) { return a * (a + 1) / 2 + b; }

    // ---------------------------------------------------------------- register-tile products -> staging
    // W t : lane (r,c) sums over its columns; partial for output row r+32a goes to st[(r+32a)*SP + c]
    static __device__ __forceinline__ void stage_W(const T (&w)[NE], const T *tcol, T *st, int r, int c, int CS, int SP) {
        T tv[TT];
        wg_read<TT>(tcol + c * CS, tv);
#pragma unroll
        for (int a = 0; a < TT; a++) {
            T acc = 0;
#pragma unroll
            for (int b = 0; b <= a; b++) acc = wg_fma(w[idx(a, b)], tv[b], acc);
            st[(r * TT + a) * SP + c] = acc;  // the TT outputs of a row group lie in consecutive staging rows (in-wave reduction, see run())
        }
    }
    // orders the LDS operations of the calling wavefront for the compiler (the hardware executes them in order)
    static __device__ __forceinline__ void wave_lds_order() {
#ifdef SQPH_SIM
        __syncthreads();  // the emulator runs lanes as fibres: a workgroup barrier is the ordering it has
#else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
    }
    // W' y : lane (r,c) sums over its rows; partial for output column c+32b goes to st[(c+32b)*SP + r]
    static __device__ __forceinline__ void stage_WT(const T (&w)[NE], const T *yrow, T *st, int r, int c, int CS, int SP) {
        T yv[TT];
        wg_read<TT>(yrow + r * CS, yv);
#pragma unroll
        for (int b = 0; b < TT; b++) {
            T acc = 0;
#pragma unroll
            for (int a = b; a < TT; a++) acc = wg_fma(w[idx(a, b)], yv[a], acc);
            st[(c + 32 * b) * SP + ((r + c) & 31)] = acc;  // rotated inside the row: with the even stride the plain column
                                                           // index would put lanes c and c+8 on the same banks; the sum is order-free
        }
    }
    // sum of the 32 partials of output j by the 4 lanes 4j..4j+3 (every lane of the quad gets the total)
    static __device__ __forceinline__ T quad_sum(const T *st, int j, int ql, int SP) {
        T p[8];
        wg_read<8>(st + j * SP + 8 * ql, p);  // ds_read_b128 moves twice the bytes per LDS cycle of ds_read2_b64
        T s0 = p[0] + p[4], s1 = p[1] + p[5], s2 = p[2] + p[6], s3 = p[3] + p[7];
        T s = (s0 + s1) + (s2 + s3);
        s += xchg<1>(s);
        s += xchg<2>(s);
        return s;
    }

    // ---------------------------------------------------------------- sparse products (matrices in LDS): csr_row_dot_m / csc_col_dot_m below
    // Measured dead ends (tools/phase_timing_csr.py, config 5): fetching six entries per lane with all index loads, then
    // all gathers in flight is no faster — the sparse phases are bound by the LDS throughput of the CU (16 waves share one LDS
    // pipeline; a 64-lane gather of doubles costs ~12 cycles of it against 4 for a linear read), not by latency; holding a lane's
    // CSC slice (values + packed indices) in registers takes 1.0 k cycles off the A'w phase but pushes tile entries into scratch,
    // which costs the same elsewhere.

    // ---------------------------------------------------------------- load balancing of the sparse phases
    // A row (column) of A is not tied to a fixed lane pair (quad): it gets 1, 2, 4 or 8 lanes by its length, such that no lane
    // carries more than K entries, K the smallest bound for which the whole matrix fits the 1,024 lanes (K = 6 at n = 200, m = 400,
    // 5 % density: rows of 7..12 entries on two lanes, longer ones on four).  With fixed pairs / quads the longest row of a wavefront
    // set its pace — 16 of mean 10 — and every phase ended with the workgroup waiting for the slowest wavefront; a perfectly
    // regular pattern runs the kernel 19 % faster (tools/bench_csr.py, SQPH_BENCH_REGULAR_PATTERN).
    // Map word of a lane (16 bits): element in bits 0-8, part in 9-11, log2(lanes of the element) in 12-13, valid in 15.
    static constexpr int MAP_VALID = 1 << 15;
    static __device__ __forceinline__ int lanes_for(int len, int K) {
        const int need = (len + K - 1) / K;
        return need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 4096;  // 4096: does not fit whatever the others need
    }
#ifdef SQPH_SIM
    static inline int wave_shfl(int v, int src) { return (int)(uint32_t)::sqph_sim::wave_exchange((uint64_t)(uint32_t)v, src); }
#else
    static __device__ __forceinline__ int wave_shfl(int v, int src) { return __shfl(v, src); }
#endif
    // ptr[0..count]: CSR row pointers or CSC column pointers in LDS; map[1024]: 16-bit words (LDS); hist: 576 ints of LDS scratch.
    // Every lane of the workgroup calls this.  Groups are laid out by size (8, 4, 2, 1), so each is aligned to its size.
    static __device__ __forceinline__ void build_lane_map(const int *ptr, int count, unsigned short *map, int *hist) {
        const int t = threadIdx.x;
        constexpr int NC = 14, HL = 544;  // lengths are <= 512 (m <= 512, n <= 224)
        constexpr int KC[NC] = {4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 48, 64, 128, 256};
        int *need = hist + HL;
        for (int e = t; e < HL + 16; e += NT) hist[e] = 0;
        map[t] = 0;
        __syncthreads();
        // lanes needed under each candidate bound, from the histogram of the lengths (integer counts: order-independent)
        if (t < count) lds_atomic_inc(&hist[ptr[t + 1] - ptr[t]]);
        __syncthreads();
        if (t < 64) {
            int loc[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) loc[c] = 0;
            for (int len = t; len < HL; len += 64) {
                const int h = hist[len];
                if (h) {
#pragma unroll
                    for (int c = 0; c < NC; c++) loc[c] += h * lanes_for(len, KC[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                int v = loc[c];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = sim_or_shfl_up(v, d);
                    if (t >= d) v += o;
                }
                if (t == 63) need[c] = v;
            }
        }
        __syncthreads();
        int K = KC[NC - 1];
#pragma unroll
        for (int c = NC - 1; c >= 0; c--)
            if (need[c] <= NT) K = KC[c];  // the smallest bound that fits (the need falls with K; 256 always fits: count <= 512)
        if (t < 64) {  // wavefront 0: eight consecutive elements per lane, class offsets by a scan over the 64 lanes
            int cnt[4] = {0, 0, 0, 0}, cls[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = 8 * t + k;
                cls[k] = -1;
                if (i < count) {
                    const int p = lanes_for(ptr[i + 1] - ptr[i], K);
                    const int c = p == 1 ? 0 : p == 2 ? 1 : p == 4 ? 2 : 3;
                    cls[k] = c;
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) cnt[cc] += (cc == c) ? 1 : 0;
                }
            }
            int incl[4], tot[4];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                int v = cnt[cc];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = sim_or_shfl_up(v, d);
                    if (t >= d) v += o;
                }
                incl[cc] = v;
                tot[cc] = wave_shfl(v, 63);
            }
            int base[4];
            base[3] = 0;
            base[2] = 8 * tot[3];
            base[1] = base[2] + 4 * tot[2];
            base[0] = base[1] + 2 * tot[1];
            int run[4];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) run[cc] = incl[cc] - cnt[cc];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int c = cls[k];
                if (c >= 0) {
                    int b0 = 0, rn = 0;
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) {
                        if (cc == c) {
                            b0 = base[cc];
                            rn = run[cc];
                            run[cc] += 1;
                        }
                    }
                    const int p = 1 << c, lane0 = b0 + rn * p;
                    for (int part = 0; part < p; part++)
                        map[lane0 + part] = (unsigned short)(MAP_VALID | (c << 12) | (part << 9) | (8 * t + k));
                }
            }
        }
        __syncthreads();
    }
    // sum over the lanes of an aligned group of 1 << lg lanes (lg lane-varying, <= 3); every lane of the wave must call this
    static __device__ __forceinline__ T group_sum(T s, int lg) {
        T o = xchg<1>(s);
        s += lg >= 1 ? o : T(0);
        o = xchg<2>(s);
        s += lg >= 2 ? o : T(0);
        o = xchg<4>(s);
        s += lg >= 3 ? o : T(0);
        return s;
    }
    // (A v)_i by the lanes the map gives row i; v plain-indexed in LDS.  Every lane of the wave must call this.
    static __device__ __forceinline__ T csr_row_dot_m(const int *rowptr, const unsigned short *col, const T *val, const T *v, int mp) {
        T a0 = 0, a1 = 0;
        const int lg = (mp >> 12) & 3;
        if (mp & MAP_VALID) {
            const int i = mp & 511, p = 1 << lg;
            const int e1 = rowptr[i + 1];
            int e = rowptr[i] + ((mp >> 9) & 7);
            for (; e + p < e1; e += 2 * p) {
                a0 = wg_fma(val[e], v[col[e]], a0);
                a1 = wg_fma(val[e + p], v[col[e + p]], a1);
            }
            if (e < e1) a0 = wg_fma(val[e], v[col[e]], a0);
        }
        return group_sum(a0 + a1, lg);
    }
    // (A' v)_j by the lanes the map gives column j
    static __device__ __forceinline__ T csc_col_dot_m(const int *colptr, const unsigned *csc, const T *val, const T *v, int mp) {
        T a0 = 0, a1 = 0;
        const int lg = (mp >> 12) & 3;
        if (mp & MAP_VALID) {
            const int j = mp & 511, p = 1 << lg;
            const int e1 = colptr[j + 1];
            int e = colptr[j] + ((mp >> 9) & 7);
            for (; e + p < e1; e += 2 * p) {
                const unsigned p0 = csc[e], p1 = csc[e + p];
                a0 = wg_fma(val[p0 & 0xffffu], v[p0 >> 16], a0);
                a1 = wg_fma(val[p1 & 0xffffu], v[p1 >> 16], a1);
            }
            if (e < e1) {
                const unsigned p0 = csc[e];
                a0 = wg_fma(val[p0 & 0xffffu], v[p0 >> 16], a0);
            }
        }
        return group_sum(a0 + a1, lg);
    }

    // ---------------------------------------------------------------- set-up
    // CSR of this QP -> LDS, CSC index by counting sort (entries of a column ordered by row: deterministic sums)
    static __device__ __forceinline__ void load_sparse(const CsrArgs<TIN> &ca, int qp, int n, int m, const CsrLayout<TT> &L, unsigned char *smem) {
        const int t = threadIdx.x;
        T *lds = reinterpret_cast<T *>(smem);
        int *li = reinterpret_cast<int *>(smem);
        int *rowptr = li + L.o_rowptr, *colptr = li + L.o_colptr, *ccur = li + L.o_ccur;
        unsigned *csc = reinterpret_cast<unsigned *>(li + L.o_csc);
        unsigned short *col = reinterpret_cast<unsigned short *>(li + L.o_col);
        T *val = lds + L.o_val;
        const int *grp = ca.rowptr + (long long)qp * ca.s_rowptr;
        const int *gci = ca.colind + (long long)qp * ca.s_colind;
        const TIN *gv = ca.val + (long long)qp * ca.s_val;
        for (int i = t; i <= m; i += NT) rowptr[i] = grp[i];
        for (int j = t; j <= L.NP; j += NT) colptr[j] = 0;
        __syncthreads();
        const int nnz = rowptr[m];
        for (int e = t; e < nnz; e += NT) {
            const int j = gci[e];
            col[e] = (unsigned short)j;
            val[e] = (T)gv[e];
            lds_atomic_inc(&colptr[j + 1]);  // integer counts: order-independent
        }
        __syncthreads();
        // exclusive scan of the column counts (n <= 224 values): wave 0, 4 per lane
        if (t < 64) {
            int v[4], s = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = 4 * t + k;
                v[k] = (j < L.NP) ? colptr[j + 1] : 0;
                s += v[k];
            }
            int incl = s;  // inclusive scan over the 64 lanes
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = sim_or_shfl_up(incl, d);
                if ((t & 63) >= d) incl += o;
            }
            int run = incl - s;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = 4 * t + k;
                if (j < L.NP) {
                    colptr[j + 1] = run + v[k];
                    ccur[j] = run;
                }
                run += v[k];
            }
        }
        __syncthreads();
        // fill: slot order inside a column is arbitrary here (it depends on the order the atomics were served) ...
        unsigned *tmp = reinterpret_cast<unsigned *>(lds + L.o_stage);  // the partial-sum staging area is idle during the set-up
        const bool ranked = (size_t)nnz * sizeof(unsigned) <= (size_t)L.o_tcol * sizeof(T);
        unsigned *fill = ranked ? tmp : csc;
        for (int i = t; i < m; i += NT) {
            for (int e = rowptr[i]; e < rowptr[i + 1]; e++) {
                const int j = col[e];
                const int slot = lds_atomic_inc(&ccur[j]);
                fill[slot] = ((unsigned)i << 16) | (unsigned)e;
            }
        }
        __syncthreads();
        // ... and made canonical (ordered by row inside every column: deterministic sums) ...
        if (ranked) {
            // ... by ranking: the keys of a column are distinct, so the final slot of an entry is the number of smaller keys in its
            // column — one lane per ENTRY, all 1024 lanes busy (round 2 sorted every column by insertion with one lane per COLUMN:
            // 200 lanes, ~200 dependent LDS operations each; the set-up's load phase was 127 k cycles)
            for (int s0 = t; s0 < nnz; s0 += NT) {
                const unsigned key = tmp[s0];
                const int j = col[key & 0xffffu];
                const int e0 = colptr[j], e1 = colptr[j + 1];
                int rank = 0;
                for (int f = e0; f < e1; f++) rank += tmp[f] < key ? 1 : 0;
                csc[e0 + rank] = key;
            }
        } else {
            for (int j = t; j < n; j += NT) {
                const int e0 = colptr[j], e1 = colptr[j + 1];
                for (int e = e0 + 1; e < e1; e++) {
                    const unsigned key = csc[e];
                    int f = e - 1;
                    while (f >= e0 && csc[f] > key) {
                        csc[f + 1] = csc[f];
                        f--;
                    }
                    csc[f + 1] = key;
                }
            }
        }
        __syncthreads();
    }

#ifdef SQPH_SIM
    static inline int sim_or_shfl_up(int v, int d) {
        const int lane = (int)(threadIdx.x & 63);
        const uint64_t r = ::sqph_sim::wave_exchange((uint64_t)(uint32_t)v, lane >= d ? lane - d : lane);
        return (int)(uint32_t)r;
    }
#else
    static __device__ __forceinline__ int sim_or_shfl_up(int v, int d) { return __shfl_up(v, d); }
#endif

    // S = P_sym + sigma I + A' diag(rho) A  ->  register tile (lower tile-triangle), one 32-column panel at a time.
    // Half-wave r owns column j = 32p + r of the panel: for every CSC entry (i, pos) of that column, in row order,
    // its 32 lanes add rho_i A_ij * (row i of A) into the panel column (distinct k per lane: rows are duplicate-free).
    static __device__ __forceinline__ void form_S(const TIN *__restrict__ gP, int n, T sigma, const CsrLayout<TT> &L, unsigned char *smem,
                                                  int t, T (&w)[NE]) {
        const int c = t & 31, r = t >> 5;
        T *lds = reinterpret_cast<T *>(smem);
        const int *li = reinterpret_cast<const int *>(smem);
        const int *rowptr = li + L.o_rowptr, *colptr = li + L.o_colptr;
        const unsigned *csc = reinterpret_cast<const unsigned *>(li + L.o_csc);
        const unsigned short *col = reinterpret_cast<const unsigned short *>(li + L.o_col);
        const T *val = lds + L.o_val, *rho = lds + L.o_rho;
        T *Sp = lds + L.o_stage;
        const int LDP = L.LDP;
#pragma unroll
        for (int p = 0; p < TT; p++) {
            __syncthreads();
            for (int e = t; e < 32 * LDP; e += NT) Sp[e] = 0;
            __syncthreads();
            const int j = 32 * p + r;
            if (j < n) {
                // software-pipelined by one entry: the index / coefficient loads of entry e + 1 (two dependent LDS round trips) are
                // issued before the read-modify-writes of entry e, which the compiler will not move loads across
                const int e1 = colptr[j + 1];
                int e = colptr[j];
                if (e < e1) {
                    unsigned pk = csc[e];
                    int i = (int)(pk >> 16);
                    T coef = rho[i] * val[pk & 0xffffu];
                    int f0 = rowptr[i], f1 = rowptr[i + 1];
                    for (; e < e1; e++) {
                        const unsigned pkn = csc[e + 1 < e1 ? e + 1 : e];
                        const int in = (int)(pkn >> 16);
                        const T coefn = rho[in] * val[pkn & 0xffffu];
                        const int f0n = rowptr[in], f1n = rowptr[in + 1];
                        for (int f = f0 + c; f < f1; f += 32) {
                            const int k = col[f];
                            if (k >= j) Sp[r * LDP + k] = wg_fma(coef, val[f], Sp[r * LDP + k]);
                        }
                        coef = coefn;
                        f0 = f0n;
                        f1 = f1n;
                    }
                }
            }
            __syncthreads();
            // + lower triangle of P (only it reaches the reference's factor, Eigen::LDLT<.,Lower>, qp.hpp:129) + sigma I.
            // All of a lane's loads are issued before the first of them is used: the panel additions are LDS read-modify-writes the
            // compiler will not move a global load across, and one HBM round trip per element (seven per lane and panel, one after
            // the other) was half of this function's time.
            {
                T pv[TT];
#pragma unroll
                for (int q = 0; q < TT; q++) {
                    const int e = t + NT * q;
                    const int jj = e / n, k = e - jj * n, jc = 32 * p + jj;
                    pv[q] = (e < 32 * n && jc < n && k >= jc) ? (T)gP[(long)jc * n + k] : T(0);
                }
#pragma unroll
                for (int q = 0; q < TT; q++) {
                    const int e = t + NT * q;
                    const int jj = e / n, k = e - jj * n, jc = 32 * p + jj;
                    if (e < 32 * n && jc < n && k >= jc) Sp[jj * LDP + k] += pv[q] + (k == jc ? sigma : T(0));
                }
            }
            __syncthreads();
#pragma unroll
            for (int a = p; a < TT; a++) w[idx(a, p)] = Sp[c * LDP + r + 32 * a];  // rows >= n of the panel are zero
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- DENSE A (run<CHECKS, DENSE = true>)
    // Dense problems beyond the register-tiled kernels' shapes (112 < n <= 224, or m beyond their row counts, m <= 512): the factor
    // W lives in the CU's register file exactly as above, A (column-major m x n, 640 KB at n = 200, m = 400) stays in global memory
    // and is streamed twice per iteration with coalesced loads — once down its columns for A' w (one wavefront per column, w in
    // registers, a wave reduction per column) and once across them for A x~ (a lane per row and column half, x~ broadcast from LDS).
    // Such a solve is HBM / Infinity-Cache bound by construction (1.28 MB per QP and iteration); the generic kernel it replaces for
    // these shapes streamed W and a row-major copy of A as well (1.9 MB) from eight or sixteen waves.
    static __device__ __forceinline__ void form_S_dense(const TIN *__restrict__ gA, const TIN *__restrict__ gP, int n, int m, T sigma,
                                                        const CsrLayout<TT> &L, unsigned char *smem, int t, T (&w)[NE]) {
        const int c = t & 31, r = t >> 5;
        T *lds = reinterpret_cast<T *>(smem);
        const T *rho = lds + L.o_rho;
        T *Ab = lds + L.o_stage;  // RB rows of A, column j at (j & 31) * 8 + (j >> 5): a lane's TT row-side and column-side operands are contiguous
        constexpr int LDA = 256;
        constexpr int RB = CsrLayout<TT>::o_tcol / LDA < 16 ? CsrLayout<TT>::o_tcol / LDA : 16;  // rows of A per block (16 from TT = 4 up)
        static_assert(RB >= 1 && RB * LDA <= CsrLayout<TT>::o_tcol, "the block of A rows fits the staging area");
#pragma unroll
        for (int e = 0; e < NE; e++) w[e] = 0;
        for (int i0 = 0; i0 < m; i0 += RB) {
            __syncthreads();
            for (int e = t; e < RB * NP; e += NT) {  // RB consecutive rows of a column are RB consecutive lanes
                const int ib = e % RB, j = e / RB, i = i0 + ib;
                Ab[ib * LDA + (j & 31) * 8 + (j >> 5)] = (j < n && i < m) ? (T)gA[(long)j * m + i] : T(0);
            }
            __syncthreads();
#pragma unroll 2
            for (int ib = 0; ib < RB; ib++) {
                if (i0 + ib >= m) break;
                const T ri = rho[i0 + ib];
                T ar[TT], ac[TT];
                wg_read<TT>(Ab + ib * LDA + r * 8, ar);
                wg_read<TT>(Ab + ib * LDA + c * 8, ac);
#pragma unroll
                for (int a = 0; a < TT; a++) {
                    const T sa = ar[a] * ri;
#pragma unroll
                    for (int b = 0; b <= a; b++) w[idx(a, b)] = wg_fma(sa, ac[b], w[idx(a, b)]);
                }
            }
        }
        __syncthreads();
        // + lower triangle of P + sigma I (Eigen::LDLT<.,Lower>, qp.hpp:129); entries above the diagonal are zero as in form_S
#pragma unroll
        for (int a = 0; a < TT; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) {
                const int i = r + 32 * a, j = c + 32 * b;
                const bool low = i < n && j < n && i >= j;
                const T pij = low ? (T)gP[(long)j * n + i] + (i == j ? sigma : T(0)) : T(0);
                w[idx(a, b)] = low ? w[idx(a, b)] + pij : T(0);
            }
    }
    // A' v for every column j < n: wavefront j % 16 takes column j (coalesced loads down the column, v in registers, one wave
    // reduction per column) and hands  add[j] + (A' v)_j  to out[out_index(j)]; v plain-indexed in LDS.  GATHER: column-gather order.
    template <bool GATHER>
    static __device__ __forceinline__ void dense_ATv(const TIN *__restrict__ gA, int n, int m, const T *v, const T *add, T *out, int CS) {
        const int t = threadIdx.x, l = t & 63, wave = t >> 6;
        T vr[8];
#pragma unroll
        for (int s = 0; s < 8; s++) vr[s] = (l + 64 * s < m) ? v[l + 64 * s] : T(0);
        for (int j = wave; j < n; j += 32) {  // two columns in flight
            const int j2 = j + 16;
            const TIN *c0 = gA + (long)j * m, *c1 = gA + (long)(j2 < n ? j2 : j) * m;
            T x0[8], x1[8];
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int i = l + 64 * s;
                x0[s] = i < m ? (T)c0[i] : T(0);
                x1[s] = i < m ? (T)c1[i] : T(0);
            }
            T a0 = 0, a1 = 0;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                a0 = wg_fma(x0[s], vr[s], a0);
                a1 = wg_fma(x1[s], vr[s], a1);
            }
            a0 = wave_sum(a0);
            a1 = wave_sum(a1);
            if (l == 0) {
                out[GATHER ? (j & 31) * CS + (j >> 5) : j] = (add ? add[j] : T(0)) + a0;
                if (j2 < n) out[GATHER ? (j2 & 31) * CS + (j2 >> 5) : j2] = (add ? add[j2] : T(0)) + a1;
            }
        }
    }
    static __device__ __forceinline__ T wave_sum(T s) {
        s += xchg<1>(s);
        s += xchg<2>(s);
        s += xchg<4>(s);
        s += xchg<8>(s);
        s += xchg<16>(s);
        s += xchg<32>(s);
        return s;
    }
    // partial sums of A v: lane t takes row t & 511 and the column half t >> 9 (coalesced across the rows); zp[2][512]
    static __device__ __forceinline__ void dense_Av_partial(const TIN *__restrict__ gA, int n, int m, const T *v, T *zp) {
        const int t = threadIdx.x, i = t & 511, h = t >> 9;
        const int nh = (n + 1) >> 1, j0 = h * nh, j1 = (j0 + nh < n) ? j0 + nh : n;
        T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (i < m) {
            const TIN *p = gA + i;
            int j = j0;
            for (; j + 8 <= j1; j += 8) {
                T x[8];
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = (T)p[(long)(j + k) * m];
                T vv[8];
                wg_read8_any(v + j, vv);
                a0 = wg_fma(x[0], vv[0], a0); a1 = wg_fma(x[1], vv[1], a1); a2 = wg_fma(x[2], vv[2], a2); a3 = wg_fma(x[3], vv[3], a3);
                a0 = wg_fma(x[4], vv[4], a0); a1 = wg_fma(x[5], vv[5], a1); a2 = wg_fma(x[6], vv[6], a2); a3 = wg_fma(x[7], vv[7], a3);
            }
            for (; j < j1; j++) a0 = wg_fma((T)p[(long)j * m], v[j], a0);
        }
        zp[h * 512 + i] = (a0 + a1) + (a2 + a3);
    }
    static __device__ __forceinline__ void wg_read8_any(const T *p, T (&v)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = p[k];
    }

    // pivots k = 32 AK + rk, rk = 0..31 (see eliminate): g[j] = W-part of row k (j < k) | d + 1 (j = k) | column k of the trailing
    // matrix (j > k); every entry (i,j), i > k, j <= i, gets  e -= (g[i]/d) * g[j].  One barrier per pivot (g is double-buffered).
    template <int AK>
    static __device__ __forceinline__ void pivot_rows(int n, T *g, T *dsv, int NPl, int r, int c, bool &good, T (&w)[NE]) {
        if constexpr (AK < TT) {
#pragma unroll 1
            for (int rk = 0; rk < 32; rk++) {
                const int k = 32 * AK + rk;
                if (k >= n) break;
                T *gk = g + (k & 1) * (NPl + 1);
                if (c == rk) {  // my column group holds column k: entries (a, AK), rows i = r + 32a > k
                    if (r > rk) gk[r + 32 * AK] = w[idx(AK, AK)];
#pragma unroll
                    for (int a = AK + 1; a < TT; a++) gk[r + 32 * a] = w[idx(a, AK)];
                }
                if (r == rk) {  // my row group holds row k: entries (AK, b), columns j = c + 32b < k, and the pivot
#pragma unroll
                    for (int b = 0; b < AK; b++) gk[c + 32 * b] = w[idx(AK, b)];
                    if (c < rk) gk[c + 32 * AK] = w[idx(AK, AK)];
                    if (c == rk) {
                        gk[k] = w[idx(AK, AK)] + T(1);
                        gk[NPl] = w[idx(AK, AK)];
                        dsv[k] = w[idx(AK, AK)];
                    }
                }
                __syncthreads();
                const T d = gk[NPl];
                // a bad pivot is only recorded (block-uniform: every lane reads the same word); leaving the loop from here
                // costs the whole tile its registers (the extra exit made the allocator spill 120 VGPRs)
                // (a pivot of the Jacobi-scaled matrix below 1e-290 counts as non-positive: its reciprocal would overflow)
                if (!(d > T(1e-290)) || !(d * T(0) == T(0))) good = false;
                const T dinv = fast_rcp(d);  // admm_wg_kernel.h: v_rcp_f64 + two Newton steps (5 instead of 11 instructions)
                T gc[TT];
#pragma unroll
                for (int b = 0; b < TT; b++) gc[b] = gk[c + 32 * b];
                // rows of tile row AK with i <= k get l_i = 0 (branch-free: a conditional update keeps old and new tile rows alive
                // side by side and spilled the tile); the tile rows above AK are finished, the ones below are all beyond k
#pragma unroll
                for (int a = AK; a < TT; a++) {
                    const T gi = gk[r + 32 * a];
                    const T li = (a > AK || r > rk) ? -(gi * dinv) : T(0);
#pragma unroll
                    for (int b = 0; b <= a; b++) w[idx(a, b)] = wg_fma(li, gc[b], w[idx(a, b)]);
                }
            }
            pivot_rows<AK + 1>(n, g, dsv, NPl, r, c, good, w);
        }
    }

    // In-register factorisation of the tile: Jacobi scaling, forward elimination of [S | I] in place (admm_generic.h:
    // factor_schur), final scaling to W = D^-1/2 L^-1 D_J^-1/2.  One broadcast vector g per pivot k:
    //   g[j] = W-part of row k (j < k) | d + 1 (j = k) | column k of the trailing matrix (j > k)
    // and every entry (i,j), i > k, j <= i, gets  e -= (g[i]/d) * g[j].   Returns false on a bad pivot (block-uniform).
    static __device__ __forceinline__ bool eliminate(int n, const CsrLayout<TT> &L, T *lds, int t, T (&w)[NE]) {
        T *g = lds + L.o_g, *sj = lds + L.o_sj, *dsv = lds + L.o_ds;
        const int NPl = L.NP;
        const int c = t & 31, r = t >> 5;
        // diagonal -> sj
        if (t < NPl) sj[t] = T(1);
        __syncthreads();
        if (r == c) {
#pragma unroll
            for (int a = 0; a < TT; a++)
                if (r + 32 * a < n) sj[r + 32 * a] = w[idx(a, a)];
        }
        __syncthreads();
        bool bad = false;
        for (int j = t & 63; j < n; j += 64) {
            const T d = sj[j];
            if (!(d > T(0)) || !(d * T(0) == T(0))) bad = true;
        }
        {   // block-uniform verdict
            T *red = lds + L.o_red;
            T f = bad ? T(1) : T(0);
            f = wave_nanmax(f);
            if ((t & 63) == 0) red[t >> 6] = f;
            __syncthreads();
            T any = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) any = nanmax(any, red[k]);
            __syncthreads();
            if (any != T(0)) return false;
        }
        if (t < n) sj[t] = T(1) / (T)sqrt((double)sj[t]);
        __syncthreads();
        {
            T sr[TT], sc[TT];
#pragma unroll
            for (int a = 0; a < TT; a++) {
                sr[a] = (r + 32 * a < n) ? sj[r + 32 * a] : T(0);
                sc[a] = (c + 32 * a < n) ? sj[c + 32 * a] : T(0);
            }
#pragma unroll
            for (int a = 0; a < TT; a++)
#pragma unroll
                for (int b = 0; b <= a; b++) w[idx(a, b)] = w[idx(a, b)] * sr[a] * sc[b];
        }
        bool good = true;
        // The pivots are walked tile row by tile row: k = 32 ak + rk with ak a COMPILE-TIME constant (seven specialised copies of the
        // pivot step at TT = 7) and rk the run-time inner loop.  With ak known, the owners publish exactly the TT - ak column entries
        // and ak + 1 row entries that exist (the run-time version tested all 28 tile entries against ak: ~60 predicated stores per
        // pivot), and only the tile rows a >= ak are updated (28, 27, 25, 22, 18, 13, 7 FMAs instead of 28) — the loop is bound by
        // the instruction issue of the 16 waves (round 2: 2.65 k cycles per pivot at ~100 instructions).
        pivot_rows<0>(n, g, dsv, NPl, r, c, good, w);
        // entries of g for columns/rows outside [0,n) are never written: zero from the initial clear of the caller
        __syncthreads();
        {
            T rs[TT], sc[TT];
#pragma unroll
            for (int a = 0; a < TT; a++) {
                rs[a] = (r + 32 * a < n) ? T(1) / (T)sqrt((double)dsv[r + 32 * a]) : T(0);
                sc[a] = (c + 32 * a < n) ? sj[c + 32 * a] : T(0);
            }
#pragma unroll
            for (int a = 0; a < TT; a++)
#pragma unroll
                for (int b = 0; b <= a; b++) {
                    const int i = r + 32 * a, j = c + 32 * b;
                    const T v = (i > j) ? w[idx(a, b)] * rs[a] : (i == j ? rs[a] : T(0));
                    w[idx(a, b)] = v * sc[b];
                }
        }
        __syncthreads();
        return good;
    }

    // tile <-> global workspace (the factor survives between setup() and solve() calls there), col-major n x n
    static __device__ __forceinline__ void store_tile(T *__restrict__ gW, int n, int r, int c, const T (&w)[NE]) {
#pragma unroll
        for (int a = 0; a < TT; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) {
                const int i = r + 32 * a, j = c + 32 * b;
                if (i < n && j < n && i >= j) gW[(long)j * n + i] = w[idx(a, b)];
            }
    }
    static __device__ __forceinline__ void load_tile(const T *__restrict__ gW, int n, int r, int c, T (&w)[NE]) {
#pragma unroll
        for (int a = 0; a < TT; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) {
                const int i = r + 32 * a, j = c + 32 * b;
                w[idx(a, b)] = (i < n && j < n && i >= j) ? gW[(long)j * n + i] : T(0);
            }
    }

    // P x with the full P (both triangles, qp.cpp:324), streamed from global memory: lane (r,c) takes rows c + 32a
    // (coalesced) and columns r + 32b; partial sums over r staged like stage_WT.  x in row-gather order in yrow.
    static __device__ __forceinline__ void stage_P_gmem(const TIN *__restrict__ gP, int n, const T *yrow, T *st, int r, int c, int CS,
                                                        int SP) {
        T xv[TT];
        wg_read<TT>(yrow + r * CS, xv);
        for (int a = 0; a < TT; a++) {
            const int i = c + 32 * a;
            T acc = 0;
            if (i < n) {
#pragma unroll
                for (int b = 0; b < TT; b++) {
                    const int j = r + 32 * b;
                    if (j < n) acc = wg_fma((T)gP[(long)j * n + i], xv[b], acc);
                }
            }
            st[(c + 32 * a) * SP + ((r + c) & 31)] = acc;
        }
    }

    // CHECKS = false: the instantiation for calls that never look at the residuals (check_termination == 0, no adaptive rho); without
    // the check block the tile stays out of scratch (no VGPR spilled instead of 6): 36.9 -> 35.7 ms per 8,192 x 200 iterations (config 5).
    // Instantiated in csr_nocheck.hip only.
    // DENSE = true: A is the dense column-major matrix of the KArgs (see form_S_dense), ca is unused
    template <bool CHECKS = true, bool DENSE = false>
    static __device__ void run(const KArgs<T, TIN> &a, const CsrArgs<TIN> &ca, unsigned char *smem) {
        const int qp = blockIdx.x;
        if (qp >= a.batch) return;
        const int n = a.n, m = a.m;
        const CsrLayout<TT> L = CsrLayout<TT>::make(m, DENSE ? 0 : ca.nnz_cap);
        const TIN *gA = DENSE ? a.A + (long)qp * a.sA : nullptr;
        T *lds = reinterpret_cast<T *>(smem);
        int *li = reinterpret_cast<int *>(smem);
        const int *rowptr = li + L.o_rowptr, *colptr = li + L.o_colptr;
        const unsigned *csc = reinterpret_cast<const unsigned *>(li + L.o_csc);
        const unsigned short *col = reinterpret_cast<const unsigned short *>(li + L.o_col);
        const T *val = lds + L.o_val;
        T *st = lds + L.o_stage, *tcol = lds + L.o_tcol, *yrow = lds + L.o_yrow, *xt = lds + L.o_xt, *wv = lds + L.o_wv, *ux = lds + L.o_ux;
        T *qv = lds + L.o_qv, *lov = lds + L.o_lo, *upv = lds + L.o_up, *rinvv = lds + L.o_rinv;
        const int CS = L.CS, SP = L.SP;

        const TIN *gP = a.P + (long)qp * a.sP;
        const TIN *gq = a.q + (long)qp * a.sq;
        const TIN *gl = a.l + (long)qp * a.sl;
        const TIN *gu = a.u + (long)qp * a.su;
        T *sx = a.x + (long)qp * n;
        T *sz = a.z + (long)qp * m;
        T *sy = a.y + (long)qp * m;
        T *srho = a.rho_vec + (long)qp * m;
        int *sct = a.ctype + (long)qp * m;
        T *gW = a.Sinv + (long)qp * 2 * n * n;

        sqph_info info = a.info[qp];
        T rho_s = a.rho[qp];
        const int mode = a.mode;
        if (!(mode & (MODE_SETUP | MODE_UPDATE)) && (info.status == SQPH_UNINITIALIZED || info.status == SQPH_NUMERICAL_ISSUES))
            return;  // qp.cpp:68-71 (block-uniform)

        // element owners: the lane quad 4j..4j+3 tracks x_j (and reduces the partial sums of the W phases); the lanes the ROW MAP gives
        // constraint row i track z_i, y_i, rho_i (all of them keep a copy, the one with part 0 writes), see build_lane_map.
        // Lane indices are re-derived from a laundered thread id in every phase (SQPH_LANE): kept live across the solve
        // they and the LDS addresses computed from them crowd the W tile out of the 128 VGPRs a lane has.
#ifdef SQPH_PHASE_TIMING
        unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
        const unsigned long long tstart = tprev;
#define SQPH_CTICK(k) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tprev; tprev = tn_; }
#else
#define SQPH_CTICK(k)
#endif
        int lmap;  // row map in the low half, column map in the high half
        if constexpr (DENSE) {  // a lane pair per constraint row (m <= 512); the column map is not used
            const int t0 = (int)threadIdx.x;
            lmap = (t0 >> 1) < m ? (MAP_VALID | (1 << 12) | ((t0 & 1) << 9) | (t0 >> 1)) : 0;
        } else {
        load_sparse(ca, qp, n, m, L, smem);
            unsigned short *tmap = reinterpret_cast<unsigned short *>(lds + L.o_stage);  // the staging area is idle during the set-up
            int *scratch = reinterpret_cast<int *>(tmap + 2 * NT);
            build_lane_map(rowptr, m, tmap, scratch);
            build_lane_map(colptr, n, tmap + NT, scratch);
            lmap = (int)tmap[threadIdx.x] | ((int)tmap[NT + threadIdx.x] << 16);
            __syncthreads();
        }
        SQPH_CTICK(0)
#define SQPH_ROWMAP(mp, im, lead, mown) \
    const int mp = lmap & 0xffff, im = mp & 511; \
    const bool mown = (mp & MAP_VALID) != 0, lead = mown && ((mp >> 9) & 7) == 0

        T x = 0, z = 0, y = 0, rho = T(1);
        bool rho_differs = false;  // against the vector the resident factor was built with (MODE_SAME_MATRICES)
        {
        SQPH_LANE(t);
        const int jn = t >> 2, ql = t & 3;
        SQPH_ROWMAP(mp, im, lead, mown);
        const bool nown = jn < n;
        // q, l, u, 1/rho of the owned elements live in LDS (read once per iteration): the register budget of a
        // 1024-lane workgroup is 128 VGPRs per lane and the W tile takes 56 of them
        if (ql == 0 && jn < L.NP) qv[jn] = nown ? (T)gq[jn] : T(0);
        if (lead) {
            lov[im] = (T)gl[im];
            upv[im] = (T)gu[im];
            rinvv[im] = T(1);
        }
        __syncthreads();

        if (mode & (MODE_SETUP | MODE_UPDATE)) {
            rho_s = a.rho0;
            if (mown) {
                const T lo = lov[im], up = upv[im];
                int ctype = SQPH_INEQUALITY_CONSTRAINT;
                if (lo < -a.loose_thresh && up > a.loose_thresh)
                    ctype = SQPH_LOOSE_BOUNDS;
                else if (up - lo < a.eq_tol)
                    ctype = SQPH_EQUALITY_CONSTRAINT;
                rho = rho_for_type<T>(ctype, rho_s, a.rho_min, a.rho_eq_factor);
                if (lead) {
                    rho_differs = !(rho == srho[im]);
                    rinvv[im] = T(1) / rho;
                    sct[im] = ctype;
                    srho[im] = rho;
                }
            }
            info.rho_updates += 1;
            if (!(mode & MODE_SETUP)) {
                if (nown) x = sx[jn];
                if (mown) {
                    z = sz[im];
                    y = sy[im];
                }
            }
        } else {
            if (nown) x = sx[jn];
            if (mown) {
                z = sz[im];
                y = sy[im];
                rho = srho[im];
                if (lead) rinvv[im] = T(1) / rho;
            }
        }
        }

        T w[NE];
        bool need_factor = (mode & (MODE_SETUP | MODE_UPDATE | MODE_REFACTOR)) != 0;
        if ((mode & MODE_SAME_MATRICES) && (mode & (MODE_SETUP | MODE_UPDATE)) && !(mode & MODE_REFACTOR) &&
            info.status != SQPH_NUMERICAL_ISSUES && info.status != SQPH_UNINITIALIZED) {
            // sqph_setup_solve_reuse_csr: same P and A as the resident factor, which is the one this set-up would build unless some
            // row's freshly classified rho differs from the vector it was built with (see admm_csrb_kernel.h)
            if (threadIdx.x == 0) st[0] = T(0);
            __syncthreads();
            if (rho_differs) st[0] = T(1);
            __syncthreads();
            if (st[0] == T(0)) {
                need_factor = false;
                info.status = SQPH_UNSOLVED;  // qp.cpp:39-43
            }
            __syncthreads();
        }
        bool solving = false;
        bool state_dirty = (mode & MODE_SETUP) != 0;
        if (!need_factor) {
            SQPH_LANE(t);
            load_tile(gW, n, t >> 5, t & 31, w);
        }
        const T alpha = a.alpha, sigma = a.sigma, oma = T(1) - a.alpha;
        int iter = 1;
        int next_check = a.check_termination > 0 ? a.check_termination : -1;
        int next_adapt = (a.adaptive_rho && a.adaptive_rho_interval > 0) ? a.adaptive_rho_interval : -1;
        for (;;) {
            if (need_factor) {
                __syncthreads();
                // nothing but the tile should be live while it is being built: the iterates are parked in the state
                // arrays (where they end up anyway) and picked up again afterwards
                SQPH_LANE(t);
                const int jn = t >> 2, ql = t & 3;
                SQPH_ROWMAP(mp, im, lead, mown);
                const bool nown = jn < n;
                if (nown && ql == 0) sx[jn] = x;
                if (lead) {
                    sz[im] = z;
                    sy[im] = y;
                    srho[im] = rho;
                    lds[L.o_rho + im] = rho;
                }
                for (int e = t; e < 2 * (L.NP + 1); e += NT) lds[L.o_g + e] = 0;
                __syncthreads();
                SQPH_CTICK(10)
                bool ok;
                {
                    int n_f = n;
                    const TIN *gP_f = gP;
                    SQPH_OPAQUE_S(n_f); SQPH_OPAQUE_S(gP_f);
                    SQPH_LANE(tf);
                    if constexpr (DENSE) {
                        int m_f = m;
                        const TIN *gA_f = gA;
                        SQPH_OPAQUE_S(m_f); SQPH_OPAQUE_S(gA_f);
                        form_S_dense(gA_f, gP_f, n_f, m_f, sigma, L, smem, tf, w);
                    } else {
                        form_S(gP_f, n_f, sigma, L, smem, tf, w);
                    }
                    SQPH_CTICK(1)
                    ok = eliminate(n_f, L, lds, tf, w);
                    SQPH_CTICK(2)
                    if (!(mode & MODE_NO_FACTOR_STORE)) store_tile(gW, n_f, tf >> 5, tf & 31, w);  // kept for later solve() calls
                }
                __syncthreads();
                {   // pick the parked iterates up again
                    SQPH_LANE(t2);
                    const int jn_f = t2 >> 2;
                    SQPH_ROWMAP(mp_f, im_f, lead_f, mown_f);
                    (void)lead_f;
                    x = jn_f < n ? sx[jn_f] : T(0);
                    z = mown_f ? sz[im_f] : T(0);
                    y = mown_f ? sy[im_f] : T(0);
                    rho = mown_f ? srho[im_f] : T(1);
                }
                need_factor = false;
                if (!solving) {
                    if (mode & (MODE_SETUP | MODE_UPDATE)) info.status = ok ? SQPH_UNSOLVED : SQPH_NUMERICAL_ISSUES;  // qp.cpp:39-43, 57-61
                    else if (!ok) info.status = SQPH_NUMERICAL_ISSUES;  // solve() rebuilding a factor that was not kept
                } else if (!ok) {
                    info.status = SQPH_NUMERICAL_ISSUES;  // qp.cpp:139-142
                    break;
                } else {
                    iter++;
                }
            }
            if (!(mode & MODE_SOLVE) || info.status == SQPH_NUMERICAL_ISSUES || info.status == SQPH_UNINITIALIZED) break;
            if (!solving) {
                solving = true;
                state_dirty = true;
                if ((mode & MODE_COLD_RESET) && !a.warm_start) x = z = y = 0;
            }
            // w = R (z - R^-1 y)  [rhs tail of qp.cpp:275 pre-multiplied by R], plain-indexed for the CSC gather
            __syncthreads();
            {
                SQPH_LANE(t);
                SQPH_ROWMAP(mp, im, lead, mown);
                (void)mown;
                if (lead) wv[im] = rho * (z - rinvv[im] * y);
                // u = sigma x - q for the lanes that sum column j of A' w (they are not the quad that tracks x_j)
                const int jn = t >> 2;
                if ((t & 3) == 0 && jn < L.NP) ux[jn] = jn < n ? sigma * x - qv[jn] : T(0);
                for (int e = t; e < 32 * CS; e += NT) tcol[e] = T(0);  // the entries of the padding columns stay zero (only columns < n are written)
            }
            SQPH_CTICK(10)
            for (; iter <= a.max_iter; iter++) {
                __syncthreads();
                SQPH_CTICK(8)
                // every phase re-derives its lane indices from a laundered thread id: kept live across the loop, the LDS
                // addresses they feed would not fit next to the tile (they were spilled to scratch and reloaded per phase)
                if constexpr (DENSE) {
                    int n_d = n, m_d = m;
                    const TIN *gA_d = gA;
                    SQPH_OPAQUE_S(n_d); SQPH_OPAQUE_S(m_d); SQPH_OPAQUE_S(gA_d);
                    dense_ATv<true>(gA_d, n_d, m_d, wv, ux, tcol, CS);
                } else {   // t = (sigma x - q) + A' w, published in column-gather order
                    const int cm = (lmap >> 16) & 0xffff;
                    const T s = csc_col_dot_m(colptr, csc, val, wv, cm);
                    const int j = cm & 511;
                    if ((cm & MAP_VALID) && ((cm >> 9) & 7) == 0) tcol[(j & 31) * CS + (j >> 5)] = ux[j] + s;
                }
                __syncthreads();
                SQPH_CTICK(3)
                {
                    SQPH_LANE(tl);
                    stage_W(w, tcol, st, tl >> 5, tl & 31, CS, SP);
                }
                // The 32 partial sums of an output row r + 32 a were all produced by the 32 lanes of ONE half-wave (row group r), and
                // stage_WT of that half-wave is the only reader of those y1 values: the reduction stays inside the wavefront — quad
                // (a, part) of the half-wave sums row r + 32 a — and needs no workgroup barrier, only program order (LDS operations of
                // a wavefront execute in order)
                wave_lds_order();
                SQPH_CTICK(4)
                {
                    SQPH_LANE(tl);
                    const int rr = tl >> 5, cc = tl & 31, a = cc >> 2, ql = cc & 3;
                    const int av = a < TT ? a : 0;
                    const T y1 = quad_sum(st, rr * TT + av, ql, SP);
                    if (ql == 0 && a < TT) yrow[rr * CS + a] = rr + 32 * av < n ? y1 : T(0);
                }
                __syncthreads();
                SQPH_CTICK(5)
                {
                    SQPH_LANE(tl);
                    stage_WT(w, yrow, st, tl >> 5, tl & 31, CS, SP);
                }
                __syncthreads();
                SQPH_CTICK(6)
                {   // x~ = W' y1: plain-indexed for the CSR gather; x relaxation (qp.cpp:96)
                    SQPH_LANE(tl);
                    const int jn = tl >> 2, ql = tl & 3;
                    const bool nown = jn < n;
                    const T xtj = quad_sum(st, jn < L.NP ? jn : 0, ql, SP);
                    if (ql == 0 && jn < L.NP) xt[jn] = nown ? xtj : T(0);
                    if (nown) x = alpha * xtj + oma * x;
                    if (ql == 0 && nown) ux[jn] = sigma * x - qv[jn];  // next iteration's u (read after two barriers)
                }
                __syncthreads();
                SQPH_CTICK(7)
                if constexpr (DENSE) {  // partial sums of A x~ by row and column half, through the (idle) staging area
                    int n_d = n, m_d = m;
                    const TIN *gA_d = gA;
                    SQPH_OPAQUE_S(n_d); SQPH_OPAQUE_S(m_d); SQPH_OPAQUE_S(gA_d);
                    dense_Av_partial(gA_d, n_d, m_d, xt, st);
                    __syncthreads();
                }
                {   // z~ = A x~ ; z, y updates (qp.cpp:99-103, 278-281)
                    SQPH_ROWMAP(mp, im, lead, mown);
                    T zt;
                    if constexpr (DENSE) zt = mown ? st[im] + st[512 + im] : T(0);
                    else zt = csr_row_dot_m(rowptr, col, val, xt, mp);
                    if (mown) {
                        const T zr = alpha * zt + oma * z;
                        T zn = zr + rinvv[im] * y;
                        const T lo = lov[im], up = upv[im];
                        zn = zn < lo ? lo : zn;
                        zn = zn > up ? up : zn;
                        y = y + rho * (zr - zn);
                        z = zn;
                        if (lead) wv[im] = rho * (z - rinvv[im] * y);  // next iteration's w (read after the loop-top barrier)
                    }
                }
                bool check = false, adapt = false;
                if constexpr (CHECKS) {
                    if (--next_check == 0) {
                        check = true;
                        next_check = a.check_termination;
                    }
                    if (--next_adapt == 0) {
                        adapt = true;
                        next_adapt = a.adaptive_rho_interval;
                    }
                }
                if (CHECKS && (check || adapt)) {
                    // update_state + residuals, qp.cpp:316-331, 353-361
                    __syncthreads();
                    SQPH_LANE(tl);
                    const int t = tl, jn = tl >> 2, ql = tl & 3;
                    SQPH_ROWMAP(mp, im, lead, mown);
                    const bool nown = jn < n;
                    if (ql == 0 && jn < L.NP) {
                        xt[jn] = nown ? x : T(0);
                        yrow[(jn & 31) * CS + (jn >> 5)] = nown ? x : T(0);
                    }
                    if (lead) wv[im] = y;
                    __syncthreads();
                    T Ax;
                    if constexpr (DENSE) {
                        int n_d = n, m_d = m;
                        const TIN *gA_d = gA;
                        SQPH_OPAQUE_S(n_d); SQPH_OPAQUE_S(m_d); SQPH_OPAQUE_S(gA_d);
                        dense_Av_partial(gA_d, n_d, m_d, xt, st);
                        dense_ATv<false>(gA_d, n_d, m_d, wv, (const T *)nullptr, tcol, CS);  // A' y, plain-indexed
                        __syncthreads();
                        Ax = mown ? st[im] + st[512 + im] : T(0);
                        __syncthreads();  // the staging area is handed to the P x partial sums next
                    } else {
                    Ax = csr_row_dot_m(rowptr, col, val, xt, mp);
                    {   // A' y by the column map's lanes, handed to the quads that track x through the (idle) column-gather vector
                        const int cm = (lmap >> 16) & 0xffff;
                        const T sATy = csc_col_dot_m(colptr, csc, val, wv, cm);
                        if ((cm & MAP_VALID) && ((cm >> 9) & 7) == 0) tcol[cm & 511] = sATy;
                    }
                    }
                    {
                        int n_c = n, r_c = tl >> 5, c_c = tl & 31;
                        const TIN *gP_c = gP;
                        SQPH_OPAQUE_S(n_c); SQPH_OPAQUE_S(gP_c);
                        stage_P_gmem(gP_c, n_c, yrow, st, r_c, c_c, CS, SP);
                    }
                    __syncthreads();
                    const T Px = quad_sum(st, jn < L.NP ? jn : 0, ql, SP);
                    const T ATy = nown ? tcol[jn] : T(0);
                    T v[7] = {0, 0, 0, 0, 0, 0, 0};
                    if (mown) {
                        v[0] = tabs(Ax);
                        v[1] = tabs(z);
                        v[2] = tabs(Ax - z);
                    }
                    if (nown) {
                        const T q = qv[jn];
                        v[3] = tabs(Px);
                        v[4] = tabs(ATy);
                        v[5] = tabs(q);
                        v[6] = tabs(Px + q + ATy);
                    }
                    {
                        T *red = lds + L.o_red;
#pragma unroll
                        for (int e = 0; e < 7; e++) v[e] = wave_nanmax(v[e]);
                        if ((t & 63) == 0) {
#pragma unroll
                            for (int e = 0; e < 7; e++) red[e * 16 + (t >> 6)] = v[e];
                        }
                        __syncthreads();
#pragma unroll
                        for (int e = 0; e < 7; e++) {
                            T mval = red[e * 16];
#pragma unroll
                            for (int k = 1; k < 16; k++) mval = nanmax(mval, red[e * 16 + k]);
                            v[e] = mval;
                        }
                        __syncthreads();
                    }
                    const T nrm_prim = nanmax(v[0], v[1]);
                    const T nrm_dual = nanmax(v[3], nanmax(v[4], v[5]));
                    info.res_prim = (double)v[2];
                    info.res_dual = (double)v[6];
                    if (check) {
                        if (v[2] <= a.eps_abs + a.eps_rel * nrm_prim && v[6] <= a.eps_abs + a.eps_rel * nrm_dual) {
                            info.status = SQPH_SOLVED;
                            break;
                        }
                    }
                    if (adapt) {
                        const T eps = a.regul;
                        const T rp_norm = v[2] / (nrm_prim + eps);
                        const T rd_norm = v[6] / (nrm_dual + eps);
                        T new_rho = rho_s * (T)sqrt((double)(rp_norm / (rd_norm + eps)));
                        new_rho = new_rho < a.rho_max ? new_rho : a.rho_max;
                        new_rho = new_rho > a.rho_min ? new_rho : a.rho_min;
                        info.rho_estimate = (double)new_rho;
                        if (new_rho < rho_s / a.rho_tol || new_rho > rho_s * a.rho_tol) {
                            rho_s = new_rho;
                            if (mown) {
                                rho = rho_for_type<T>(sct[im], rho_s, a.rho_min, a.rho_eq_factor);
                                if (lead) rinvv[im] = T(1) / rho;
                            }
                            info.rho_updates += 1;
                            need_factor = true;
                            break;  // leave WITHOUT advancing iter; the factor block does it
                        }
                    }
                    __syncthreads();
                    if (lead) wv[im] = rho * (z - rinvv[im] * y);  // the check borrowed wv for y
                }
            }
            if (!need_factor) break;
        }
        if (solving) {
            if (iter > a.max_iter) info.status = SQPH_MAX_ITER_EXCEEDED;
            info.iter = iter;
        }
#ifdef SQPH_PHASE_TIMING
        tacc[9] = __builtin_amdgcn_s_memtime() - tstart;
        if (threadIdx.x < 48 && (threadIdx.x & 3) == 0) x = (T)tacc[threadIdx.x >> 2];  // debug build only: wave 0's phase ticks instead of x[0..12)
#endif
        SQPH_LANE(t);
        if (state_dirty) {
            const int jn = t >> 2;
            SQPH_ROWMAP(mp, im, lead, mown);
            (void)mown;
            if (jn < n && (t & 3) == 0) sx[jn] = x;
            if (lead) {
                sz[im] = z;
                sy[im] = y;
                srho[im] = rho;
            }
        }
        if (t == 0) {
            a.info[qp] = info;
            a.rho[qp] = rho_s;
        }
    }
};

template <typename TIN>
struct CsrLaunch {
    KArgs<double, TIN> a;
    CsrArgs<TIN> ca;
};

template <typename TIN, int TT>
__global__ __launch_bounds__(1024) void admm_csr_kernel(CsrLaunch<TIN> p) {
    SQPH_DYN_SMEM(smem_raw);
    CsrKernel<TIN, TT>::template run<true>(p.a, p.ca, smem_raw);
}
// the same without the residual-check block (see CsrKernel::run); instantiated in csr_nocheck.hip only
template <typename TIN, int TT>
__global__ __launch_bounds__(1024) void admm_csr_nocheck_kernel(CsrLaunch<TIN> p) {
    SQPH_DYN_SMEM(smem_raw);
    CsrKernel<TIN, TT>::template run<false>(p.a, p.ca, smem_raw);
}
// dense A streamed from global memory (CsrKernel::run<CHECKS, DENSE = true>); instantiated in csr_dense.hip only
template <typename TIN, int TT>
__global__ __launch_bounds__(1024) void admm_csrd_kernel(CsrLaunch<TIN> p) {
    SQPH_DYN_SMEM(smem_raw);
    CsrKernel<TIN, TT>::template run<true, true>(p.a, p.ca, smem_raw);
}
template <typename TIN, int TT>
__global__ __launch_bounds__(1024) void admm_csrd_nocheck_kernel(CsrLaunch<TIN> p) {
    SQPH_DYN_SMEM(smem_raw);
    CsrKernel<TIN, TT>::template run<false, true>(p.a, p.ca, smem_raw);
}
// > 0 launched, 0 shape not covered (n > 256 or m > 512), < 0 HIP error
template <typename TIN>
int csrd_try_launch(const KArgs<double, TIN> &a, hipStream_t stream, const char **name);
extern template int csrd_try_launch<double>(const KArgs<double, double> &, hipStream_t, const char **);
extern template int csrd_try_launch<float>(const KArgs<double, float> &, hipStream_t, const char **);

// launches it where a tile edge TT is compiled in: > 0 launched, 0 no such edge, < 0 HIP error (hipGetLastError has it)
template <typename TIN>
int csr_nocheck_launch(int TT, int m, int nnz_cap, int batch, hipStream_t stream, const CsrLaunch<TIN> &p);
extern template int csr_nocheck_launch<double>(int, int, int, int, hipStream_t, const CsrLaunch<double> &);
extern template int csr_nocheck_launch<float>(int, int, int, int, hipStream_t, const CsrLaunch<float> &);

// tile edges compiled into the library (n <= 32*TT): first fit wins
#if defined(SQPH_SLIM) && defined(SQPH_SLIM_CSR)
#define SQPH_CSR_SHAPES(X) X(7)
#elif defined(SQPH_SLIM)
#define SQPH_CSR_SHAPES(X)
#else
#define SQPH_CSR_SHAPES(X) X(4) X(7)
#endif
// tile edges of the dense-A mode (csr_dense.hip): problems with n <= 128 that the register-tiled kernels do not take (m too large), n <= 224 and n <= 256
#ifdef SQPH_SLIM
#define SQPH_CSRD_SHAPES(X) X(7)
#else
#define SQPH_CSRD_SHAPES(X) X(4) X(7) X(8)
#endif
// additional small edges for the host SIMT emulation in the CPU test-suite
#define SQPH_CSR_SIM_SHAPES(X) X(1) X(2) X(4) X(7) X(8)

#ifdef SQPH_SIM
template <typename TIN>
inline int sim_run_csrd(const KArgs<double, TIN> &a) {
    if (a.m > 512) return -1;
    const CsrArgs<TIN> ca{nullptr, nullptr, nullptr, 0, 0, 0, 0};
#define SQPH_SIM_CASE(TT_)                                                                                              \
    if (a.n <= 32 * TT_) {                                                                                              \
        const CsrLayout<TT_> L = CsrLayout<TT_>::make(a.m, 0);                                                          \
        if (a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0))                               \
            ::sqph_sim::launch(admm_csrd_nocheck_kernel<TIN, TT_>, dim3(a.batch), dim3(1024), L.bytes, CsrLaunch<TIN>{a, ca}); \
        else                                                                                                            \
            ::sqph_sim::launch(admm_csrd_kernel<TIN, TT_>, dim3(a.batch), dim3(1024), L.bytes, CsrLaunch<TIN>{a, ca});  \
        return 0;                                                                                                       \
    }
    SQPH_CSR_SIM_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

#ifdef SQPH_SIM
template <typename TIN>
inline int sim_run_csr(const KArgs<double, TIN> &a, const CsrArgs<TIN> &ca) {
    if (a.m > 512) return -1;
#define SQPH_SIM_CASE(TT_)                                                                                              \
    if (a.n <= 32 * TT_) {                                                                                              \
        const CsrLayout<TT_> L = CsrLayout<TT_>::make(a.m, ca.nnz_cap);                                                      \
        if (a.check_termination <= 0 && !(a.adaptive_rho && a.adaptive_rho_interval > 0))                               \
            ::sqph_sim::launch(admm_csr_nocheck_kernel<TIN, TT_>, dim3(a.batch), dim3(1024), L.bytes, CsrLaunch<TIN>{a, ca}); \
        else                                                                                                            \
            ::sqph_sim::launch(admm_csr_kernel<TIN, TT_>, dim3(a.batch), dim3(1024), L.bytes, CsrLaunch<TIN>{a, ca});   \
        return 0;                                                                                                       \
    }
    SQPH_CSR_SIM_SHAPES(SQPH_SIM_CASE)
#undef SQPH_SIM_CASE
    return -1;
}
#endif

}  // namespace sqph
