// libsqp_hip.so — host side of the C-ABI declared in include/sqp_hip.h.
// Owns the per-batch device state of N QPSolver instances and launches the ADMM kernels.
// There is deliberately NO CPU execution path in this library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
// RCCL: types only — the library itself is opened with dlopen on the first cross-device gather, so an installation without the
// RCCL headers still compiles this file (the few types the dlopen'd entry points need are restated below)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclFloat64 = 8 } ncclDataType_t;
#endif

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sqp_hip.h"
#include "admm_generic.h"
#include "admm_csr_kernel.h"
#include "admm_csrb_kernel.h"
#include "admm_dispatch.h"
#include "kargs.h"

namespace {

thread_local std::string g_err;

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

size_t dsize(int dtype) { return dtype == SQPH_F32 ? sizeof(float) : sizeof(double); }

__global__ void cvt_f64_to_f32(const double *__restrict__ src, float *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
__global__ void cvt_f32_to_f64(const float *__restrict__ src, double *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

// gather up to four device segments (8-byte words) into one contiguous buffer: one D2H instead of four
__global__ void pack_words(const unsigned long long *s0, size_t n0, const unsigned long long *s1, size_t n1, const unsigned long long *s2,
                           size_t n2, const unsigned long long *s3, size_t n3, unsigned long long *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n0) dst[i] = s0[i];
    else if (i < n0 + n1) dst[i] = s1[i - n0];
    else if (i < n0 + n1 + n2) dst[i] = s2[i - n0 - n1];
    else if (i < n0 + n1 + n2 + n3) dst[i] = s3[i - n0 - n1 - n2];
}

constexpr size_t SMALL_CALL_BYTES = 4u << 20;
constexpr size_t ZERO_COPY_BYTES = 256u << 10;  // beyond this the kernels' re-reads of A and P at termination checks should hit HBM, not PCIe

// Small host-memspace calls (the SQP driver's per-iteration subproblem batches) are latency-bound: the kernels read the
// problem data straight from a pinned, device-mapped host buffer and the results are packed straight into another one
// (no H2D / D2H copy commands), and the host waits by polling the stream for a bounded time before it falls back to the
// blocking wait (whose wake-up latency is tens of microseconds).  Experiment builds: SQPH_NO_ZEROCOPY=1 restores the copy path (A/B).
bool zero_copy_enabled() {
#ifdef SQPH_EXPERIMENTS
    static const bool off = getenv("SQPH_NO_ZEROCOPY") != nullptr;
    return !off;
#else
    return true;
#endif
}
hipError_t wait_stream_low_latency(hipStream_t st) {
    for (int spin = 0; spin < 20000; spin++) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
    }
    return hipStreamSynchronize(st);
}

// CSR -> dense column-major expansion of the constraint matrices (one thread per (QP, row): duplicates within a row
// are summed in storage order, so the result is deterministic).  `dst` must be zero-filled.
template <typename TIN>
__global__ void csr_expand(int batch, int n, int m, const int *__restrict__ rowptr, const int *__restrict__ colind,
                           const TIN *__restrict__ val, long long s_rowptr, long long s_colind, long long s_val,
                           long long nnz_cap, TIN *__restrict__ dst, int *__restrict__ bad) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)batch * m) return;
    const int b = (int)(t / m), i = (int)(t - (long long)b * m);
    const int *rp = rowptr + b * s_rowptr;
    const int *ci = colind + b * s_colind;
    const TIN *v = val + b * s_val;
    TIN *d = dst + (long long)b * m * n;
    const int e0 = rp[i], e1 = rp[i + 1];
    if (e0 < 0 || e1 < e0 || e1 > nnz_cap || (i == 0 && e0 != 0)) { atomicOr(bad, 1); return; }  // as csr_check
    for (int e = e0; e < e1; e++) {
        const int j = ci[e];
        if (j < 0 || j >= n) { atomicOr(bad, 2); continue; }
        d[(long long)j * m + i] += v[e];
    }
}

// Compressed-column P -> dense column-major (sqph_*_csr_sp): one thread per (QP, column); `dst` must be zero-filled.
// bad: bit 0 = column pointers malformed, bit 1 = row index out of range, bit 2 = rows of a column not strictly increasing,
// bit 3 = the pattern is not symmetric (entry (i, j) without (j, i): a triangle was passed instead of the full matrix),
// bit 4 = the values are not (P_ij != P_ji bit for bit: the routes that read column i as row i would differ from the dense ones).
template <typename TIN>
__global__ void csc_expand_P(int batch, int n, const int *__restrict__ colptr, const int *__restrict__ rowind, const TIN *__restrict__ val,
                             long long s_colptr, long long s_rowind, long long s_val, long long nnz_cap, TIN *__restrict__ dst,
                             int *__restrict__ bad) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)batch * n) return;
    const int b = (int)(t / n), j = (int)(t - (long long)b * n);
    const int *cp = colptr + b * s_colptr;
    const int *ri = rowind + b * s_rowind;
    const TIN *v = val + b * s_val;
    TIN *d = dst + (long long)b * n * n + (long long)j * n;
    const int e0 = cp[j], e1 = cp[j + 1];
    if (e0 < 0 || e1 < e0 || e1 > nnz_cap || (j == 0 && e0 != 0)) { atomicOr(bad, 1); return; }
    int prev = -1;
    for (int e = e0; e < e1; e++) {
        const int i = ri[e];
        if (i < 0 || i >= n) { atomicOr(bad, 2); continue; }
        if (i <= prev) atomicOr(bad, 4);
        prev = i;
        if (i != j) {  // the mirror entry (j, i): binary search in column i (a malformed column i is reported by its own thread)
            int lo = cp[i], hi = cp[i + 1];
            bool found = false;
            if (lo >= 0 && hi <= nnz_cap) {
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1, r = ri[mid];
                    if (r == j) {
                        found = true;
                        if (!(v[mid] == v[e])) atomicOr(bad, 16);  // P_ij != P_ji: the in-place route reads column i as row i
                        break;
                    }
                    if (r < j) lo = mid + 1;
                    else hi = mid;
                }
            }
            if (!found) atomicOr(bad, 8);
        }
        if (dst) d[i] = v[e];  // (dst == nullptr: structure check only — the block-row kernel reads the columns in place)
    }
}

// Structural check of a CSR batch for the native sparse kernel: bit 0 = row pointers malformed, bit 1 = column index
// out of range, bit 2 = a row is not strictly increasing in its column indices (legal, but takes the expand path).
__global__ void csr_check(int batch, int n, int m, long long nnz_cap, const int *__restrict__ rowptr, const int *__restrict__ colind,
                          long long s_rowptr, long long s_colind, int *__restrict__ flags) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)batch * m) return;
    const int b = (int)(t / m), i = (int)(t - (long long)b * m);
    const int *rp = rowptr + b * s_rowptr;
    const int *ci = colind + b * s_colind;
    const int e0 = rp[i], e1 = rp[i + 1];
    int f = 0;
    if (e0 < 0 || e1 < e0 || e1 > nnz_cap || (i == 0 && e0 != 0)) {
        f = 1;
    } else {
        int prev = -1;
        for (int e = e0; e < e1; e++) {
            const int j = ci[e];
            if (j < 0 || j >= n) f |= 2;
            if (j <= prev) f |= 4;
            prev = j;
        }
    }
    if (f) atomicOr(flags, f);
}

}  // namespace

// every device allocation of the library.  Experiment builds with SQPH_POISON_WS set fill it with 0xFF (NaN as doubles, -1 as ints) so
// that a kernel reading workspace it never wrote fails instead of depending on what the allocator handed out (tools/xp/poison_ws.sh)
static inline int ws_poison_byte() {
#ifdef SQPH_EXPERIMENTS
    static const int v = getenv("SQPH_POISON_WS") ? 0xFF : 0;
    return v;
#else
    return 0;
#endif
}
template <typename P>
static inline hipError_t ws_malloc(P **p, size_t bytes) {
    hipError_t e = hipMalloc((void **)p, bytes);
#ifdef SQPH_EXPERIMENTS
    if (e == hipSuccess && ws_poison_byte()) e = hipMemset(*p, 0xFF, bytes);
#endif
    return e;
}

struct sqph_solver {
    int device = 0, n = 0, m = 0, cap = 0, dtype = SQPH_F64, flags = 0;
    int num_simds = 1024;  // 4 per CU
    hipStream_t stream = nullptr;
    bool stream_owned = false;
    sqph_settings settings{};
    // persistent device state
    void *x = nullptr, *z = nullptr, *y = nullptr, *rho_vec = nullptr, *rho = nullptr;
    int *ctype = nullptr;
    sqph_info *info = nullptr;
    void *Sinv = nullptr, *At = nullptr;
    // device staging for host-memspace problem data
    void *sP = nullptr, *sq = nullptr, *sA = nullptr, *sl = nullptr, *su = nullptr;
    // CSR entry points: staged index/value arrays (host memspace) and the expanded dense A
    void *cRow = nullptr, *cCol = nullptr, *cVal = nullptr, *cA = nullptr;
    int *cBad = nullptr;
    size_t cCol_cap = 0, cVal_cap = 0;
    // sqph_*_csr_sp: staged compressed-column arrays of P (host memspace) and the dense P they expand to
    void *pPtr = nullptr, *pInd = nullptr, *pVal = nullptr, *cP = nullptr;
    size_t pInd_cap = 0, pVal_cap = 0;
    // small host-memspace calls (the SQP driver's n = 2..50 subproblems): one pinned staging buffer each way, one
    // H2D / D2H per call instead of one per array
    void *hpin = nullptr, *dpin = nullptr, *hout = nullptr, *dout = nullptr;
    void *hpin_dev = nullptr, *hout_dev = nullptr;  // device addresses of the two pinned buffers (zero-copy path)
    size_t hpin_cap = 0, hout_cap = 0;
    hipEvent_t pin_ev = nullptr;
    bool pin_busy = false;
    std::string err;
    const char *kernel_name = "none";
    bool factor_resident = false;  // the workspace holds the factor of the last setup/update (see SQPH_FLAG_KEEP_FACTOR)
    char factor_family = 0;        // kernel family that built it: 't' register-tiled / lane (canonical W), 'c' sparse, 'g' generic
    double *trace = nullptr;       // verbose trace of one QP: [0] = count, then 4 doubles per termination check
    int trace_qp = 0, trace_cap = 0;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;  // one pair per launch while timing is on
    size_t ev_used = 0;
};

#define SQPH_FAIL(s, code, ...)                        \
    do {                                               \
        char buf_[512];                                \
        snprintf(buf_, sizeof(buf_), __VA_ARGS__);     \
        if (s) (s)->err = buf_;                        \
        g_err = buf_;                                  \
        return (code);                                 \
    } while (0)

#define SQPH_HIP(s, call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) SQPH_FAIL(s, SQPH_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int sqph_version(void) { return SQPH_VERSION; }

void sqph_default_settings(sqph_settings *s) {
    // QPSolverSettings defaults, /root/reference/include/solvers/qp.hpp:38-53
    s->rho = 1e-1;
    s->sigma = 1e-6;
    s->alpha = 1.0;
    s->eps_rel = 1e-3;
    s->eps_abs = 1e-3;
    s->max_iter = 1000;
    s->check_termination = 25;
    s->warm_start = 0;
    s->adaptive_rho = 0;
    s->adaptive_rho_tolerance = 5;
    s->adaptive_rho_interval = 25;
    s->verbose = 0;
}

const char *sqph_global_error(void) { return g_err.c_str(); }
const char *sqph_last_error(const sqph_solver *s) { return s ? s->err.c_str() : g_err.c_str(); }
const char *sqph_kernel_name(const sqph_solver *s) { return s ? s->kernel_name : "none"; }

long long sqph_algorithmic_bytes(int n, int m, int dtype) {
    const long long e = (long long)dsize(dtype);
    return e * ((long long)n * n + n + (long long)m * n + 2LL * m) + e * ((long long)n + m) + (long long)sizeof(sqph_info);
}

int sqph_constr_type_init(int dtype, int m, const void *l, const void *u, int *out) {
    // static QPSolver::constr_type_init, /root/reference/src/qp.cpp:283-294
    if (m < 0 || (m > 0 && (!l || !u || !out))) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "constr_type_init: null argument");
    for (int i = 0; i < m; i++) {
        if (dtype == SQPH_F32) {
            const float li = ((const float *)l)[i], ui = ((const float *)u)[i];
            out[i] = (li < -1e16f && ui > 1e16f) ? SQPH_LOOSE_BOUNDS
                     : (ui - li < 1e-4f)        ? SQPH_EQUALITY_CONSTRAINT
                                                : SQPH_INEQUALITY_CONSTRAINT;
        } else {
            const double li = ((const double *)l)[i], ui = ((const double *)u)[i];
            out[i] = (li < -1e16 && ui > 1e16) ? SQPH_LOOSE_BOUNDS
                     : (ui - li < 1e-4)       ? SQPH_EQUALITY_CONSTRAINT
                                              : SQPH_INEQUALITY_CONSTRAINT;
        }
    }
    return SQPH_OK;
}

int sqph_create(sqph_solver **out, int device, int n, int m, int batch_capacity, int dtype, int flags) {
    if (!out) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "sqph_create: out is null");
    *out = nullptr;
    if (n <= 0 || m < 0 || batch_capacity <= 0) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "sqph_create: bad shape n=%d m=%d batch=%d", n, m, batch_capacity);
    if (dtype != SQPH_F64 && dtype != SQPH_F32) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "sqph_create: bad dtype %d", dtype);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_NO_DEVICE, "sqph_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "sqph_create: device %d out of range (%d devices)", device, ndev);

    sqph_solver *s = new (std::nothrow) sqph_solver();
    if (!s) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "sqph_create: out of host memory");
    s->device = device;
    s->n = n;
    s->m = m;
    s->cap = batch_capacity;
    s->dtype = dtype;
    s->flags = flags;
    sqph_default_settings(&s->settings);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) s->num_simds = 4 * cus;
    }

    DeviceGuard g(device);
    const size_t e = sizeof(double), B = (size_t)batch_capacity;  // state/workspace: always fp64
    const size_t mm = (size_t)(m > 0 ? m : 1);
    hipError_t err = hipSuccess;
    int fill = ws_poison_byte();  // (0 outside the experiment builds)
    auto alloc = [&](void **p, size_t bytes) {
        if (err == hipSuccess) err = ws_malloc(p, bytes);
        if (err == hipSuccess) err = hipMemset(*p, fill, bytes);
    };
    fill = 0;  // the iterates of a new instance are zero (qp.hpp:74); everything else is written before it is read
    alloc(&s->x, B * n * e);
    alloc(&s->z, B * mm * e);
    alloc(&s->y, B * mm * e);
    fill = ws_poison_byte();
    alloc(&s->rho_vec, B * mm * e);
    alloc(&s->rho, B * e);
    alloc((void **)&s->ctype, B * mm * sizeof(int));
    fill = 0;
    alloc((void **)&s->info, B * sizeof(sqph_info));
    fill = ws_poison_byte();
    alloc(&s->Sinv, B * 2 * (size_t)n * n * e);
    if (err == hipSuccess) {
        // every instance starts UNINITIALIZED (qp.hpp:74): status field = 4, rest 0
        sqph_info *h = new (std::nothrow) sqph_info[B];
        if (!h) {
            g_err = "sqph_create: out of host memory";
            sqph_destroy(s);
            return SQPH_ERR_INVALID;
        }
        memset(h, 0, B * sizeof(sqph_info));
        for (size_t i = 0; i < B; i++) h[i].status = SQPH_UNINITIALIZED;
        err = hipMemcpy(s->info, h, B * sizeof(sqph_info), hipMemcpyHostToDevice);
        delete[] h;
    }
    if (err != hipSuccess) {
        g_err = std::string("sqph_create: device allocation failed: ") + hipGetErrorString(err);
        sqph_destroy(s);
        return SQPH_ERR_HIP;
    }
    *out = s;
    return SQPH_OK;
}

void sqph_destroy(sqph_solver *s) {
    if (!s) return;
    DeviceGuard g(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    else (void)hipDeviceSynchronize();
    if (s->stream_owned) (void)hipStreamDestroy(s->stream);
    void *ptrs[] = {s->trace, s->x, s->z, s->y, s->rho_vec, s->rho, s->ctype, s->info, s->Sinv, s->At, s->sP, s->sq, s->sA, s->sl, s->su, s->cRow, s->cCol, s->cVal, s->cA, s->cBad, s->pPtr, s->pInd, s->pVal, s->cP};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (s->hpin) (void)hipHostFree(s->hpin);
    if (s->hout) (void)hipHostFree(s->hout);
    if (s->dpin) (void)hipFree(s->dpin);
    if (s->dout) (void)hipFree(s->dout);
    if (s->pin_ev) (void)hipEventDestroy(s->pin_ev);
    for (auto &p : s->evs) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    delete s;
}

int sqph_set_stream(sqph_solver *s, void *hip_stream) {
    if (!s) return SQPH_ERR_INVALID;
    if (s->stream_owned) {
        DeviceGuard g(s->device);
        (void)hipStreamSynchronize(s->stream);
        (void)hipStreamDestroy(s->stream);
        s->stream_owned = false;
    }
    s->stream = (hipStream_t)hip_stream;
    return SQPH_OK;
}

int sqph_set_trace_qp(sqph_solver *s, int index) {
    if (!s) return SQPH_ERR_INVALID;
    if (index < 0 || index >= s->cap) SQPH_FAIL(s, SQPH_ERR_INVALID, "sqph_set_trace_qp: index %d outside [0, %d)", index, s->cap);
    s->trace_qp = index;
    return SQPH_OK;
}

int sqph_get_trace(sqph_solver *s, double *records, int cap_records, int *count) {
    if (!s || !count) return SQPH_ERR_INVALID;
    *count = 0;
    if (!s->trace) return SQPH_OK;
    DeviceGuard g(s->device);
    std::vector<double> h(1 + 4 * (size_t)s->trace_cap);
    SQPH_HIP(s, hipMemcpyAsync(h.data(), s->trace, h.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    SQPH_HIP(s, hipStreamSynchronize(s->stream));
    int k = (int)h[0];
    if (k > s->trace_cap) k = s->trace_cap;
    *count = k;
    for (int i = 0; i < k && i < cap_records && records; i++)
        for (int e = 0; e < 4; e++) records[4 * i + e] = h[1 + 4 * i + e];
    return SQPH_OK;
}

int sqph_own_stream(sqph_solver *s) {
    if (!s) return SQPH_ERR_INVALID;
    if (s->stream_owned) return SQPH_OK;
    DeviceGuard g(s->device);
    hipStream_t st = nullptr;
    SQPH_HIP(s, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    s->stream = st;
    s->stream_owned = true;
    return SQPH_OK;
}

void sqph_shard_bounds(long long total, int parts, int part, long long *lo, long long *hi) {
    if (parts <= 0) parts = 1;
    const long long base = total / parts, rem = total % parts;
    const long long a = (long long)part * base + (part < rem ? part : rem);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (part < rem ? 1 : 0);
}

int sqph_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sqph_set_settings(sqph_solver *s, const sqph_settings *st) {
    if (!s || !st) return SQPH_ERR_INVALID;
    if (!(st->rho > 0) || !(st->sigma > 0) || !(st->alpha > 0 && st->alpha < 2) || st->max_iter < 0 ||
        st->check_termination < 0 || (st->adaptive_rho && st->adaptive_rho_interval <= 0))
        SQPH_FAIL(s, SQPH_ERR_INVALID, "sqph_set_settings: out-of-range setting (rho,sigma>0; 0<alpha<2; max_iter,check_termination>=0; adaptive_rho_interval>0)");
    s->settings = *st;
    return SQPH_OK;
}

int sqph_get_settings(const sqph_solver *s, sqph_settings *st) {
    if (!s || !st) return SQPH_ERR_INVALID;
    *st = s->settings;
    return SQPH_OK;
}

int sqph_enable_timing(sqph_solver *s, int on) {
    if (!s) return SQPH_ERR_INVALID;
    s->timing = on != 0;
    s->ev_used = 0;
    return SQPH_OK;
}

int sqph_last_kernel_ms(sqph_solver *s, float *ms) {
    if (!s || !ms) return SQPH_ERR_INVALID;
    if (!s->timing || s->ev_used == 0) SQPH_FAIL(s, SQPH_ERR_INVALID, "sqph_last_kernel_ms: timing not enabled or nothing launched");
    DeviceGuard g(s->device);
    auto &p = s->evs[s->ev_used - 1];
    SQPH_HIP(s, hipEventSynchronize(p.second));
    SQPH_HIP(s, hipEventElapsedTime(ms, p.first, p.second));
    return SQPH_OK;
}

int sqph_collect_kernel_ms(sqph_solver *s, float *ms, int cap, int *count) {
    if (!s || !count) return SQPH_ERR_INVALID;
    DeviceGuard g(s->device);
    int k = 0;
    for (size_t i = 0; i < s->ev_used; i++) {
        auto &p = s->evs[i];
        SQPH_HIP(s, hipEventSynchronize(p.second));
        float t = 0;
        SQPH_HIP(s, hipEventElapsedTime(&t, p.first, p.second));
        if (ms && k < cap) ms[k] = t;
        k++;
    }
    *count = k;
    s->ev_used = 0;
    return SQPH_OK;
}

int sqph_synchronize(sqph_solver *s) {
    if (!s) return SQPH_ERR_INVALID;
    DeviceGuard g(s->device);
    SQPH_HIP(s, hipStreamSynchronize(s->stream));
    return SQPH_OK;
}

int sqph_device_state(sqph_solver *s, void **x, void **y, void **z, sqph_info **info) {
    if (!s) return SQPH_ERR_INVALID;
    if (x) *x = s->x;
    if (y) *y = s->y;
    if (z) *z = s->z;
    if (info) *info = s->info;
    return SQPH_OK;
}

int sqph_get_solution(sqph_solver *s, int batch, int memspace, void *x, void *y, void *z, sqph_info *info) {
    if (!s) return SQPH_ERR_INVALID;
    if (batch < 0 || batch > s->cap) SQPH_FAIL(s, SQPH_ERR_INVALID, "sqph_get_solution: batch %d exceeds capacity %d", batch, s->cap);
    DeviceGuard g(s->device);
    const size_t B = (size_t)batch;
    const hipMemcpyKind kind = memspace == SQPH_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    struct Item { void *dst; const void *src; size_t elems; };
    const Item items[3] = {{x, s->x, B * s->n}, {y, s->y, B * s->m}, {z, s->z, B * s->m}};
    {
        const size_t w0 = x ? B * s->n : 0, w1 = y ? B * s->m : 0, w2 = z ? B * s->m : 0, w3 = info ? B * (sizeof(sqph_info) / 8) : 0;
        const size_t words = w0 + w1 + w2 + w3;
        if (memspace == SQPH_HOST && words > 0 && words * 8 <= SMALL_CALL_BYTES) {
            if (words * 8 > s->hout_cap) {
                if (s->hout) (void)hipHostFree(s->hout);
                if (s->dout) (void)hipFree(s->dout);
                s->hout = s->dout = s->hout_dev = nullptr;
                s->hout_cap = 0;
                const size_t cap = words * 8 < 65536 ? 65536 : words * 8;
                SQPH_HIP(s, hipHostMalloc(&s->hout, cap, hipHostMallocMapped | hipHostMallocCoherent));
                SQPH_HIP(s, ws_malloc(&s->dout, cap));
                if (hipHostGetDevicePointer(&s->hout_dev, s->hout, 0) != hipSuccess) s->hout_dev = nullptr;
                s->hout_cap = cap;
            }
            const bool zc = zero_copy_enabled() && s->hout_dev;
            hipLaunchKernelGGL(pack_words, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s->stream,
                               (const unsigned long long *)s->x, w0, (const unsigned long long *)s->y, w1,
                               (const unsigned long long *)s->z, w2, (const unsigned long long *)s->info, w3,
                               (unsigned long long *)(zc ? s->hout_dev : s->dout));
            SQPH_HIP(s, hipGetLastError());
            if (!zc) SQPH_HIP(s, hipMemcpyAsync(s->hout, s->dout, words * 8, hipMemcpyDeviceToHost, s->stream));
            SQPH_HIP(s, wait_stream_low_latency(s->stream));
            const double *h = (const double *)s->hout;
            void *dsts[3] = {x, y, z};
            const size_t ws[3] = {w0, w1, w2};
            for (int k = 0; k < 3; k++) {
                if (!ws[k]) continue;
                if (s->dtype == SQPH_F64) {
                    memcpy(dsts[k], h, ws[k] * 8);
                } else {  // QPSolver<float>: state is kept in fp64 on the device, narrowed on the way out
                    float *d = (float *)dsts[k];
                    for (size_t i = 0; i < ws[k]; i++) d[i] = (float)h[i];
                }
                h += ws[k];
            }
            if (w3) memcpy(info, h, w3 * 8);
            return SQPH_OK;
        }
    }
    for (const Item &it : items) {
        if (!it.dst || it.elems == 0) continue;
        if (s->dtype == SQPH_F64) {
            SQPH_HIP(s, hipMemcpyAsync(it.dst, it.src, it.elems * sizeof(double), kind, s->stream));
        } else if (memspace == SQPH_DEVICE) {
            hipLaunchKernelGGL(cvt_f64_to_f32, dim3((unsigned)((it.elems + 255) / 256)), dim3(256), 0, s->stream,
                               (const double *)it.src, (float *)it.dst, it.elems);
            SQPH_HIP(s, hipGetLastError());
        } else {
            // QPSolver<float>: state is kept in fp64 on the device, narrowed on the way out
            std::vector<double> tmp(it.elems);
            SQPH_HIP(s, hipMemcpyAsync(tmp.data(), it.src, it.elems * sizeof(double), hipMemcpyDeviceToHost, s->stream));
            SQPH_HIP(s, hipStreamSynchronize(s->stream));
            float *d = (float *)it.dst;
            for (size_t i = 0; i < it.elems; i++) d[i] = (float)tmp[i];
        }
    }
    if (info) SQPH_HIP(s, hipMemcpyAsync(info, s->info, B * sizeof(sqph_info), kind, s->stream));
    if (memspace == SQPH_HOST) SQPH_HIP(s, hipStreamSynchronize(s->stream));
    return SQPH_OK;
}

int sqph_set_state(sqph_solver *s, int batch, int memspace, const void *x, const void *z, const void *y) {
    if (!s) return SQPH_ERR_INVALID;
    if (batch < 0 || batch > s->cap) SQPH_FAIL(s, SQPH_ERR_INVALID, "sqph_set_state: batch %d exceeds capacity %d", batch, s->cap);
    DeviceGuard g(s->device);
    const size_t B = (size_t)batch;
    const hipMemcpyKind kind = memspace == SQPH_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    struct Item { void *dst; const void *src; size_t elems; };
    const Item items[3] = {{s->x, x, B * s->n}, {s->z, z, B * s->m}, {s->y, y, B * s->m}};
    for (const Item &it : items) {
        if (!it.src || it.elems == 0) continue;
        if (s->dtype == SQPH_F64) {
            SQPH_HIP(s, hipMemcpyAsync(it.dst, it.src, it.elems * sizeof(double), kind, s->stream));
        } else if (memspace == SQPH_DEVICE) {
            hipLaunchKernelGGL(cvt_f32_to_f64, dim3((unsigned)((it.elems + 255) / 256)), dim3(256), 0, s->stream,
                               (const float *)it.src, (double *)it.dst, it.elems);
            SQPH_HIP(s, hipGetLastError());
        } else {
            std::vector<double> tmp(it.elems);
            const float *f = (const float *)it.src;
            for (size_t i = 0; i < it.elems; i++) tmp[i] = (double)f[i];
            SQPH_HIP(s, hipMemcpyAsync(it.dst, tmp.data(), it.elems * sizeof(double), hipMemcpyHostToDevice, s->stream));
            SQPH_HIP(s, hipStreamSynchronize(s->stream));
        }
    }
    if (memspace == SQPH_HOST) SQPH_HIP(s, hipStreamSynchronize(s->stream));
    return SQPH_OK;
}

}  // extern "C"

namespace {

// Scalar constants of the reference class (qp.hpp:136-141) and the settings are rounded through the
// interface Scalar (TIN) first, then widened to the fp64 the kernels compute in.
struct CsrDesc {  // device-resident CSR arrays of the native sparse path
    const int *rowptr, *colind;
    const void *val;
    long long s_rowptr, s_colind, s_val;
    int nnz_cap, TT;
    int NB;  // > 0: the block-row kernel (admm_csrb_kernel.h) with this block-row count; 0: the 32 x 32 lane-grid kernel with tile edge TT
    // P in compressed columns, device-resident (sqph_*_csr_sp on the block-row kernel: read in place by its sparse-P instantiations)
    const int *p_colptr = nullptr, *p_rowind = nullptr;
    const void *p_val = nullptr;
    long long s_pcolptr = 0, s_prowind = 0, s_pval = 0;
};

template <typename TIN>
int launch_typed(sqph_solver *s, const sqph_qp_batch *qp, int mode, const void *P, const void *q, const void *A,
                 const void *l, const void *u, long long sP, long long sq, long long sA, long long sl, long long su,
                 const CsrDesc *csr = nullptr) {
    using namespace sqph;
    using T = double;
    KArgs<T, TIN> a{};
    a.n = s->n;
    a.m = s->m;
    a.batch = qp->batch;
    a.mode = mode | ((s->flags & SQPH_FLAG_LEGACY_COLD_START) ? MODE_COLD_RESET : 0);
    // factor residency: a fused setup+solve writes no factor unless asked to; solve() on such a handle — or on one whose factor
    // was built by another kernel family (the layouts differ) — rebuilds it first (MODE_REFACTOR, set per family below)
    const bool fused = (mode & (MODE_SETUP | MODE_UPDATE)) && (mode & MODE_SOLVE);
    if (fused && !(s->flags & SQPH_FLAG_KEEP_FACTOR)) a.mode |= MODE_NO_FACTOR_STORE;
    const bool solve_only = !(mode & (MODE_SETUP | MODE_UPDATE));
    const int mode_in = a.mode;
    auto for_family = [&](char fam) {  // mode bits for a launch by kernel family `fam`: always derived from the call's own mode
        int mbits = mode_in & ~(MODE_REFACTOR | (fam == 'g' ? MODE_NO_FACTOR_STORE : 0));
        if (solve_only && (!s->factor_resident || s->factor_family != fam)) mbits |= MODE_REFACTOR;
        if ((mbits & MODE_SAME_MATRICES) && (!s->factor_resident || s->factor_family != fam)) mbits &= ~MODE_SAME_MATRICES;
        return mbits;
    };
    auto launched_by = [&](char fam, int mbits) {
        if ((mbits & (MODE_SETUP | MODE_UPDATE | MODE_REFACTOR))) {
            s->factor_resident = fam == 'g' || !(mbits & MODE_NO_FACTOR_STORE);  // the generic kernel iterates out of the workspace
            s->factor_family = fam;
        }
    };
    a.P = (const TIN *)P; a.q = (const TIN *)q; a.A = (const TIN *)A; a.l = (const TIN *)l; a.u = (const TIN *)u;
    a.sP = sP; a.sq = sq; a.sA = sA; a.sl = sl; a.su = su;
    a.x = (T *)s->x; a.z = (T *)s->z; a.y = (T *)s->y; a.rho_vec = (T *)s->rho_vec; a.ctype = s->ctype;
    a.rho = (T *)s->rho; a.info = s->info; a.Sinv = (T *)s->Sinv;
    const sqph_settings &st = s->settings;
    a.rho0 = (T)(TIN)st.rho; a.sigma = (T)(TIN)st.sigma; a.alpha = (T)(TIN)st.alpha;
    a.eps_rel = (T)(TIN)st.eps_rel; a.eps_abs = (T)(TIN)st.eps_abs; a.rho_tol = (T)(TIN)st.adaptive_rho_tolerance;
    a.rho_min = (T)(TIN)1e-6; a.rho_max = (T)(TIN)1e+6; a.eq_tol = (T)(TIN)1e-4; a.rho_eq_factor = (T)(TIN)1e+3;
    a.loose_thresh = (T)(TIN)1e+16; a.regul = (T)std::numeric_limits<TIN>::epsilon();
    a.max_iter = st.max_iter; a.check_termination = st.check_termination; a.warm_start = st.warm_start;
    a.adaptive_rho = st.adaptive_rho; a.adaptive_rho_interval = st.adaptive_rho_interval;

#ifdef SQPH_EXPERIMENTS
    {   // experiment builds only: knobs SQPH_XP0..7, and (SQPH_XDBG=<file>) 8 words per workgroup dumped after the launch
        for (int k = 0; k < 8; k++) {
            char nm[16];
            snprintf(nm, sizeof(nm), "SQPH_XP%d", k);
            a.xp[k] = getenv(nm) ? atoi(getenv(nm)) : 0;
        }
        a.xdbg = nullptr;
        if (getenv("SQPH_XDBG")) {
            static unsigned long long *dbg = nullptr;
            static size_t dbg_cap = 0;
            if ((size_t)qp->batch > dbg_cap) {
                if (dbg) (void)hipFree(dbg);
                SQPH_HIP(s, ws_malloc((void **)&dbg, (size_t)qp->batch * 8 * sizeof(unsigned long long)));
                dbg_cap = qp->batch;
            }
            SQPH_HIP(s, hipMemsetAsync(dbg, 0, (size_t)qp->batch * 8 * sizeof(unsigned long long), s->stream));
            a.xdbg = dbg;
        }
    }
#endif
    const bool verbose = st.verbose != 0 && (mode & MODE_SOLVE);
    if (verbose) {
        const int cap = (st.check_termination > 0 ? st.max_iter / st.check_termination : 0) + 2;
        if (cap > s->trace_cap) {
            if (s->trace) (void)hipFree(s->trace);
            s->trace = nullptr;
            s->trace_cap = 0;
            SQPH_HIP(s, ws_malloc((void **)&s->trace, (1 + 4 * (size_t)cap) * sizeof(double)));
            s->trace_cap = cap;
        }
        SQPH_HIP(s, hipMemsetAsync(s->trace, 0, sizeof(double), s->stream));
        a.trace = s->trace;
        a.trace_qp = s->trace_qp;
        a.trace_cap = s->trace_cap;
    } else if (s->trace) {
        SQPH_HIP(s, hipMemsetAsync(s->trace, 0, sizeof(double), s->stream));  // the last call recorded nothing
    }

    if (s->timing) {
        if (s->ev_used == s->evs.size()) {
            hipEvent_t e0, e1;
            SQPH_HIP(s, hipEventCreate(&e0));
            SQPH_HIP(s, hipEventCreate(&e1));
            s->evs.emplace_back(e0, e1);
        }
        SQPH_HIP(s, hipEventRecord(s->evs[s->ev_used].first, s->stream));
    }

    bool launched = false;
    if (csr) {
        CsrLaunch<TIN> p;
        a.mode = for_family('c');
        p.a = a;
        p.ca = CsrArgs<TIN>{csr->rowptr, csr->colind, (const TIN *)csr->val, csr->s_rowptr, csr->s_colind, csr->s_val, csr->nnz_cap,
                            csr->p_colptr, csr->p_rowind, (const TIN *)csr->p_val, csr->s_pcolptr, csr->s_prowind, csr->s_pval};
        const bool sparse_P = csr->p_colptr != nullptr;
        const bool nocheck = st.check_termination <= 0 && !(st.adaptive_rho && st.adaptive_rho_interval > 0);
        if (csr->NB > 0) {  // block-row kernel: 512 lanes per QP, W as MFMA blocks in registers (csrb.hip)
            const int rc = sparse_P ? csrb_sp_launch<TIN>(csr->NB, nocheck, s->m, csr->nnz_cap, qp->batch, s->stream, p)
                                    : csrb_launch<TIN>(csr->NB, nocheck, s->m, csr->nnz_cap, qp->batch, s->stream, p);
            if (rc < 0) SQPH_FAIL(s, SQPH_ERR_HIP, "sparse kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
            launched = rc > 0;
            if (launched && sparse_P)
                s->kernel_name = csr->NB == 14 ? "csb_nb14_sp" : csr->NB == 13 ? "csb_nb13_sp" : csr->NB == 9 ? "csb_nb9_sp" : csr->NB == 5 ? "csb_nb5_sp" : "csb_sp";
            else if (launched) s->kernel_name = csr->NB == 14 ? "csb_nb14" : csr->NB == 13 ? "csb_nb13" : csr->NB == 9 ? "csb_nb9" : csr->NB == 5 ? "csb_nb5" : "csb";
        }
        if (!launched && sparse_P) SQPH_FAIL(s, SQPH_ERR_UNSUPPORTED, "internal: sparse P handed to a kernel that reads it dense");
        {
        }
        // calls that never look at the residuals take the instantiation without the check block (csr_nocheck.hip)
        if (!launched && nocheck) {
            const int rc = csr_nocheck_launch<TIN>(csr->TT, s->m, csr->nnz_cap, qp->batch, s->stream, p);
            if (rc < 0) SQPH_FAIL(s, SQPH_ERR_HIP, "sparse kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
            if (rc > 0) {
                s->kernel_name = csr->TT == 7 ? "csr_t7" : csr->TT == 4 ? "csr_t4" : "csr";
                launched = true;
            }
        }
#define SQPH_CSR_CASE(TT_)                                                                                                          \
    if (!launched && csr->TT == TT_) {                                                                                              \
        const CsrLayout<TT_> L = CsrLayout<TT_>::make(s->m, csr->nnz_cap);                                                          \
        SQPH_HIP(s, hipFuncSetAttribute((const void *)admm_csr_kernel<TIN, TT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes)); \
        hipLaunchKernelGGL((admm_csr_kernel<TIN, TT_>), dim3(qp->batch), dim3(1024), L.bytes, s->stream, p);                        \
        SQPH_HIP(s, hipGetLastError());                                                                                             \
        s->kernel_name = "csr_t" #TT_;                                                                                              \
        launched = true;                                                                                                            \
    }
        SQPH_CSR_SHAPES(SQPH_CSR_CASE)
#undef SQPH_CSR_CASE
        if (!launched) SQPH_FAIL(s, SQPH_ERR_UNSUPPORTED, "no sparse kernel for tile edge %d", csr->TT);
        launched_by('c', a.mode);
    }
    if (!launched && verbose && !(s->flags & SQPH_FLAG_FORCE_GENERIC)) {
        // only the one-QP-per-lane and the generic kernel record the trace
        a.mode = for_family('t');
        int rc = lane_try_launch<TIN>(a, s->stream, &s->kernel_name);
        if (rc < 0) SQPH_FAIL(s, SQPH_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        launched = rc > 0;
        if (launched) launched_by('t', a.mode);
    }
    if (!launched && !verbose && !(s->flags & SQPH_FLAG_FORCE_GENERIC)) {
        int rc = 0;
        a.mode = for_family('t');
        if (sizeof(TIN) == 4 && (s->flags & SQPH_FLAG_F32_ARITH)) rc = lane_try_launch<TIN, float>(a, s->stream, &s->kernel_name);
        if constexpr (sizeof(TIN) == 4) {
            if (rc == 0 && (s->flags & SQPH_FLAG_F32_ARITH)) rc = wgf_try_launch(a, s->stream, &s->kernel_name);  // wg_f32.hip
        }
        if (rc == 0) rc = lane_try_launch<TIN>(a, s->stream, &s->kernel_name);
        if (rc == 0) rc = g16_try_launch<TIN>(a, s->stream, &s->kernel_name);
        if (rc == 0) rc = wg_try_launch<TIN>(a, s->stream, &s->kernel_name);
        if (rc < 0) SQPH_FAIL(s, SQPH_ERR_HIP, "tiled kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        launched = rc > 0;
        if (launched) launched_by('t', a.mode);
    }
    if (!launched && !verbose && !(s->flags & SQPH_FLAG_FORCE_GENERIC)) {
        // dense problems beyond the register-tiled shapes, n <= 256 and m <= 512: W in the CU's registers, A streamed (csr_dense.hip)
        a.mode = for_family('c');
        const int rc = csrd_try_launch<TIN>(a, s->stream, &s->kernel_name);
        if (rc < 0) SQPH_FAIL(s, SQPH_ERR_HIP, "dense CU-wide kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        launched = rc > 0;
        if (launched) launched_by('c', a.mode);
    }
    if (!launched) {
        if (!s->At) {
            const size_t bytes = (size_t)s->cap * (size_t)(s->m > 0 ? s->m : 1) * s->n * sizeof(T);
            SQPH_HIP(s, ws_malloc(&s->At, bytes));
        }
        a.At = (T *)s->At;
        a.mode = for_family('g');
        const int big = s->n > s->m ? s->n : s->m;
#ifdef SQPH_EXPERIMENTS
        static const int nt_env = getenv("SQPH_GENERIC_NT") ? atoi(getenv("SQPH_GENERIC_NT")) : 0;
#else
        constexpr int nt_env = 0;
#endif
        // 4 waves were 8 waves per CU at these sizes (LDS): too few loads in flight to stream the matrices — measured 1.2-2.9x with 8 / 16
        int nt = nt_env > 0 ? nt_env : (big <= 128 ? 64 : ((long long)s->n * s->m >= 20000 ? 1024 : 512));
        while (nt > 256 && generic_lds_elems<T>(s->n, s->m, nt) * sizeof(T) > 160 * 1024) nt /= 2;  // the per-thread scratch must leave room for the vectors
        const size_t lds = generic_lds_elems<T>(s->n, s->m, nt) * sizeof(T);
        if (lds > 160 * 1024) SQPH_FAIL(s, SQPH_ERR_UNSUPPORTED, "n=%d m=%d needs %zu B of LDS (>160 KiB)", s->n, s->m, lds);
        if (lds > 64 * 1024)
            SQPH_HIP(s, hipFuncSetAttribute((const void *)admm_generic_kernel<T, TIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((admm_generic_kernel<T, TIN>), dim3(qp->batch), dim3(nt), lds, s->stream, a);
        SQPH_HIP(s, hipGetLastError());
        s->kernel_name = nt == 64 ? "generic_w1" : nt == 256 ? "generic_w4" : nt == 512 ? "generic_w8" : "generic_w16";
        launched_by('g', a.mode);
    }
    if (s->timing) {
        SQPH_HIP(s, hipEventRecord(s->evs[s->ev_used].second, s->stream));
        s->ev_used++;
    }
#ifdef SQPH_EXPERIMENTS
    if (a.xdbg) {
        SQPH_HIP(s, hipStreamSynchronize(s->stream));
        std::vector<unsigned long long> h((size_t)qp->batch * 8);
        SQPH_HIP(s, hipMemcpy(h.data(), a.xdbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(getenv("SQPH_XDBG"), "wb")) {
            fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
            fclose(f);
        }
    }
#endif
    return SQPH_OK;
}

int run(sqph_solver *s, const sqph_qp_batch *qp, int mode, const char *what, const CsrDesc *csr = nullptr) {
    if (!s) return SQPH_ERR_INVALID;
    if (!qp) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: qp is null", what);
    if (qp->batch < 0 || qp->batch > s->cap) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: batch %d exceeds capacity %d", what, qp->batch, s->cap);
    if (qp->batch == 0) return SQPH_OK;
    if (!qp->P || !qp->q || (s->m > 0 && ((!csr && !qp->A) || !qp->l || !qp->u))) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: null problem pointer", what);
    if (qp->stride_P < 0 || qp->stride_q < 0 || qp->stride_A < 0 || qp->stride_l < 0 || qp->stride_u < 0)
        SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: negative stride", what);
    if (qp->memspace != SQPH_HOST && qp->memspace != SQPH_DEVICE) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: bad memspace %d", what, qp->memspace);

    DeviceGuard g(s->device);
    const void *P = qp->P, *q = qp->q, *A = qp->A, *l = qp->l, *u = qp->u;
    long long sP = qp->stride_P, sq = qp->stride_q, sA = qp->stride_A, sl = qp->stride_l, su = qp->stride_u;
    const size_t e = dsize(s->dtype);
    const size_t n = s->n, m = s->m;
    bool staged = false, pin_zero_copy = false;
    if (qp->memspace == SQPH_HOST) {
        // small calls: pack everything into one pinned buffer (read in place by the kernel, or one H2D)
        struct PItem { const void *src; size_t elems; long long *stride; const void **out; size_t off, bytes; };
        PItem pit[5] = {{qp->P, n * n, &sP, &P, 0, 0}, {qp->q, n, &sq, &q, 0, 0}, {qp->A, csr ? 0 : m * n, &sA, &A, 0, 0},
                        {qp->l, m, &sl, &l, 0, 0}, {qp->u, m, &su, &u, 0, 0}};
        size_t total = 0;
        for (auto &it : pit) {
            it.bytes = it.elems * e * (*it.stride == 0 ? 1 : (size_t)qp->batch);
            it.off = total;
            total += (it.bytes + 15) & ~(size_t)15;
        }
        if (total > 0 && total <= SMALL_CALL_BYTES) {
            if (total > s->hpin_cap) {
                if (s->pin_busy) SQPH_HIP(s, hipEventSynchronize(s->pin_ev));
                s->pin_busy = false;
                if (s->hpin) (void)hipHostFree(s->hpin);
                if (s->dpin) (void)hipFree(s->dpin);
                s->hpin = s->dpin = s->hpin_dev = nullptr;
                s->hpin_cap = 0;
                const size_t cap = total < 65536 ? 65536 : total;
                SQPH_HIP(s, hipHostMalloc(&s->hpin, cap, hipHostMallocMapped | hipHostMallocCoherent));
                SQPH_HIP(s, ws_malloc(&s->dpin, cap));
                if (hipHostGetDevicePointer(&s->hpin_dev, s->hpin, 0) != hipSuccess) s->hpin_dev = nullptr;
                s->hpin_cap = cap;
            }
            if (!s->pin_ev) SQPH_HIP(s, hipEventCreateWithFlags(&s->pin_ev, hipEventDisableTiming));
            if (s->pin_busy) SQPH_HIP(s, hipEventSynchronize(s->pin_ev));  // the previous call has consumed the buffer
            s->pin_busy = false;
            const bool zc = zero_copy_enabled() && s->hpin_dev && !csr && total <= ZERO_COPY_BYTES;
            for (auto &it : pit) {
                if (it.elems == 0) continue;
                char *dstp = (char *)s->hpin + it.off;
                if (*it.stride == 0 || (size_t)*it.stride == it.elems) {
                    memcpy(dstp, it.src, it.bytes);
                } else {
                    for (int b = 0; b < qp->batch; b++)
                        memcpy(dstp + (size_t)b * it.elems * e, (const char *)it.src + (size_t)b * (size_t)*it.stride * e, it.elems * e);
                    *it.stride = (long long)it.elems;
                }
                *it.out = (const char *)(zc ? s->hpin_dev : s->dpin) + it.off;
            }
            if (!zc) {
                SQPH_HIP(s, hipMemcpyAsync(s->dpin, s->hpin, total, hipMemcpyHostToDevice, s->stream));
                SQPH_HIP(s, hipEventRecord(s->pin_ev, s->stream));
                s->pin_busy = true;
            }
            pin_zero_copy = zc;  // the kernel itself reads the pinned buffer: it is busy until the launch below has finished
            staged = true;
        }
    }
    if (qp->memspace == SQPH_HOST && !staged) {
        // stage host data: one H2D per array into packed device buffers owned by the solver
        struct Item { const void *src; void **dst; size_t elems; long long *stride; };
        Item items[5] = {{qp->P, &s->sP, n * n, &sP}, {qp->q, &s->sq, n, &sq}, {qp->A, &s->sA, csr ? 0 : m * n, &sA},
                         {qp->l, &s->sl, m, &sl}, {qp->u, &s->su, m, &su}};
        for (auto &it : items) {
            if (it.elems == 0) continue;
            if (!*it.dst) SQPH_HIP(s, ws_malloc(it.dst, (size_t)s->cap * it.elems * e));
            if (*it.stride == 0) {
                SQPH_HIP(s, hipMemcpyAsync(*it.dst, it.src, it.elems * e, hipMemcpyHostToDevice, s->stream));
            } else if ((size_t)*it.stride == it.elems) {
                SQPH_HIP(s, hipMemcpyAsync(*it.dst, it.src, (size_t)qp->batch * it.elems * e, hipMemcpyHostToDevice, s->stream));
            } else {
                SQPH_HIP(s, hipMemcpy2DAsync(*it.dst, it.elems * e, it.src, (size_t)*it.stride * e, it.elems * e, (size_t)qp->batch,
                                             hipMemcpyHostToDevice, s->stream));
                *it.stride = (long long)it.elems;
            }
        }
        P = s->sP; q = s->sq; A = s->sA; l = s->sl; u = s->su;
        // pageable host memory: the async copies above have already consumed the caller's buffers
        // only after the stream reaches them; make the borrow end with the call.
        SQPH_HIP(s, hipStreamSynchronize(s->stream));
    }
    const int rc = s->dtype == SQPH_F32 ? launch_typed<float>(s, qp, mode, P, q, A, l, u, sP, sq, sA, sl, su, csr)
                                        : launch_typed<double>(s, qp, mode, P, q, A, l, u, sP, sq, sA, sl, su, csr);
    if (rc == SQPH_OK && pin_zero_copy) {
        SQPH_HIP(s, hipEventRecord(s->pin_ev, s->stream));
        s->pin_busy = true;
    }
    return rc;
}

// sqph_*_csr_sp: P in compressed-column form.  stage_sparse_P validates the descriptor and makes the arrays device-resident (*dv);
// with host memspace q, l, u are staged as well and *d becomes a device-memspace batch (the CSR arrays of A stay with run_csr).
// place_sparse_P then either checks the structure only (expand = false: the block-row kernel's sparse-P instantiations read the
// columns in place) or expands them into the handle's dense workspace, which d->P then points to (every other route).
struct SparsePDev {
    const int *colptr, *rowind;
    const void *val;
    long long s_ptr, s_ind, s_val, nnz_max;
};
static int stage_sparse_P(sqph_solver *s, const sqph_csr_batch *c, const sqph_csc_P *sp, sqph_qp_batch *d, SparsePDev *dv, const char *what) {
    const size_t e = dsize(s->dtype), B = (size_t)c->batch, n = s->n, m = s->m;
    if (!sp->colptr || (sp->nnz_max > 0 && (!sp->rowind || !sp->val))) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: null pointer in the sparse P", what);
    if (sp->stride_colptr < 0 || sp->stride_rowind < 0 || sp->stride_val < 0 || sp->nnz_max < 0)
        SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: negative stride / nnz_max in the sparse P", what);
    if ((sp->stride_rowind && sp->stride_rowind < sp->nnz_max) || (sp->stride_val && sp->stride_val < sp->nnz_max))
        SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: sparse P: stride_rowind/stride_val smaller than nnz_max", what);
    if ((sp->stride_colptr == 0) != (sp->stride_rowind == 0))
        SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: sparse P: colptr and rowind must both be shared or both per-QP", what);
    if (sp->stride_colptr && sp->stride_colptr < (long long)n + 1) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: sparse P: stride_colptr smaller than n+1", what);
    if (sp->stride_val == 0 && sp->stride_colptr != 0) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: sparse P: shared values need a shared pattern", what);
    if (c->memspace == SQPH_HOST && (!c->q || (m > 0 && (!c->l || !c->u)))) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: null problem pointer", what);
    if (c->stride_q < 0 || c->stride_l < 0 || c->stride_u < 0) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: negative stride", what);

    DeviceGuard g(s->device);
    const int *colptr = sp->colptr, *rowind = sp->rowind;
    const void *val = sp->val;
    long long s_ptr = sp->stride_colptr, s_ind = sp->stride_rowind, s_val = sp->stride_val;
    if (c->memspace == SQPH_HOST) {
        const size_t nind = s_ind ? B * (size_t)s_ind : (size_t)sp->nnz_max;
        const size_t nval = s_val ? B * (size_t)s_val : (size_t)sp->nnz_max;
        if (!s->pPtr) SQPH_HIP(s, ws_malloc(&s->pPtr, (size_t)s->cap * (n + 1) * sizeof(int)));
        if (s_ptr && (size_t)s_ptr != n + 1) {
            SQPH_HIP(s, hipMemcpy2DAsync(s->pPtr, (n + 1) * sizeof(int), colptr, (size_t)s_ptr * sizeof(int), (n + 1) * sizeof(int), B,
                                         hipMemcpyHostToDevice, s->stream));
            s_ptr = (long long)(n + 1);
        } else {
            SQPH_HIP(s, hipMemcpyAsync(s->pPtr, colptr, (s_ptr ? B * (n + 1) : n + 1) * sizeof(int), hipMemcpyHostToDevice, s->stream));
        }
        if (nind > s->pInd_cap) {
            if (s->pInd) (void)hipFree(s->pInd);
            s->pInd = nullptr;
            s->pInd_cap = 0;
            SQPH_HIP(s, ws_malloc(&s->pInd, (nind ? nind : 1) * sizeof(int)));
            s->pInd_cap = nind;
        }
        if (nval > s->pVal_cap) {
            if (s->pVal) (void)hipFree(s->pVal);
            s->pVal = nullptr;
            s->pVal_cap = 0;
            SQPH_HIP(s, ws_malloc(&s->pVal, (nval ? nval : 1) * e));
            s->pVal_cap = nval;
        }
        if (nind) SQPH_HIP(s, hipMemcpyAsync(s->pInd, rowind, nind * sizeof(int), hipMemcpyHostToDevice, s->stream));
        if (nval) SQPH_HIP(s, hipMemcpyAsync(s->pVal, val, nval * e, hipMemcpyHostToDevice, s->stream));
        colptr = (const int *)s->pPtr; rowind = (const int *)s->pInd; val = s->pVal;
        // q, l, u: the dense remainder, staged by hand (the dense entry would expect a host P)
        struct Item { const void *src; void **dst; size_t elems; long long stride; const void **out; long long *sout; };
        Item items[3] = {{c->q, &s->sq, n, c->stride_q, &d->q, &d->stride_q}, {c->l, &s->sl, m, c->stride_l, &d->l, &d->stride_l},
                         {c->u, &s->su, m, c->stride_u, &d->u, &d->stride_u}};
        for (auto &it : items) {
            if (it.elems == 0) continue;
            if (!*it.dst) SQPH_HIP(s, ws_malloc(it.dst, (size_t)s->cap * it.elems * e));
            if (it.stride == 0) {
                SQPH_HIP(s, hipMemcpyAsync(*it.dst, it.src, it.elems * e, hipMemcpyHostToDevice, s->stream));
            } else if ((size_t)it.stride == it.elems) {
                SQPH_HIP(s, hipMemcpyAsync(*it.dst, it.src, B * it.elems * e, hipMemcpyHostToDevice, s->stream));
            } else {
                SQPH_HIP(s, hipMemcpy2DAsync(*it.dst, it.elems * e, it.src, (size_t)it.stride * e, it.elems * e, B, hipMemcpyHostToDevice, s->stream));
                *it.sout = (long long)it.elems;
            }
            *it.out = *it.dst;
        }
        d->memspace = SQPH_DEVICE;
    }
    *dv = SparsePDev{colptr, rowind, val, s_ptr, s_ind, s_val, sp->nnz_max};
    return SQPH_OK;
}
static int place_sparse_P(sqph_solver *s, int batch, const SparsePDev &dv, bool expand, sqph_qp_batch *d, const char *what) {
    const size_t e = dsize(s->dtype), B = (size_t)batch, n = s->n;
    DeviceGuard g(s->device);
    const bool shared = dv.s_val == 0;
    const size_t nexp = shared ? 1 : B, ncheck = dv.s_ptr == 0 ? 1 : B;  // (a shared pattern is checked once)
    if (expand && !s->cP) SQPH_HIP(s, ws_malloc(&s->cP, (size_t)s->cap * n * n * e));
    if (!s->cBad) SQPH_HIP(s, ws_malloc((void **)&s->cBad, sizeof(int)));
    if (expand) SQPH_HIP(s, hipMemsetAsync(s->cP, 0, nexp * n * n * e, s->stream));
    SQPH_HIP(s, hipMemsetAsync(s->cBad, 0, sizeof(int), s->stream));
    const size_t nk = expand ? nexp : ncheck;
    const unsigned blocks = (unsigned)((nk * n + 255) / 256);
    if (s->dtype == SQPH_F32)
        hipLaunchKernelGGL((csc_expand_P<float>), dim3(blocks), dim3(256), 0, s->stream, (int)nk, (int)n, dv.colptr, dv.rowind, (const float *)dv.val,
                           dv.s_ptr, dv.s_ind, dv.s_val, (long long)dv.nnz_max, expand ? (float *)s->cP : (float *)nullptr, s->cBad);
    else
        hipLaunchKernelGGL((csc_expand_P<double>), dim3(blocks), dim3(256), 0, s->stream, (int)nk, (int)n, dv.colptr, dv.rowind, (const double *)dv.val,
                           dv.s_ptr, dv.s_ind, dv.s_val, (long long)dv.nnz_max, expand ? (double *)s->cP : (double *)nullptr, s->cBad);
    SQPH_HIP(s, hipGetLastError());
    int bad = 0;
    SQPH_HIP(s, hipMemcpyAsync(&bad, s->cBad, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    SQPH_HIP(s, hipStreamSynchronize(s->stream));  // (also ends the borrow of the pageable host arrays staged above)
    if (bad)
        SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: malformed sparse P (%s)", what,
                  (bad & 1) ? "column pointers not monotone" : (bad & 2) ? "row index out of range"
                  : (bad & 4) ? "row indices of a column not strictly increasing"
                  : (bad & 8) ? "pattern not symmetric: the full matrix is expected, not a triangle" : "values not symmetric: P(i,j) != P(j,i)");
    if (expand) {
        d->P = s->cP;
        d->stride_P = shared ? 0 : (long long)(n * n);
    } else {
        d->P = s->cBad;  // (never read: the kernel takes P from the CsrDesc; run() wants a non-null pointer)
        d->stride_P = 0;
    }
    return SQPH_OK;
}

// CSR entry points: expand A on the device, then the dense path.
static int run_csr_impl(sqph_solver *s, const sqph_csr_batch *c, int mode, const char *what, const sqph_csc_P *sp);
int run_csr(sqph_solver *s, const sqph_csr_batch *c, int mode, const char *what, const sqph_csc_P *sp = nullptr) {
    const int rc = run_csr_impl(s, c, mode, what, sp);
    if (rc != SQPH_OK && s && c && c->memspace == SQPH_HOST) {
        // an error return ends the borrow of the caller's pageable host arrays like a success does: asynchronous copies out of
        // them may have been enqueued before the failing check (the error text set by the failing call is kept)
        DeviceGuard g(s->device);
        (void)hipStreamSynchronize(s->stream);
        (void)hipGetLastError();
    }
    return rc;
}
static int run_csr_impl(sqph_solver *s, const sqph_csr_batch *c, int mode, const char *what, const sqph_csc_P *sp) {
    if (!s) return SQPH_ERR_INVALID;
    if (!c) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: qp is null", what);
    if (c->batch < 0 || c->batch > s->cap) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: batch %d exceeds capacity %d", what, c->batch, s->cap);
    if (c->batch == 0) return SQPH_OK;
    if (c->memspace != SQPH_HOST && c->memspace != SQPH_DEVICE) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: bad memspace %d", what, c->memspace);
    sqph_qp_batch d{};
    d.batch = c->batch; d.memspace = c->memspace;
    d.P = c->P; d.q = c->q; d.l = c->l; d.u = c->u;
    d.stride_P = c->stride_P; d.stride_q = c->stride_q; d.stride_l = c->stride_l; d.stride_u = c->stride_u;
    SparsePDev spd{};
    bool sp_pending = false;  // P still in compressed columns: placed (checked / expanded) once the route is known
    if (sp) {  // P sparse as well: device-resident columns; host q, l, u staged (d becomes a device-memspace batch)
        const int rc = stage_sparse_P(s, c, sp, &d, &spd, what);
        if (rc != SQPH_OK) return rc;
        sp_pending = true;
    }
    auto dense_P = [&]() -> int {  // the routes that read P dense: expand it into the handle's workspace
        if (!sp_pending) return SQPH_OK;
        sp_pending = false;
        return place_sparse_P(s, c->batch, spd, true, &d, what);
    };
    if (s->m == 0) {
        const int rc = dense_P();
        return rc != SQPH_OK ? rc : run(s, &d, mode, what);
    }
    if (!c->A_rowptr || (c->nnz_max > 0 && (!c->A_colind || !c->A_val))) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: null CSR pointer", what);
    if (c->stride_rowptr < 0 || c->stride_colind < 0 || c->stride_val < 0 || c->nnz_max < 0)
        SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: negative stride / nnz_max", what);
    if ((c->stride_colind && c->stride_colind < c->nnz_max) || (c->stride_val && c->stride_val < c->nnz_max))
        SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: stride_colind/stride_val smaller than nnz_max", what);
    if ((c->stride_rowptr == 0) != (c->stride_colind == 0)) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: rowptr and colind must both be shared or both per-QP", what);
    if (c->stride_rowptr && c->stride_rowptr < s->m + 1) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: stride_rowptr smaller than m+1", what);
    if (c->stride_val == 0 && c->stride_rowptr != 0) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: shared values need a shared pattern", what);

    DeviceGuard g(s->device);
    const size_t e = dsize(s->dtype), B = (size_t)c->batch, n = s->n, m = s->m;
    const int *rowptr = c->A_rowptr, *colind = c->A_colind;
    const void *val = c->A_val;
    long long s_row = c->stride_rowptr, s_col = c->stride_colind, s_val = c->stride_val;
    if (c->memspace == SQPH_HOST) {
        const size_t nrow = s_row ? B * (size_t)s_row : m + 1;
        const size_t ncol = s_col ? B * (size_t)s_col : (size_t)c->nnz_max;
        const size_t nval = s_val ? B * (size_t)s_val : (size_t)c->nnz_max;
        if (!s->cRow) SQPH_HIP(s, ws_malloc(&s->cRow, (size_t)s->cap * (m + 1) * sizeof(int)));
        if (s_row && (size_t)s_row != m + 1) {
            SQPH_HIP(s, hipMemcpy2DAsync(s->cRow, (m + 1) * sizeof(int), rowptr, (size_t)s_row * sizeof(int), (m + 1) * sizeof(int), B,
                                         hipMemcpyHostToDevice, s->stream));
            s_row = (long long)(m + 1);
        } else {
            SQPH_HIP(s, hipMemcpyAsync(s->cRow, rowptr, (s_row ? B * (m + 1) : m + 1) * sizeof(int), hipMemcpyHostToDevice, s->stream));
        }
        (void)nrow;
        if (ncol > s->cCol_cap) {
            if (s->cCol) (void)hipFree(s->cCol);
            s->cCol = nullptr;
            SQPH_HIP(s, ws_malloc(&s->cCol, (ncol ? ncol : 1) * sizeof(int)));
            s->cCol_cap = ncol;
        }
        if (nval > s->cVal_cap) {
            if (s->cVal) (void)hipFree(s->cVal);
            s->cVal = nullptr;
            SQPH_HIP(s, ws_malloc(&s->cVal, (nval ? nval : 1) * e));
            s->cVal_cap = nval;
        }
        if (ncol) SQPH_HIP(s, hipMemcpyAsync(s->cCol, colind, ncol * sizeof(int), hipMemcpyHostToDevice, s->stream));
        if (nval) SQPH_HIP(s, hipMemcpyAsync(s->cVal, val, nval * e, hipMemcpyHostToDevice, s->stream));
        rowptr = (const int *)s->cRow; colind = (const int *)s->cCol; val = s->cVal;
    }
    // native sparse kernel (admm_csr_kernel.h) where it applies: shapes beyond the dense register-tiled kernels, rows
    // sorted and duplicate-free, CSR + CSC index + vectors within one CU's LDS
    if (!(s->flags & (SQPH_FLAG_CSR_EXPAND | SQPH_FLAG_FORCE_GENERIC)) && !s->settings.verbose && !(m <= 128 && n <= 64) && m <= 512 && n <= 224 &&
        c->nnz_max >= 1 && c->nnz_max <= 65535) {
        int TT = 0;
        size_t lds_bytes = 0;
#define SQPH_CSR_PICK(TT_)                                                              \
    if (!TT && (int)n <= 32 * TT_) {                                                    \
        TT = TT_;                                                                       \
        lds_bytes = sqph::CsrLayout<TT_>::make((int)m, (int)c->nnz_max).bytes;          \
    }
        SQPH_CSR_SHAPES(SQPH_CSR_PICK)
#undef SQPH_CSR_PICK
        // the block-row kernel where a block-row count is compiled in and its LDS map fits (SQPH_CSR_FORM=tile in the environment
        // keeps the 32 x 32 lane-grid kernel: A/B measurements)
        int NB = 0;
        {
            static const bool tile_form = getenv("SQPH_CSR_FORM") && !strcmp(getenv("SQPH_CSR_FORM"), "tile");
#define SQPH_CSB_PICK(NB_)                                                                                          \
    if (!NB && !tile_form && (int)n <= 16 * NB_ && sqph::CsbLayout<NB_>::make((int)m, (int)c->nnz_max).bytes <= 160 * 1024) NB = NB_;
            SQPH_CSB_SHAPES(SQPH_CSB_PICK)
#undef SQPH_CSB_PICK
        }
        if (NB || (TT && lds_bytes <= 160 * 1024)) {
            if (!s->cBad) SQPH_HIP(s, ws_malloc((void **)&s->cBad, sizeof(int)));
            SQPH_HIP(s, hipMemsetAsync(s->cBad, 0, sizeof(int), s->stream));
            const size_t npat = s_row ? B : 1;
            hipLaunchKernelGGL(csr_check, dim3((unsigned)((npat * m + 255) / 256)), dim3(256), 0, s->stream, (int)npat, (int)n, (int)m,
                               (long long)c->nnz_max, rowptr, colind, s_row, s_col, s->cBad);
            SQPH_HIP(s, hipGetLastError());
            int bad = 0;
            SQPH_HIP(s, hipMemcpyAsync(&bad, s->cBad, sizeof(int), hipMemcpyDeviceToHost, s->stream));
            SQPH_HIP(s, hipStreamSynchronize(s->stream));
            if (bad & 3) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: malformed CSR (%s)", what, (bad & 1) ? "row pointers not monotone" : "column index out of range");
            if (!(bad & 4)) {
                CsrDesc cd{rowptr, colind, val, s_row, s_col, s_val ? s_val : 0, (int)c->nnz_max, TT, NB};
                if (sp_pending && NB) {  // the block-row kernel reads the compressed columns in place: structure check only
                    sp_pending = false;
                    const int rc = place_sparse_P(s, c->batch, spd, false, &d, what);
                    if (rc != SQPH_OK) return rc;
                    cd.p_colptr = spd.colptr; cd.p_rowind = spd.rowind; cd.p_val = spd.val ? spd.val : (const void *)s->cBad;
                    cd.s_pcolptr = spd.s_ptr; cd.s_prowind = spd.s_ind; cd.s_pval = spd.s_val;
                }
                const int rc = dense_P();  // (the 32 x 32 lane-grid kernel)
                if (rc != SQPH_OK) return rc;
                return run(s, &d, mode, what, &cd);
            }
        }
    }
    {
        const int rc = dense_P();  // expand + dense route
        if (rc != SQPH_OK) return rc;
    }
    const bool shared = s_val == 0;
    const size_t nexp = shared ? 1 : B;
    if (!s->cA) SQPH_HIP(s, ws_malloc(&s->cA, (size_t)s->cap * m * n * e));
    if (!s->cBad) SQPH_HIP(s, ws_malloc((void **)&s->cBad, sizeof(int)));
    SQPH_HIP(s, hipMemsetAsync(s->cA, 0, nexp * m * n * e, s->stream));
    SQPH_HIP(s, hipMemsetAsync(s->cBad, 0, sizeof(int), s->stream));
    const unsigned blocks = (unsigned)((nexp * m + 255) / 256);
    if (s->dtype == SQPH_F32)
        hipLaunchKernelGGL((csr_expand<float>), dim3(blocks), dim3(256), 0, s->stream, (int)nexp, (int)n, (int)m, rowptr, colind,
                           (const float *)val, s_row, s_col, s_val, (long long)c->nnz_max, (float *)s->cA, s->cBad);
    else
        hipLaunchKernelGGL((csr_expand<double>), dim3(blocks), dim3(256), 0, s->stream, (int)nexp, (int)n, (int)m, rowptr, colind,
                           (const double *)val, s_row, s_col, s_val, (long long)c->nnz_max, (double *)s->cA, s->cBad);
    SQPH_HIP(s, hipGetLastError());
    int bad = 0;
    SQPH_HIP(s, hipMemcpyAsync(&bad, s->cBad, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    SQPH_HIP(s, hipStreamSynchronize(s->stream));
    if (bad) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: malformed CSR (%s)", what, (bad & 1) ? "row pointers not monotone" : "column index out of range");

    // dense remainder of the problem: stage host arrays through the regular path, A is already on the device
    if (d.memspace == SQPH_HOST) {
        sqph_qp_batch h = d;  // P, q, l, u from the host; a dummy A pointer is staged below
        // stage P,q,l,u by hand (the dense entry would also copy A)
        struct Item { const void *src; void **dst; size_t elems; long long *stride; };
        long long sP = d.stride_P, sq = d.stride_q, sl = d.stride_l, su = d.stride_u;
        Item items[4] = {{d.P, &s->sP, n * n, &sP}, {d.q, &s->sq, n, &sq}, {d.l, &s->sl, m, &sl}, {d.u, &s->su, m, &su}};
        if (!d.P || !d.q || !d.l || !d.u) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: null problem pointer", what);
        if (sP < 0 || sq < 0 || sl < 0 || su < 0) SQPH_FAIL(s, SQPH_ERR_INVALID, "%s: negative stride", what);
        for (auto &it : items) {
            if (!*it.dst) SQPH_HIP(s, ws_malloc(it.dst, (size_t)s->cap * it.elems * e));
            if (*it.stride == 0) {
                SQPH_HIP(s, hipMemcpyAsync(*it.dst, it.src, it.elems * e, hipMemcpyHostToDevice, s->stream));
            } else if ((size_t)*it.stride == it.elems) {
                SQPH_HIP(s, hipMemcpyAsync(*it.dst, it.src, B * it.elems * e, hipMemcpyHostToDevice, s->stream));
            } else {
                SQPH_HIP(s, hipMemcpy2DAsync(*it.dst, it.elems * e, it.src, (size_t)*it.stride * e, it.elems * e, B, hipMemcpyHostToDevice, s->stream));
                *it.stride = (long long)it.elems;
            }
        }
        SQPH_HIP(s, hipStreamSynchronize(s->stream));
        h.memspace = SQPH_DEVICE;
        h.P = s->sP; h.q = s->sq; h.l = s->sl; h.u = s->su;
        h.stride_P = sP; h.stride_q = sq; h.stride_l = sl; h.stride_u = su;
        d = h;
    }
    d.A = s->cA;
    d.stride_A = shared ? 0 : (long long)(m * n);
    return run(s, &d, mode, what);
}

}  // namespace


// ---- RCCL leg of the single-process multi-GPU gather (north_star: "final RCCL gather over xGMI") -------------------------------
// librccl is opened on first use (dlopen: a process that never gathers across devices does not load it) and ONE communicator per
// visible device is created with ncclCommInitAll.  A cross-device sqph_gather_post is then a grouped ncclSend (producer's
// communicator, producer's stream, right behind its solve) / ncclRecv (root's communicator, the gather's own stream) per array —
// point to point over xGMI, the same direct-to-root pattern as the torch.distributed path (sqp_solver_amd/dist.py).  Where RCCL
// cannot be had (library missing, initialisation refused) the post falls back to hipMemcpyPeerAsync; sqph_gather_transport()
// says which one ran.
namespace {
struct RcclApi {
    std::mutex mu;  // communicators are not thread-safe: posts from the shards' host threads enqueue one at a time
    bool tried = false, ok = false;
    void *lib = nullptr;
    int ndev = 0;
    std::vector<ncclComm_t> comms;
    std::string why;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // call with mu held
    bool init() {
        if (tried) return ok;
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { why = "librccl not found"; return false; }
#define SQPH_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, sym)); if (!field) { why = std::string("librccl lacks ") + sym; return false; }
        SQPH_SYM(CommInitAll, "ncclCommInitAll") SQPH_SYM(CommDestroy, "ncclCommDestroy") SQPH_SYM(GroupStart, "ncclGroupStart")
        SQPH_SYM(GroupEnd, "ncclGroupEnd") SQPH_SYM(Send, "ncclSend") SQPH_SYM(Recv, "ncclRecv") SQPH_SYM(GetErrorString, "ncclGetErrorString")
#undef SQPH_SYM
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { why = "no HIP device"; return false; }
        int cur = 0;
        (void)hipGetDevice(&cur);
        std::vector<int> devs(ndev);
        for (int d = 0; d < ndev; d++) devs[d] = d;
        comms.assign(ndev, nullptr);
        const ncclResult_t r = CommInitAll(comms.data(), ndev, devs.data());
        (void)hipSetDevice(cur);
        if (r != ncclSuccess) { why = std::string("ncclCommInitAll: ") + GetErrorString(r); comms.clear(); return false; }
        ok = true;
        return true;
    }
};
RcclApi &rccl() {
    static RcclApi *r = new RcclApi();  // leaked on purpose: communicators must not be torn down from a static destructor
    return *r;
}
}  // namespace

struct sqph_gather {
    int device = 0, n = 0, m = 0, flags = 0;
    long long total = 0;
    double *x = nullptr, *y = nullptr;
    sqph_info *info = nullptr;
    hipStream_t stream = nullptr;      // the root's side of the RCCL receives: ONE stream.  All ncclRecv of a post are issued inside one
                                       // ncclGroupStart / End on the root's communicator: RCCL fuses a group's operations per
                                       // communicator into ONE kernel (its channels drive the seven xGMI links concurrently) launched
                                       // on the first stream it saw and makes any other stream wait for it — per-peer receive
                                       // streams (round 5) bought nothing and cost seven event records per post
    std::mutex mu;                     // shards post from their own host threads (MultiGpuBatchQPSolver::run_host)
    std::vector<hipEvent_t> pending;   // one per posted copy, recorded on the stream it was enqueued on  (guarded by mu)
    std::vector<int> pending_dev;      //                                                                  (guarded by mu)
    const char *transport = "none";    // of the last post
    std::string err;
};

extern "C" {
int sqph_gather_create_ex(sqph_gather **out, int device, int n, int m, long long total, int flags) {
    if (!out) return SQPH_ERR_INVALID;
    *out = nullptr;
    if (n <= 0 || m < 0 || total <= 0) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "sqph_gather_create: bad shape");
    if ((flags & SQPH_GATHER_RCCL_ALWAYS) && (flags & SQPH_GATHER_NO_RCCL)) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "sqph_gather_create: contradictory flags");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_NO_DEVICE, "sqph_gather_create: no HIP device visible");
    if (device < 0 || device >= ndev) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "sqph_gather_create: device %d out of range", device);
    sqph_gather *g = new (std::nothrow) sqph_gather();
    if (!g) SQPH_FAIL((sqph_solver *)nullptr, SQPH_ERR_INVALID, "sqph_gather_create: out of host memory");
    g->device = device; g->n = n; g->m = m; g->total = total; g->flags = flags;
    DeviceGuard dg(device);
    const size_t mm = (size_t)(m > 0 ? m : 1);
    hipError_t e = ws_malloc((void **)&g->x, (size_t)total * n * sizeof(double));
    if (e == hipSuccess) e = ws_malloc((void **)&g->y, (size_t)total * mm * sizeof(double));
    if (e == hipSuccess) e = ws_malloc((void **)&g->info, (size_t)total * sizeof(sqph_info));
    if (e == hipSuccess) {
        e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    }
    if (e != hipSuccess) {
        g_err = std::string("sqph_gather_create: ") + hipGetErrorString(e);
        sqph_gather_destroy(g);
        return SQPH_ERR_HIP;
    }
    *out = g;
    return SQPH_OK;
}
int sqph_gather_create(sqph_gather **out, int device, int n, int m, long long total) { return sqph_gather_create_ex(out, device, n, m, total, 0); }

void sqph_gather_destroy(sqph_gather *g) {
    if (!g) return;
    for (size_t i = 0; i < g->pending.size(); i++) {
        DeviceGuard dg(g->pending_dev[i]);
        (void)hipEventSynchronize(g->pending[i]);
        (void)hipEventDestroy(g->pending[i]);
    }
    DeviceGuard dg(g->device);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    if (g->x) (void)hipFree(g->x);
    if (g->y) (void)hipFree(g->y);
    if (g->info) (void)hipFree(g->info);
    delete g;
}

const char *sqph_gather_transport(const sqph_gather *g) { return g ? g->transport : "none"; }

// One RCCL group for ALL shards of a call (producer sides on the producers' streams behind their solves, root sides on the root's
// ONE receive stream), then one event per producer stream and one on the root's.  Shards on the root's own device — and everything
// where RCCL cannot be had — are device-to-device / peer copies on the producer's stream.
// Ordering: posts are serialised by the process-wide RcclApi::mu (a communicator must not be entered by two host threads at once), so
// the groups of concurrent posts reach the root's receive stream one after the other, each complete; on a stream RCCL executes groups in
// issue order, and every sender's stream orders its send behind its own solve.  Nothing else is needed for correctness: the gather
// buffers of two posts never overlap (disjoint [offset, offset + count) ranges, checked above per shard).
int sqph_gather_post_many(sqph_gather *g, sqph_solver *const *srcs, const long long *offsets, const int *counts, int k) {
    if (!g || !srcs || !offsets || !counts || k < 0) return SQPH_ERR_INVALID;
    for (int i = 0; i < k; i++) {
        sqph_solver *src = srcs[i];
        if (!src) return SQPH_ERR_INVALID;
        if (src->n != g->n || src->m != g->m) SQPH_FAIL(src, SQPH_ERR_INVALID, "sqph_gather_post: shape mismatch");
        if (counts[i] < 0 || counts[i] > src->cap || offsets[i] < 0 || offsets[i] + counts[i] > g->total)
            SQPH_FAIL(src, SQPH_ERR_INVALID, "sqph_gather_post: range [%lld, %lld) outside the gather buffers / solver capacity", offsets[i], offsets[i] + counts[i]);
    }
    const size_t n = g->n, m = g->m;
    std::vector<char> via((size_t)k, 0);
    const char *rccl_err = nullptr;
    sqph_solver *err_src = k ? srcs[0] : nullptr;
    bool want_rccl = false;
    for (int i = 0; i < k; i++)
        if (counts[i] > 0 && !(g->flags & SQPH_GATHER_NO_RCCL) && (srcs[i]->device != g->device || (g->flags & SQPH_GATHER_RCCL_ALWAYS))) want_rccl = true;
    if (want_rccl) {
        RcclApi &R = rccl();
        // the enqueue is serialised per process: every post uses the ROOT's communicator (its receive side), and a communicator
        // must not be entered from two host threads at once; the lock covers the enqueue only (microseconds), never a transfer
        std::lock_guard<std::mutex> lk(R.mu);
        if (R.init()) {
            ncclResult_t r = R.GroupStart();
            for (int i = 0; i < k; i++) {
                sqph_solver *src = srcs[i];
                const size_t c = (size_t)counts[i];
                const bool eligible = c > 0 && (src->device != g->device || (g->flags & SQPH_GATHER_RCCL_ALWAYS)) && src->device < R.ndev && g->device < R.ndev;
                if (!eligible) continue;
                hipStream_t rs = g->stream;
                const auto xfer = [&](const void *from, void *to, size_t elems, ncclDataType_t dt) {
                    if (r == ncclSuccess) r = R.Send(from, elems, dt, g->device, R.comms[src->device], src->stream);
                    if (r == ncclSuccess) r = R.Recv(to, elems, dt, src->device, R.comms[g->device], rs);
                };
                xfer(src->x, g->x + (size_t)offsets[i] * n, c * n, ncclFloat64);
                if (m) xfer(src->y, g->y + (size_t)offsets[i] * m, c * m, ncclFloat64);
                xfer(src->info, g->info + offsets[i], c * sizeof(sqph_info), ncclInt8);
                via[(size_t)i] = 1;
                if (r != ncclSuccess) err_src = src;
            }
            const ncclResult_t re = R.GroupEnd();
            if (r == ncclSuccess) r = re;
            // an error after a partly issued group: the send / receive kernels that were enqueued still have to be waited for —
            // the events below are recorded either way, the error is returned after them
            if (r != ncclSuccess) rccl_err = R.GetErrorString(r);
        }
    }
    // (any failure from here on still hands the events already created to the gather: nothing leaks, nothing enqueued goes unwaited)
    std::vector<hipEvent_t> evs;
    std::vector<int> evdev;
    bool any_rccl = false, any_copy = false;
    const auto hand_over = [&]() {
        std::lock_guard<std::mutex> lk(g->mu);
        if (any_rccl || any_copy) g->transport = any_rccl ? "rccl" : "peer-copy";  // (one host thread per shard may post: under the gather's mutex)
        for (size_t i = 0; i < evs.size(); i++) {
            g->pending.push_back(evs[i]);
            g->pending_dev.push_back(evdev[i]);
        }
        evs.clear();
        evdev.clear();
    };
#define SQPH_HIP_POST(src_, call)                                                                             \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            hand_over();                                                                                      \
            SQPH_FAIL(src_, SQPH_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));                     \
        }                                                                                                     \
    } while (0)
    for (int i = 0; i < k; i++) {
        sqph_solver *src = srcs[i];
        const size_t c = (size_t)counts[i];
        if (!c) continue;
        const bool cross = src->device != g->device;
        if (via[(size_t)i] && !any_rccl) {  // ONE event on the root's receive stream covers the whole group
            any_rccl = true;
            DeviceGuard dr(g->device);
            hipEvent_t ev_root = nullptr;
            SQPH_HIP_POST(src, hipEventCreateWithFlags(&ev_root, hipEventDisableTiming));
            evs.push_back(ev_root);
            evdev.push_back(g->device);
            SQPH_HIP_POST(src, hipEventRecord(ev_root, g->stream));
        }
        DeviceGuard dg(src->device);
        if (!via[(size_t)i]) {
            any_copy = true;
            if (cross) {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, src->device, g->device);
                if (can) (void)hipDeviceEnablePeerAccess(g->device, 0);  // idempotent (an "already enabled" error is cleared below)
                (void)hipGetLastError();
            }
            SQPH_HIP_POST(src, hipMemcpyPeerAsync(g->x + (size_t)offsets[i] * n, g->device, src->x, src->device, c * n * sizeof(double), src->stream));
            if (m) SQPH_HIP_POST(src, hipMemcpyPeerAsync(g->y + (size_t)offsets[i] * m, g->device, src->y, src->device, c * m * sizeof(double), src->stream));
            SQPH_HIP_POST(src, hipMemcpyPeerAsync(g->info + offsets[i], g->device, src->info, src->device, c * sizeof(sqph_info), src->stream));
        }
        hipEvent_t ev = nullptr;
        SQPH_HIP_POST(src, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        evs.push_back(ev);
        evdev.push_back(src->device);
        SQPH_HIP_POST(src, hipEventRecord(ev, src->stream));
    }
#undef SQPH_HIP_POST
    hand_over();
    if (rccl_err) SQPH_FAIL(err_src, SQPH_ERR_HIP, "sqph_gather_post: RCCL: %s", rccl_err);
    return SQPH_OK;
}

int sqph_gather_post(sqph_gather *g, sqph_solver *src, long long offset, int count) {
    if (!g || !src) return SQPH_ERR_INVALID;
    return sqph_gather_post_many(g, &src, &offset, &count, 1);
}

static int gather_wait(sqph_gather *g) {
    // take the posted events out under the lock, wait for them outside it (a concurrent post lands in the next wait)
    std::vector<hipEvent_t> evs;
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        evs.swap(g->pending);
        devs.swap(g->pending_dev);
    }
    int rc = SQPH_OK;
    for (size_t i = 0; i < evs.size(); i++) {
        DeviceGuard dg(devs[i]);
        const hipError_t e = rc == SQPH_OK ? hipEventSynchronize(evs[i]) : hipSuccess;
        (void)hipEventDestroy(evs[i]);  // every event is released, also behind a failed one
        if (e != hipSuccess) {
            g_err = std::string("sqph_gather: ") + hipGetErrorString(e);
            rc = SQPH_ERR_HIP;
        }
    }
    return rc;
}

int sqph_gather_device_ptrs(sqph_gather *g, void **x, void **y, sqph_info **info) {
    if (!g) return SQPH_ERR_INVALID;
    const int rc = gather_wait(g);
    if (rc != SQPH_OK) return rc;
    if (x) *x = g->x;
    if (y) *y = g->y;
    if (info) *info = g->info;
    return SQPH_OK;
}

int sqph_gather_fetch(sqph_gather *g, int dtype, void *x, void *y, sqph_info *info) {
    if (!g) return SQPH_ERR_INVALID;
    if (dtype != SQPH_F64 && dtype != SQPH_F32) return SQPH_ERR_INVALID;
    const int rc = gather_wait(g);
    if (rc != SQPH_OK) return rc;
    DeviceGuard dg(g->device);
    const size_t T = (size_t)g->total;
    struct Item { void *dst; const double *src; size_t elems; };
    const Item items[2] = {{x, g->x, T * g->n}, {y, g->y, T * g->m}};
    for (const Item &it : items) {
        if (!it.dst || !it.elems) continue;
        if (dtype == SQPH_F64) {
            SQPH_HIP((sqph_solver *)nullptr, hipMemcpy(it.dst, it.src, it.elems * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            std::vector<double> tmp(it.elems);
            SQPH_HIP((sqph_solver *)nullptr, hipMemcpy(tmp.data(), it.src, it.elems * sizeof(double), hipMemcpyDeviceToHost));
            float *d = (float *)it.dst;
            for (size_t i = 0; i < it.elems; i++) d[i] = (float)tmp[i];
        }
    }
    if (info) SQPH_HIP((sqph_solver *)nullptr, hipMemcpy(info, g->info, T * sizeof(sqph_info), hipMemcpyDeviceToHost));
    return SQPH_OK;
}
}  // extern "C"

extern "C" {
int sqph_setup_solve_reuse_csr(sqph_solver *s, const sqph_csr_batch *qp) {
    return run_csr(s, qp, sqph::MODE_SETUP | sqph::MODE_SOLVE | sqph::MODE_SAME_MATRICES, "sqph_setup_solve_reuse_csr");
}
int sqph_setup_solve_reuse_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P) {
    if (!s) return SQPH_ERR_INVALID;
    if (!P) SQPH_FAIL(s, SQPH_ERR_INVALID, "sqph_setup_solve_reuse_csr_sp: P is null");
    return run_csr(s, qp, sqph::MODE_SETUP | sqph::MODE_SOLVE | sqph::MODE_SAME_MATRICES, "sqph_setup_solve_reuse_csr_sp", P);
}
int sqph_setup_csr(sqph_solver *s, const sqph_csr_batch *qp) { return run_csr(s, qp, sqph::MODE_SETUP, "sqph_setup_csr"); }
int sqph_update_qp_csr(sqph_solver *s, const sqph_csr_batch *qp) { return run_csr(s, qp, sqph::MODE_UPDATE, "sqph_update_qp_csr"); }
int sqph_solve_csr(sqph_solver *s, const sqph_csr_batch *qp) { return run_csr(s, qp, sqph::MODE_SOLVE, "sqph_solve_csr"); }
int sqph_setup_solve_csr(sqph_solver *s, const sqph_csr_batch *qp) {
    return run_csr(s, qp, sqph::MODE_SETUP | sqph::MODE_SOLVE, "sqph_setup_solve_csr");
}
int sqph_update_solve_csr(sqph_solver *s, const sqph_csr_batch *qp) {
    return run_csr(s, qp, sqph::MODE_UPDATE | sqph::MODE_SOLVE, "sqph_update_solve_csr");
}
#define SQPH_NEED_SP(what) \
    if (s && !P) SQPH_FAIL(s, SQPH_ERR_INVALID, what ": P is null")
int sqph_setup_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P) {
    SQPH_NEED_SP("sqph_setup_csr_sp");
    return run_csr(s, qp, sqph::MODE_SETUP, "sqph_setup_csr_sp", P);
}
int sqph_update_qp_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P) {
    SQPH_NEED_SP("sqph_update_qp_csr_sp");
    return run_csr(s, qp, sqph::MODE_UPDATE, "sqph_update_qp_csr_sp", P);
}
int sqph_solve_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P) {
    SQPH_NEED_SP("sqph_solve_csr_sp");
    return run_csr(s, qp, sqph::MODE_SOLVE, "sqph_solve_csr_sp", P);
}
int sqph_setup_solve_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P) {
    SQPH_NEED_SP("sqph_setup_solve_csr_sp");
    return run_csr(s, qp, sqph::MODE_SETUP | sqph::MODE_SOLVE, "sqph_setup_solve_csr_sp", P);
}
int sqph_update_solve_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P) {
    SQPH_NEED_SP("sqph_update_solve_csr_sp");
    return run_csr(s, qp, sqph::MODE_UPDATE | sqph::MODE_SOLVE, "sqph_update_solve_csr_sp", P);
}
#undef SQPH_NEED_SP
int sqph_setup(sqph_solver *s, const sqph_qp_batch *qp) { return run(s, qp, sqph::MODE_SETUP, "sqph_setup"); }
int sqph_update_qp(sqph_solver *s, const sqph_qp_batch *qp) { return run(s, qp, sqph::MODE_UPDATE, "sqph_update_qp"); }
int sqph_solve(sqph_solver *s, const sqph_qp_batch *qp) { return run(s, qp, sqph::MODE_SOLVE, "sqph_solve"); }
int sqph_setup_solve(sqph_solver *s, const sqph_qp_batch *qp) {
    return run(s, qp, sqph::MODE_SETUP | sqph::MODE_SOLVE, "sqph_setup_solve");
}
int sqph_update_solve(sqph_solver *s, const sqph_qp_batch *qp) {
    return run(s, qp, sqph::MODE_UPDATE | sqph::MODE_SOLVE, "sqph_update_solve");
}
int sqph_setup_solve_reuse(sqph_solver *s, const sqph_qp_batch *qp) {
    return run(s, qp, sqph::MODE_SETUP | sqph::MODE_SOLVE | sqph::MODE_SAME_MATRICES, "sqph_setup_solve_reuse");
}
}
