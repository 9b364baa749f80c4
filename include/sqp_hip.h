/*
 * sqp_hip.h — C-ABI of libsqp_hip.so: MI355X-native batched ADMM QP-subproblem solver.
 *
 * This is the drop-in boundary for the reference's QP hot path.  The reference has no
 * FFI layer of its own (it is a C++11/Eigen class library), so each entry point below
 * replaces a *method of qp_solver::QPSolver<Scalar>* applied to a whole batch of
 * same-(n,m) problems; the C++ facade in include/sqp_hip/qp.hpp re-creates the class on
 * top of these calls (see INTEGRATION.md for the binding a maintainer would add).
 *
 *   reference interface (file:line under /root/reference)      C-ABI entry point
 *   ---------------------------------------------------------  ---------------------------
 *   QPSolverSettings<Scalar>      include/solvers/qp.hpp:36-54     sqph_settings
 *   QPSolverInfo<Scalar>          include/solvers/qp.hpp:72-80     sqph_info
 *   QPSolverStatus                include/solvers/qp.hpp:70        SQPH_SOLVED ... (same order)
 *   QuadraticProblem<Scalar>      include/solvers/qp.hpp:19-34     sqph_qp_batch (borrowed ptrs)
 *   QPSolver::setup(qp)           src/qp.cpp:11-44                 sqph_setup
 *   QPSolver::update_qp(qp)       src/qp.cpp:46-62                 sqph_update_qp
 *   QPSolver::solve(qp)           src/qp.cpp:64-157                sqph_solve
 *   setup(qp); solve(qp)          src/sqp.cpp:221-222 (run_solve_qp) sqph_setup_solve (one launch)
 *   update_qp(qp); solve(qp)      src/qp.cpp:46-62, 64-157         sqph_update_solve (one launch, iterates kept)
 *   primal_solution()/dual_solution()/info()  qp.hpp:160-170       sqph_get_solution / sqph_device_state
 *   settings()                    qp.hpp:166-167                   sqph_set_settings / sqph_get_settings
 *   static constr_type_init(l,u,type)  src/qp.cpp:283-294          sqph_constr_type_init (host utility)
 *   legacy sparse QP<n,m> (Eigen::SparseMatrix A), setup/update_qp/solve
 *                                 include/unsupported/qp_solver.hpp:17-32,215-330   sqph_*_csr (sqph_csr_batch: dense P, CSR A), sqph_*_csr_sp (+ sqph_csc_P: P sparse too)
 *   non-const x / y / z accessors (warm starts)  qp.hpp:160-164    sqph_set_state
 *   the SOC re-solve: same P, A, new bounds      src/sqp.cpp:244-276 (TODO :273)   sqph_setup_solve_reuse (+ _csr, _csr_sp)
 *   settings.verbose / print_status              src/qp.cpp:72-76,113-117,373-383  sqph_set_trace_qp / sqph_get_trace
 *   one solver object per problem, spread over threads (and devices) by the caller
 *                                 qp.hpp:217-247                   sqph_device_count / sqph_shard_bounds / sqph_own_stream /
 *                                                                  sqph_gather_create, _post, _fetch, _device_ptrs, _destroy
 *
 * The reference's callers are served source-compatibly by include/sqp_hip/compat/ (solvers/qp.hpp, unsupported/qp_solver.hpp,
 * solvers/sqp.hpp, solvers/bfgs.hpp): its own test files compile unchanged against that tree.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.  dtype selects Scalar.
 *   - every QP of a batch has the same (n, m).  Matrices are per-QP COLUMN-MAJOR (Eigen's
 *     default, qp.hpp:27), QP-major across the batch: problem b starts at base + b*stride
 *     (strides in elements; stride 0 = the same array is shared by every QP of the batch).
 *   - problem data is borrowed for the duration of the call only (host memspace: copied to
 *     the device inside the call; device memspace: must stay valid until the stream reaches
 *     the end of the enqueued work).
 *   - all calls are asynchronous on the solver's stream except where noted.
 *   - return value: 0 = ok, <0 = API misuse or HIP error (message via sqph_last_error).
 *     Per-QP numerical status is in sqph_info::status, exactly as in the reference.
 *   - q, l, u given to sqph_solve may differ from those of the preceding sqph_setup /
 *     sqph_update_qp (the reference re-reads them from solve()'s argument, src/qp.cpp:89,
 *     100,112).  P may differ as well while the factor is the resident one (after
 *     sqph_setup / sqph_update_qp, or on a SQPH_FLAG_KEEP_FACTOR handle) and rho is not
 *     adapted: as in the reference it then enters the residuals only (src/qp.cpp:324,360;
 *     the factor is setup()'s).  A must be the matrix of the preceding set-up: the iteration
 *     runs on B = A W', rebuilt from the call's A (the reference reads solve()'s A for the
 *     residuals alone).  The batch entry points cannot check this (device pointers); the
 *     drop-in class qp_solver::QPSolver<Scalar>::solve (include/sqp_hip/qp.hpp) compares its
 *     argument with the A of the preceding setup()/update_qp() and throws
 *     std::invalid_argument on a difference instead of solving another problem silently.
 *   - reproducibility: repeated calls on the same inputs are bit-identical (no atomics whose order is left to the hardware).
 *     The kernel is chosen by shape, settings AND, for m <= 4 with the one-QP-per-lane kernel, by the batch size of the call (a
 *     four-lanes-per-QP form serves batches <= 2,048): the same QP may then differ in its last bits between a small and a large
 *     batch (another summation order of A'w; both within the parity bar).  On the sparse route the block-row kernel places the
 *     register-resident entries of A by LDS bank where settings.max_iter >= 150 (SQPH_CSB_PLACE_MIN_ITERS) and keeps them in storage
 *     order below: the per-lane summation order of A x and A'w — the last bits of a solve — therefore also depends on max_iter.
 *   - thread safety: distinct handles may be used from distinct host threads concurrently
 *     (tests/cpp/qp_facade_test.cpp: testTwoHostThreadsTwoHandles); one handle must not be.
 *     sqph_global_error() is per thread, sqph_last_error(s) per handle.
 */
#ifndef SQP_HIP_H
#define SQP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SQPH_VERSION 1

/* Scalar type of a solver instance (QPSolver<double> / QPSolver<float>, src/qp.cpp:385-386) */
enum { SQPH_F64 = 0, SQPH_F32 = 1 };
/* where caller buffers live */
enum { SQPH_HOST = 0, SQPH_DEVICE = 1 };
/* QPSolverStatus, qp.hpp:70 — same numeric order */
enum { SQPH_SOLVED = 0, SQPH_MAX_ITER_EXCEEDED = 1, SQPH_UNSOLVED = 2, SQPH_NUMERICAL_ISSUES = 3, SQPH_UNINITIALIZED = 4 };
/* ConstraintType, qp.hpp:134 — same numeric order */
enum { SQPH_INEQUALITY_CONSTRAINT = 0, SQPH_EQUALITY_CONSTRAINT = 1, SQPH_LOOSE_BOUNDS = 2 };
/* error codes */
enum {
    SQPH_OK = 0,
    SQPH_ERR_INVALID = -1,     /* bad argument / misuse */
    SQPH_ERR_HIP = -2,         /* HIP runtime error */
    SQPH_ERR_UNSUPPORTED = -3, /* shape outside what the kernels cover */
    SQPH_ERR_NO_DEVICE = -4    /* no HIP device: there is NO CPU fallback */
};

/* QPSolverSettings<Scalar>, qp.hpp:36-54; values are converted to Scalar on use */
typedef struct sqph_settings {
    double rho;                    /* 1e-1 */
    double sigma;                  /* 1e-6 */
    double alpha;                  /* 1.0  */
    double eps_rel;                /* 1e-3 */
    double eps_abs;                /* 1e-3 */
    int max_iter;                  /* 1000 */
    int check_termination;         /* 25, 0 = never */
    int warm_start;                /* 0 */
    int adaptive_rho;              /* 0 */
    double adaptive_rho_tolerance; /* 5 */
    int adaptive_rho_interval;     /* 25 */
    int verbose;                   /* 0; non-zero: record the per-check status line of one QP (sqph_get_trace) */
} sqph_settings;

/* QPSolverInfo<Scalar>, qp.hpp:72-80 (40 bytes; Scalar fields widened to double) */
typedef struct sqph_info {
    int status;
    int iter;
    int rho_updates;
    int _pad;
    double rho_estimate;
    double res_prim;
    double res_dual;
} sqph_info;

/* A batch of QuadraticProblem<Scalar> (qp.hpp:19-34): non-owning pointers. */
typedef struct sqph_qp_batch {
    int batch;    /* number of QPs, <= capacity given at creation */
    int memspace; /* SQPH_HOST or SQPH_DEVICE */
    const void *P; /* n x n, col-major; only its lower triangle enters the factor (LDLT<Lower>) */
    const void *q; /* n */
    const void *A; /* m x n, col-major */
    const void *l; /* m */
    const void *u; /* m */
    long long stride_P, stride_q, stride_A, stride_l, stride_u; /* elements; 0 = shared */
} sqph_qp_batch;

/* The same batch with the constraint matrix in CSR (BASELINE config 5; the reference's sparse variant keeps P and A as
 * Eigen::SparseMatrix, include/unsupported/qp_solver.hpp:17-32,363-394 — same ADMM, sparse KKT storage).  P stays dense.
 * Row i of QP b holds entries A_colind/A_val[rowptr[i] .. rowptr[i+1]) (0-based; duplicates within a row are summed).
 * stride_rowptr = 0 and stride_colind = 0 share one sparsity pattern across the batch (stride_val then = nnz). */
typedef struct sqph_csr_batch {
    int batch;
    int memspace;
    const void *P;        /* n x n dense, col-major */
    const void *q;        /* n */
    const int *A_rowptr;  /* m + 1 */
    const int *A_colind;  /* nnz (<= stride_colind when per-QP) */
    const void *A_val;    /* nnz */
    const void *l;        /* m */
    const void *u;        /* m */
    long long stride_P, stride_q, stride_rowptr, stride_colind, stride_val, stride_l, stride_u; /* elements; 0 = shared */
    long long nnz_max;    /* capacity of one QP's colind/val arrays (= nnz for a shared pattern) */
} sqph_csr_batch;

/* P of a sqph_csr_batch in compressed-column form — the storage of the Eigen::SparseMatrix<Scalar> the legacy sparse class keeps P
 * in (include/unsupported/qp_solver.hpp:24-25).  The FULL symmetric matrix is given (the reference multiplies by it,
 * unsupported/qp_solver.hpp:532,568, and inserts all of it into the KKT matrix, :376), n + 1 column pointers, rows strictly
 * increasing inside a column.  Same memspace as the batch it goes with; strides in elements, 0 = shared by every QP (a shared
 * pattern with per-QP values is stride_colptr = stride_rowind = 0, stride_val >= nnz_max). */
typedef struct sqph_csc_P {
    const int *colptr;   /* n + 1 */
    const int *rowind;   /* nnz (<= nnz_max) */
    const void *val;     /* nnz */
    long long stride_colptr, stride_rowind, stride_val;
    long long nnz_max;   /* capacity of one QP's rowind / val arrays */
} sqph_csc_P;

typedef struct sqph_solver sqph_solver;

/* Behaviour flags for sqph_create */
enum {
    /* 0: supported-class semantics, src/qp.cpp:78-82 — `warm_start=false` does NOT reset x,z,y
     *    in solve() (the reference calls the static Zero() factory and drops the result);
     * 1: legacy-class semantics, include/unsupported/qp_solver.hpp:256-260 — it does reset. */
    SQPH_FLAG_LEGACY_COLD_START = 1,
    /* force the generic (global-memory) kernel even where a register-tiled one exists */
    SQPH_FLAG_FORCE_GENERIC = 2,
    /* (value 4 is retired: it selected a superseded single-wave kernel family) */
    /* CSR entry points: always expand A to dense on the device instead of using the native sparse kernel */
    SQPH_FLAG_CSR_EXPAND = 8,
    /* Keep the KKT factor of a fused sqph_setup_solve* call resident in the workspace (n*n doubles written per QP).  Without
     * it a fused call writes no factor, and a later sqph_solve* on that handle rebuilds the factor first — same arithmetic,
     * same results, one extra factorisation.  sqph_setup / sqph_update_qp always leave their factor resident.  Set this for
     * callers that follow a fused call with sqph_solve on new q, l, u (SQP second-order correction, MPC). */
    SQPH_FLAG_KEEP_FACTOR = 16,
    /* AN APPROXIMATE MODE, OUTSIDE THE PARITY CONTRACT.  QPSolver<float> with single-precision arithmetic where a kernel for it exists
     * (the reference instantiates QPSolver<float>, src/qp.cpp:385-386; the parity contract of this library for it is the DEFAULT
     * below: fp32 at the interface, fp64 arithmetic, results within the float reference's own rounding of the fp64 solution):
     *   - n <= 4, m <= 6 (one QP per lane): iterates, factor and residuals in fp32; agrees with the reference's QPSolver<float>
     *     to ~1e-3 (the Schur-complement factor loses ~3 digits more in fp32 than the reference's KKT LDL');
     *   - m <= 40, n <= 24 and m <= 112, n <= 56 (the BASELINE dense shapes; wg_f32.hip): the operator tiles, the operand vectors
     *     and the partial sums of the iteration's two stages in fp32, the factorisation, the iterates and the residual checks in
     *     fp64.  Measured against the fp64 solution: x and z as accurate as the reference's float path (0.5-1.8x its error), the
     *     DUAL y 2.7-4.6x LESS accurate (1e-4 .. 6e-4): the fp32 rounding of the operand rho z - y is an error in y of eps32 |rho z|.
     *     That misses the 2x-of-the-float-reference bar SURVEY section 8 row f4 set for a parity-grade fp32 path, for 12 % of
     *     speed at (50,100): use it where a dual accurate to three digits is enough, never for a parity claim.  The tests pin
     *     the measured behaviour (tests/test_gpu_parity.py), they do not certify it as the reference's.
     * Default (flag clear): fp32 at the interface only, fp64 arithmetic.  Ignored for dtype SQPH_F64 and for shapes without an
     * fp32 kernel (they iterate in fp64). */
    SQPH_FLAG_F32_ARITH = 32
};

void sqph_default_settings(sqph_settings *s);

/* Create a batched solver bound to HIP device `device`: `batch_capacity` instances of
 * QPSolver<Scalar> for n variables and m constraints. Every instance starts UNINITIALIZED. */
int sqph_create(sqph_solver **out, int device, int n, int m, int batch_capacity, int dtype, int flags);
void sqph_destroy(sqph_solver *s);

/* Stream the kernels/copies are enqueued on (a hipStream_t; NULL = the null stream).  With SQPH_DEVICE problem data a call only
 * enqueues work there (one kernel launch; no allocation or synchronisation after the first call of a shape), so it can be captured
 * into a hipGraph and replayed on updated inputs (tests/test_gpu_parity.py::test_hip_graph_capture_of_the_fused_call). */
int sqph_set_stream(sqph_solver *s, void *hip_stream);

int sqph_set_settings(sqph_solver *s, const sqph_settings *settings);
int sqph_get_settings(const sqph_solver *s, sqph_settings *settings);

int sqph_setup(sqph_solver *s, const sqph_qp_batch *qp);       /* QPSolver::setup for each QP     */
int sqph_update_qp(sqph_solver *s, const sqph_qp_batch *qp);   /* QPSolver::update_qp             */
int sqph_solve(sqph_solver *s, const sqph_qp_batch *qp);       /* QPSolver::solve                 */
int sqph_setup_solve(sqph_solver *s, const sqph_qp_batch *qp); /* setup()+solve(), single launch  */
/* setup()+solve() of QPs whose P and A are those of this handle's previous setup/update (only q, l, u differ): what the SQP
 * driver's second-order correction asks for (src/sqp.cpp:244-276; "only l and u change", TODO at :273).  Semantics and results
 * are exactly sqph_setup_solve's (iterates reset, constraints re-classified, rho back to settings.rho); the factorisation is
 * skipped for every QP whose rho vector comes out equal to the one the resident factor was built with.  Needs the factor
 * resident (SQPH_FLAG_KEEP_FACTOR, or a preceding sqph_setup/sqph_update_qp); otherwise identical to sqph_setup_solve. */
int sqph_setup_solve_reuse(sqph_solver *s, const sqph_qp_batch *qp);
/* update_qp()+solve() in one launch: the constraints are re-classified, rho goes back to settings.rho and the factor is rebuilt as
 * in sqph_update_qp (src/qp.cpp:46-62), the iterates x, z, y of every QP are KEPT and the solve starts from them — the route the
 * reference intends for successive subproblems (its SQP driver sets warm_start = true, src/sqp.cpp:16, and then defeats it by
 * calling setup(), src/sqp.cpp:221 / src/qp.cpp:16-18).  sqp::BatchSQP uses it when sqp_settings_t::warm_start_qp is set. */
int sqph_update_solve(sqph_solver *s, const sqph_qp_batch *qp);

/* CSR-A variants of the four calls above (legacy sparse QPSolver, unsupported/qp_solver.hpp:215-330). */
int sqph_setup_csr(sqph_solver *s, const sqph_csr_batch *qp);
int sqph_update_qp_csr(sqph_solver *s, const sqph_csr_batch *qp);
int sqph_solve_csr(sqph_solver *s, const sqph_csr_batch *qp);
int sqph_setup_solve_csr(sqph_solver *s, const sqph_csr_batch *qp);
int sqph_update_solve_csr(sqph_solver *s, const sqph_csr_batch *qp);  /* = sqph_update_solve: update_qp() + solve(), iterates kept */
/* The same five with P sparse as well (qp->P and qp->stride_P are ignored and may be NULL / 0).  Where the block-row kernel runs
 * (n <= 224, m <= 512, shapes beyond the dense register-tiled kernels: BASELINE config 5) its sparse-P instantiations read the
 * compressed columns in place — set-up (lower triangle into S) and dual residual (P x, column i being row i of the symmetric
 * matrix) — in the summation order of the dense path; on every other route one scatter pass expands P into the handle's n x n
 * workspace (one matrix when shared) and the call continues as its dense-P twin.  Either way the results are bit-identical to
 * passing that dense P, and what crosses the boundary (and PCIe, for host memspace) is 12 nnz(P) + 4 (n + 1) bytes per QP instead
 * of 8 n^2.  P must be symmetric.  A malformed structure (column pointers not monotone / beyond nnz_max, row index out of range or
 * not strictly increasing, a pattern that is not symmetric — a triangle instead of the full matrix — or mirror entries whose values differ) is SQPH_ERR_INVALID, detected on the device before anything is solved. */
int sqph_setup_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P);
int sqph_update_qp_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P);
int sqph_solve_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P);
int sqph_setup_solve_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P);
int sqph_update_solve_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P);
/* sqph_setup_solve_reuse on the sparse route: setup()+solve() of QPs whose P and A are those of this handle's previous set-up (the
 * SQP second-order correction of a sparse subproblem: only q, l, u differ, src/sqp.cpp:244-276, TODO :273).  Results are exactly
 * sqph_setup_solve_csr's; the factorisation is skipped for every QP whose freshly classified rho vector equals the one the resident
 * factor was built with (needs SQPH_FLAG_KEEP_FACTOR or a preceding sqph_setup_csr / sqph_update_qp_csr served by the same
 * kernel family; otherwise — and on routes without the fast path — identical to sqph_setup_solve_csr). */
int sqph_setup_solve_reuse_csr(sqph_solver *s, const sqph_csr_batch *qp);
int sqph_setup_solve_reuse_csr_sp(sqph_solver *s, const sqph_csr_batch *qp, const sqph_csc_P *P);

/* Copy out primal x [batch][n], dual y [batch][m], z [batch][m] and info [batch]; any pointer
 * may be NULL. With SQPH_HOST this call synchronises the stream before returning. */
int sqph_get_solution(sqph_solver *s, int batch, int memspace, void *x, void *y, void *z, sqph_info *info);

/* Overwrite the iterates (warm-start injection; the reference exposes them through the
 * non-const primal_solution()/dual_solution() accessors). NULL = leave unchanged. */
int sqph_set_state(sqph_solver *s, int batch, int memspace, const void *x, const void *z, const void *y);

/* Device-resident state arrays (QP-major, valid until sqph_destroy): no copy.  The arrays are fp64 whatever the
 * solver's interface dtype (a QPSolver<float> iterates in fp64 on the device and narrows on the way out). */
int sqph_device_state(sqph_solver *s, void **x, void **y, void **z, sqph_info **info);

int sqph_synchronize(sqph_solver *s);

/* settings.verbose (reference QP_SOLVER_PRINTING, src/qp.cpp:72-76, 113-117, 152-156, 373-383): when non-zero, solve calls
 * record for ONE QP of the batch (sqph_set_trace_qp, default 0) the line print_status prints at every termination check —
 * iteration, objective 0.5 x'Px + q'x, primal and dual residual.  The library prints nothing itself; sqph_get_trace returns the
 * records of the last call (4 doubles each: iter, obj, res_prim, res_dual) and the C++ facade prints them in the reference's
 * format.  Verbose calls run on the generic or the one-QP-per-lane kernel (the ones that record) — a debugging mode. */
int sqph_set_trace_qp(sqph_solver *s, int index);
int sqph_get_trace(sqph_solver *s, double *records, int cap_records, int *count);

/* ---- multi-GPU (single process, one handle per device): SURVEY.md §8(e).  The batch shards contiguously over the devices
 * (sqph_shard_bounds), every device solves its shard on its own stream with no data-path exchange, and the only
 * communication is the collection of the result records on a root device: peer-to-peer copies over xGMI
 * (hipMemcpyPeerAsync), posted on the producing solver's stream right behind its solve. */
/* Contiguous block split of `total` QPs over `parts` devices/ranks (the first total % parts get one more): [lo, hi). */
void sqph_shard_bounds(long long total, int parts, int part, long long *lo, long long *hi);
/* Number of HIP devices visible to this process (0 when none / on error). */
int sqph_device_count(void);
/* Give the solver a private non-blocking stream on its device (destroyed with the solver) so that solvers on different
 * devices — or several solvers on one — run concurrently from one host thread. */
int sqph_own_stream(sqph_solver *s);
/* Root-side buffers for the gathered records of `total` QPs (fp64 x [total][n], y [total][m], sqph_info [total]) on `device`. */
typedef struct sqph_gather sqph_gather;
int sqph_gather_create(sqph_gather **out, int device, int n, int m, long long total);
void sqph_gather_destroy(sqph_gather *g);
/* The same with flags: 0, or SQPH_GATHER_RCCL_ALWAYS (also shards that live on the root's device go through RCCL, as a send to
 * self: exercises the RCCL leg on a one-GPU box), or SQPH_GATHER_NO_RCCL (peer copies only). */
enum { SQPH_GATHER_RCCL_ALWAYS = 1, SQPH_GATHER_NO_RCCL = 2 };
int sqph_gather_create_ex(sqph_gather **out, int device, int n, int m, long long total, int flags);
/* Enqueue, on src's stream, the transfer of src's first `count` result records to positions [offset, offset+count) of the
 * gather buffers.  A shard on another device than the root's is sent with RCCL (grouped ncclSend / ncclRecv, point to point
 * over xGMI; librccl is opened on first use) — hipMemcpyPeerAsync where RCCL is not available; a shard on the root's own
 * device is a device-to-device copy.  Asynchronous; safe to call from one host thread per shard. */
int sqph_gather_post(sqph_gather *g, sqph_solver *src, long long offset, int count);
/* The same for k shards in ONE call: every cross-device shard's send / receive pair goes into one RCCL group, the root's
 * receives run on one stream per source device — seven peers use their seven xGMI links at once instead of queueing on one
 * receive stream.  What a caller that launches its shards from one host thread should use (MultiGpuBatchQPSolver::
 * setup_solve_device does). */
int sqph_gather_post_many(sqph_gather *g, sqph_solver *const *srcs, const long long *offsets, const int *counts, int k);
/* "rccl", "peer-copy" or "none": what the last sqph_gather_post on g used. */
const char *sqph_gather_transport(const sqph_gather *g);
/* Wait for every posted copy, then copy the gathered records to host buffers (NULL = skip; x/y narrowed to `dtype`). */
int sqph_gather_fetch(sqph_gather *g, int dtype, void *x, void *y, sqph_info *info);
/* Wait for every posted copy and expose the device buffers (valid until sqph_gather_destroy). */
int sqph_gather_device_ptrs(sqph_gather *g, void **x, void **y, sqph_info **info);

/* Name of the kernel variant the last launch used ("generic_w1", "wg2_16x8_7x7s_w2", "lane_2x3_exact", "csr_t7", ...). */
const char *sqph_kernel_name(const sqph_solver *s);
/* Last launch duration helpers: records HIP events around every launch when enabled. */
int sqph_enable_timing(sqph_solver *s, int on);
/* Milliseconds of the last launched kernel (synchronises on its stop event). */
int sqph_last_kernel_ms(sqph_solver *s, float *ms);
/* Durations (ms) of every launch since timing was enabled / last collected; writes up to
 * `cap` values, returns the number of launches in *count and clears the list. */
int sqph_collect_kernel_ms(sqph_solver *s, float *ms, int cap, int *count);

const char *sqph_last_error(const sqph_solver *s);
/* error text for failures where no solver exists yet (sqph_create) */
const char *sqph_global_error(void);

/* static QPSolver::constr_type_init(l, u, constr_type), src/qp.cpp:283-294 (host, no device) */
int sqph_constr_type_init(int dtype, int m, const void *l, const void *u, int *constr_type);

/* Algorithmic HBM bytes of one setup+solve of one QP:
 * sizeof(Scalar)*(n*n + n + m*n + 2m) read + sizeof(Scalar)*(n+m) written + sizeof(sqph_info). */
long long sqph_algorithmic_bytes(int n, int m, int dtype);

int sqph_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SQP_HIP_H */
