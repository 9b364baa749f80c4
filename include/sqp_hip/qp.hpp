// qp.hpp — source-compatible C++ facades of the reference's QP solver classes on top of the C-ABI
// (include/sqp_hip.h, libsqp_hip.so).  Header-only, Eigen-optional.
//
//   reference class                                         facade here
//   ------------------------------------------------------  -------------------------------------------
//   qp_solver::QuadraticProblem<Scalar>  solvers/qp.hpp:19-34    qp_solver::QuadraticProblem<Scalar>
//   qp_solver::QPSolverSettings<Scalar>  solvers/qp.hpp:36-54    qp_solver::QPSolverSettings<Scalar>
//   qp_solver::QPSolverInfo<Scalar>      solvers/qp.hpp:72-80    qp_solver::QPSolverInfo<Scalar>
//   qp_solver::QPSolver<Scalar>          solvers/qp.hpp:118-248  qp_solver::QPSolver<Scalar>      (batch of 1)
//   (new)                                                      qp_solver::BatchQPSolver<Scalar> (batch of N, the GPU's natural unit)
//   qp_solver::QP<n,m,Scalar> + QPSolver<QP<n,m>>  unsupported/qp_solver.hpp:18-49,135-592
//                                                              qp_solver::legacy::QP<n,m,Scalar>, legacy::QPSolver<QPType>
//
// Two build modes:
//   * Eigen on the include path (SQP_HIP_HAVE_EIGEN): the DROP-IN mode.  QuadraticProblem holds the reference's five
//     `const Eigen::Matrix*` members (`qp.P = &P;`), primal_solution()/dual_solution() return mutable Eigen vectors (a
//     vector the caller modified is sent back to the device before the next solve(), like writing to the reference's
//     x/y members), constr_type_init takes (const Vector&, const Vector&, Eigen::VectorXi&), the legacy QP<n,m> has Eigen
//     members.  include/sqp_hip/compat/ mirrors the reference's include paths: `-I include/sqp_hip/compat` in place of
//     the reference's `include/` makes `#include "solvers/qp.hpp"` (supported class) and
//     `#include "unsupported/qp_solver.hpp"` / `"solvers/qp_solver.hpp"` (legacy class, SQP_HIP_LEGACY_API) resolve here.
//   * no Eigen: the same classes over raw column-major pointers (RawQuadraticProblem; what the GPU box's tests use).
// The supported and the legacy class share the name qp_solver::QPSolver in the reference (two alternative headers); here
// they live in the inline namespace `supported` and the namespace `legacy` — SQP_HIP_LEGACY_API flips which one is inline.
//
// Method names, argument meaning, status values, iteration bookkeeping and the cold-start quirk of each class follow the
// reference: the supported class does NOT reset x,z,y in solve() when warm_start=false (src/qp.cpp:78-82 is a no-op), the
// legacy class does (unsupported/qp_solver.hpp:256-260).
//
// Behavioural difference (documented in sqp_hip.h and INTEGRATION.md): the device factorises the Schur complement
// S = P + sigma I + A'RA, which must be positive definite; a P that makes S indefinite yields NUMERICAL_ISSUES where the
// reference's pivoted LDL' of the indefinite KKT matrix would go on iterating (on a non-convex problem).
#pragma once
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../sqp_hip.h"

#if !defined(SQP_HIP_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define SQP_HIP_HAVE_EIGEN 1
#endif
#endif

#ifdef SQP_HIP_LEGACY_API
#define SQP_HIP_INLINE_SUPPORTED
#define SQP_HIP_INLINE_LEGACY inline
#else
#define SQP_HIP_INLINE_SUPPORTED inline
#define SQP_HIP_INLINE_LEGACY
#endif

namespace qp_solver {

SQP_HIP_INLINE_SUPPORTED namespace supported {

typedef enum { SOLVED, MAX_ITER_EXCEEDED, UNSOLVED, NUMERICAL_ISSUES, UNINITIALIZED } QPSolverStatus;

// Problem data over raw pointers: P is n x n, A is m x n, both column-major (Eigen's default layout).  Borrowed.
template <typename Scalar = double>
struct RawQuadraticProblem {
    int n = 0, m = 0;
    const Scalar *P = nullptr;  // only the lower triangle enters the factor (reference: Eigen::LDLT<.,Lower>)
    const Scalar *q = nullptr;
    const Scalar *A = nullptr;
    const Scalar *l = nullptr;
    const Scalar *u = nullptr;
};

#ifdef SQP_HIP_HAVE_EIGEN
// The reference's struct, member for member (solvers/qp.hpp:19-34): five non-owning pointers to Eigen objects that must
// outlive setup()/solve().
template <typename Scalar = double>
struct QuadraticProblem {
    using Vector = Eigen::Matrix<Scalar, Eigen::Dynamic, 1>;
    using Matrix = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic>;
    const Matrix *P;
    const Vector *q;
    const Matrix *A;
    const Vector *l;
    const Vector *u;
};
#else
template <typename Scalar = double>
using QuadraticProblem = RawQuadraticProblem<Scalar>;
#endif

template <typename Scalar>
struct QPSolverSettings {
    Scalar rho = 1e-1;
    Scalar sigma = 1e-6;
    Scalar alpha = 1.0;
    Scalar eps_rel = 1e-3;
    Scalar eps_abs = 1e-3;
    int max_iter = 1000;
    int check_termination = 25;
    bool warm_start = false;
    bool adaptive_rho = false;
    Scalar adaptive_rho_tolerance = 5;
    int adaptive_rho_interval = 25;
    bool verbose = false;

    void print() const {
        printf("ADMM settings:\n  sigma %.2e\n  rho %.2e\n  alpha %.2f\n  eps_rel %.1e\n  eps_abs %.1e\n  max_iter %d\n  adaptive_rho %d\n  warm_start %d\n",
               (double)sigma, (double)rho, (double)alpha, (double)eps_rel, (double)eps_abs, max_iter, (int)adaptive_rho, (int)warm_start);
    }
};

template <typename Scalar>
struct QPSolverInfo {
    QPSolverStatus status = UNINITIALIZED;
    int iter = 0;
    int rho_updates = 0;
    Scalar rho_estimate = 0;
    Scalar res_prim = 0;
    Scalar res_dual = 0;

    void print() const {
        static const char *names[] = {"SOLVED", "MAX_ITER_EXCEEDED", "UNSOLVED", "NUMERICAL_ISSUES", "UNINITIALIZED"};
        printf("ADMM info:\n  status %s\n  iter %d\n  rho_updates %d\n  rho_estimate %f\n  res_prim %f\n  res_dual %f\n",
               names[(int)status], iter, rho_updates, (double)rho_estimate, (double)res_prim, (double)res_dual);
    }
};

namespace detail {
template <typename Scalar> struct dtype_of;
template <> struct dtype_of<double> { static constexpr int value = SQPH_F64; };
template <> struct dtype_of<float> { static constexpr int value = SQPH_F32; };

inline void check(int rc, sqph_solver *s, const char *what) {
    if (rc != SQPH_OK) throw std::runtime_error(std::string(what) + ": " + (s ? sqph_last_error(s) : sqph_global_error()));
}
}  // namespace detail

// N independent QPSolver<Scalar> instances for same-(n,m) problems, solved by one kernel launch per call.
template <typename SCALAR>
class BatchQPSolver {
   public:
    using Scalar = SCALAR;
    using Settings = QPSolverSettings<Scalar>;
    using Info = QPSolverInfo<Scalar>;

    BatchQPSolver(int n, int m, int batch, int device = 0, int flags = 0) : n_(n), m_(m), batch_(batch) {
        detail::check(sqph_create(&h_, device, n, m, batch, detail::dtype_of<Scalar>::value, flags), nullptr, "sqph_create");
        x_.resize((size_t)batch * n);
        y_.resize((size_t)batch * (m > 0 ? m : 1));
        z_.resize((size_t)batch * (m > 0 ? m : 1));
        info_.resize(batch);
        raw_info_.resize(batch);
    }
    ~BatchQPSolver() { sqph_destroy(h_); }
    BatchQPSolver(const BatchQPSolver &) = delete;
    BatchQPSolver &operator=(const BatchQPSolver &) = delete;

    Settings &settings() { return settings_; }
    const Settings &settings() const { return settings_; }

    // Problem b of the batch starts at base + b*stride (elements); stride 0 shares the array.
    struct Batch {
        int batch;
        int memspace;  // SQPH_HOST or SQPH_DEVICE
        const Scalar *P, *q, *A, *l, *u;
        long long stride_P, stride_q, stride_A, stride_l, stride_u;
    };
    Batch packed(int batch, const Scalar *P, const Scalar *q, const Scalar *A, const Scalar *l, const Scalar *u, int memspace = SQPH_HOST) const {
        return Batch{batch, memspace, P, q, A, l, u, (long long)n_ * n_, n_, (long long)m_ * n_, m_, m_};
    }

    void setup(const Batch &b) { call(sqph_setup, b, "sqph_setup"); }
    void update_qp(const Batch &b) { call(sqph_update_qp, b, "sqph_update_qp"); }
    void solve(const Batch &b) { call(sqph_solve, b, "sqph_solve"); }
    void setup_solve(const Batch &b) { call(sqph_setup_solve, b, "sqph_setup_solve"); }  // what SQP::run_solve_qp does (src/sqp.cpp:221-222)
    // the same for QPs whose P and A are those of the previous setup (only q, l, u differ — the SQP second-order correction,
    // src/sqp.cpp:244-276): the resident factor is reused where the rho vector did not move (create with SQPH_FLAG_KEEP_FACTOR)
    void setup_solve_reuse(const Batch &b) { call(sqph_setup_solve_reuse, b, "sqph_setup_solve_reuse"); }
    // update_qp() + solve() in one launch: the iterates of the previous call are the starting point (src/qp.cpp:46-62)
    void update_solve(const Batch &b) { call(sqph_update_solve, b, "sqph_update_solve"); }

    // The same calls with the constraint matrices in CSR (legacy sparse class, unsupported/qp_solver.hpp:17-32; BASELINE
    // config 5).  Per-QP arrays: rowptr [m+1], colind/val [nnz_max]; packed_csr lays QPs back to back.
    struct CsrBatch {
        int batch;
        int memspace;
        const Scalar *P, *q;
        const int *rowptr, *colind;
        const Scalar *val, *l, *u;
        long long stride_P, stride_q, stride_rowptr, stride_colind, stride_val, stride_l, stride_u, nnz_max;
    };
    CsrBatch packed_csr(int batch, const Scalar *P, const Scalar *q, const int *rowptr, const int *colind, const Scalar *val, long long nnz_max,
                        const Scalar *l, const Scalar *u, int memspace = SQPH_HOST) const {
        return CsrBatch{batch, memspace, P, q, rowptr, colind, val, l, u, (long long)n_ * n_, n_, m_ + 1, nnz_max, nnz_max, m_, m_, nnz_max};
    }
    void setup_csr(const CsrBatch &b) { call_csr(sqph_setup_csr, b, "sqph_setup_csr"); }
    void update_qp_csr(const CsrBatch &b) { call_csr(sqph_update_qp_csr, b, "sqph_update_qp_csr"); }
    void solve_csr(const CsrBatch &b) { call_csr(sqph_solve_csr, b, "sqph_solve_csr"); }
    void setup_solve_csr(const CsrBatch &b) { call_csr(sqph_setup_solve_csr, b, "sqph_setup_solve_csr"); }
    void update_solve_csr(const CsrBatch &b) { call_csr(sqph_update_solve_csr, b, "sqph_update_solve_csr"); }
    void setup_solve_reuse_csr(const CsrBatch &b) { call_csr(sqph_setup_solve_reuse_csr, b, "sqph_setup_solve_reuse_csr"); }  // the SOC re-solve, sparse route
    // ... and with P sparse as well (the legacy sparse class keeps P as Eigen::SparseMatrix, unsupported/qp_solver.hpp:24-25): the
    // full symmetric matrix in compressed-column form, colptr [n+1], rowind / val [nnz_max] per QP; b.P is ignored.
    struct CscP {
        const int *colptr, *rowind;
        const Scalar *val;
        long long stride_colptr, stride_rowind, stride_val, nnz_max;
    };
    CscP packed_csc_P(const int *colptr, const int *rowind, const Scalar *val, long long nnz_max) const {
        return CscP{colptr, rowind, val, n_ + 1, nnz_max, nnz_max, nnz_max};
    }
    void setup_csr(const CsrBatch &b, const CscP &P) { call_csr_sp(sqph_setup_csr_sp, b, P, "sqph_setup_csr_sp"); }
    void update_qp_csr(const CsrBatch &b, const CscP &P) { call_csr_sp(sqph_update_qp_csr_sp, b, P, "sqph_update_qp_csr_sp"); }
    void solve_csr(const CsrBatch &b, const CscP &P) { call_csr_sp(sqph_solve_csr_sp, b, P, "sqph_solve_csr_sp"); }
    void setup_solve_csr(const CsrBatch &b, const CscP &P) { call_csr_sp(sqph_setup_solve_csr_sp, b, P, "sqph_setup_solve_csr_sp"); }
    void update_solve_csr(const CsrBatch &b, const CscP &P) { call_csr_sp(sqph_update_solve_csr_sp, b, P, "sqph_update_solve_csr_sp"); }
    void setup_solve_reuse_csr(const CsrBatch &b, const CscP &P) { call_csr_sp(sqph_setup_solve_reuse_csr_sp, b, P, "sqph_setup_solve_reuse_csr_sp"); }

    // results of the last call (host copies, fetched lazily)
    const Scalar *primal_solution(int b) { fetch(); return &x_[(size_t)b * n_]; }
    const Scalar *dual_solution(int b) { fetch(); return &y_[(size_t)b * m_]; }
    const Scalar *z(int b) { fetch(); return &z_[(size_t)b * m_]; }  // the reference keeps z as solver state (qp.hpp:224)
    const Info &info(int b) { fetch(); return info_[b]; }
    // overwrite iterates of the first `batch` instances (nullptr = leave): the reference's writable primal/dual accessors
    void set_state(int batch, const Scalar *x, const Scalar *z, const Scalar *y) {
        detail::check(sqph_set_state(h_, batch, SQPH_HOST, x, z, y), h_, "sqph_set_state");
        fetched_ = false;
    }
    // settings().verbose: which QP of the batch is traced (default 0), and the records of the last solve call — one per
    // termination check: {iter, objective, res_prim, res_dual} (reference print_status, src/qp.cpp:373-383)
    void set_trace_qp(int b) { detail::check(sqph_set_trace_qp(h_, b), h_, "sqph_set_trace_qp"); }
    std::vector<double> trace() {
        int count = 0;
        detail::check(sqph_get_trace(h_, nullptr, 0, &count), h_, "sqph_get_trace");
        std::vector<double> rec((size_t)4 * count);
        if (count) detail::check(sqph_get_trace(h_, rec.data(), count, &count), h_, "sqph_get_trace");
        return rec;
    }
    // prints what the reference prints for a verbose solve: the per-check table, then the info record
    void print_trace(int b = 0) {
        const std::vector<double> rec = trace();
        for (size_t k = 0; k < rec.size() / 4; k++) {
            if (k == 0) printf("iter   obj       rp        rd\n");
            printf("%4d  %.2e  %.2e  %.2e\n", (int)rec[4 * k], rec[4 * k + 1], rec[4 * k + 2], rec[4 * k + 3]);
        }
        info(b).print();
    }
    sqph_solver *handle() { return h_; }
    int n() const { return n_; }
    int m() const { return m_; }

   private:
    void push_settings() {
        sqph_settings st;
        st.rho = settings_.rho; st.sigma = settings_.sigma; st.alpha = settings_.alpha;
        st.eps_rel = settings_.eps_rel; st.eps_abs = settings_.eps_abs;
        st.max_iter = settings_.max_iter; st.check_termination = settings_.check_termination;
        st.warm_start = settings_.warm_start; st.adaptive_rho = settings_.adaptive_rho;
        st.adaptive_rho_tolerance = settings_.adaptive_rho_tolerance;
        st.adaptive_rho_interval = settings_.adaptive_rho_interval; st.verbose = settings_.verbose;
        detail::check(sqph_set_settings(h_, &st), h_, "sqph_set_settings");
    }
    static sqph_csr_batch c_csr(const CsrBatch &b) {
        sqph_csr_batch c;
        c.batch = b.batch; c.memspace = b.memspace;
        c.P = b.P; c.q = b.q; c.A_rowptr = b.rowptr; c.A_colind = b.colind; c.A_val = b.val; c.l = b.l; c.u = b.u;
        c.stride_P = b.stride_P; c.stride_q = b.stride_q; c.stride_rowptr = b.stride_rowptr; c.stride_colind = b.stride_colind;
        c.stride_val = b.stride_val; c.stride_l = b.stride_l; c.stride_u = b.stride_u; c.nnz_max = b.nnz_max;
        return c;
    }
    template <typename F>
    void call_csr(F fn, const CsrBatch &b, const char *what) {
        push_settings();
        const sqph_csr_batch c = c_csr(b);
        detail::check(fn(h_, &c), h_, what);
        last_batch_ = b.batch;
        fetched_ = false;
    }
    template <typename F>
    void call_csr_sp(F fn, const CsrBatch &b, const CscP &P, const char *what) {
        push_settings();
        const sqph_csr_batch c = c_csr(b);
        const sqph_csc_P sp{P.colptr, P.rowind, P.val, P.stride_colptr, P.stride_rowind, P.stride_val, P.nnz_max};
        detail::check(fn(h_, &c, &sp), h_, what);
        last_batch_ = b.batch;
        fetched_ = false;
    }
    template <typename F>
    void call(F fn, const Batch &b, const char *what) {
        push_settings();
        sqph_qp_batch qb;
        qb.batch = b.batch; qb.memspace = b.memspace;
        qb.P = b.P; qb.q = b.q; qb.A = b.A; qb.l = b.l; qb.u = b.u;
        qb.stride_P = b.stride_P; qb.stride_q = b.stride_q; qb.stride_A = b.stride_A; qb.stride_l = b.stride_l; qb.stride_u = b.stride_u;
        detail::check(fn(h_, &qb), h_, what);
        last_batch_ = b.batch;
        fetched_ = false;
    }
    void fetch() {
        if (fetched_) return;
        detail::check(sqph_get_solution(h_, last_batch_, SQPH_HOST, x_.data(), m_ ? y_.data() : nullptr, m_ ? z_.data() : nullptr, raw_info_.data()), h_, "sqph_get_solution");
        for (int b = 0; b < last_batch_; b++) {
            info_[b].status = (QPSolverStatus)raw_info_[b].status;
            info_[b].iter = raw_info_[b].iter;
            info_[b].rho_updates = raw_info_[b].rho_updates;
            info_[b].rho_estimate = (Scalar)raw_info_[b].rho_estimate;
            info_[b].res_prim = (Scalar)raw_info_[b].res_prim;
            info_[b].res_dual = (Scalar)raw_info_[b].res_dual;
        }
        fetched_ = true;
    }

    int n_, m_, batch_, last_batch_ = 0;
    bool fetched_ = true;
    sqph_solver *h_ = nullptr;
    Settings settings_;
    std::vector<Scalar> x_, y_, z_;
    std::vector<Info> info_;
    std::vector<sqph_info> raw_info_;
};

// Drop-in for the supported class, reference include/solvers/qp.hpp:118-173 (one problem per instance).
template <typename SCALAR>
class QPSolver {
   public:
    using Scalar = SCALAR;
    using QP = QuadraticProblem<Scalar>;
    using RawQP = RawQuadraticProblem<Scalar>;
    using Settings = QPSolverSettings<Scalar>;
    using Info = QPSolverInfo<Scalar>;
#ifdef SQP_HIP_HAVE_EIGEN
    using Vector = Eigen::Matrix<Scalar, Eigen::Dynamic, 1>;
    using Matrix = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic>;
#else
    using Vector = std::vector<Scalar>;
#endif

    enum { INEQUALITY_CONSTRAINT, EQUALITY_CONSTRAINT, LOOSE_BOUNDS } ConstraintType;
    static constexpr Scalar RHO_MIN = 1e-6;
    static constexpr Scalar RHO_MAX = 1e+6;
    static constexpr Scalar RHO_TOL = 1e-4;
    static constexpr Scalar RHO_EQ_FACTOR = 1e+3;
    static constexpr Scalar LOOSE_BOUNDS_THRESH = 1e+16;
    static constexpr Scalar DIV_BY_ZERO_REGUL = std::numeric_limits<Scalar>::epsilon();

    explicit QPSolver(int device = 0, int flags = 0) : device_(device), flags_(flags) {}
    ~QPSolver() { delete impl_; }
    QPSolver(const QPSolver &) = delete;
    QPSolver &operator=(const QPSolver &) = delete;

    void setup(const RawQP &qp) {
        if (!impl_ || impl_->n() != qp.n || impl_->m() != qp.m) {
            // a new shape starts a fresh instance, like the resize() cascade of src/qp.cpp:13-29
            delete impl_;
            impl_ = new BatchQPSolver<Scalar>(qp.n, qp.m, 1, device_, flags_);
        }
        push(false);
        impl_->setup(impl_->packed(1, qp.P, qp.q, qp.A, qp.l, qp.u));
        A_factored_.assign(qp.A, qp.A + (size_t)qp.n * qp.m);
        pull();
    }
    void update_qp(const RawQP &qp) {
        if (!impl_) return;
        push(true);
        impl_->update_qp(impl_->packed(1, qp.P, qp.q, qp.A, qp.l, qp.u));
        A_factored_.assign(qp.A, qp.A + (size_t)qp.n * qp.m);
        pull();
    }
    void solve(const RawQP &qp) {
        if (!impl_) return;  // UNINITIALIZED: solve() returns silently, src/qp.cpp:68-71
        if (info_.status == UNINITIALIZED || info_.status == NUMERICAL_ISSUES) return;
        // The reference iterates on the factor of setup()'s A and reads solve()'s A for the residuals only (src/qp.cpp:319-331,
        // 355-360); the device iterates on B = A W', so another A here would silently solve another problem: rejected instead
        // (update_qp() is the call that takes a new A, as in the reference's own use, src/qp.cpp:46-62).
        if (qp.n != impl_->n() || qp.m != impl_->m()) throw std::invalid_argument("QPSolver::solve: problem shape differs from setup()'s");
        for (size_t k = 0; k < A_factored_.size(); k++)
            if (!(qp.A[k] == A_factored_[k]) && !(qp.A[k] != qp.A[k] && A_factored_[k] != A_factored_[k]))
                throw std::invalid_argument("QPSolver::solve: A differs from the A of the preceding setup()/update_qp(); call update_qp()");
        if (settings_.verbose) settings_.print();  // QP_SOLVER_PRINTING, src/qp.cpp:72-76
        push(true);
        impl_->solve(impl_->packed(1, qp.P, qp.q, qp.A, qp.l, qp.u));
        pull();
        if (settings_.verbose) impl_->print_trace(0);  // print_status lines + info_.print(), src/qp.cpp:113-117, 152-156
    }
#ifdef SQP_HIP_HAVE_EIGEN
    void setup(const QP &qp) { setup(raw(qp)); }
    void update_qp(const QP &qp) { update_qp(raw(qp)); }
    void solve(const QP &qp) { solve(raw(qp)); }
    // static QPSolver::constr_type_init(l, u, constr_type), src/qp.cpp:283-294 — the caller sizes constr_type
    static void constr_type_init(const Vector &l, const Vector &u, Eigen::VectorXi &constr_type) {
        constr_type_init((int)l.rows(), l.data(), u.data(), constr_type.data());
    }
#endif

    // mutable like the reference's (qp.hpp:160-164): what the caller writes here is the state the next solve() starts from
    const Vector &primal_solution() const { return x_; }
    Vector &primal_solution() { return x_; }
    const Vector &dual_solution() const { return y_; }
    Vector &dual_solution() { return y_; }
    Settings &settings() { return settings_; }
    const Settings &settings() const { return settings_; }
    Info &info() { return info_; }
    const Info &info() const { return info_; }

    static void constr_type_init(int m, const Scalar *l, const Scalar *u, int *constr_type) {
        detail::check(sqph_constr_type_init(detail::dtype_of<Scalar>::value, m, l, u, constr_type), nullptr, "sqph_constr_type_init");
    }

   private:
#ifdef SQP_HIP_HAVE_EIGEN
    static RawQP raw(const QP &qp) {
        RawQP r;
        r.n = (int)qp.P->rows(); r.m = (int)qp.A->rows();
        r.P = qp.P->data(); r.q = qp.q->data(); r.A = qp.A->data(); r.l = qp.l->data(); r.u = qp.u->data();
        return r;
    }
#endif
    void push(bool send_state) {
        impl_->settings() = settings_;
        if (!send_state) return;
        // iterates the caller changed through the accessors since the last call go back to the device
        const int n = impl_->n(), m = impl_->m();
        bool dx = (int)x_.size() != n, dy = (int)y_.size() != m;
        for (int i = 0; i < n && !dx; i++) dx = !(x_[i] == x_seen_[i]);
        for (int i = 0; i < m && !dy; i++) dy = !(y_[i] == y_seen_[i]);
        if ((int)x_.size() != n || (int)y_.size() != m) return;  // resized by the caller: nothing sensible to send
        if (dx || dy) impl_->set_state(1, dx ? &x_[0] : nullptr, nullptr, (dy && m) ? &y_[0] : nullptr);
    }
    void pull() {
        const int n = impl_->n(), m = impl_->m();
#ifdef SQP_HIP_HAVE_EIGEN
        x_.resize(n);
        y_.resize(m);
#else
        x_.resize(n);
        y_.resize(m);
#endif
        for (int i = 0; i < n; i++) x_[i] = impl_->primal_solution(0)[i];
        for (int i = 0; i < m; i++) y_[i] = impl_->dual_solution(0)[i];
        x_seen_.assign(impl_->primal_solution(0), impl_->primal_solution(0) + n);
        y_seen_.assign(impl_->dual_solution(0), impl_->dual_solution(0) + m);
        info_ = impl_->info(0);
    }
    int device_, flags_;
    BatchQPSolver<Scalar> *impl_ = nullptr;
    Settings settings_;
    Info info_;
    Vector x_, y_;
    std::vector<Scalar> x_seen_, y_seen_;
    std::vector<Scalar> A_factored_;  // the A the resident factor was built from (solve() checks its argument against it)
};

}  // namespace supported

// Fixed-size legacy API, reference include/unsupported/qp_solver.hpp:18-49,135-592.
SQP_HIP_INLINE_LEGACY namespace legacy {

// the legacy header's own names for settings / info / status (unsupported/qp_solver.hpp:51-127): no NUMERICAL_ISSUES
template <typename Scalar>
using qp_sover_settings_t = supported::QPSolverSettings<Scalar>;
typedef enum { SOLVED, MAX_ITER_EXCEEDED, UNSOLVED, UNINITIALIZED } status_t;
template <typename Scalar>
struct qp_solver_info_t {
    status_t status = UNINITIALIZED;
    int iter = 0;
    int rho_updates = 0;
    Scalar rho_estimate = 0;
    Scalar res_prim = 0;
    Scalar res_dual = 0;
};

template <int N_, int M_, typename Scalar_ = double>
struct QP {
    using Scalar = Scalar_;
    enum { n = N_, m = M_ };
#if defined(SQP_HIP_HAVE_EIGEN) && defined(QP_SOLVER_USE_SPARSE)
    // the sparse variant of the legacy class, unsupported/qp_solver.hpp:17-32: P and A are Eigen::SparseMatrix; the column
    // counts only size the reference's KKT matrix and are not needed here.  P is densified (n x n) and A handed over in CSR
    // (sqph_*_csr) by QPSolver below.
    Eigen::SparseMatrix<Scalar> P;
    Eigen::Matrix<int, N_, 1> P_col_nnz;
    Eigen::Matrix<Scalar, N_, 1> q;
    Eigen::SparseMatrix<Scalar> A;
    Eigen::Matrix<int, N_, 1> A_col_nnz;
    Eigen::Matrix<Scalar, M_, 1> l, u;
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#elif defined(SQP_HIP_HAVE_EIGEN)
    Eigen::Matrix<Scalar, N_, N_> P;  // members as in unsupported/qp_solver.hpp:35-48
    Eigen::Matrix<Scalar, N_, 1> q;
    Eigen::Matrix<Scalar, M_, N_> A;
    Eigen::Matrix<Scalar, M_, 1> l, u;
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#else
    Scalar P[N_ * N_];  // column-major
    Scalar q[N_];
    Scalar A[M_ * N_];  // column-major
    Scalar l[M_], u[M_];
#endif
};

namespace detail {
#ifdef SQP_HIP_HAVE_EIGEN
template <typename T> inline const typename T::Scalar *ptr(const T &v) { return v.data(); }
template <typename T> inline typename T::Scalar *ptr(T &v) { return v.data(); }
#else
template <typename S> inline const S *ptr(const S *v) { return v; }
template <typename S> inline S *ptr(S *v) { return v; }
#endif
}  // namespace detail

// LinearSolver / UpLo are accepted for source compatibility and ignored: the device always factors the Schur complement.
#if defined(SQP_HIP_HAVE_EIGEN) && defined(QP_SOLVER_USE_SPARSE)
template <typename QPType, template <typename, int, typename...> class LinearSolver = Eigen::SimplicialLDLT, int LinearSolver_UpLo = Eigen::Lower>
#elif defined(SQP_HIP_HAVE_EIGEN)
template <typename QPType, template <typename, int, typename...> class LinearSolver = Eigen::LDLT, int LinearSolver_UpLo = Eigen::Lower>
#else
template <typename QPType>
#endif
class QPSolver {
   public:
    enum { n = QPType::n, m = QPType::m };
    using qp_t = QPType;
    using Scalar = typename QPType::Scalar;
    using settings_t = qp_sover_settings_t<Scalar>;
    using info_t = qp_solver_info_t<Scalar>;
    static constexpr Scalar RHO_MIN = 1e-6;
    static constexpr Scalar RHO_MAX = 1e+6;
    static constexpr Scalar RHO_TOL = 1e-4;
    static constexpr Scalar RHO_EQ_FACTOR = 1e+3;
    static constexpr Scalar LOOSE_BOUNDS_THRESH = 1e+16;
    static constexpr Scalar DIV_BY_ZERO_REGUL = std::numeric_limits<Scalar>::epsilon();
    // public state, as in the reference (unsupported/qp_solver.hpp:172-200)
    int iter = 0;
#ifdef SQP_HIP_HAVE_EIGEN
    using var_t = Eigen::Matrix<Scalar, n, 1>;
    using constraint_t = Eigen::Matrix<Scalar, m, 1>;
    using dual_t = Eigen::Matrix<Scalar, m, 1>;
    var_t x;
    constraint_t z;
    dual_t y;
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#else
    Scalar x[n], z[m > 0 ? m : 1], y[m > 0 ? m : 1];
#endif
    enum { INEQUALITY_CONSTRAINT, EQUALITY_CONSTRAINT, LOOSE_BOUNDS } constr_type[m > 0 ? m : 1];
    settings_t _settings;
    info_t _info;

    explicit QPSolver(int device = 0) : impl_(n, m, 1, device, SQPH_FLAG_LEGACY_COLD_START) {
        for (int i = 0; i < n; i++) x[i] = 0;
        for (int i = 0; i < m; i++) z[i] = y[i] = 0;
    }
    void setup(const qp_t &qp) { run(OP_SETUP, qp, false); }
    void update_qp(const qp_t &qp) { run(OP_UPDATE, qp, true); }
    void solve(const qp_t &qp) {
        if (_info.status == UNINITIALIZED) return;  // unsupported/qp_solver.hpp:246-249
        run(OP_SOLVE, qp, true);
    }
#ifdef SQP_HIP_HAVE_EIGEN
    const var_t &primal_solution() const { return x; }
    var_t &primal_solution() { return x; }
    const dual_t &dual_solution() const { return y; }
    dual_t &dual_solution() { return y; }
#else
    const Scalar *primal_solution() const { return x; }
    Scalar *primal_solution() { return x; }
    const Scalar *dual_solution() const { return y; }
    Scalar *dual_solution() { return y; }
#endif
    const settings_t &settings() const { return _settings; }
    settings_t &settings() { return _settings; }
    const info_t &info() const { return _info; }
    info_t &info() { return _info; }

   private:
    enum op_t { OP_SETUP, OP_UPDATE, OP_SOLVE };
#if defined(SQP_HIP_HAVE_EIGEN) && defined(QP_SOLVER_USE_SPARSE)
    // Eigen's column-major A -> CSR (rows in order, columns ascending inside a row); P stays sparse: its compressed columns as they
    // are (sqph_*_csr_sp expands them on the device)
    void dispatch(op_t op, const qp_t &qp) {
        typedef Eigen::SparseMatrix<Scalar> SpMat;
        std::vector<int> pcol((size_t)n + 1, 0), prow;
        std::vector<Scalar> pval;
        for (int k = 0; k < (int)qp.P.outerSize(); k++) {
            for (typename SpMat::InnerIterator it(qp.P, k); it; ++it) {
                prow.push_back((int)it.row());
                pval.push_back(it.value());
            }
            pcol[(size_t)k + 1] = (int)prow.size();
        }
        const long long pnnz = (long long)prow.size();
        if (prow.empty()) { prow.push_back(0); pval.push_back(Scalar(0)); }
        std::vector<int> rowptr((size_t)m + 1, 0);
        for (int k = 0; k < (int)qp.A.outerSize(); k++)
            for (typename SpMat::InnerIterator it(qp.A, k); it; ++it) rowptr[(size_t)it.row() + 1]++;
        for (int i = 0; i < m; i++) rowptr[(size_t)i + 1] += rowptr[(size_t)i];
        const int nnz = rowptr[(size_t)m];
        std::vector<int> colind((size_t)(nnz > 0 ? nnz : 1)), cur(rowptr.begin(), rowptr.end() - 1);
        std::vector<Scalar> val((size_t)(nnz > 0 ? nnz : 1));
        for (int k = 0; k < (int)qp.A.outerSize(); k++)
            for (typename SpMat::InnerIterator it(qp.A, k); it; ++it) {
                const int e = cur[(size_t)it.row()]++;
                colind[(size_t)e] = (int)it.col();
                val[(size_t)e] = it.value();
            }
        const auto b = impl_.packed_csr(1, nullptr, detail::ptr(qp.q), rowptr.data(), colind.data(), val.data(), nnz > 0 ? nnz : 1,
                                        detail::ptr(qp.l), detail::ptr(qp.u));
        // (an empty P — an LP — hands over one dummy entry that the all-zero column pointers never reference: nnz_max = 0 would turn
        // the per-QP value / row-index strides into "shared", which the C-ABI rejects next to a per-QP colptr)
        const auto sp = impl_.packed_csc_P(pcol.data(), prow.data(), pval.data(), pnnz > 0 ? pnnz : 1);
        if (op == OP_SETUP) impl_.setup_csr(b, sp);
        else if (op == OP_UPDATE) impl_.update_qp_csr(b, sp);
        else impl_.solve_csr(b, sp);
    }
#else
    void dispatch(op_t op, const qp_t &qp) {
        const auto b = impl_.packed(1, detail::ptr(qp.P), detail::ptr(qp.q), detail::ptr(qp.A), detail::ptr(qp.l), detail::ptr(qp.u));
        if (op == OP_SETUP) impl_.setup(b);
        else if (op == OP_UPDATE) impl_.update_qp(b);
        else impl_.solve(b);
    }
#endif
    void run(op_t op, const qp_t &qp, bool send_state) {
        impl_.settings() = _settings;
        if (send_state) {  // x, z, y are public members in the reference: what the caller left there is the solver's state
            bool d = false;
            for (int i = 0; i < n && !d; i++) d = !(x[i] == seen_[i]);
            for (int i = 0; i < m && !d; i++) d = !(z[i] == seen_[n + i]) || !(y[i] == seen_[n + m + i]);
            if (d) impl_.set_state(1, detail::ptr(x), m > 0 ? detail::ptr(z) : nullptr, m > 0 ? detail::ptr(y) : nullptr);
        }
        dispatch(op, qp);
        for (int i = 0; i < n; i++) seen_[i] = x[i] = impl_.primal_solution(0)[i];
        for (int i = 0; i < m; i++) seen_[n + i] = z[i] = impl_.z(0)[i];
        for (int i = 0; i < m; i++) seen_[n + m + i] = y[i] = impl_.dual_solution(0)[i];
        if (m > 0) {
            int ct[m > 0 ? m : 1];
            sqph_constr_type_init(supported::detail::dtype_of<Scalar>::value, m, detail::ptr(qp.l), detail::ptr(qp.u), ct);
            for (int i = 0; i < m; i++) constr_type[i] = static_cast<decltype(INEQUALITY_CONSTRAINT)>(ct[i]);
        }
        const auto &bi = impl_.info(0);
        // the legacy enum has no NUMERICAL_ISSUES (unsupported/qp_solver.hpp:84-89): a failed factorisation leaves UNSOLVED
        _info.status = bi.status == supported::SOLVED              ? SOLVED
                       : bi.status == supported::MAX_ITER_EXCEEDED ? MAX_ITER_EXCEEDED
                       : bi.status == supported::UNINITIALIZED     ? UNINITIALIZED
                                                                   : UNSOLVED;
        _info.iter = bi.iter; _info.rho_updates = bi.rho_updates; _info.rho_estimate = bi.rho_estimate;
        _info.res_prim = bi.res_prim; _info.res_dual = bi.res_dual;
        iter = _info.iter;
    }
    supported::BatchQPSolver<Scalar> impl_;
    Scalar seen_[n + 2 * (m > 0 ? m : 1)] = {};
};
}  // namespace legacy

}  // namespace qp_solver
