// qp.hpp — source-compatible C++ facades of the reference's QP solver classes on top of the C-ABI
// (include/sqp_hip.h, libsqp_hip.so).  Header-only, Eigen-optional.
//
//   reference class                                         facade here
//   ------------------------------------------------------  -------------------------------------------
//   qp_solver::QuadraticProblem<Scalar>  solvers/qp.hpp:19-34    qp_solver::QuadraticProblem<Scalar> (raw pointers + dims)
//   qp_solver::QPSolverSettings<Scalar>  solvers/qp.hpp:36-54    qp_solver::QPSolverSettings<Scalar>
//   qp_solver::QPSolverInfo<Scalar>      solvers/qp.hpp:72-80    qp_solver::QPSolverInfo<Scalar>
//   qp_solver::QPSolver<Scalar>          solvers/qp.hpp:118-248  qp_solver::QPSolver<Scalar>      (batch of 1)
//   (new)                                                      qp_solver::BatchQPSolver<Scalar> (batch of N, the GPU's natural unit)
//   qp_solver::QP<n,m,Scalar> + QPSolver<QP<n,m>>  unsupported/qp_solver.hpp:18-49,135-592
//                                                              qp_solver::legacy::QP<n,m,Scalar>, legacy::QPSolver<QPType>
//
// Method names, argument meaning, status values, iteration bookkeeping and the cold-start quirk of each
// class follow the reference: the supported class does NOT reset x,z,y in solve() when warm_start=false
// (src/qp.cpp:78-82 is a no-op), the legacy class does (unsupported/qp_solver.hpp:256-260).
// Eigen users: matrices are taken through data() pointers; Eigen's default column-major layout is what
// the C-ABI expects, so `qp.P = P.data(); qp.n = P.rows();` is all the glue needed.
#pragma once
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../sqp_hip.h"

// Eigen is optional: when its headers are on the include path the problem struct can be built straight from the
// reference's five `const Eigen::Matrix*` (include/solvers/qp.hpp:29-33).  (Not compiled in this repository's own
// test environment, which has no Eigen.)
#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define SQP_HIP_HAVE_EIGEN 1
#endif
#endif

namespace qp_solver {

typedef enum { SOLVED, MAX_ITER_EXCEEDED, UNSOLVED, NUMERICAL_ISSUES, UNINITIALIZED } QPSolverStatus;

template <typename Scalar = double>
struct QuadraticProblem {
    int n = 0, m = 0;           // P is n x n, A is m x n (column-major)
    const Scalar *P = nullptr;  // only the lower triangle enters the factor (reference: Eigen::LDLT<.,Lower>)
    const Scalar *q = nullptr;
    const Scalar *A = nullptr;
    const Scalar *l = nullptr;
    const Scalar *u = nullptr;

    QuadraticProblem() = default;
#ifdef SQP_HIP_HAVE_EIGEN
    using Matrix = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic>;  // column-major, as qp.hpp:21
    using Vector = Eigen::Matrix<Scalar, Eigen::Dynamic, 1>;
    // borrowed, like the reference: the Eigen objects must outlive setup()/solve()
    QuadraticProblem(const Matrix *P_, const Vector *q_, const Matrix *A_, const Vector *l_, const Vector *u_)
        : n((int)P_->rows()), m((int)A_->rows()), P(P_->data()), q(q_->data()), A(A_->data()), l(l_->data()), u(u_->data()) {}
#endif
};

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

    // results of the last call (host copies, fetched lazily)
    const Scalar *primal_solution(int b) { fetch(); return &x_[(size_t)b * n_]; }
    const Scalar *dual_solution(int b) { fetch(); return &y_[(size_t)b * m_]; }
    const Scalar *z(int b) { fetch(); return &z_[(size_t)b * m_]; }  // the reference keeps z as solver state (qp.hpp:224)
    const Info &info(int b) { fetch(); return info_[b]; }
    sqph_solver *handle() { return h_; }
    int n() const { return n_; }
    int m() const { return m_; }

   private:
    template <typename F>
    void call_csr(F fn, const CsrBatch &b, const char *what) {
        push_settings();
        sqph_csr_batch c;
        c.batch = b.batch; c.memspace = b.memspace;
        c.P = b.P; c.q = b.q; c.A_rowptr = b.rowptr; c.A_colind = b.colind; c.A_val = b.val; c.l = b.l; c.u = b.u;
        c.stride_P = b.stride_P; c.stride_q = b.stride_q; c.stride_rowptr = b.stride_rowptr; c.stride_colind = b.stride_colind;
        c.stride_val = b.stride_val; c.stride_l = b.stride_l; c.stride_u = b.stride_u; c.nnz_max = b.nnz_max;
        detail::check(fn(h_, &c), h_, what);
        last_batch_ = b.batch;
        fetched_ = false;
    }
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
    template <typename F>
    void call(F fn, const Batch &b, const char *what) {
        sqph_settings st;
        st.rho = settings_.rho; st.sigma = settings_.sigma; st.alpha = settings_.alpha;
        st.eps_rel = settings_.eps_rel; st.eps_abs = settings_.eps_abs;
        st.max_iter = settings_.max_iter; st.check_termination = settings_.check_termination;
        st.warm_start = settings_.warm_start; st.adaptive_rho = settings_.adaptive_rho;
        st.adaptive_rho_tolerance = settings_.adaptive_rho_tolerance;
        st.adaptive_rho_interval = settings_.adaptive_rho_interval; st.verbose = settings_.verbose;
        detail::check(sqph_set_settings(h_, &st), h_, "sqph_set_settings");
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
    using Settings = QPSolverSettings<Scalar>;
    using Info = QPSolverInfo<Scalar>;

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

    void setup(const QP &qp) {
        if (!impl_ || impl_->n() != qp.n || impl_->m() != qp.m) {
            // a new shape starts a fresh instance, like the resize() cascade of src/qp.cpp:13-29
            delete impl_;
            impl_ = new BatchQPSolver<Scalar>(qp.n, qp.m, 1, device_, flags_);
        }
        push();
        impl_->setup(impl_->packed(1, qp.P, qp.q, qp.A, qp.l, qp.u));
        pull();
    }
    void update_qp(const QP &qp) {
        if (!impl_) return;
        push();
        impl_->update_qp(impl_->packed(1, qp.P, qp.q, qp.A, qp.l, qp.u));
        pull();
    }
    void solve(const QP &qp) {
        if (!impl_) return;  // UNINITIALIZED: solve() returns silently, src/qp.cpp:68-71
        push();
        impl_->solve(impl_->packed(1, qp.P, qp.q, qp.A, qp.l, qp.u));
        pull();
    }

    const std::vector<Scalar> &primal_solution() const { return x_; }
    const std::vector<Scalar> &dual_solution() const { return y_; }
    Settings &settings() { return settings_; }
    const Settings &settings() const { return settings_; }
    Info &info() { return info_; }
    const Info &info() const { return info_; }

    // static QPSolver::constr_type_init(l, u, constr_type), src/qp.cpp:283-294
    static void constr_type_init(int m, const Scalar *l, const Scalar *u, int *constr_type) {
        detail::check(sqph_constr_type_init(detail::dtype_of<Scalar>::value, m, l, u, constr_type), nullptr, "sqph_constr_type_init");
    }

   private:
    void push() { impl_->settings() = settings_; }
    void pull() {
        x_.assign(impl_->primal_solution(0), impl_->primal_solution(0) + impl_->n());
        y_.assign(impl_->dual_solution(0), impl_->dual_solution(0) + impl_->m());
        info_ = impl_->info(0);
    }
    int device_, flags_;
    BatchQPSolver<Scalar> *impl_ = nullptr;
    Settings settings_;
    Info info_;
    std::vector<Scalar> x_, y_;
};

// Fixed-size legacy API, reference include/unsupported/qp_solver.hpp:18-49,135-592.
namespace legacy {
template <int N_, int M_, typename Scalar_ = double>
struct QP {
    using Scalar = Scalar_;
    enum { n = N_, m = M_ };
    Scalar P[N_ * N_];  // column-major
    Scalar q[N_];
    Scalar A[M_ * N_];  // column-major
    Scalar l[M_], u[M_];
};

template <typename QPType>
class QPSolver {
   public:
    enum { n = QPType::n, m = QPType::m };
    using Scalar = typename QPType::Scalar;
    using settings_t = QPSolverSettings<Scalar>;
    using info_t = QPSolverInfo<Scalar>;
    // public state, as in the reference (unsupported/qp_solver.hpp:172-200)
    int iter = 0;
    Scalar x[n], z[m > 0 ? m : 1], y[m > 0 ? m : 1];
    int constr_type[m > 0 ? m : 1];  // INEQUALITY_CONSTRAINT / EQUALITY_CONSTRAINT / LOOSE_BOUNDS
    settings_t _settings;
    info_t _info;

    explicit QPSolver(int device = 0) : impl_(n, m, 1, device, SQPH_FLAG_LEGACY_COLD_START) {}
    void setup(const QPType &qp) { run(&BatchQPSolver<Scalar>::setup, qp); }
    void update_qp(const QPType &qp) { run(&BatchQPSolver<Scalar>::update_qp, qp); }
    void solve(const QPType &qp) { run(&BatchQPSolver<Scalar>::solve, qp); }
    const Scalar *primal_solution() const { return x; }
    const Scalar *dual_solution() const { return y; }
    settings_t &settings() { return _settings; }
    info_t &info() { return _info; }

   private:
    template <typename F>
    void run(F fn, const QPType &qp) {
        impl_.settings() = _settings;
        (impl_.*fn)(impl_.packed(1, qp.P, qp.q, qp.A, qp.l, qp.u));
        for (int i = 0; i < n; i++) x[i] = impl_.primal_solution(0)[i];
        for (int i = 0; i < m; i++) y[i] = impl_.dual_solution(0)[i];
        for (int i = 0; i < m; i++) z[i] = impl_.z(0)[i];
        if (m > 0) sqph_constr_type_init(detail::dtype_of<Scalar>::value, m, qp.l, qp.u, constr_type);
        _info = impl_.info(0);
        if (_info.status == NUMERICAL_ISSUES) _info.status = UNSOLVED;  // the legacy enum has no NUMERICAL_ISSUES (unsupported:84-89)
        iter = _info.iter;
    }
    BatchQPSolver<Scalar> impl_;
};
}  // namespace legacy

}  // namespace qp_solver
