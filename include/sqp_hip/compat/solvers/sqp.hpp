// Drop-in for the reference's include/solvers/sqp.hpp (`-I <repo>/include/sqp_hip/compat`): sqp::SQP<Scalar>, sqp::NonLinearProblem,
// sqp::sqp_settings_t, sqp::Info with the reference's names, members and Eigen types (sqp.hpp:13-38, 62-140), so that callers of
// SQP::solve() compile unchanged.  The outer loop is the batched driver of include/sqp_hip/sqp.hpp run with ONE instance (the
// same statements as src/sqp.cpp; its QP subproblems go to the GPU through libsqp_hip); many instances at once: sqp::raw::BatchSQP.
//
// The declarations of sqp_settings_t, Info, NonLinearProblem and SQP's public members (names, defaults and their doc comments) restate
// the interface of msplr/sqp_solver's include/solvers/sqp.hpp — Copyright (c) 2019 Michael Spieler, MIT License — because source
// compatibility with its callers is the contract of this header; the implementation behind them is this repository's own.
#pragma once
#define SQP_HIP_SQP_DROPIN 1
#include <cstdio>
#include <functional>
#include <limits>
#include <memory>
#include <vector>

#include <Eigen/Dense>

#include "qp.hpp"  // <solvers/qp.hpp>, as the reference's sqp.hpp includes it
#include "../../sqp.hpp"

namespace sqp {

template <typename T>
class SQP;

template <typename Scalar>
struct sqp_settings_t {  // sqp.hpp:13-31
    Scalar tau = 0.5;       /**< line search iteration decrease, 0 < tau < 1 */
    Scalar eta = 0.25;      /**< line search parameter, 0 < eta < 1 */
    Scalar rho = 0.5;       /**< line search parameter, 0 < rho < 1 */
    Scalar eps_prim = 1e-4; /**< primal step termination threshold, eps_prim > 0 */
    Scalar eps_dual = 1e-4; /**< dual step termination threshold, eps_dual > 0 */
    int max_iter = 100;
    int line_search_max_iter = 20;
    bool second_order_correction = false;
    std::function<void(SQP<Scalar> &)> iteration_callback;

    bool validate() {  // (the reference's test, bug included: eps_* < 0.0 — sqp.hpp:25-30)
        return 0.0 < tau && tau < 1.0 && 0.0 < eta && eta < 1.0 && 0.0 < rho && rho < 1.0 && eps_prim < 0.0 && eps_dual < 0.0 &&
               max_iter > 0 && line_search_max_iter > 0;
    }
};

typedef enum { SOLVED, MAX_ITER_EXCEEDED, INVALID_SETTINGS } Status;

struct Info {  // sqp.hpp:35-60
    int iter;
    int qp_solver_iter;
    Status status;

    void print() {
        printf("SQP info:\n");
        printf("  iter: %d\n", iter);
        printf("  qp_solver_iter: %d\n", qp_solver_iter);
        printf("  status: ");
        switch (status) {
            case SOLVED: printf("SOLVED\n"); break;
            case MAX_ITER_EXCEEDED: printf("MAX_ITER_EXCEEDED\n"); break;
            case INVALID_SETTINGS: printf("INVALID_SETTINGS\n"); break;
            default: printf("UNKNOWN\n"); break;
        }
    }
};

template <typename Scalar_ = double>
struct NonLinearProblem {  // sqp.hpp:62-76
    using Scalar = Scalar_;
    using Matrix = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic>;
    using Vector = Eigen::Matrix<Scalar, Eigen::Dynamic, 1>;

    int num_var;
    int num_constr;

    virtual void objective(const Vector &x, Scalar &obj) = 0;
    virtual void objective_linearized(const Vector &x, Vector &grad, Scalar &obj) = 0;
    virtual void constraint(const Vector &x, Vector &c, Vector &l, Vector &u) = 0;
    virtual void constraint_linearized(const Vector &x, Matrix &Jc, Vector &c, Vector &l, Vector &u) = 0;
};

template <typename Scalar_>
class SQP {
   public:
    using Scalar = Scalar_;
    using Matrix = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic>;
    using Vector = Eigen::Matrix<Scalar, Eigen::Dynamic, 1>;
    using Problem = NonLinearProblem<Scalar>;
    using Settings = sqp_settings_t<Scalar>;

    static constexpr Scalar DIV_BY_ZERO_REGUL = std::numeric_limits<Scalar>::epsilon();
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW

    explicit SQP(int device = 0) : device_(device) {
        info_.iter = 0; info_.qp_solver_iter = 0; info_.status = MAX_ITER_EXCEEDED;
        // TODO(mi) of the reference: "Performance strongly depends on QP solver settings" — its constructor's values, src/sqp.cpp:15-23
        qp_settings_.warm_start = true;
        qp_settings_.check_termination = 10;
        qp_settings_.eps_abs = 1e-4;
        qp_settings_.eps_rel = 1e-4;
        qp_settings_.max_iter = 100;
        qp_settings_.adaptive_rho = true;
        qp_settings_.adaptive_rho_interval = 50;
        qp_settings_.alpha = 1.6;
    }
    ~SQP() = default;

    void solve(Problem &prob, const Vector &x0, const Vector &lambda0) {  // src/sqp.cpp:26-31
        x_ = x0;
        lambda_ = lambda0;
        run_solve(prob);
    }
    void solve(Problem &prob) {  // src/sqp.cpp:33-41
        x_ = Vector::Zero(prob.num_var);
        lambda_ = Vector::Zero(prob.num_constr);
        run_solve(prob);
    }

    inline const Vector &primal_solution() const { return x_; }
    inline Vector &primal_solution() { return x_; }
    inline const Vector &dual_solution() const { return lambda_; }
    inline Vector &dual_solution() { return lambda_; }
    inline const Settings &settings() const { return settings_; }
    inline Settings &settings() { return settings_; }
    inline const Info &info() const { return info_; }
    inline Info &info() { return info_; }
    // the settings of the QP subproblem solver (the reference exposes them as qp_solver_.settings() of its public member); they
    // take effect at the next solve()
    qp_solver::QPSolverSettings<Scalar> &qp_settings() { return qp_settings_; }

    // Solver state variables (public in the reference: "// private:" is commented out, sqp.hpp:104; the iteration callback reads them)
    Vector x_;
    Vector lambda_;

    void run_solve(Problem &prob) {  // src/sqp.cpp:43-101 through the batched driver, one instance
        const int n = prob.num_var, m = prob.num_constr;
        if (!drv_ || n != n_ || m != m_) {
            drv_.reset(new raw::BatchSQP<Scalar>(n, m, 1, device_));
            n_ = n;
            m_ = m;
        }
        auto &ds = drv_->settings();
        ds.tau = settings_.tau; ds.eta = settings_.eta; ds.rho = settings_.rho; ds.eps_prim = settings_.eps_prim;
        ds.eps_dual = settings_.eps_dual; ds.max_iter = settings_.max_iter; ds.line_search_max_iter = settings_.line_search_max_iter;
        ds.second_order_correction = settings_.second_order_correction;
        drv_->qp_settings() = qp_settings_;
        Adaptor ad(prob);
        std::vector<raw::NonLinearProblem<Scalar> *> probs(1, &ad);
        info_.qp_solver_iter = 0;
        if (settings_.iteration_callback) settings_.iteration_callback(*this);  // src/sqp.cpp:65-67
        drv_->set_step_callback(settings_.iteration_callback ? &SQP::on_step : nullptr, this);
        x_.resize(n);        // (callers may hand over fixed-size vectors of the right length)
        lambda_.resize(m);
        drv_->solve(probs, x_.data(), m > 0 ? lambda_.data() : nullptr);
        for (int i = 0; i < n; i++) x_[i] = drv_->primal_solution(0)[i];
        for (int i = 0; i < m; i++) lambda_[i] = drv_->dual_solution(0)[i];
        const raw::Info &bi = drv_->info(0);
        info_.iter = bi.iter;
        info_.qp_solver_iter = bi.qp_solver_iter;
        info_.status = bi.status == raw::SOLVED ? SOLVED : bi.status == raw::MAX_ITER_EXCEEDED ? MAX_ITER_EXCEEDED : INVALID_SETTINGS;
    }

   private:
    // the reference's Eigen-typed problem behind the raw-pointer interface of the batched driver (Jacobian column-major = Eigen's)
    struct Adaptor : raw::NonLinearProblem<Scalar> {
        Problem &p;
        Vector x, g, c, l, u;
        Matrix J;
        explicit Adaptor(Problem &prob) : p(prob), x(prob.num_var), g(prob.num_var), c(prob.num_constr), l(prob.num_constr), u(prob.num_constr),
                                          J(prob.num_constr, prob.num_var) {
            this->num_var = prob.num_var;
            this->num_constr = prob.num_constr;
        }
        void in(const Scalar *xs) { for (int i = 0; i < this->num_var; i++) x[i] = xs[i]; }
        void out(Scalar *cs, Scalar *ls, Scalar *us) {
            for (int i = 0; i < this->num_constr; i++) { cs[i] = c[i]; ls[i] = l[i]; us[i] = u[i]; }
        }
        void objective(const Scalar *xs, Scalar &obj) override { in(xs); p.objective(x, obj); }
        void objective_linearized(const Scalar *xs, Scalar *grad, Scalar &obj) override {
            in(xs);
            p.objective_linearized(x, g, obj);
            for (int i = 0; i < this->num_var; i++) grad[i] = g[i];
        }
        void constraint(const Scalar *xs, Scalar *cs, Scalar *ls, Scalar *us) override { in(xs); p.constraint(x, c, l, u); out(cs, ls, us); }
        void constraint_linearized(const Scalar *xs, Scalar *Jc, Scalar *cs, Scalar *ls, Scalar *us) override {
            in(xs);
            p.constraint_linearized(x, J, c, l, u);
            out(cs, ls, us);
            for (int j = 0; j < this->num_var; j++)
                for (int i = 0; i < this->num_constr; i++) Jc[(size_t)j * this->num_constr + i] = J(i, j);
        }
    };
    static void on_step(void *user, int, int iter, const Scalar *x, const Scalar *lambda) {  // src/sqp.cpp:88-90
        SQP *self = static_cast<SQP *>(user);
        for (int i = 0; i < self->n_; i++) self->x_[i] = x[i];
        for (int i = 0; i < self->m_; i++) self->lambda_[i] = lambda[i];
        self->info_.iter = iter;
        self->settings_.iteration_callback(*self);
    }
    int device_;
    int n_ = -1, m_ = -1;
    std::unique_ptr<raw::BatchSQP<Scalar>> drv_;
    Settings settings_;
    qp_solver::QPSolverSettings<Scalar> qp_settings_;
    Info info_;
};

}  // namespace sqp
