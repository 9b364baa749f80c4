// Drop-in check of the SUPPORTED class: this file is written the way a caller of the reference writes —
// `#include "solvers/qp.hpp"`, Eigen objects, `qp.P = &P;` — and is compiled with `-I include/sqp_hip/compat` plus an Eigen on
// the include path (real Eigen where there is one; tests/cpp/eigen_stub here).  Cases follow the reference's
// tests/qp_solver_test.cpp:43-156 and src/sqp.cpp:210-242 (how SQP::run_solve_qp uses the class).
// Exit 0 = all passed, 3 = no HIP device (after the host-only case passed).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "solvers/qp.hpp"

using namespace qp_solver;

#define CHECK(cond)                                                                 \
    do {                                                                            \
        if (!(cond)) {                                                              \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

template <typename Scalar>
struct Fixture {  // owns the Eigen objects; the QuadraticProblem only points at them (qp.hpp:29-33)
    using Matrix = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic>;
    using Vector = Eigen::Matrix<Scalar, Eigen::Dynamic, 1>;
    Matrix P{2, 2}, A{3, 2};
    Vector q{2}, l{3}, u{3};
    Eigen::Matrix<Scalar, 2, 1> SOLUTION;
    QuadraticProblem<Scalar> qp;
    Fixture() {
        P << 4, 1, 1, 2;
        q << 1, 1;
        A << 1, 1, 1, 0, 0, 1;
        l << 1, 0, 0;
        u << 1, (Scalar)0.7, (Scalar)0.7;
        SOLUTION << (Scalar)0.3, (Scalar)0.7;
        qp.P = &P;
        qp.q = &q;
        qp.A = &A;
        qp.l = &l;
        qp.u = &u;
    }
};

static void constraint_classes() {  // static constr_type_init(const Vector&, const Vector&, Eigen::VectorXi&), qp.hpp:173
    using Solver = QPSolver<double>;
    Eigen::VectorXd l(5), u(5);
    const double T = Solver::LOOSE_BOUNDS_THRESH;
    l(0) = -10 * T; u(0) = 10 * T;
    l(1) = -1; u(1) = 10 * T;
    l(2) = -10 * T; u(2) = 2;
    l(3) = -3; u(3) = 4;
    l(4) = 42; u(4) = 42;
    Eigen::VectorXi type(5);
    Solver::constr_type_init(l, u, type);
    const int expect[5] = {Solver::LOOSE_BOUNDS, Solver::INEQUALITY_CONSTRAINT, Solver::INEQUALITY_CONSTRAINT, Solver::INEQUALITY_CONSTRAINT,
                           Solver::EQUALITY_CONSTRAINT};
    for (int i = 0; i < l.rows(); i++) CHECK(type[i] == expect[i]);
}

template <typename Scalar>
static void simple_qp() {
    Fixture<Scalar> f;
    QPSolver<Scalar> solver;
    solver.settings().max_iter = 1000;
    solver.setup(f.qp);
    solver.solve(f.qp);
    Eigen::Matrix<Scalar, 2, 1> sol = solver.primal_solution();  // dynamic -> fixed, as the reference's tests do
    CHECK(sol.isApprox(f.SOLUTION, 1e-2));
    CHECK(solver.info().iter < solver.settings().max_iter);
    CHECK(solver.info().status == SOLVED);
    // feasibility band with Eigen expressions on the borrowed matrices
    Eigen::Matrix<Scalar, 3, 1> lower = (*f.qp.A) * sol - (*f.qp.l);
    Eigen::Matrix<Scalar, 3, 1> upper = (*f.qp.A) * sol - (*f.qp.u);
    CHECK(lower.minCoeff() >= -1e-2 && upper.maxCoeff() <= 1e-2);
}

static void adaptive_rho_and_resolve() {
    Fixture<double> f;
    QPSolver<double> solver;
    solver.settings().warm_start = false;
    solver.settings().max_iter = 1000;
    solver.settings().rho = 0.1;
    solver.settings().adaptive_rho = false;
    solver.setup(f.qp);
    solver.solve(f.qp);
    const int prev_iter = solver.info().iter;
    CHECK(prev_iter == 125);  // oracle-pinned
    solver.settings().adaptive_rho = true;
    solver.settings().adaptive_rho_interval = 10;
    solver.solve(f.qp);
    auto info = solver.info();
    CHECK(info.iter < solver.settings().max_iter && info.iter < prev_iter && info.status == SOLVED);
}

static void run_solve_qp_pattern() {
    // the reference's only production caller (src/sqp.cpp:210-242): a stack QuadraticProblem of five pointers, setup, solve,
    // info().iter accumulated, NUMERICAL_ISSUES checked, solutions copied into Eigen vectors
    Fixture<double> f;
    QPSolver<double> qp_solver_;
    Eigen::VectorXd prim(2), dual(3);
    int qp_solver_iter = 0;
    for (int outer = 0; outer < 3; outer++) {
        QuadraticProblem<double> qp;
        qp.P = &f.P; qp.q = &f.q; qp.A = &f.A; qp.l = &f.l; qp.u = &f.u;
        qp_solver_.setup(qp);
        qp_solver_.solve(qp);
        qp_solver_iter += qp_solver_.info().iter;
        CHECK(qp_solver_.info().status != NUMERICAL_ISSUES);
        prim = qp_solver_.primal_solution();
        dual = qp_solver_.dual_solution();
        f.q(0) += 0.05;  // the next "linearisation"
    }
    CHECK(qp_solver_iter > 0 && prim.rows() == 2 && dual.rows() == 3);
    CHECK(std::fabs(prim(0) + prim(1) - 1.0) < 1e-2);  // the equality row
    CHECK(prim.norm() > 0);
}

static void writable_iterates() {
    // primal_solution()/dual_solution() are non-const references to solver state in the reference (qp.hpp:160-164): what the
    // caller writes there is what the next solve() starts from.  Two solvers, one started from an injected point.
    Fixture<double> f;
    QPSolver<double> a, b;
    for (QPSolver<double> *s : {&a, &b}) {
        s->settings().max_iter = 30;
        s->settings().check_termination = 0;
        s->setup(f.qp);
    }
    a.solve(f.qp);  // from zero
    b.primal_solution()(0) = 0.25;
    b.primal_solution()(1) = 0.75;
    b.dual_solution()(0) = -2.0;
    b.solve(f.qp);  // from the injected x, y
    const Eigen::VectorXd xa = a.primal_solution(), xb = b.primal_solution();
    CHECK(!xa.isApprox(xb, 1e-12));                         // the injection reached the device ...
    CHECK((xa - xb).norm() < 0.2 && xb.isApprox(f.SOLUTION, 0.2));  // ... and both runs head for the same solution
}

static void indefinite_P_is_numerical_issues() {
    // DOCUMENTED DIFFERENCE (sqp_hip.h): the device factors S = P + sigma I + A'RA and needs it positive definite; the
    // reference's pivoted LDL' of the KKT matrix accepts an indefinite P and iterates on the non-convex problem.
    Fixture<double> f;
    f.P << 1, 0, 0, -50;  // S(1,1) = -50 + sigma + rho-weighted column norm < 0
    QPSolver<double> solver;
    solver.setup(f.qp);
    CHECK(solver.info().status == NUMERICAL_ISSUES);
    solver.solve(f.qp);  // no-op on NUMERICAL_ISSUES, src/qp.cpp:68-71
    CHECK(solver.info().status == NUMERICAL_ISSUES && solver.info().iter == 0);
}

int main() {
    try {
        constraint_classes();  // host-only
        simple_qp<double>();
        simple_qp<float>();
        adaptive_rho_and_resolve();
        run_solve_qp_pattern();
        writable_iterates();
        indefinite_P_is_numerical_issues();
    } catch (const std::runtime_error &e) {
        if (std::string(e.what()).find("no HIP device") != std::string::npos) {
            fprintf(stderr, "no HIP device: %s\n", e.what());
            return 3;
        }
        fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    printf("qp_dropin_test: all passed\n");
    return 0;
}
