// Drop-in check of the LEGACY fixed-size class: written against the reference's unsupported/qp_solver.hpp API — QP<n,m> with
// Eigen members, QPSolver<QPType>, public x/z/y/iter/constr_type — compiled with `-I include/sqp_hip/compat` and an Eigen on the
// include path (tests/cpp/eigen_stub here).  Cases follow tests/unsupported/qp_solver_test.cpp and
// tests/qp_solver_sparse_test.cpp:68-98 (multiple solve, update_qp) on the dense class.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "unsupported/qp_solver.hpp"

using namespace qp_solver;

#define CHECK(cond)                                                                 \
    do {                                                                            \
        if (!(cond)) {                                                              \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

template <typename S>
struct Simple : QP<2, 3, S> {
    Eigen::Matrix<S, 2, 1> SOLUTION;
    Simple() {
        this->P << 4, 1, 1, 2;
        this->q << 1, 1;
        this->A << 1, 1, 1, 0, 0, 1;
        this->l << 1, 0, 0;
        this->u << 1, (S)0.7, (S)0.7;
        SOLUTION << (S)0.3, (S)0.7;
    }
};

template <typename S>
static void simple_qp() {
    Simple<S> qp;
    QPSolver<Simple<S>> prob;
    prob.settings().max_iter = 1000;
    prob.setup(qp);
    prob.solve(qp);
    Eigen::Matrix<S, 2, 1> sol = prob.primal_solution();
    CHECK(sol.isApprox(qp.SOLUTION, 1e-2));
    CHECK(prob.iter < prob.settings().max_iter);
    CHECK(prob.info().status == SOLVED);
    Eigen::Matrix<S, 3, 1> lower = qp.A * sol - qp.l, upper = qp.A * sol - qp.u;
    CHECK(lower.minCoeff() >= -1e-2 && upper.maxCoeff() <= 1e-2);
}

static void multiple_solve_and_update() {
    Simple<double> qp;
    QPSolver<Simple<double>, Eigen::LDLT, Eigen::Lower> prob;  // the three-argument form compiles (arguments ignored)
    prob.setup(qp);
    prob.solve(qp);
    CHECK(prob.info().status == SOLVED);
    const int it1 = prob.iter;
    prob.solve(qp);  // the legacy class really resets x, z, y when warm_start == false (unsupported/qp_solver.hpp:256-260)
    CHECK(prob.info().status == SOLVED && prob.iter == it1);
    qp.P.setIdentity();
    qp.q << 0, 0;
    qp.SOLUTION << 0.5, 0.5;
    prob.update_qp(qp);
    prob.solve(qp);
    CHECK(prob.primal_solution().isApprox(qp.SOLUTION, 1e-2));
    CHECK(prob.info().status == SOLVED);
    // public state: z is the projected A x, the dual of the equality row is non-zero
    CHECK(std::fabs(prob.z(0) - 1.0) < 1e-2 && prob.z(1) >= -1e-9 && prob.z(2) <= 0.7 + 1e-9);
}

static void constraint_classes_through_setup() {
    using qp_t = QP<5, 5, double>;
    using solver_t = QPSolver<qp_t>;
    solver_t prob;
    qp_t qp;
    qp.P.setIdentity();
    qp.q.setConstant(-1);
    qp.A.setIdentity();
    int expect[5];
    qp.l(0) = -1e+17; qp.u(0) = 1e+17; expect[0] = solver_t::LOOSE_BOUNDS;
    qp.l(1) = -101;   qp.u(1) = 1e+17; expect[1] = solver_t::INEQUALITY_CONSTRAINT;
    qp.l(2) = -1e+17; qp.u(2) = 123;   expect[2] = solver_t::INEQUALITY_CONSTRAINT;
    qp.l(3) = -1;     qp.u(3) = 1;     expect[3] = solver_t::INEQUALITY_CONSTRAINT;
    qp.l(4) = 42;     qp.u(4) = 42;    expect[4] = solver_t::EQUALITY_CONSTRAINT;
    CHECK(prob.info().status == UNINITIALIZED);
    prob.solve(qp);  // before setup: silently returns (unsupported/qp_solver.hpp:246-249)
    CHECK(prob.info().status == UNINITIALIZED && prob.iter == 0);
    prob.setup(qp);
    CHECK(prob.info().status == UNSOLVED);
    for (int i = 0; i < qp.l.rows(); i++) CHECK(prob.constr_type[i] == expect[i]);
}

int main() {
    try {
        simple_qp<double>();
        simple_qp<float>();
        multiple_solve_and_update();
        constraint_classes_through_setup();
    } catch (const std::runtime_error &e) {
        if (std::string(e.what()).find("no HIP device") != std::string::npos) {
            fprintf(stderr, "no HIP device: %s\n", e.what());
            return 3;
        }
        fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    printf("qp_dropin_legacy_test: all passed\n");
    return 0;
}
