// Drop-in check of the LEGACY class's SPARSE variant (QP_SOLVER_USE_SPARSE: Eigen::SparseMatrix P and A members,
// unsupported/qp_solver.hpp:24-25) on the edge the reference's own tests/qp_solver_sparse_test.cpp does not touch: an EMPTY P
// (an LP: P.nonZeros() == 0) and a P without stored diagonal entries.  Compiled with `-I include/sqp_hip/compat` and the
// Eigen stand-in of tests/cpp/eigen_stub.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#define QP_SOLVER_USE_SPARSE
#include "solvers/qp_solver.hpp"

using namespace qp_solver;

#define CHECK(cond)                                                                 \
    do {                                                                            \
        if (!(cond)) {                                                              \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

// min -x0 - x1  s.t.  x0 + x1 <= 1, 0 <= x0 <= 0.7, 0 <= x1 <= 0.7 : any point of the face x0 + x1 = 1; objective -1
struct BoxLP : QP<2, 3, double> {
    BoxLP() {
        Eigen::MatrixXd Pd(2, 2), Ad(3, 2);
        Pd << 0, 0, 0, 0;
        Ad << 1, 1, 1, 0, 0, 1;
        this->P = Pd.sparseView();  // no stored entry at all
        this->q << -1, -1;
        this->A = Ad.sparseView();
        this->l << -1e20, 0, 0;
        this->u << 1, 0.7, 0.7;
    }
};

static void lp_with_empty_P() {
    BoxLP qp;
    CHECK(qp.P.nonZeros() == 0);
    QPSolver<BoxLP> prob;
    prob.settings().max_iter = 4000;
    prob.setup(qp);
    CHECK(prob.info().status == UNSOLVED);
    prob.solve(qp);
    CHECK(prob.info().status == SOLVED);
    Eigen::Matrix<double, 2, 1> x = prob.primal_solution();
    CHECK(std::fabs(x(0) + x(1) - 1.0) < 1e-2);
    CHECK(x(0) >= -1e-2 && x(0) <= 0.7 + 1e-2 && x(1) >= -1e-2 && x(1) <= 0.7 + 1e-2);
    // update_qp() and a second solve on the same (still empty) P
    qp.q << -1, -2;  // now x1 is preferred: the vertex (0.3, 0.7)
    prob.update_qp(qp);
    prob.solve(qp);
    CHECK(prob.info().status == SOLVED);
    x = prob.primal_solution();
    CHECK(std::fabs(x(0) - 0.3) < 2e-2 && std::fabs(x(1) - 0.7) < 2e-2);
}

// P with off-diagonal entries only on a problem the constraints make convex on the feasible set is NOT what this checks:
// the diagonal is simply absent from the storage of a positive semi-definite P = [[0, 0], [0, 2]]
struct NoStoredDiagonal : QP<2, 3, double> {
    NoStoredDiagonal() {
        Eigen::MatrixXd Pd(2, 2), Ad(3, 2);
        Pd << 0, 0, 0, 2;
        Ad << 1, 1, 1, 0, 0, 1;
        this->P = Pd.sparseView();  // one stored entry, column 0 is empty
        this->q << 1, -1;
        this->A = Ad.sparseView();
        this->l << 1, 0, 0;
        this->u << 1, 0.7, 0.7;
    }
};

static void column_without_entries() {
    NoStoredDiagonal qp;
    CHECK(qp.P.nonZeros() == 1);
    QPSolver<NoStoredDiagonal> prob;
    prob.settings().max_iter = 4000;
    prob.setup(qp);
    prob.solve(qp);
    CHECK(prob.info().status == SOLVED);
    // x0 + x1 = 1: minimise x0 - x1 + x1^2 = 1 - 2 x1 + x1^2 over x1 in [0.3, 0.7] -> x1 = 0.7
    Eigen::Matrix<double, 2, 1> x = prob.primal_solution();
    CHECK(std::fabs(x(0) - 0.3) < 1e-2 && std::fabs(x(1) - 0.7) < 1e-2);
}

int main() {
    try {
        lp_with_empty_P();
        column_without_entries();
    } catch (const std::runtime_error &e) {
        if (std::string(e.what()).find("no HIP device") != std::string::npos) {
            fprintf(stderr, "no HIP device: %s\n", e.what());
            return 3;
        }
        fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    printf("qp_dropin_legacy_sparse_test: all passed\n");
    return 0;
}
