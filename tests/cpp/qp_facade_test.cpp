// The reference's GTest cases for the QP solver (tests/qp_solver_test.cpp:43-156 and
// tests/unsupported/qp_solver_test.cpp) restated against the C++ facade include/sqp_hip/qp.hpp,
// i.e. host C++ -> C-ABI -> HIP kernels.  Plain asserts; exit code 0 = all passed, 3 = no HIP device.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sqp_hip/qp.hpp"

using namespace qp_solver;

#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            exit(1);                                                         \
        }                                                                    \
    } while (0)

// the canonical fixture, tests/qp_solver_test.cpp:19-31 (column-major storage like Eigen)
template <typename Scalar>
struct SimpleQP : QuadraticProblem<Scalar> {
    Scalar Pd[4] = {4, 1, 1, 2};
    Scalar qd[2] = {1, 1};
    Scalar Ad[6] = {1, 1, 0, 1, 0, 1};  // A = [[1,1],[1,0],[0,1]] column-major
    Scalar ld[3] = {1, 0, 0};
    Scalar ud[3] = {1, (Scalar)0.7, (Scalar)0.7};
    Scalar SOLUTION[2] = {(Scalar)0.3, (Scalar)0.7};
    SimpleQP() {
        this->n = 2; this->m = 3;
        this->P = Pd; this->q = qd; this->A = Ad; this->l = ld; this->u = ud;
    }
};

template <typename V, typename Scalar>
static bool is_approx(const V &a, const Scalar *b, int k, double prec) {  // Eigen's isApprox
    double d = 0, na = 0, nb = 0;
    for (int i = 0; i < k; i++) { d += (a[i] - b[i]) * (a[i] - b[i]); na += a[i] * a[i]; nb += b[i] * b[i]; }
    return std::sqrt(d) <= prec * std::sqrt(na < nb ? na : nb);
}

static void testSimpleQP() {
    SimpleQP<double> qp;
    QPSolver<double> solver;
    solver.settings().max_iter = 1000;
    solver.setup(qp);
    solver.solve(qp);
    CHECK(is_approx(solver.primal_solution(), qp.SOLUTION, 2, 1e-2));
    CHECK(solver.info().iter < solver.settings().max_iter);
    CHECK(solver.info().status == SOLVED);
    CHECK(solver.info().iter == 125);  // oracle-pinned
}
static void testSolveWithAnotherAIsRejected() {  // src/qp.cpp:319-331: the reference would iterate on setup()'s A; here the call is refused
    SimpleQP<double> qp;
    QPSolver<double> solver;
    solver.setup(qp);
    solver.solve(qp);
    qp.ld[1] = -0.1;  // q, l, u may change between setup() and solve() (src/qp.cpp:89,100)
    solver.solve(qp);
    CHECK(solver.info().status == SOLVED);
    qp.Ad[3] = 2.0;
    bool thrown = false;
    try {
        solver.solve(qp);
    } catch (const std::invalid_argument &) {
        thrown = true;
    }
    CHECK(thrown);
    solver.update_qp(qp);  // the call that takes a new A
    solver.solve(qp);
    CHECK(solver.info().status == SOLVED);
}
static void testVerboseTrace() {  // QP_SOLVER_PRINTING: settings, "iter obj rp rd" table at every check, info (src/qp.cpp:72-76,113-117,152-156)
    SimpleQP<double> qp;
    QPSolver<double> solver;
    solver.settings().verbose = true;
    solver.setup(qp);
    solver.solve(qp);  // prints to stdout; the test driver greps for the table
    CHECK(solver.info().status == SOLVED && solver.info().iter == 125);
}
static void testSinglePrecisionFloat() {
    SimpleQP<float> qp;
    QPSolver<float> solver;
    solver.setup(qp);
    solver.solve(qp);
    CHECK(is_approx(solver.primal_solution(), qp.SOLUTION, 2, 1e-2));
    CHECK(solver.info().iter < solver.settings().max_iter);
    CHECK(solver.info().status == SOLVED);
}
static void testConstraintViolation() {
    SimpleQP<double> qp;
    QPSolver<double> solver;
    solver.settings().eps_rel = 1e-4f;
    solver.settings().eps_abs = 1e-4f;
    solver.setup(qp);
    solver.solve(qp);
    const auto &x = solver.primal_solution();
    const double Ax[3] = {x[0] + x[1], x[0], x[1]};
    for (int i = 0; i < 3; i++) {
        CHECK(Ax[i] - qp.ld[i] >= -1e-3);
        CHECK(Ax[i] - qp.ud[i] <= 1e-3);
    }
}
static void testAdaptiveRho() {
    SimpleQP<double> qp;
    QPSolver<double> solver;
    solver.settings().adaptive_rho = true;
    solver.settings().adaptive_rho_interval = 10;
    solver.setup(qp);
    solver.solve(qp);
    CHECK(solver.info().status == SOLVED);
}
static void testAdaptiveRhoImprovesConvergence() {
    SimpleQP<double> qp;
    QPSolver<double> solver;
    solver.settings().warm_start = false;
    solver.settings().max_iter = 1000;
    solver.settings().rho = 0.1;
    solver.settings().adaptive_rho = false;
    solver.setup(qp);
    solver.solve(qp);
    const int prev_iter = solver.info().iter;
    solver.settings().adaptive_rho = true;
    solver.settings().adaptive_rho_interval = 10;
    solver.solve(qp);
    CHECK(solver.info().iter < solver.settings().max_iter);
    CHECK(solver.info().iter < prev_iter);
    CHECK(solver.info().status == SOLVED);
}
static void TestConstraint() {
    using Solver = QPSolver<double>;
    const double T = Solver::LOOSE_BOUNDS_THRESH;
    const double l[5] = {-10 * T, -1, -10 * T, -3, 42};
    const double u[5] = {10 * T, 10 * T, 2, 4, 42};
    const int expect[5] = {Solver::LOOSE_BOUNDS, Solver::INEQUALITY_CONSTRAINT, Solver::INEQUALITY_CONSTRAINT,
                           Solver::INEQUALITY_CONSTRAINT, Solver::EQUALITY_CONSTRAINT};
    int type[5];
    Solver::constr_type_init(5, l, u, type);
    for (int i = 0; i < 5; i++) CHECK(type[i] == expect[i]);
}
static void testLegacyFixedSize() {  // tests/unsupported/qp_solver_test.cpp:29-41 + sparse test's multiple-solve semantics
    legacy::QP<2, 3, double> qp = {{4, 1, 1, 2}, {1, 1}, {1, 1, 0, 1, 0, 1}, {1, 0, 0}, {1, 0.7, 0.7}};
    legacy::QPSolver<legacy::QP<2, 3, double>> prob;
    prob.settings().max_iter = 1000;
    prob.setup(qp);
    prob.solve(qp);
    const double sol[2] = {0.3, 0.7};
    CHECK(is_approx(prob.primal_solution(), sol, 2, 1e-2));
    CHECK(prob.iter < prob.settings().max_iter);
    CHECK(prob.info().status == SOLVED);
    const int it1 = prob.iter;
    prob.solve(qp);  // legacy class resets x,z,y when warm_start == false
    CHECK(prob.info().status == SOLVED && prob.iter == it1);
    // public state of the legacy class: z and the constraint classes
    CHECK(std::fabs(prob.z[0] - 1.0) < 1e-2 && prob.z[1] >= -1e-9 && prob.z[2] <= 0.7 + 1e-9);
    CHECK(prob.constr_type[0] == SQPH_EQUALITY_CONSTRAINT && prob.constr_type[1] == SQPH_INEQUALITY_CONSTRAINT);
}
static void testBatch() {
    const int B = 256;
    BatchQPSolver<double> solver(2, 3, B);
    SimpleQP<double> qp;
    std::vector<double> q(2 * B), l(3 * B), u(3 * B);
    for (int b = 0; b < B; b++) {
        q[2 * b] = 1 + 0.001 * b; q[2 * b + 1] = 1;
        for (int i = 0; i < 3; i++) { l[3 * b + i] = qp.ld[i]; u[3 * b + i] = qp.ud[i]; }
    }
    auto batch = solver.packed(B, qp.Pd, q.data(), qp.Ad, l.data(), u.data());
    batch.stride_P = 0;  // P and A shared by the whole batch
    batch.stride_A = 0;
    solver.setup_solve(batch);
    for (int b = 0; b < B; b++) {
        CHECK(solver.info(b).status == SOLVED);
        CHECK(std::fabs(solver.primal_solution(b)[0] + solver.primal_solution(b)[1] - 1.0) < 5e-3);
    }
}

static void testSparseCsr() {  // tests/qp_solver_sparse_test.cpp:34-98 through the CSR entry points (legacy semantics)
    SimpleQP<double> qp;
    const int rowptr[4] = {0, 2, 3, 4}, colind[4] = {0, 1, 0, 1};
    const double val[4] = {1, 1, 1, 1};
    BatchQPSolver<double> prob(2, 3, 1, 0, SQPH_FLAG_LEGACY_COLD_START);
    prob.settings().max_iter = 1000;
    prob.settings().adaptive_rho = true;
    auto b = prob.packed_csr(1, qp.Pd, qp.qd, rowptr, colind, val, 4, qp.ld, qp.ud);
    prob.setup_csr(b);
    prob.solve_csr(b);
    CHECK(is_approx(prob.primal_solution(0), qp.SOLUTION, 2, 1e-2));
    CHECK(prob.info(0).iter < prob.settings().max_iter && prob.info(0).status == SOLVED);
    prob.solve_csr(b);  // testCanMultipleSolve
    CHECK(prob.info(0).status == SOLVED);
    const double Pid[4] = {1, 0, 0, 1}, q0[2] = {0, 0}, sol2[2] = {0.5, 0.5};  // testCanUpdateQP
    auto b2 = prob.packed_csr(1, Pid, q0, rowptr, colind, val, 4, qp.ld, qp.ud);
    prob.update_qp_csr(b2);
    prob.solve_csr(b2);
    CHECK(is_approx(prob.primal_solution(0), sol2, 2, 1e-2) && prob.info(0).status == SOLVED);
    // P sparse as well (sqph_*_csr_sp): the same problem with P = [[4, 1], [1, 2]] in compressed columns gives the same bits
    const int pcol[3] = {0, 2, 4}, prow[4] = {0, 1, 0, 1};
    const double pval[4] = {qp.Pd[0], qp.Pd[1], qp.Pd[2], qp.Pd[3]};
    BatchQPSolver<double> dense(2, 3, 1, 0, SQPH_FLAG_LEGACY_COLD_START), sparse(2, 3, 1, 0, SQPH_FLAG_LEGACY_COLD_START);
    dense.settings().max_iter = sparse.settings().max_iter = 1000;
    dense.setup_csr(b);
    dense.solve_csr(b);
    auto bs = sparse.packed_csr(1, nullptr, qp.qd, rowptr, colind, val, 4, qp.ld, qp.ud);
    const auto sp = sparse.packed_csc_P(pcol, prow, pval, 4);
    sparse.setup_csr(bs, sp);
    sparse.solve_csr(bs, sp);
    CHECK(sparse.info(0).status == SOLVED && sparse.info(0).iter == dense.info(0).iter);
    CHECK(sparse.primal_solution(0)[0] == dense.primal_solution(0)[0] && sparse.primal_solution(0)[1] == dense.primal_solution(0)[1]);
    const int bad_row[4] = {0, 2, 0, 1};  // row index out of range: rejected, the message names it
    bool threw = false;
    try {
        sparse.setup_csr(bs, sparse.packed_csc_P(pcol, bad_row, pval, 4));
    } catch (const std::exception &e) {
        threw = std::string(e.what()).find("row index out of range") != std::string::npos;
    }
    CHECK(threw);
}

// distinct handles may be driven from distinct host threads.  Two threads, each with a handle of its own (own
// stream, own state), solve different batches 40 times concurrently; every result must equal the same solve done alone.  The
// process-wide error channel is per thread: a failing sqph_create on each thread reports that thread's arguments.
static void twoThreadWorker(int id, std::vector<double> *out, std::string *err) {
    const int B = 192 + 64 * id;
    BatchQPSolver<double> solver(2, 3, B);
    SimpleQP<double> qp;
    std::vector<double> q(2 * B), l(3 * B), u(3 * B);
    for (int b = 0; b < B; b++) {
        q[2 * b] = 1 + 0.001 * b + 0.5 * id; q[2 * b + 1] = 1 - 0.25 * id;
        for (int i = 0; i < 3; i++) { l[3 * b + i] = qp.ld[i]; u[3 * b + i] = qp.ud[i]; }
    }
    auto batch = solver.packed(B, qp.Pd, q.data(), qp.Ad, l.data(), u.data());
    batch.stride_P = 0;
    batch.stride_A = 0;
    for (int rep = 0; rep < 40; rep++) {
        solver.setup_solve(batch);
        sqph_solver *bad = nullptr;
        CHECK(sqph_create(&bad, 0, -(7 + id), 3, 1, SQPH_F64, 0) != SQPH_OK && bad == nullptr);  // (writes the thread's error string)
    }
    *err = sqph_global_error();
    out->resize(2 * B);
    for (int b = 0; b < B; b++) { (*out)[2 * b] = solver.primal_solution(b)[0]; (*out)[2 * b + 1] = solver.primal_solution(b)[1]; }
}
static void testTwoHostThreadsTwoHandles() {
    std::vector<double> alone[2], together[2];
    std::string e_alone[2], e_together[2];
    for (int id = 0; id < 2; id++) twoThreadWorker(id, &alone[id], &e_alone[id]);
    std::thread t0(twoThreadWorker, 0, &together[0], &e_together[0]), t1(twoThreadWorker, 1, &together[1], &e_together[1]);
    t0.join();
    t1.join();
    for (int id = 0; id < 2; id++) {
        CHECK(alone[id].size() == together[id].size() && !alone[id].empty());
        for (size_t k = 0; k < alone[id].size(); k++) CHECK(alone[id][k] == together[id][k]);  // bit-identical
        CHECK(e_together[id] == e_alone[id]);
        CHECK(e_together[id].find(id == 0 ? "n=-7" : "n=-8") != std::string::npos);
    }
}

int main() {
    try {
        TestConstraint();  // host-only, no device needed
        testSimpleQP();
        testSolveWithAnotherAIsRejected();
        testVerboseTrace();
        testSinglePrecisionFloat();
        testConstraintViolation();
        testAdaptiveRho();
        testAdaptiveRhoImprovesConvergence();
        testLegacyFixedSize();
        testBatch();
        testSparseCsr();
        testTwoHostThreadsTwoHandles();
    } catch (const std::runtime_error &e) {
        if (std::string(e.what()).find("no HIP device") != std::string::npos) {
            fprintf(stderr, "no HIP device: %s\n", e.what());
            return 3;
        }
        fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    printf("qp_facade_test: all passed\n");
    return 0;
}
