// Kernel argument block shared by every ADMM kernel variant (host + device).
#pragma once
#include "../../include/sqp_hip.h"

namespace sqph {

enum : int {
    MODE_SETUP = 1,       // QPSolver::setup   (src/qp.cpp:11-44): zero x,z,y; classify; rho; factor
    MODE_UPDATE = 2,      // QPSolver::update_qp (src/qp.cpp:46-62): classify; rho; factor; keep x,z,y
    MODE_SOLVE = 4,       // QPSolver::solve   (src/qp.cpp:64-157)
    MODE_COLD_RESET = 8,  // legacy class: solve() zeroes x,z,y when !warm_start (unsupported/qp_solver.hpp:256-260)
    // factor residency (host policy, capi.hip): a fused setup+solve does not write its factor to the workspace unless the
    // solver was created with SQPH_FLAG_KEEP_FACTOR; a later solve() on such an instance rebuilds it first (same arithmetic,
    // same results — the factor depends on P, A, sigma and the current rho vector only)
    MODE_NO_FACTOR_STORE = 16,
    MODE_REFACTOR = 32,
    // with MODE_SETUP / MODE_UPDATE: P and A are those of the resident factor (sqph_setup_solve_reuse: the SQP second-order
    // correction re-solves with new bounds only, src/sqp.cpp:244-276 and the TODO at :273) — a QP whose freshly classified rho
    // vector equals the one the resident factor was built with skips the factorisation.  A hint: kernels without support for
    // it factor as usual (same results either way).
    MODE_SAME_MATRICES = 64,
};

// T   = arithmetic / state type (always double in the shipped library)
// TIN = type of the caller's problem arrays (double for QPSolver<double>, float for QPSolver<float>)
template <typename T, typename TIN = T>
struct KArgs {
    int n, m, batch, mode;
    // borrowed problem data (device pointers), per-QP column-major, element strides (0 = shared)
    const TIN *P, *q, *A, *l, *u;
    long long sP, sq, sA, sl, su;
    // persistent per-QP solver state, QP-major
    T *x;         // [batch][n]
    T *z;         // [batch][m]
    T *y;         // [batch][m]
    T *rho_vec;   // [batch][m]
    int *ctype;   // [batch][m]
    T *rho;       // [batch]      current scalar rho (QPSolver::rho, qp.hpp:228)
    sqph_info *info;  // [batch]
    // factor workspace
    T *Sinv;      // [batch][2*n*n] factor W of S = P + sigma I + A' diag(rho) A, S^-1 = W'W (tiled kernels use n*n)
    T *At;        // [batch][m*n]  row-major copy of A (generic kernel only; may be null for tiled kernels)
    // settings converted to Scalar
    T rho0, sigma, alpha, eps_rel, eps_abs, rho_tol;
    // class constants (qp.hpp:136-141), rounded through Scalar on the host
    T rho_min, rho_max, eq_tol, rho_eq_factor, loose_thresh, regul;
    int max_iter, check_termination, warm_start, adaptive_rho, adaptive_rho_interval;
    // verbose trace of ONE QP (settings.verbose; reference print_status, src/qp.cpp:373-383): at every termination check the
    // kernel appends {iter, objective 0.5 x'Px + q'x, res_prim, res_dual} to trace[1 + 4 k ..], k = trace[0]++ (< trace_cap).
    // Null = off.  Only the generic and the one-QP-per-lane kernels record; the host routes verbose calls to them.
    double *trace;
    int trace_qp, trace_cap;
#ifdef SQPH_EXPERIMENTS
    // experiment builds only (tools/slim_build.sh -DSQPH_EXPERIMENTS): knobs from the environment (SQPH_XP0..7) and a
    // per-workgroup debug record buffer; never part of the shipped library
    int xp[8];
    unsigned long long *xdbg;
#endif
};

}  // namespace sqph
