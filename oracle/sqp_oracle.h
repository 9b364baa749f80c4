/* TEST INFRASTRUCTURE — NOT PRODUCT CODE. See sqp_oracle.c. */
#ifndef SQP_ORACLE_H
#define SQP_ORACLE_H
#include "qp_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

enum { SQPO_SOLVED = 0, SQPO_MAX_ITER_EXCEEDED = 1, SQPO_INVALID_SETTINGS = 2 }; /* sqp.hpp:33 */

typedef struct sqpo_settings { /* sqp_settings_t, sqp.hpp:13-23 */
    double tau, eta, rho, eps_prim, eps_dual;
    int max_iter, line_search_max_iter, second_order_correction;
} sqpo_settings;

typedef struct sqpo_info { /* sqp::Info, sqp.hpp:35-38 */
    int iter, qp_solver_iter, status;
} sqpo_info;

/* NonLinearProblem<double>, sqp.hpp:62-76, as C callbacks; Jc is num_constr x num_var column-major */
typedef struct sqpo_problem {
    int num_var, num_constr;
    void *user;
    void (*objective)(void *user, const double *x, double *obj);
    void (*objective_linearized)(void *user, const double *x, double *grad, double *obj);
    void (*constraint)(void *user, const double *x, double *c, double *l, double *u);
    void (*constraint_linearized)(void *user, const double *x, double *Jc, double *c, double *l, double *u);
} sqpo_problem;

void sqpo_default_settings(sqpo_settings *s);
/* trajectory record, once per outer iteration after the line search (p_lambda is the dual STEP, sqp.cpp:79); NULL = off.
 * Process-global (test infrastructure, single-threaded use). */
typedef void (*sqpo_trace_fn)(void *user, int iter, const double *p, const double *p_lambda, double alpha, int qp_iter);
void sqpo_set_trace(sqpo_trace_fn f, void *user);
/* 1: the QP subproblems are solved by the x87 extended-precision instance of the QP oracle (a yard-stick for "which outcomes of
 * the outer loop are decided by the rounding of a QP iterate"); 0 (default): the double instance, as the reference. */
void sqpo_set_qp_extended(int on);
/* BFGS_update, bfgs.hpp:14-41, on a column-major n x n matrix (test access) */
void sqpo_bfgs_update(double *B, int n, const double *s, const double *y);
void sqpo_solve(const sqpo_problem *prob, const sqpo_settings *settings, const double *x0, const double *lambda0,
                double *x_out, double *lambda_out, sqpo_info *info_out);

#ifdef __cplusplus
}
#endif
#endif
