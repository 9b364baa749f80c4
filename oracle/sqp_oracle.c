/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's SQP outer loop, the caller of the QP hot path:
 *   sqp::SQP<double>            /root/reference/src/sqp.cpp:13-343, include/solvers/sqp.hpp:13-161
 *   BFGS_update                 /root/reference/include/solvers/bfgs.hpp:14-41
 * on top of the QP oracle (qp_oracle.c).  Used to pin the batched host driver
 * (include/sqp_hip/sqp.hpp, BASELINE config 4) instance by instance, and checked itself against the
 * known answers of the reference's SQP tests (tests/sqp_test.cpp, tests/sqp_test_autodiff.cpp).
 * Matrices are column-major like Eigen's.
 */
#include "sqp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

static sqpo_trace_fn g_trace = 0;
static void *g_trace_user = 0;
static int g_qp_extended = 0;
void sqpo_set_qp_extended(int on) { g_qp_extended = on; }
void sqpo_set_trace(sqpo_trace_fn f, void *user) {
    g_trace = f;
    g_trace_user = user;
}

/* sqp_settings_t defaults, sqp.hpp:13-23 */
void sqpo_default_settings(sqpo_settings *s) {
    s->tau = 0.5;
    s->eta = 0.25;
    s->rho = 0.5;
    s->eps_prim = 1e-4;
    s->eps_dual = 1e-4;
    s->max_iter = 100;
    s->line_search_max_iter = 20;
    s->second_order_correction = 0;
}

/* is_posdef via Eigen::LLT (sqp.cpp:115-122): NumericalIssue iff a pivot is <= 0 */
static int is_posdef(const double *H, int n, double *work) {
    memcpy(work, H, sizeof(double) * (size_t)n * n);
    for (int k = 0; k < n; k++) {
        double x = work[k * n + k];
        for (int j = 0; j < k; j++) x -= work[j * n + k] * work[j * n + k]; /* L(k,j)^2, lower stored col-major */
        if (!(x > 0.0)) return 0;
        x = sqrt(x);
        work[k * n + k] = x;
        for (int i = k + 1; i < n; i++) {
            double v = work[k * n + i];
            for (int j = 0; j < k; j++) v -= work[j * n + i] * work[j * n + k];
            work[k * n + i] = v / x;
        }
    }
    return 1;
}

/* Damped BFGS, bfgs.hpp:14-41 */
static void bfgs_update(double *B, int n, const double *s, const double *y, double *Bs, double *r) {
    double sBs = 0, sy = 0, sr;
    for (int i = 0; i < n; i++) {
        double a = 0;
        for (int j = 0; j < n; j++) a += B[j * n + i] * s[j];
        Bs[i] = a;
    }
    for (int i = 0; i < n; i++) {
        sBs += s[i] * Bs[i];
        sy += s[i] * y[i];
    }
    if (sy < 0.2 * sBs) {
        const double theta = 0.8 * sBs / (sBs - sy);
        for (int i = 0; i < n; i++) r[i] = theta * y[i] + (1 - theta) * Bs[i];
        sr = theta * sy + (1 - theta) * sBs;
    } else {
        for (int i = 0; i < n; i++) r[i] = y[i];
        sr = sy;
    }
    if (sr < DBL_EPSILON) return;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) B[j * n + i] += -Bs[i] * Bs[j] / sBs + r[i] * r[j] / sr;
}

/* constraint_norm, sqp.cpp:310-318 */
static double constraint_norm(const double *c, const double *l, const double *u, int m) {
    double v = DBL_EPSILON;
    double a = 0, b = 0;
    for (int i = 0; i < m; i++) a += (l[i] - c[i]) > 0.0 ? (l[i] - c[i]) : 0.0;
    for (int i = 0; i < m; i++) b += (c[i] - u[i]) > 0.0 ? (c[i] - u[i]) : 0.0;
    v += a;
    v += b;
    return v;
}

typedef struct {
    int n, m;
    double *x, *lambda, *step_prev, *grad_L, *delta_grad_L, *Hess, *grad_obj, *Jac, *constr, *l, *u;
    double obj, primal_step_norm, dual_step_norm;
    double *p, *p_lambda, *ql, *qu, *tmp_n, *tmp_n2, *tmp_m, *work_nn, *x_step, *c_step, *d;
    qpo_solver_f64 *qp;
    qpo_solver_f80 *qp80; /* only with sqpo_set_qp_extended(1): the same QP algorithm in x87 extended precision */
    sqpo_settings settings;
    sqpo_info info;
} sqp_state;

static double *dalloc(int k) { return (double *)calloc((size_t)(k > 0 ? k : 1), sizeof(double)); }

/* run_solve_qp, sqp.cpp:210-242: setup() + solve() on the persistent QP solver member */
static int run_solve_qp(sqp_state *s, const double *P, const double *q, const double *A, const double *l, const double *u,
                        double *prim, double *dual) {
    if (s->qp80) {
        /* yard-stick mode: identical statement order, the QP arithmetic in long double.  Shows which outcomes of the outer
         * loop hinge on the rounding of the QP iterate (nothing else in the outer loop changes). */
        const int n = s->n, m = s->m;
        long double *b = (long double *)malloc(sizeof(long double) * (size_t)(n * n + n + m * n + 2 * m + 1));
        long double *Pl = b, *ql = Pl + n * n, *Al = ql + n, *ll = Al + m * n, *ul = ll + m;
        for (int i = 0; i < n * n; i++) Pl[i] = P[i];
        for (int i = 0; i < n; i++) ql[i] = q[i];
        for (int i = 0; i < m * n; i++) Al[i] = A[i];
        for (int i = 0; i < m; i++) { ll[i] = l[i]; ul[i] = u[i]; }
        qpo_setup_f80(s->qp80, n, m, Pl, ql, Al, ll, ul);
        qpo_solve_f80(s->qp80, Pl, ql, Al, ll, ul);
        free(b);
        s->info.qp_solver_iter += qpo_info_ptr_f80(s->qp80)->iter;
        if (qpo_info_ptr_f80(s->qp80)->status == QPO_NUMERICAL_ISSUES) return 0;
        for (int i = 0; i < n; i++) prim[i] = (double)qpo_primal_f80(s->qp80)[i];
        for (int i = 0; i < m; i++) dual[i] = (double)qpo_dual_f80(s->qp80)[i];
        return 1;
    }
    qpo_setup_f64(s->qp, s->n, s->m, P, q, A, l, u);
    qpo_solve_f64(s->qp, P, q, A, l, u);
    s->info.qp_solver_iter += qpo_info_ptr_f64(s->qp)->iter;
    if (qpo_info_ptr_f64(s->qp)->status == QPO_NUMERICAL_ISSUES) return 0;
    memcpy(prim, qpo_primal_f64(s->qp), sizeof(double) * (size_t)s->n);
    memcpy(dual, qpo_dual_f64(s->qp), sizeof(double) * (size_t)s->m);
    return 1;
}

/* solve_qp, sqp.cpp:139-208 (+ second_order_correction 244-276) */
static void solve_qp(sqp_state *s, const sqpo_problem *prob) {
    const int n = s->n, m = s->m;
    prob->objective_linearized(prob->user, s->x, s->grad_obj, &s->obj);
    prob->constraint_linearized(prob->user, s->x, s->Jac, s->constr, s->l, s->u);
    for (int i = 0; i < n; i++) s->delta_grad_L[i] = -s->grad_L[i];
    for (int j = 0; j < n; j++) {
        double a = 0;
        for (int i = 0; i < m; i++) a += s->Jac[j * m + i] * s->lambda[i];
        s->grad_L[j] = s->grad_obj[j] + a;
    }
    if (s->info.iter == 1) {
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) s->Hess[j * n + i] = (i == j) ? 1.0 : 0.0;
    } else {
        for (int i = 0; i < n; i++) s->delta_grad_L[i] += s->grad_L[i];
        bfgs_update(s->Hess, n, s->step_prev, s->delta_grad_L, s->tmp_n, s->tmp_n2);
    }
    if (!is_posdef(s->Hess, n, s->work_nn)) {
        double tau = 1e-3;
        while (!is_posdef(s->Hess, n, s->work_nn)) {
            for (int i = 0; i < n; i++) s->Hess[i * n + i] += tau;
            tau *= 10;
        }
    }
    for (int i = 0; i < m; i++) {
        s->ql[i] = s->l[i] - s->constr[i];
        s->qu[i] = s->u[i] - s->constr[i];
    }
    run_solve_qp(s, s->Hess, s->grad_obj, s->Jac, s->ql, s->qu, s->p, s->p_lambda);
    if (s->settings.second_order_correction) {
        for (int i = 0; i < n; i++) s->x_step[i] = s->x[i] + s->p[i];
        prob->constraint(prob->user, s->x_step, s->c_step, s->l, s->u);
        for (int i = 0; i < m; i++) {
            double a = 0;
            for (int j = 0; j < n; j++) a += s->Jac[j * m + i] * s->p[j];
            s->d[i] = s->c_step[i] - a;
        }
        for (int i = 0; i < m; i++) {
            s->ql[i] = s->l[i] - s->d[i];
            s->qu[i] = s->u[i] - s->d[i];
        }
        run_solve_qp(s, s->Hess, s->grad_obj, s->Jac, s->ql, s->qu, s->p, s->p_lambda);
    }
}

/* line_search, sqp.cpp:277-308 */
static double line_search(sqp_state *s, const sqpo_problem *prob) {
    const int n = s->n, m = s->m;
    const double tau = s->settings.tau;
    const double constr_l1 = constraint_norm(s->constr, s->l, s->u, m);
    double gp = 0, pHp = 0;
    for (int i = 0; i < n; i++) gp += s->grad_obj[i] * s->p[i];
    for (int i = 0; i < n; i++) {
        double a = 0;
        for (int j = 0; j < n; j++) a += s->Hess[j * n + i] * s->p[j];
        pHp += s->p[i] * a;
    }
    const double mu = (gp + 0.5 * pHp) / ((1 - s->settings.rho) * constr_l1);
    const double phi_l1 = s->obj + mu * constr_l1;
    const double Dp_phi_l1 = gp - mu * constr_l1;
    double alpha = 1.0;
    for (int i = 1; i < s->settings.line_search_max_iter; i++) {
        double obj_step;
        for (int k = 0; k < n; k++) s->x_step[k] = s->x[k] + alpha * s->p[k];
        prob->objective(prob->user, s->x_step, &obj_step);
        prob->constraint(prob->user, s->x_step, s->constr, s->l, s->u); /* constraint_norm(x, prob) overwrites constr_, l_, u_ */
        const double phi_l1_step = obj_step + mu * constraint_norm(s->constr, s->l, s->u, m);
        if (phi_l1_step <= phi_l1 + alpha * s->settings.eta * Dp_phi_l1) break;
        alpha = tau * alpha;
    }
    return alpha;
}

static double inf_norm(const double *v, int k) {
    double r = 0;
    for (int i = 0; i < k; i++) r = fabs(v[i]) > r ? fabs(v[i]) : r;
    return r;
}

/* test access to the BFGS restatement (the reference tests it on its own, tests/bfgs_test.cpp) */
void sqpo_bfgs_update(double *B, int n, const double *s, const double *y) {
    double *w = dalloc(2 * n);
    bfgs_update(B, n, s, y, w, w + n);
    free(w);
}

/* SQP::solve(prob, x0, lambda0) -> run_solve, sqp.cpp:26-101 */
void sqpo_solve(const sqpo_problem *prob, const sqpo_settings *settings, const double *x0, const double *lambda0,
                double *x_out, double *lambda_out, sqpo_info *info_out) {
    sqp_state S;
    sqp_state *s = &S;
    memset(s, 0, sizeof(S));
    const int n = s->n = prob->num_var, m = s->m = prob->num_constr;
    s->settings = *settings;
    s->x = dalloc(n); s->lambda = dalloc(m); s->step_prev = dalloc(n); s->grad_L = dalloc(n); s->delta_grad_L = dalloc(n);
    s->Hess = dalloc(n * n); s->grad_obj = dalloc(n); s->Jac = dalloc(m * n); s->constr = dalloc(m); s->l = dalloc(m); s->u = dalloc(m);
    s->p = dalloc(n); s->p_lambda = dalloc(m); s->ql = dalloc(m); s->qu = dalloc(m); s->tmp_n = dalloc(n); s->tmp_n2 = dalloc(n);
    s->tmp_m = dalloc(m); s->work_nn = dalloc(n * n); s->x_step = dalloc(n); s->c_step = dalloc(m); s->d = dalloc(m);
    if (x0) memcpy(s->x, x0, sizeof(double) * (size_t)n);
    if (lambda0) memcpy(s->lambda, lambda0, sizeof(double) * (size_t)m);
    /* QP settings of the SQP constructor, sqp.cpp:15-23 */
    s->qp = qpo_create_f64();
    qpo_settings *qs = qpo_settings_ptr_f64(s->qp);
    qs->warm_start = 1; qs->check_termination = 10; qs->eps_abs = 1e-4; qs->eps_rel = 1e-4; qs->max_iter = 100;
    qs->adaptive_rho = 1; qs->adaptive_rho_interval = 50; qs->alpha = 1.6;
    if (g_qp_extended) {
        s->qp80 = qpo_create_f80();
        *qpo_settings_ptr_f80(s->qp80) = *qs;
    }
    s->info.qp_solver_iter = 0;
    s->info.status = SQPO_MAX_ITER_EXCEEDED;

    int iter;
    for (iter = 1; iter <= s->settings.max_iter; iter++) {
        s->info.iter = iter;
        solve_qp(s, prob);
        for (int i = 0; i < m; i++) s->p_lambda[i] -= s->lambda[i];
        const double alpha = line_search(s, prob);
        if (g_trace) g_trace(g_trace_user, iter, s->p, s->p_lambda, alpha, s->info.qp_solver_iter);
        for (int i = 0; i < n; i++) s->x[i] += alpha * s->p[i];
        for (int i = 0; i < m; i++) s->lambda[i] += alpha * s->p_lambda[i];
        for (int i = 0; i < n; i++) s->step_prev[i] = alpha * s->p[i];
        s->primal_step_norm = alpha * inf_norm(s->p, n);
        s->dual_step_norm = alpha * inf_norm(s->p_lambda, m);
        if (getenv("SQPO_TRACE")) {
            fprintf(stderr, "sqpo iter %d alpha %.3e obj %.6e p", iter, alpha, s->obj);
            for (int i = 0; i < n; i++) fprintf(stderr, " %.9g", s->p[i]);
            fprintf(stderr, " | x");
            for (int i = 0; i < n; i++) fprintf(stderr, " %.9g", s->x[i]);
            fprintf(stderr, "\n");
        }
        /* termination_criteria, sqp.cpp:124-131 + max_constraint_violation 329-343 */
        double c_max = 0;
        prob->constraint(prob->user, s->x, s->constr, s->l, s->u);
        if (m > 0) {
            double a = -INFINITY, b = -INFINITY;
            for (int i = 0; i < m; i++) a = (s->l[i] - s->constr[i]) > a ? (s->l[i] - s->constr[i]) : a;
            for (int i = 0; i < m; i++) b = (s->constr[i] - s->u[i]) > b ? (s->constr[i] - s->u[i]) : b;
            c_max = fmax(c_max, a);
            c_max = fmax(c_max, b);
        }
        if (s->primal_step_norm <= s->settings.eps_prim && s->dual_step_norm <= s->settings.eps_dual && c_max <= s->settings.eps_prim) {
            s->info.status = SQPO_SOLVED;
            break;
        }
    }
    if (iter > s->settings.max_iter) s->info.status = SQPO_MAX_ITER_EXCEEDED;
    s->info.iter = iter;
    memcpy(x_out, s->x, sizeof(double) * (size_t)n);
    if (lambda_out) memcpy(lambda_out, s->lambda, sizeof(double) * (size_t)m);
    if (info_out) *info_out = s->info;
    qpo_destroy_f64(s->qp);
    if (s->qp80) qpo_destroy_f80(s->qp80);
    double *ptrs[] = {s->x, s->lambda, s->step_prev, s->grad_L, s->delta_grad_L, s->Hess, s->grad_obj, s->Jac, s->constr, s->l, s->u,
                      s->p, s->p_lambda, s->ql, s->qu, s->tmp_n, s->tmp_n2, s->tmp_m, s->work_nn, s->x_step, s->c_step, s->d};
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(ptrs[i]);
}
