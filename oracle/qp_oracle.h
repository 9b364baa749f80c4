/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See qp_oracle_impl.h.
 *
 * C interface of the CPU oracle: a restatement of qp_solver::QPSolver<Scalar>
 * (/root/reference/include/solvers/qp.hpp:36-108,118-173) for Scalar=double
 * (_f64) and Scalar=float (_f32).
 *
 * Parity pin status: pinned against the reference's own known-answer tests
 * (tests/qp_solver_test.cpp:43-156, tests/unsupported/qp_solver_test.cpp,
 * tests/qp_solver_sparse_test.cpp:68-98; all 1e-2 solution tests + the
 * constraint-classification table) and against the analytic KKT solution of
 * that fixture.  The LDL^T layer itself is *unpinned*: Eigen is absent from
 * the image and the reference holds no test at that level.
 */
#ifndef QP_ORACLE_H
#define QP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* QPSolverStatus order, qp.hpp:70 */
enum { QPO_SOLVED = 0, QPO_MAX_ITER_EXCEEDED = 1, QPO_UNSOLVED = 2, QPO_NUMERICAL_ISSUES = 3, QPO_UNINITIALIZED = 4 };
/* ConstraintType order, qp.hpp:134 */
enum { QPO_INEQUALITY_CONSTRAINT = 0, QPO_EQUALITY_CONSTRAINT = 1, QPO_LOOSE_BOUNDS = 2 };

/* QPSolverSettings, qp.hpp:36-54 (values are cast to Scalar on use) */
typedef struct qpo_settings {
    double rho, sigma, alpha, eps_rel, eps_abs;
    int max_iter, check_termination, warm_start, adaptive_rho;
    double adaptive_rho_tolerance;
    int adaptive_rho_interval, verbose;
} qpo_settings;

/* QPSolverInfo, qp.hpp:72-80 */
typedef struct qpo_info {
    int status, iter, rho_updates, _pad;
    double rho_estimate, res_prim, res_dual;
} qpo_info;

void qpo_default_settings(qpo_settings *s);

#define QPO_DECL(SFX_, T)                                                                                   \
    typedef struct qpo_solver##SFX_ qpo_solver##SFX_;                                                       \
    qpo_solver##SFX_ *qpo_create##SFX_(void);                                                               \
    void qpo_destroy##SFX_(qpo_solver##SFX_ *);                                                             \
    qpo_settings *qpo_settings_ptr##SFX_(qpo_solver##SFX_ *);                                               \
    qpo_info *qpo_info_ptr##SFX_(qpo_solver##SFX_ *);                                                       \
    void qpo_set_legacy_cold_start##SFX_(qpo_solver##SFX_ *, int on);                                       \
    const T *qpo_primal##SFX_(const qpo_solver##SFX_ *);                                                    \
    const T *qpo_dual##SFX_(const qpo_solver##SFX_ *);                                                      \
    const T *qpo_z##SFX_(const qpo_solver##SFX_ *);                                                         \
    const int *qpo_constr_type##SFX_(const qpo_solver##SFX_ *);                                             \
    const T *qpo_rho_vec##SFX_(const qpo_solver##SFX_ *);                                                   \
    void qpo_set_state##SFX_(qpo_solver##SFX_ *, const T *x, const T *z, const T *y);                       \
    void qpo_constr_type_init##SFX_(int m, const T *l, const T *u, int *constr_type);                       \
    void qpo_setup##SFX_(qpo_solver##SFX_ *, int n, int m, const T *P, const T *q, const T *A, const T *l,  \
                         const T *u);                                                                       \
    void qpo_update_qp##SFX_(qpo_solver##SFX_ *, const T *P, const T *q, const T *A, const T *l,            \
                             const T *u);                                                                   \
    void qpo_solve##SFX_(qpo_solver##SFX_ *, const T *P, const T *q, const T *A, const T *l, const T *u);   \
    void qpo_solve_batch##SFX_(int n, int m, int batch, const T *P, const T *q, const T *A, const T *l,     \
                               const T *u, const qpo_settings *settings, T *x_out, T *y_out, T *z_out,      \
                               qpo_info *info_out, int nthreads);                                           \
    int qpo_ldlt_factor_solve##SFX_(int size, const T *K, T *L_out, int *transp_out, T *rhs_inout);

QPO_DECL(_f64, double)
QPO_DECL(_f32, float)
QPO_DECL(_f80, long double) /* extended-precision yard-stick, see qp_oracle.c */

int qpo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
