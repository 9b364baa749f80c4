/* TEST INFRASTRUCTURE.  The CPU oracle (oracle/qp_oracle.c) driven directly from C so that it can run under
 * -fsanitize=address,undefined (host sanitizers on the restatement): the reference's SimpleQP in the three
 * scalar instances, random QPs with every row class through setup / solve / update_qp / solve with adaptive rho, the batch
 * entry point on two threads, the LDL' helper.  Exit code 0 = finite results, expected statuses; the sanitizers abort otherwise. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../oracle/qp_oracle.h"

static unsigned long long rng = 88172645463325252ull;
static double urand(void) {
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (double)(rng >> 11) / 9007199254740992.0;
}
static double nrand(void) { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static int simple_f64(void) {
    const double P[4] = {4, 1, 1, 2}, q[2] = {1, 1}, A[6] = {1, 1, 0, 1, 0, 1}, l[3] = {1, 0, 0}, u[3] = {1, 0.7, 0.7};
    qpo_solver_f64 *s = qpo_create_f64();
    qpo_setup_f64(s, 2, 3, P, q, A, l, u);
    qpo_solve_f64(s, P, q, A, l, u);
    const double *x = qpo_primal_f64(s);
    CHECK(qpo_info_ptr_f64(s)->status == QPO_SOLVED && fabs(x[0] - 0.3) < 1e-2 && fabs(x[1] - 0.7) < 1e-2);
    qpo_destroy_f64(s);
    return 0;
}
static int simple_f32(void) {
    const float P[4] = {4, 1, 1, 2}, q[2] = {1, 1}, A[6] = {1, 1, 0, 1, 0, 1}, l[3] = {1, 0, 0}, u[3] = {1, 0.7f, 0.7f};
    qpo_solver_f32 *s = qpo_create_f32();
    qpo_setup_f32(s, 2, 3, P, q, A, l, u);
    qpo_solve_f32(s, P, q, A, l, u);
    const float *x = qpo_primal_f32(s);
    CHECK(qpo_info_ptr_f32(s)->status == QPO_SOLVED && fabsf(x[0] - 0.3f) < 1e-2f && fabsf(x[1] - 0.7f) < 1e-2f);
    qpo_destroy_f32(s);
    return 0;
}
static int simple_f80(void) {
    const long double P[4] = {4, 1, 1, 2}, q[2] = {1, 1}, A[6] = {1, 1, 0, 1, 0, 1}, l[3] = {1, 0, 0}, u[3] = {1, 0.7L, 0.7L};
    qpo_solver_f80 *s = qpo_create_f80();
    qpo_setup_f80(s, 2, 3, P, q, A, l, u);
    qpo_solve_f80(s, P, q, A, l, u);
    CHECK(qpo_info_ptr_f80(s)->status == QPO_SOLVED);
    qpo_destroy_f80(s);
    return 0;
}
/* column-major P (n x n, SPD), A (m x n), rows: equalities, one-sided, loose, boxes */
static void random_qp(int n, int m, double *P, double *q, double *A, double *l, double *u) {
    double *G = (double *)malloc(sizeof(double) * n * n), *x0 = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n * n; i++) G[i] = nrand();
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += G[i + k * n] * G[j + k * n];
            P[i + j * n] = s / n + (i == j ? 0.1 : 0.0);
        }
    for (int j = 0; j < n; j++) { q[j] = nrand(); x0[j] = nrand(); }
    for (int i = 0; i < m * n; i++) A[i] = nrand();
    for (int i = 0; i < m; i++) {
        double c = 0;
        for (int j = 0; j < n; j++) c += A[i + j * m] * x0[j];
        l[i] = c - urand(); u[i] = c + urand();
        if (i % 10 == 0) l[i] = u[i] = c;
        if (i % 10 == 1) u[i] = INFINITY;
        if (i % 10 == 2) { l[i] = -1e20; u[i] = 1e20; }
    }
    free(G); free(x0);
}
static int random_paths(void) {
    const int n = 12, m = 21, B = 6;
    double *P = (double *)malloc(sizeof(double) * B * n * n), *q = (double *)malloc(sizeof(double) * B * n), *A = (double *)malloc(sizeof(double) * B * m * n);
    double *l = (double *)malloc(sizeof(double) * B * m), *u = (double *)malloc(sizeof(double) * B * m);
    for (int b = 0; b < B; b++) random_qp(n, m, P + b * n * n, q + b * n, A + b * m * n, l + b * m, u + b * m);
    qpo_solver_f64 *s = qpo_create_f64();
    qpo_settings_ptr_f64(s)->adaptive_rho = 1;
    qpo_settings_ptr_f64(s)->adaptive_rho_interval = 10;
    qpo_setup_f64(s, n, m, P, q, A, l, u);
    qpo_solve_f64(s, P, q, A, l, u);
    CHECK(qpo_info_ptr_f64(s)->status == QPO_SOLVED || qpo_info_ptr_f64(s)->status == QPO_MAX_ITER_EXCEEDED);
    qpo_update_qp_f64(s, P + n * n, q + n, A + m * n, l + m, u + m);
    qpo_solve_f64(s, P + n * n, q + n, A + m * n, l + m, u + m);
    for (int j = 0; j < n; j++) CHECK(isfinite(qpo_primal_f64(s)[j]));
    for (int i = 0; i < m; i++) CHECK(isfinite(qpo_dual_f64(s)[i]) && qpo_constr_type_f64(s)[i] >= 0 && qpo_rho_vec_f64(s)[i] > 0);
    qpo_destroy_f64(s);
    /* the batch entry point (OpenMP over QPs) */
    double *x = (double *)malloc(sizeof(double) * B * n), *y = (double *)malloc(sizeof(double) * B * m), *z = (double *)malloc(sizeof(double) * B * m);
    qpo_info *info = (qpo_info *)malloc(sizeof(qpo_info) * B);
    qpo_settings st;
    qpo_default_settings(&st);
    qpo_solve_batch_f64(n, m, B, P, q, A, l, u, &st, x, y, z, info, 2);
    for (int b = 0; b < B; b++) CHECK(info[b].status == QPO_SOLVED || info[b].status == QPO_MAX_ITER_EXCEEDED);
    /* the LDL' helper on a quasi-definite matrix */
    {
        const int N = 5;
        double K[25], L[25], rhs[5];
        int tr[5];
        for (int i = 0; i < 25; i++) K[i] = 0;
        for (int i = 0; i < N; i++) { K[i + i * N] = i < 3 ? 2.0 + i : -1.0 - i; rhs[i] = 1.0 + i; }
        K[3 + 0 * N] = K[0 + 3 * N] = 0.5; K[4 + 1 * N] = K[1 + 4 * N] = -0.25;
        CHECK(qpo_ldlt_factor_solve_f64(N, K, L, tr, rhs) != 0); /* (non-zero = Eigen::Success) */
        for (int i = 0; i < N; i++) CHECK(isfinite(rhs[i]));
    }
    free(P); free(q); free(A); free(l); free(u); free(x); free(y); free(z); free(info);
    return 0;
}
int main(void) {
    if (simple_f64() || simple_f32() || simple_f80() || random_paths()) return 1;
    printf("oracle sanitize run passed\n");
    return 0;
}
