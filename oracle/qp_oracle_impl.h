/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, Eigen-free) of the reference ADMM QP solver
 *   qp_solver::QPSolver<Scalar>            /root/reference/src/qp.cpp:11-386
 *                                           /root/reference/include/solvers/qp.hpp:118-248
 * plus the dense pivoted LDL^T it delegates to (Eigen::LDLT<Matrix,Lower>,
 * third-party, NOT vendored in /root/reference: `find_package(Eigen3 3.3)`
 * CMakeLists.txt:12; call sites qp.hpp:129, qp.cpp:90,242,253).  Eigen is not
 * installed in the build image, so the reference itself is unbuildable here
 * (no oracle/_ref) and the LDL^T arithmetic is restated from Eigen's published
 * algorithm (Cholesky/LDLT.h, unblocked lower variant; see ldlt_compute below).
 *
 * This file is included twice by qp_oracle.c with
 *     #define SCALAR double / float     #define SFX(name) name##_f64 / name##_f32
 * mirroring the two explicit instantiations at src/qp.cpp:385-386.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this code, and only as the checker.
 */

/* ------------------------------------------------------------------------ */
/* Dense LDL^T with diagonal pivoting, lower storage, column-major.          */
/* Follows the unblocked in-place algorithm of Eigen::LDLT<...,Lower>:       */
/*   step k: pick p = argmax_{i>=k} |M_ii| over the *stored* diagonal (first */
/*   maximum wins), apply the symmetric transposition k<->p to the lower     */
/*   triangle, then the left-looking column update                            */
/*       t_j   = D_j * L_kj            (j<k)                                  */
/*       M_kk -= L_k,0:k . t                                                   */
/*       M_{k+1:,k} -= L_{k+1:,0:k} t ;  M_{k+1:,k} /= M_kk  (if M_kk != 0)   */
/*   returns 0 ("NumericalIssue") iff a non-zero pivot follows a zero pivot   */
/*   or a zero pivot has a non-zero column.                                   */
/* ------------------------------------------------------------------------ */
typedef struct SFX(qpo_ldlt) {
    int size;
    SCALAR *mat;   /* size*size col-major; lower = L (unit diag implied), diag = D */
    int *transp;   /* transpositions[k] */
    SCALAR *temp;  /* size */
    int ok;        /* 1 = Success, 0 = NumericalIssue */
} SFX(qpo_ldlt);

static void SFX(ldlt_alloc)(SFX(qpo_ldlt) *f, int size) {
    f->size = size;
    f->mat = (SCALAR *)malloc(sizeof(SCALAR) * (size_t)size * (size_t)size);
    f->transp = (int *)malloc(sizeof(int) * (size_t)size);
    f->temp = (SCALAR *)malloc(sizeof(SCALAR) * (size_t)size);
    f->ok = 0;
}

static void SFX(ldlt_free)(SFX(qpo_ldlt) *f) {
    free(f->mat);
    free(f->transp);
    free(f->temp);
    f->mat = NULL;
    f->transp = NULL;
    f->temp = NULL;
}

#define MAT(i, j) mat[(size_t)(j) * (size_t)size + (size_t)(i)]

/* factor the matrix already stored in f->mat (only the lower triangle is read) */
static int SFX(ldlt_compute_inplace)(SFX(qpo_ldlt) *f) {
    const int size = f->size;
    SCALAR *mat = f->mat;
    SCALAR *temp = f->temp;
    int ret = 1;
    int found_zero_pivot = 0;

    if (size <= 1) {
        for (int i = 0; i < size; i++) f->transp[i] = i;
        f->ok = 1;
        return 1;
    }

    for (int k = 0; k < size; k++) {
        /* largest stored diagonal entry of the trailing block */
        int p = k;
        SCALAR best = SABS(MAT(k, k));
        for (int i = k + 1; i < size; i++) {
            SCALAR a = SABS(MAT(i, i));
            if (a > best) {
                best = a;
                p = i;
            }
        }
        f->transp[k] = p;
        if (p != k) {
            /* symmetric row/col swap restricted to the lower triangle */
            for (int j = 0; j < k; j++) {
                SCALAR t = MAT(k, j);
                MAT(k, j) = MAT(p, j);
                MAT(p, j) = t;
            }
            for (int i = p + 1; i < size; i++) {
                SCALAR t = MAT(i, k);
                MAT(i, k) = MAT(i, p);
                MAT(i, p) = t;
            }
            {
                SCALAR t = MAT(k, k);
                MAT(k, k) = MAT(p, p);
                MAT(p, p) = t;
            }
            for (int i = k + 1; i < p; i++) {
                SCALAR t = MAT(i, k);
                MAT(i, k) = MAT(p, i);
                MAT(p, i) = t;
            }
        }

        const int rs = size - k - 1;
        if (k > 0) {
            SCALAR acc = 0;
            for (int j = 0; j < k; j++) {
                temp[j] = MAT(j, j) * MAT(k, j);
                acc += MAT(k, j) * temp[j];
            }
            MAT(k, k) -= acc;
            for (int j = 0; j < k; j++) {
                const SCALAR tj = temp[j];
                for (int i = k + 1; i < size; i++) MAT(i, k) -= MAT(i, j) * tj;
            }
        }

        const SCALAR akk = MAT(k, k);
        const int pivot_is_valid = SABS(akk) > (SCALAR)0;

        if (k == 0 && !pivot_is_valid) {
            /* whole diagonal is zero */
            for (int j = 0; j < size; j++) {
                f->transp[j] = j;
                for (int i = j + 1; i < size; i++)
                    if (MAT(i, j) != (SCALAR)0) ret = 0;
            }
            f->ok = ret;
            return ret;
        }

        if (rs > 0 && pivot_is_valid) {
            for (int i = k + 1; i < size; i++) MAT(i, k) /= akk;
        } else if (rs > 0) {
            for (int i = k + 1; i < size; i++)
                if (MAT(i, k) != (SCALAR)0) ret = 0;
        }

        if (found_zero_pivot && pivot_is_valid)
            ret = 0;
        else if (!pivot_is_valid)
            found_zero_pivot = 1;
    }
    /* NaN anywhere makes every comparison above false; Eigen reports
     * NumericalIssue for non-finite factors through the same predicates, so a
     * NaN pivot behaves as "invalid pivot with non-zero column". */
    f->ok = ret;
    return ret;
}

/* x <- K^{-1} x using the stored factor: P, L, pinv(D), L^T, P^T */
static void SFX(ldlt_solve_inplace)(const SFX(qpo_ldlt) *f, SCALAR *x) {
    const int size = f->size;
    const SCALAR *mat = f->mat;
    for (int k = 0; k < size; k++) {
        const int p = f->transp[k];
        if (p != k) {
            SCALAR t = x[k];
            x[k] = x[p];
            x[p] = t;
        }
    }
    /* unit lower solve, column oriented */
    for (int j = 0; j < size; j++) {
        const SCALAR xj = x[j];
        if (xj != (SCALAR)0)
            for (int i = j + 1; i < size; i++) x[i] -= MAT(i, j) * xj;
    }
    /* pseudo-inverse of D */
    const SCALAR tol = SCALAR_MIN;
    for (int i = 0; i < size; i++) {
        const SCALAR d = MAT(i, i);
        if (SABS(d) > tol)
            x[i] /= d;
        else
            x[i] = 0;
    }
    /* unit upper solve with L^T */
    for (int i = size - 1; i >= 0; i--) {
        SCALAR acc = x[i];
        for (int k = i + 1; k < size; k++) acc -= MAT(k, i) * x[k];
        x[i] = acc;
    }
    for (int k = size - 1; k >= 0; k--) {
        const int p = f->transp[k];
        if (p != k) {
            SCALAR t = x[k];
            x[k] = x[p];
            x[p] = t;
        }
    }
}
#undef MAT

/* ------------------------------------------------------------------------ */
/* Solver object == qp_solver::QPSolver<Scalar> private state (qp.hpp:217-247) */
/* ------------------------------------------------------------------------ */
typedef struct SFX(qpo_solver) {
    int n, m;
    int iter;
    SCALAR *x, *z, *y;
    SCALAR *x_tilde, *z_tilde, *z_prev;
    SCALAR *rho_vec, *rho_inv_vec;
    SCALAR rho;
    SCALAR *rhs, *x_tilde_nu;
    SCALAR max_Ax_z_norm, max_Px_ATy_q_norm;
    int *constr_type;
    SCALAR *kkt; /* (n+m)^2 col-major, lower blocks defined */
    SCALAR *tmp_n, *tmp_m;
    SFX(qpo_ldlt) lin;
    qpo_settings settings;
    qpo_info info;
    int legacy_cold_start; /* 0: src/qp.cpp:78-82 (reset is a no-op); 1: unsupported/qp_solver.hpp:256-260 */
} SFX(qpo_solver);

static void SFX(free_state)(SFX(qpo_solver) * s) {
    free(s->x); free(s->z); free(s->y);
    free(s->x_tilde); free(s->z_tilde); free(s->z_prev);
    free(s->rho_vec); free(s->rho_inv_vec);
    free(s->rhs); free(s->x_tilde_nu);
    free(s->constr_type); free(s->kkt); free(s->tmp_n); free(s->tmp_m);
    s->x = s->z = s->y = s->x_tilde = s->z_tilde = s->z_prev = NULL;
    s->rho_vec = s->rho_inv_vec = s->rhs = s->x_tilde_nu = s->kkt = s->tmp_n = s->tmp_m = NULL;
    s->constr_type = NULL;
    if (s->lin.mat) SFX(ldlt_free)(&s->lin);
}

SFX(qpo_solver) * SFX(qpo_create)(void) {
    SFX(qpo_solver) *s = (SFX(qpo_solver) *)calloc(1, sizeof(SFX(qpo_solver)));
    qpo_default_settings(&s->settings);
    s->info.status = QPO_UNINITIALIZED; /* qp.hpp:74 */
    s->info.iter = 0;
    s->info.rho_updates = 0;
    s->info.rho_estimate = 0;
    s->info.res_prim = 0;
    s->info.res_dual = 0;
    return s;
}

void SFX(qpo_destroy)(SFX(qpo_solver) * s) {
    if (!s) return;
    SFX(free_state)(s);
    free(s);
}

qpo_settings *SFX(qpo_settings_ptr)(SFX(qpo_solver) * s) { return &s->settings; }
qpo_info *SFX(qpo_info_ptr)(SFX(qpo_solver) * s) { return &s->info; }
void SFX(qpo_set_legacy_cold_start)(SFX(qpo_solver) * s, int on) { s->legacy_cold_start = on; }
const SCALAR *SFX(qpo_primal)(const SFX(qpo_solver) * s) { return s->x; }
const SCALAR *SFX(qpo_dual)(const SFX(qpo_solver) * s) { return s->y; }
const SCALAR *SFX(qpo_z)(const SFX(qpo_solver) * s) { return s->z; }
const int *SFX(qpo_constr_type)(const SFX(qpo_solver) * s) { return s->constr_type; }
const SCALAR *SFX(qpo_rho_vec)(const SFX(qpo_solver) * s) { return s->rho_vec; }
/* warm-start injection used by tests (the reference exposes x,z,y through the
 * non-const primal_solution()/dual_solution() accessors, qp.hpp:160-164) */
void SFX(qpo_set_state)(SFX(qpo_solver) * s, const SCALAR *x, const SCALAR *z, const SCALAR *y) {
    if (x) memcpy(s->x, x, sizeof(SCALAR) * (size_t)s->n);
    if (z) memcpy(s->z, z, sizeof(SCALAR) * (size_t)s->m);
    if (y) memcpy(s->y, y, sizeof(SCALAR) * (size_t)s->m);
}

/* qp.cpp:283-294 (public static, qp.hpp:173) */
void SFX(qpo_constr_type_init)(int m, const SCALAR *l, const SCALAR *u, int *constr_type) {
    const SCALAR LOOSE_BOUNDS_THRESH = (SCALAR)1e+16;
    const SCALAR RHO_TOL = (SCALAR)1e-4;
    for (int i = 0; i < m; i++) {
        if (l[i] < -LOOSE_BOUNDS_THRESH && u[i] > LOOSE_BOUNDS_THRESH) {
            constr_type[i] = QPO_LOOSE_BOUNDS;
        } else if (u[i] - l[i] < RHO_TOL) {
            constr_type[i] = QPO_EQUALITY_CONSTRAINT;
        } else {
            constr_type[i] = QPO_INEQUALITY_CONSTRAINT;
        }
    }
}

/* qp.cpp:296-314 */
static void SFX(rho_vec_update)(SFX(qpo_solver) * s, SCALAR rho0) {
    const SCALAR RHO_MIN = (SCALAR)1e-6;
    const SCALAR RHO_EQ_FACTOR = (SCALAR)1e+3;
    for (int i = 0; i < s->m; i++) {
        switch (s->constr_type[i]) {
            case QPO_LOOSE_BOUNDS:
                s->rho_vec[i] = RHO_MIN;
                break;
            case QPO_EQUALITY_CONSTRAINT:
                s->rho_vec[i] = RHO_EQ_FACTOR * rho0;
                break;
            default:
                s->rho_vec[i] = rho0;
        }
    }
    for (int i = 0; i < s->m; i++) s->rho_inv_vec[i] = (SCALAR)1 / s->rho_vec[i];
    s->rho = rho0;
    s->info.rho_updates += 1;
}

/* qp.cpp:159-189 (dense branch): lower blocks only; upper-right never written */
static void SFX(construct_KKT_mat)(SFX(qpo_solver) * s, const SCALAR *P, const SCALAR *A) {
    const int n = s->n, m = s->m, N = n + m;
    SCALAR *K = s->kkt;
    const SCALAR sigma = (SCALAR)s->settings.sigma;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++)
            K[(size_t)j * N + i] = P[(size_t)j * n + i] + (i == j ? sigma : (SCALAR)0);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++) K[(size_t)j * N + n + i] = A[(size_t)j * m + i];
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++)
            K[(size_t)(n + j) * N + n + i] = (i == j) ? (SCALAR)(-1.0) * s->rho_inv_vec[i] : (SCALAR)0;
}

/* qp.cpp:225-235 */
static void SFX(update_KKT_rho)(SFX(qpo_solver) * s) {
    const int n = s->n, m = s->m, N = n + m;
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++)
            s->kkt[(size_t)(n + j) * N + n + i] = (i == j) ? (SCALAR)(-1.0) * s->rho_inv_vec[i] : (SCALAR)0;
}

/* qp.cpp:237-259: LDLT::compute copies the matrix, factors the copy */
static int SFX(factorize_KKT)(SFX(qpo_solver) * s) {
    const int N = s->n + s->m;
    /* upper-right block of kkt is uninitialised in the reference (qp.cpp:25,185-187);
     * LDLT<Lower> never reads it.  Copy only the lower triangle, zero the rest. */
    for (int j = 0; j < N; j++)
        for (int i = 0; i < N; i++)
            s->lin.mat[(size_t)j * N + i] = (i >= j) ? s->kkt[(size_t)j * N + i] : (SCALAR)0;
    return SFX(ldlt_compute_inplace)(&s->lin);
}

/* qp.cpp:11-44 */
void SFX(qpo_setup)(SFX(qpo_solver) * s, int n, int m, const SCALAR *P, const SCALAR *q,
                    const SCALAR *A, const SCALAR *l, const SCALAR *u) {
    (void)q;
    const int N = n + m;
    if (s->x && s->n == n && s->m == m && s->lin.mat && s->lin.size == N) {
        /* same shape as the last set-up of this object (the batched driver, one object per thread): the arrays are cleared in
         * place instead of freed and calloc'ed again — the same zero-initialised state, none of the allocator traffic (two of
         * the arrays are beyond glibc's mmap threshold at n = 50, m = 100: every QP page-faulted them in under the process lock) */
        memset(s->x, 0, sizeof(SCALAR) * (size_t)(n > 0 ? n : 1));
        memset(s->z, 0, sizeof(SCALAR) * (size_t)(m > 0 ? m : 1));
        memset(s->y, 0, sizeof(SCALAR) * (size_t)(m > 0 ? m : 1));
        memset(s->x_tilde, 0, sizeof(SCALAR) * (size_t)(n > 0 ? n : 1));
        memset(s->z_tilde, 0, sizeof(SCALAR) * (size_t)(m > 0 ? m : 1));
        memset(s->z_prev, 0, sizeof(SCALAR) * (size_t)(m > 0 ? m : 1));
        memset(s->rho_vec, 0, sizeof(SCALAR) * (size_t)(m > 0 ? m : 1));
        memset(s->rho_inv_vec, 0, sizeof(SCALAR) * (size_t)(m > 0 ? m : 1));
        memset(s->rhs, 0, sizeof(SCALAR) * (size_t)(N > 0 ? N : 1));
        memset(s->x_tilde_nu, 0, sizeof(SCALAR) * (size_t)(N > 0 ? N : 1));
        memset(s->constr_type, 0, sizeof(int) * (size_t)(m > 0 ? m : 1));
        memset(s->kkt, 0, sizeof(SCALAR) * (size_t)(N > 0 ? N : 1) * (size_t)(N > 0 ? N : 1));
        memset(s->tmp_n, 0, sizeof(SCALAR) * (size_t)(n > 0 ? n : 1));
        memset(s->tmp_m, 0, sizeof(SCALAR) * (size_t)(m > 0 ? m : 1));
        s->lin.ok = 0;
    } else {
    SFX(free_state)(s);
    s->n = n;
    s->m = m;
    s->x = (SCALAR *)calloc((size_t)(n > 0 ? n : 1), sizeof(SCALAR));
    s->z = (SCALAR *)calloc((size_t)(m > 0 ? m : 1), sizeof(SCALAR));
    s->y = (SCALAR *)calloc((size_t)(m > 0 ? m : 1), sizeof(SCALAR));
    s->x_tilde = (SCALAR *)calloc((size_t)(n > 0 ? n : 1), sizeof(SCALAR));
    s->z_tilde = (SCALAR *)calloc((size_t)(m > 0 ? m : 1), sizeof(SCALAR));
    s->z_prev = (SCALAR *)calloc((size_t)(m > 0 ? m : 1), sizeof(SCALAR));
    s->rho_vec = (SCALAR *)calloc((size_t)(m > 0 ? m : 1), sizeof(SCALAR));
    s->rho_inv_vec = (SCALAR *)calloc((size_t)(m > 0 ? m : 1), sizeof(SCALAR));
    s->rhs = (SCALAR *)calloc((size_t)(N > 0 ? N : 1), sizeof(SCALAR));
    s->x_tilde_nu = (SCALAR *)calloc((size_t)(N > 0 ? N : 1), sizeof(SCALAR));
    s->constr_type = (int *)calloc((size_t)(m > 0 ? m : 1), sizeof(int));
    s->kkt = (SCALAR *)calloc((size_t)(N > 0 ? N : 1) * (size_t)(N > 0 ? N : 1), sizeof(SCALAR));
    s->tmp_n = (SCALAR *)calloc((size_t)(n > 0 ? n : 1), sizeof(SCALAR));
    s->tmp_m = (SCALAR *)calloc((size_t)(m > 0 ? m : 1), sizeof(SCALAR));
    SFX(ldlt_alloc)(&s->lin, N);
    }

    SFX(qpo_constr_type_init)(m, l, u, s->constr_type);
    SFX(rho_vec_update)(s, (SCALAR)s->settings.rho);
    SFX(construct_KKT_mat)(s, P, A);
    if (SFX(factorize_KKT)(s)) {
        s->info.status = QPO_UNSOLVED;
    } else {
        s->info.status = QPO_NUMERICAL_ISSUES;
    }
}

/* qp.cpp:46-62 */
void SFX(qpo_update_qp)(SFX(qpo_solver) * s, const SCALAR *P, const SCALAR *q, const SCALAR *A,
                        const SCALAR *l, const SCALAR *u) {
    (void)q;
    SFX(qpo_constr_type_init)(s->m, l, u, s->constr_type);
    SFX(rho_vec_update)(s, (SCALAR)s->settings.rho);
    SFX(construct_KKT_mat)(s, P, A); /* update_KKT_mat == construct_KKT_mat for dense, qp.cpp:220-222 */
    if (SFX(factorize_KKT)(s)) {
        s->info.status = QPO_UNSOLVED;
    } else {
        s->info.status = QPO_NUMERICAL_ISSUES;
    }
}

static SCALAR SFX(inf_norm)(const SCALAR *v, int k) {
    SCALAR r = 0;
    for (int i = 0; i < k; i++) {
        SCALAR a = SABS(v[i]);
        if (a > r || a != a) r = a; /* NaN propagates like Eigen's lpNorm<Infinity> maxCoeff */
    }
    return r;
}

static void SFX(gemv_A)(int m, int n, const SCALAR *A, const SCALAR *x, SCALAR *out) { /* out = A x */
    for (int i = 0; i < m; i++) out[i] = 0;
    for (int j = 0; j < n; j++) {
        const SCALAR xj = x[j];
        const SCALAR *col = A + (size_t)j * m;
        for (int i = 0; i < m; i++) out[i] += col[i] * xj;
    }
}

static void SFX(gemv_AT)(int m, int n, const SCALAR *A, const SCALAR *y, SCALAR *out) { /* out = A^T y */
    for (int j = 0; j < n; j++) {
        const SCALAR *col = A + (size_t)j * m;
        SCALAR acc = 0;
        for (int i = 0; i < m; i++) acc += col[i] * y[i];
        out[j] = acc;
    }
}

/* qp.cpp:316-331 + 353-361 */
static void SFX(update_state)(SFX(qpo_solver) * s, const SCALAR *P, const SCALAR *q, const SCALAR *A) {
    const int n = s->n, m = s->m;
    SCALAR *Ax = s->tmp_m;
    SCALAR *tn = s->tmp_n;
    SFX(gemv_A)(m, n, A, s->x, Ax);
    SCALAR norm_Ax = SFX(inf_norm)(Ax, m);
    SCALAR norm_z = SFX(inf_norm)(s->z, m);
    s->max_Ax_z_norm = norm_Ax > norm_z ? norm_Ax : norm_z; /* fmax(norm_Ax, norm_z) */

    /* residual_prim: ||A x - z||_inf */
    SCALAR rp = 0;
    for (int i = 0; i < m; i++) {
        SCALAR a = SABS(Ax[i] - s->z[i]);
        if (a > rp || a != a) rp = a;
    }

    SFX(gemv_A)(n, n, P, s->x, tn); /* P x (full P, qp.cpp:324) */
    SCALAR norm_Px = SFX(inf_norm)(tn, n);
    SCALAR *ATy = s->x_tilde; /* scratch: x_tilde is dead between iterations */
    SFX(gemv_AT)(m, n, A, s->y, ATy);
    SCALAR norm_ATy = SFX(inf_norm)(ATy, n);
    SCALAR norm_q = SFX(inf_norm)(q, n);
    { SCALAR t_ = norm_ATy > norm_q ? norm_ATy : norm_q; s->max_Px_ATy_q_norm = norm_Px > t_ ? norm_Px : t_; } /* fmax(norm_Px, fmax(norm_ATy, norm_q)) */

    SCALAR rd = 0;
    for (int j = 0; j < n; j++) {
        SCALAR a = SABS(tn[j] + q[j] + ATy[j]);
        if (a > rd || a != a) rd = a;
    }
    s->info.res_prim = rp;
    s->info.res_dual = rd;
}

/* qp.cpp:64-157 */
void SFX(qpo_solve)(SFX(qpo_solver) * s, const SCALAR *P, const SCALAR *q, const SCALAR *A,
                    const SCALAR *l, const SCALAR *u) {
    const int n = s->n, m = s->m;
    const SCALAR RHO_MIN = (SCALAR)1e-6, RHO_MAX = (SCALAR)1e+6;
    const SCALAR REGUL = SCALAR_EPS;
    int check_termination = 0;

    if (s->info.status == QPO_UNINITIALIZED || s->info.status == QPO_NUMERICAL_ISSUES) return;

    if (!s->settings.warm_start && s->legacy_cold_start) {
        /* only the legacy header really resets (unsupported/qp_solver.hpp:256-260);
         * src/qp.cpp:78-82 calls the static Zero() factory and discards the result */
        for (int i = 0; i < n; i++) s->x[i] = 0;
        for (int i = 0; i < m; i++) s->z[i] = s->y[i] = 0;
    }

    const SCALAR sigma = (SCALAR)s->settings.sigma;
    int iter;
    for (iter = 1; iter <= s->settings.max_iter; iter++) {
        const SCALAR alpha = (SCALAR)s->settings.alpha;
        for (int i = 0; i < m; i++) s->z_prev[i] = s->z[i];

        /* form_KKT_rhs, qp.cpp:272-276 */
        for (int i = 0; i < n; i++) s->rhs[i] = sigma * s->x[i] - q[i];
        for (int i = 0; i < m; i++) s->rhs[n + i] = s->z[i] - s->rho_inv_vec[i] * s->y[i];

        for (int i = 0; i < n + m; i++) s->x_tilde_nu[i] = s->rhs[i];
        SFX(ldlt_solve_inplace)(&s->lin, s->x_tilde_nu);

        for (int i = 0; i < n; i++) s->x_tilde[i] = s->x_tilde_nu[i];
        for (int i = 0; i < m; i++)
            s->z_tilde[i] = s->z_prev[i] + s->rho_inv_vec[i] * (s->x_tilde_nu[n + i] - s->y[i]);

        for (int i = 0; i < n; i++) s->x[i] = alpha * s->x_tilde[i] + ((SCALAR)1 - alpha) * s->x[i];

        for (int i = 0; i < m; i++) {
            SCALAR zi = alpha * s->z_tilde[i] + ((SCALAR)1 - alpha) * s->z_prev[i] + s->rho_inv_vec[i] * s->y[i];
            /* cwiseMax(l) then cwiseMin(u), qp.cpp:278-281 */
            zi = (zi < l[i]) ? l[i] : zi;
            zi = (zi > u[i]) ? u[i] : zi;
            s->z[i] = zi;
        }

        for (int i = 0; i < m; i++)
            s->y[i] = s->y[i] +
                      s->rho_vec[i] * (alpha * s->z_tilde[i] + ((SCALAR)1 - alpha) * s->z_prev[i] - s->z[i]);

        check_termination =
            (s->settings.check_termination != 0 && iter % s->settings.check_termination == 0);

        if (check_termination) {
            SFX(update_state)(s, P, q, A);
            const SCALAR eps_prim = (SCALAR)s->settings.eps_abs + (SCALAR)s->settings.eps_rel * s->max_Ax_z_norm;
            const SCALAR eps_dual = (SCALAR)s->settings.eps_abs + (SCALAR)s->settings.eps_rel * s->max_Px_ATy_q_norm;
            if ((SCALAR)s->info.res_prim <= eps_prim && (SCALAR)s->info.res_dual <= eps_dual) {
                s->info.status = QPO_SOLVED;
                break;
            }
        }

        if (s->settings.adaptive_rho && iter % s->settings.adaptive_rho_interval == 0) {
            if (!check_termination) SFX(update_state)(s, P, q, A);
            /* rho_estimate, qp.cpp:333-341 */
            SCALAR rp_norm = (SCALAR)s->info.res_prim / (s->max_Ax_z_norm + REGUL);
            SCALAR rd_norm = (SCALAR)s->info.res_dual / (s->max_Px_ATy_q_norm + REGUL);
            SCALAR new_rho = s->rho * SSQRT(rp_norm / (rd_norm + REGUL));
            new_rho = new_rho < RHO_MAX ? new_rho : RHO_MAX; /* fmax(RHO_MIN, fmin(new_rho, RHO_MAX)) */
            new_rho = new_rho > RHO_MIN ? new_rho : RHO_MIN;
            s->info.rho_estimate = new_rho;

            const SCALAR tol = (SCALAR)s->settings.adaptive_rho_tolerance;
            if (new_rho < s->rho / tol || new_rho > s->rho * tol) {
                SFX(rho_vec_update)(s, new_rho);
                SFX(update_KKT_rho)(s);
                if (!SFX(factorize_KKT)(s)) {
                    s->info.status = QPO_NUMERICAL_ISSUES;
                    break;
                }
            }
        }
    }

    if (iter > s->settings.max_iter) s->info.status = QPO_MAX_ITER_EXCEEDED;
    s->info.iter = iter;
    s->iter = iter;
}

/* Batched driver: one fresh solver per QP, setup()+solve() exactly as the
 * reference's only production caller does (SQP::run_solve_qp, src/sqp.cpp:210-242).
 * Arrays are QP-major: problem b starts at base + b*stride. */
void SFX(qpo_solve_batch)(int n, int m, int batch, const SCALAR *P, const SCALAR *q, const SCALAR *A,
                          const SCALAR *l, const SCALAR *u, const qpo_settings *settings, SCALAR *x_out,
                          SCALAR *y_out, SCALAR *z_out, qpo_info *info_out, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        SFX(qpo_solver) *s = SFX(qpo_create)();
        s->settings = *settings;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int b = 0; b < batch; b++) {
            const SCALAR *Pb = P + (size_t)b * n * n, *qb = q + (size_t)b * n;
            const SCALAR *Ab = A + (size_t)b * m * n, *lb = l + (size_t)b * m, *ub = u + (size_t)b * m;
            s->info.rho_updates = 0; /* a fresh QPSolver object per problem */
            s->info.rho_estimate = 0;
            s->info.res_prim = s->info.res_dual = 0;
            s->info.iter = 0;
            SFX(qpo_setup)(s, n, m, Pb, qb, Ab, lb, ub);
            SFX(qpo_solve)(s, Pb, qb, Ab, lb, ub);
            if (x_out) memcpy(x_out + (size_t)b * n, s->x, sizeof(SCALAR) * (size_t)n);
            if (y_out) memcpy(y_out + (size_t)b * m, s->y, sizeof(SCALAR) * (size_t)m);
            if (z_out) memcpy(z_out + (size_t)b * m, s->z, sizeof(SCALAR) * (size_t)m);
            if (info_out) info_out[b] = s->info;
        }
        SFX(qpo_destroy)(s);
    }
}

/* Exposed for oracle self-tests: factor an arbitrary symmetric matrix (lower
 * triangle read) and solve with it. Returns 1 on Success. */
int SFX(qpo_ldlt_factor_solve)(int size, const SCALAR *K, SCALAR *L_out, int *transp_out, SCALAR *rhs_inout) {
    SFX(qpo_ldlt) f;
    SFX(ldlt_alloc)(&f, size);
    for (int j = 0; j < size; j++)
        for (int i = 0; i < size; i++) f.mat[(size_t)j * size + i] = (i >= j) ? K[(size_t)j * size + i] : (SCALAR)0;
    int ok = SFX(ldlt_compute_inplace)(&f);
    if (L_out) memcpy(L_out, f.mat, sizeof(SCALAR) * (size_t)size * (size_t)size);
    if (transp_out) memcpy(transp_out, f.transp, sizeof(int) * (size_t)size);
    if (rhs_inout) SFX(ldlt_solve_inplace)(&f, rhs_inout);
    SFX(ldlt_free)(&f);
    return ok;
}
