/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See qp_oracle_impl.h / qp_oracle.h.
 * Instantiates the restatement for double and float, like src/qp.cpp:385-386.
 */
#include "qp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* defaults of QPSolverSettings, qp.hpp:38-53 */
void qpo_default_settings(qpo_settings *s) {
    s->rho = 1e-1;
    s->sigma = 1e-6;
    s->alpha = 1.0;
    s->eps_rel = 1e-3;
    s->eps_abs = 1e-3;
    s->max_iter = 1000;
    s->check_termination = 25;
    s->warm_start = 0;
    s->adaptive_rho = 0;
    s->adaptive_rho_tolerance = 5;
    s->adaptive_rho_interval = 25;
    s->verbose = 0;
}

int qpo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_num_procs();  /* (not omp_get_max_threads: a solve with an explicit thread count lowers that for good) */
#else
    return 1;
#endif
}

#define SCALAR double
#define SCALAR_EPS DBL_EPSILON
#define SCALAR_MIN DBL_MIN
#define SABS(x) fabs(x)
#define SSQRT(x) sqrt(x)
#define SFX(name) name##_f64
#include "qp_oracle_impl.h"
#undef SCALAR
#undef SCALAR_EPS
#undef SCALAR_MIN
#undef SABS
#undef SSQRT
#undef SFX

/* x87 80-bit extended precision instance: NOT a reference instantiation. It is the yard-stick the
 * tests use to tell how far the fp64 reference path itself is from the exact ADMM iterate on
 * ill-conditioned problems (adaptive rho can push rho_eq = 1e3*rho to 1e6+). DIV_BY_ZERO_REGUL and
 * the pseudo-inverse threshold keep their double values so the trajectory is the double one. */
#define SCALAR long double
#define SCALAR_EPS DBL_EPSILON
#define SCALAR_MIN DBL_MIN
#define SABS(x) fabsl(x)
#define SSQRT(x) sqrtl(x)
#define SFX(name) name##_f80
#include "qp_oracle_impl.h"
#undef SCALAR
#undef SCALAR_EPS
#undef SCALAR_MIN
#undef SABS
#undef SSQRT
#undef SFX

#define SCALAR float
#define SCALAR_EPS FLT_EPSILON
#define SCALAR_MIN FLT_MIN
#define SABS(x) fabsf(x)
#define SSQRT(x) ((float)sqrt((double)(x)))
#define SFX(name) name##_f32
#include "qp_oracle_impl.h"
#undef SCALAR
#undef SCALAR_EPS
#undef SCALAR_MIN
#undef SFX
