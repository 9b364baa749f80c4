// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// Runs the product's HIP kernels (sqp_solver_amd/csrc/*.h) under the host SIMT emulator so the
// CPU-only unit tests can compare kernel logic with the oracle. Built into libsqph_sim.so.
#include "hip_sim.h"

#include <limits>
// kernels (device code only; host launch code lives in capi.hip and is not compiled here)
#include "../../sqp_solver_amd/csrc/admm_generic.h"
#include "../../sqp_solver_amd/csrc/admm_lane_kernel.h"
#include "../../sqp_solver_amd/csrc/admm_wg_kernel.h"
#include "../../sqp_solver_amd/csrc/admm_csr_kernel.h"
#include "../../sqp_solver_amd/csrc/admm_csrb_kernel.h"

extern "C" {

// Flat mirror of sqph::KArgs with host pointers and double-typed settings.
struct SimArgs {
    int n, m, batch, mode;
    const void *P, *q, *A, *l, *u;
    long long sP, sq, sA, sl, su;
    void *x, *z, *y, *rho_vec;
    int *ctype;
    void *rho;
    sqph_info *info;
    void *Sinv, *At;
    double rho0, sigma, alpha, eps_rel, eps_abs, rho_tol;
    int max_iter, check_termination, warm_start, adaptive_rho, adaptive_rho_interval;
};

}  // extern "C"

namespace {
template <typename TIN>
sqph::KArgs<double, TIN> convert(const SimArgs &s) {
    using T = double;
    sqph::KArgs<T, TIN> a{};
    a.n = s.n; a.m = s.m; a.batch = s.batch; a.mode = s.mode;
    a.P = (const TIN *)s.P; a.q = (const TIN *)s.q; a.A = (const TIN *)s.A; a.l = (const TIN *)s.l; a.u = (const TIN *)s.u;
    a.sP = s.sP; a.sq = s.sq; a.sA = s.sA; a.sl = s.sl; a.su = s.su;
    a.x = (T *)s.x; a.z = (T *)s.z; a.y = (T *)s.y; a.rho_vec = (T *)s.rho_vec; a.ctype = s.ctype;
    a.rho = (T *)s.rho; a.info = s.info; a.Sinv = (T *)s.Sinv; a.At = (T *)s.At;
    a.rho0 = (T)(TIN)s.rho0; a.sigma = (T)(TIN)s.sigma; a.alpha = (T)(TIN)s.alpha; a.eps_rel = (T)(TIN)s.eps_rel;
    a.eps_abs = (T)(TIN)s.eps_abs; a.rho_tol = (T)(TIN)s.rho_tol;
    a.rho_min = (T)(TIN)1e-6; a.rho_max = (T)(TIN)1e+6; a.eq_tol = (T)(TIN)1e-4; a.rho_eq_factor = (T)(TIN)1e+3;
    a.loose_thresh = (T)(TIN)1e+16; a.regul = (T)std::numeric_limits<TIN>::epsilon();
    a.max_iter = s.max_iter; a.check_termination = s.check_termination; a.warm_start = s.warm_start;
    a.adaptive_rho = s.adaptive_rho; a.adaptive_rho_interval = s.adaptive_rho_interval;
    return a;
}

// measurement only (tests/test_f32_tiles.py): the generic kernel with fp32 ARITHMETIC — state arrays of the SimArgs are float here
int run_generic_f32_arith(const SimArgs &s, int nt) {
    using T = float;
    sqph::KArgs<T, float> a{};
    a.n = s.n; a.m = s.m; a.batch = s.batch; a.mode = s.mode;
    a.P = (const float *)s.P; a.q = (const float *)s.q; a.A = (const float *)s.A; a.l = (const float *)s.l; a.u = (const float *)s.u;
    a.sP = s.sP; a.sq = s.sq; a.sA = s.sA; a.sl = s.sl; a.su = s.su;
    a.x = (T *)s.x; a.z = (T *)s.z; a.y = (T *)s.y; a.rho_vec = (T *)s.rho_vec; a.ctype = s.ctype;
    a.rho = (T *)s.rho; a.info = s.info; a.Sinv = (T *)s.Sinv; a.At = (T *)s.At;
    a.rho0 = (T)s.rho0; a.sigma = (T)s.sigma; a.alpha = (T)s.alpha; a.eps_rel = (T)s.eps_rel; a.eps_abs = (T)s.eps_abs; a.rho_tol = (T)s.rho_tol;
    a.rho_min = 1e-6f; a.rho_max = 1e+6f; a.eq_tol = 1e-4f; a.rho_eq_factor = 1e+3f; a.loose_thresh = 1e+16f;
    a.regul = std::numeric_limits<float>::epsilon();
    a.max_iter = s.max_iter; a.check_termination = s.check_termination; a.warm_start = s.warm_start;
    a.adaptive_rho = s.adaptive_rho; a.adaptive_rho_interval = s.adaptive_rho_interval;
    const size_t lds = sqph::generic_lds_elems<float>(s.n, s.m, nt) * sizeof(float);
    sqph_sim::launch(sqph::admm_generic_kernel<float, float>, dim3(s.batch), dim3(nt), lds, a);
    return 0;
}

template <typename TIN>
int run_generic(const SimArgs &s, int nt) {
    auto a = convert<TIN>(s);
    const size_t lds = sqph::generic_lds_elems<double>(s.n, s.m, nt) * sizeof(double);
    sqph_sim::launch(sqph::admm_generic_kernel<double, TIN>, dim3(s.batch), dim3(nt), lds, a);
    return 0;
}
}  // namespace

extern "C" {

// measurement switch (tests/test_f32_tiles.py): the register-tiled kernels round their B and W' tiles through fp32 (SQPH_TILE_QUANT)
void sim_set_tile_quant(int on) { ::sqph_sim::tile_quant() = on != 0; }

// variant: 0 = generic (nt threads per QP); 2 = workgroup-tiled; 4 / 5 = four / two QPs per wavefront; 6 = one QP per lane
int sim_run(const SimArgs *s, int variant, int dtype, int nt) {
    if (variant == 9) return dtype == SQPH_F32 ? run_generic_f32_arith(*s, nt) : -1;
    if (variant == 12) return dtype == SQPH_F32 ? sqph::sim_run_wgs<float>(convert<float>(*s)) : sqph::sim_run_wgs<double>(convert<double>(*s));  // stacked operator
    if (variant == 11) return dtype == SQPH_F32 ? sqph::sim_run_csrd<float>(convert<float>(*s)) : sqph::sim_run_csrd<double>(convert<double>(*s));  // dense A streamed, W in registers
    if (variant == 10) return dtype == SQPH_F32 ? sqph::sim_run_wgf<float>(convert<float>(*s)) : -1;  // fp32-product register-tiled kernels
    if (variant == 0) return dtype == SQPH_F32 ? run_generic<float>(*s, nt) : run_generic<double>(*s, nt);
    if (variant == 4) return dtype == SQPH_F32 ? sqph::sim_run_g16<float>(convert<float>(*s)) : sqph::sim_run_g16<double>(convert<double>(*s));
    if (variant == 7) return dtype == SQPH_F32 ? sqph::sim_run_lane<float, float>(convert<float>(*s)) : -1;
    if (variant == 13) return dtype == SQPH_F32 ? sqph::sim_run_lane<float>(convert<float>(*s), true) : sqph::sim_run_lane<double>(convert<double>(*s), true);  // four lanes per QP
    if (variant == 6) return dtype == SQPH_F32 ? sqph::sim_run_lane<float>(convert<float>(*s)) : sqph::sim_run_lane<double>(convert<double>(*s));
    if (variant == 2) return dtype == SQPH_F32 ? sqph::sim_run_wg<float>(convert<float>(*s)) : sqph::sim_run_wg<double>(convert<double>(*s));
    return -1;
}

// the sparse-A kernel (admm_csr_kernel.h): A comes as CSR, s->A is ignored
int sim_run_csr(const SimArgs *s, const int *rowptr, const int *colind, const void *val, long long s_rowptr, long long s_colind,
                long long s_val, int nnz_cap, int dtype) {
    if (dtype == SQPH_F32) {
        sqph::CsrArgs<float> ca{rowptr, colind, (const float *)val, s_rowptr, s_colind, s_val, nnz_cap};
        return sqph::sim_run_csr<float>(convert<float>(*s), ca);
    }
    sqph::CsrArgs<double> ca{rowptr, colind, (const double *)val, s_rowptr, s_colind, s_val, nnz_cap};
    return sqph::sim_run_csr<double>(convert<double>(*s), ca);
}

// the block-row sparse kernel (admm_csrb_kernel.h), same arguments
int sim_run_csrb(const SimArgs *s, const int *rowptr, const int *colind, const void *val, long long s_rowptr, long long s_colind,
                 long long s_val, int nnz_cap, int dtype) {
    if (dtype == SQPH_F32) {
        sqph::CsrArgs<float> ca{rowptr, colind, (const float *)val, s_rowptr, s_colind, s_val, nnz_cap};
        return sqph::sim_run_csrb<float>(convert<float>(*s), ca);
    }
    sqph::CsrArgs<double> ca{rowptr, colind, (const double *)val, s_rowptr, s_colind, s_val, nnz_cap};
    return sqph::sim_run_csrb<double>(convert<double>(*s), ca);
}

// ... with P in compressed columns (the sparse-P instantiations, sqph_*_csr_sp)
int sim_run_csrb_sp(const SimArgs *s, const int *rowptr, const int *colind, const void *val, long long s_rowptr, long long s_colind,
                    long long s_val, int nnz_cap, int dtype, const int *pcol, const int *prow, const void *pval, long long s_pcol,
                    long long s_prow, long long s_pval) {
    if (dtype == SQPH_F32) {
        sqph::CsrArgs<float> ca{rowptr, colind, (const float *)val, s_rowptr, s_colind, s_val, nnz_cap, pcol, prow, (const float *)pval, s_pcol, s_prow, s_pval};
        return sqph::sim_run_csrb_sp<float>(convert<float>(*s), ca);
    }
    sqph::CsrArgs<double> ca{rowptr, colind, (const double *)val, s_rowptr, s_colind, s_val, nnz_cap, pcol, prow, (const double *)pval, s_pcol, s_prow, s_pval};
    return sqph::sim_run_csrb_sp<double>(convert<double>(*s), ca);
}
}
