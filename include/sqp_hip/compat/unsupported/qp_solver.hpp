// Drop-in for the reference's include/unsupported/qp_solver.hpp (the fixed-size legacy class QPSolver<QP<n,m,Scalar>>,
// unsupported/qp_solver.hpp:18-49,135-592): `-I <repo>/include/sqp_hip/compat`.  Cannot be combined with solvers/qp.hpp in
// one translation unit — as in the reference, where both headers define qp_solver::QPSolver.
// QP_SOLVER_USE_SPARSE (defined before the include, as in the reference) selects the sparse variant: QP<n,m> with
// Eigen::SparseMatrix P and A; A goes to the device in CSR (sqph_*_csr), P densified.
#pragma once
#define SQP_HIP_LEGACY_API 1
#ifdef QP_SOLVER_USE_SPARSE
#include <Eigen/Sparse>
#endif
#include "../../qp.hpp"
#ifndef SQP_HIP_HAVE_EIGEN
#error "the drop-in header needs Eigen (<Eigen/Dense>) on the include path"
#endif
