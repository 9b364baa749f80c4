// Drop-in for the reference's include/unsupported/qp_solver.hpp (the fixed-size legacy class QPSolver<QP<n,m,Scalar>>,
// unsupported/qp_solver.hpp:18-49,135-592): `-I <repo>/include/sqp_hip/compat`.  Cannot be combined with solvers/qp.hpp in
// one translation unit — as in the reference, where both headers define qp_solver::QPSolver.  Dense QP<n,m> only
// (QP_SOLVER_USE_SPARSE selects Eigen::SparseMatrix members in the reference; the CSR entry points of
// qp_solver::supported::BatchQPSolver are the sparse path here).
#pragma once
#ifdef QP_SOLVER_USE_SPARSE
#error "QP_SOLVER_USE_SPARSE: use qp_solver::BatchQPSolver::*_csr (include/sqp_hip/qp.hpp) for sparse constraint matrices"
#endif
#define SQP_HIP_LEGACY_API 1
#include "../../qp.hpp"
#ifndef SQP_HIP_HAVE_EIGEN
#error "the drop-in header needs Eigen (<Eigen/Dense>) on the include path"
#endif
