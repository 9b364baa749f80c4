// Drop-in for the reference's include/solvers/qp.hpp: put `-I <repo>/include/sqp_hip/compat` where the reference's
// `-I <reference>/include` was and link libsqp_hip.so instead of the reference's sqp_solver library.  Needs Eigen on the
// include path (the reference does too).  Provides qp_solver::QuadraticProblem / QPSolverSettings / QPSolverInfo /
// QPSolverStatus / QPSolver<Scalar> with the reference's members (include/solvers/qp.hpp:19-173).
#pragma once
#include "../../qp.hpp"
#ifndef SQP_HIP_HAVE_EIGEN
#error "the drop-in header needs Eigen (<Eigen/Dense>) on the include path; without Eigen use include/sqp_hip/qp.hpp's raw-pointer API"
#endif
