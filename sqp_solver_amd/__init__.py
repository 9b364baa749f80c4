"""sqp_solver_amd — MI355X-native batched ADMM QP-subproblem solver behind the sqp_solver API.

Only the QP hot path of msplr/sqp_solver lives here: hand-written HIP kernels
(csrc/), the C-ABI host library (include/sqp_hip.h -> lib/libsqp_hip.so) and this thin Python
mirror of the reference's QPSolver interface.
"""
from . import _capi  # noqa: F401
from .qp import (  # noqa: F401
    EQUALITY_CONSTRAINT, INEQUALITY_CONSTRAINT, LOOSE_BOUNDS, MAX_ITER_EXCEEDED, NUMERICAL_ISSUES,
    SOLVED, UNINITIALIZED, UNSOLVED, QPSolverBatch, SqphError, constr_type_init, default_settings,
)

__all__ = [
    "QPSolverBatch", "SqphError", "constr_type_init", "default_settings",
    "SOLVED", "MAX_ITER_EXCEEDED", "UNSOLVED", "NUMERICAL_ISSUES", "UNINITIALIZED",
    "INEQUALITY_CONSTRAINT", "EQUALITY_CONSTRAINT", "LOOSE_BOUNDS",
]
