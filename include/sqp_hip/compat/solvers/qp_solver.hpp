// The reference's legacy tests include the fixed-size class under this (older) path, tests/unsupported/qp_solver_test.cpp:2.
#pragma once
#include "../unsupported/qp_solver.hpp"
