// Drop-in for the reference's include/solvers/bfgs.hpp: BFGS_update(Mat&, const Vec&, const Vec&) (bfgs.hpp:14-41) on Eigen
// objects, forwarded to the driver's column-major routine (include/sqp_hip/sqp.hpp).
#pragma once
#include <vector>

#include <Eigen/Dense>

#include "../../sqp.hpp"

template <typename Mat, typename Vec>
void BFGS_update(Mat &B, const Vec &s, const Vec &y) {
    using Scalar = typename Mat::Scalar;
    const int n = (int)B.rows();
    std::vector<Scalar> w((size_t)2 * n);
    sqp::raw::bfgs_update(B.data(), n, s.data(), y.data(), w.data(), w.data() + n);
}
