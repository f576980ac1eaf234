// Row-tile-resident fused pass: one read of X per component instead of two (and, for the
// NIPALS algo, the deflation write folded into the same sweep).  See DESIGN.md section 4.
#pragma once
#include "common.hpp"

namespace plsk {

// rc: 0 = launched, 1 = shape/alignment not covered (caller falls back to the one-product
// kernels), <0 = launch error.
template <typename T>
int launch_fused_pass(hipStream_t, int /*num_cu*/, const T * /*X*/, i64 /*ldx*/, T * /*dst*/,
                      i64 /*ldd*/, i64 /*N*/, int /*K*/, const double * /*v*/, const T * /*tprev*/,
                      const double * /*pprev*/, T * /*tout*/, double * /*part*/, int /*max_rows*/,
                      double * /*sspart*/, int * /*nb*/, int * /*nss*/) {
    return 1;
}

}  // namespace plsk
