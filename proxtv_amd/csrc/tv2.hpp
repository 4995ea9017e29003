// tv2.hpp -- batched exact 1-D TV-L2 prox over fibres (tv2.hip).
#pragma once

#include "common.hpp"

namespace ptv {

// out = argmin 1/2 ||x - in||^2 + lam ||Dx||_2 along dimension `dim` of every fibre of the N-D column-major array.
// `in` and `out` must be distinct arrays.
void tv2_fibres(const double *in, double *out, const int *ns, int nds, int dim, double lam, hipStream_t s);

}  // namespace ptv
