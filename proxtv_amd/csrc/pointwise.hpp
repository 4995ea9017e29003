// pointwise.hpp -- host interface of the pointwise / reduction kernels (pointwise.hip).
#pragma once

#include "common.hpp"

namespace ptv {

constexpr int kMaxTerms = 16;  // penalty terms per PD_TV / PDR_TV call held in one kernel-argument pack
struct PtrPack {
    double *v[kMaxTerms];
};

constexpr int kReduceBlocks = 1024;  // partial sums per reduction (fixed => run-to-run deterministic)

// out[b] = sum(a[b*n .. (b+1)*n)) for b < segments       `partials` holds segments * kReduceBlocks doubles
void sum_to(const double *a, long n, long segments, double *partials, double *out, hipStream_t s);
// t[b*n + i] = sign * (2 * sums[b] / n)                    (DR initialisation, src/TV2Dopt.cpp:390-395)
void dr_fill(double *t, long n, long segments, const double *sums, double sign, hipStream_t s);
// *out = sum |a - b|
void absdiff_to(const double *a, const double *b, long n, double *partials, double *out, hipStream_t s);

// PD_TV combine (src/TVNDopt.cpp:212-227): xo = sum_i p_i / P ; z_i += xo - p_i ; *out = sum |xo - x|.
// x and xo may be the same array.
void pd_combine(const PtrPack &p, const PtrPack &z, const double *x, double *xo, int P, long n, double *partials,
                double *out, hipStream_t s);
// PDR_TV combine (src/TVNDopt.cpp:465-484): q = sum p_i/P ; xo = sum z_i/P ; z_i += 2q - xo - p_i ; *out = sum|xo - x|
void pdr_combine(const PtrPack &p, const PtrPack &z, const double *x, double *xo, int P, long n, double *partials,
                 double *out, hipStream_t s);
// x = y / P                                                 (PDR initialisation, src/TVNDopt.cpp:362-367)
void scale_to(const double *y, double *x, double divisor, long n, hipStream_t s);
// dst = src, 8 bytes per lane (counter calibration only)
void calib_copy(const double *src, double *dst, long n, hipStream_t s);
// Yang X update (src/TV2Dopt.cpp:832-833 ; src/TVNDopt.cpp:729-730): X = (Y + sum U_k + rho sum Z_k) / (1 + D rho)
void yang_x(const double *Y, const PtrPack &U, const PtrPack &Z, double *X, int D, double rho, long n, hipStream_t s);

}  // namespace ptv
