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
// The same two for ANY number of terms (the reference takes any npen: src/TVNDopt.cpp:48-110): the array pointers come from a table in
// HBM ([P] p_i then [P] z_i) instead of the kernel-argument pack; same arithmetic in the same order, one element per lane.
void pd_combine_many(double *const *table, const double *x, double *xo, int P, long n, double *partials, double *out, bool dr_variant,
                     hipStream_t s);
// x = y / P                                                 (PDR initialisation, src/TVNDopt.cpp:362-367)
void scale_to(const double *y, double *x, double divisor, long n, hipStream_t s);
// out = ca a + cb b + cc c + cd d   (null pointers are skipped; out may alias any operand).  The unfused splitting loops
// of the mixed-norm solvers are made of these.
void lincomb(double *out, const double *a, double ca, const double *b, double cb, const double *c, double cc,
             const double *d, double cd, long n, hipStream_t s);
// Tiled transpose of `slabs` column-major matrices stored back to back: out (cols x rows) = in (rows x cols)^T per slab.
// Sweeps along a strided dimension use it to run as dimension-0 sweeps on transposed copies: fibre j = slab * inc + off
// sits at j * len after the transposition of every (inc x len) slab.
// (gate != null: a no-op when *gate == 0, like the gated sweep the copy belongs to)
void slab_transpose(const double *in, double *out, long rows, long cols, long slabs, hipStream_t s, const int *gate = nullptr);
void block_transpose(const double *in, double *out, long rows, long cols, long ld_in, long ld_out, hipStream_t s);
// Edge statistics of an array along one dimension -- what the geometry policy (sweep.hip) is seeded with: a histogram of
// |y[e + inc] - y[e]| (weighted: divided by the edge's penalty) over a fixed sample of edges, kProbeBins bins of an eighth
// of an octave each (bin = biased exponent and three mantissa bits of the value, clamped to the window that starts at 2^-60),
// hist[kProbeBins] = edges sampled.  A second histogram of the same layout follows (hist + kProbeBins + 1): the total
// variation of sampled STRETCHES of 16 consecutive edges of one fibre -- where that is small against lambda a speculative
// walk finds nothing to meet the true walk at, however lively the array is elsewhere.  Integer counts over a sample that
// depends on the shape only: the same array always gives the same histograms.  `hist` (device, kProbeWords words) must be
// zero on entry.
constexpr int kProbeBins = 1024;
constexpr int kProbeWords = 2 * (kProbeBins + 1);
constexpr int kProbeLowExp = 1023 - 60;
inline int probe_bin(double v) {   // host side of the same binning
    unsigned long long b;
    memcpy(&b, &v, 8);
    const long k = (long)((b & 0x7fffffffffffffffull) >> 49) - ((long)kProbeLowExp << 3);
    return k < 0 ? 0 : (k >= kProbeBins ? kProbeBins - 1 : (int)k);
}
// (y2 != null: of the array y + c2 y2)
void edge_histogram(const double *y, const double *w, long n, long inc, int len, unsigned *hist, hipStream_t s, const double *y2 = nullptr, double c2 = 0.0);
// dst = src, 8 bytes per lane (counter calibration only)
void calib_copy(const double *src, double *dst, long n, hipStream_t s);
// Yang X update (src/TV2Dopt.cpp:832-833 ; src/TVNDopt.cpp:729-730): X = (Y + sum U_k + rho sum Z_k) / (1 + D rho)
void yang_x(const double *Y, const PtrPack &U, const PtrPack &Z, double *X, int D, double rho, long n, hipStream_t s);

// ---- Kolmogorov2_TV (src/TV2Dopt.cpp:907-1024).  The dual is kept unscaled: U = su * D with D = V - colprox(V).
// `gate` (may be null): the kernel is a no-op when *gate == 0.
// V = (su*D + sigma (X + theta (X - Xold))) / sigma     (:963-966); if `changed`: *changed = 1 when X differs from Xold
// anywhere ((Xold - X)^2 > 0: the reference's exit test `stop > 0`, :1007-1013)
void kolmo_dual_in(const double *D, double su, const double *X, const double *Xold, double sigma, double theta, double *V,
                   long n, const int *gate, int *changed, hipStream_t s);
// V = c1 (Y + c2 (X - tau su D)),  c1 = 1/(1+1/tau), c2 = 1/tau        (:981-984)
void kolmo_primal_in(const double *X, const double *D, double su, const double *Y, double tau, double c1, double c2,
                     double *V, long n, const int *gate, hipStream_t s);

// ---- CondatChambollePock2_TV (src/TV2Dopt.cpp:587-760): M x N column-major image, duals U1 ((M-1) x N), U2 (M x (N-1)).
void ccp_init(const double *Y, double *U1, double *U2, long M, long N, hipStream_t s);
// One iteration, fused: (unless `first`) the dual ascent + clip of the PREVIOUS iteration from Zold into U1n/U2n
// (every thread recomputes the four duals around its pixel, writes the two it owns), then the primal step
// (alg 0: X - tau (X - Y + g) ; 1, 2: (X + tau (Y - g)) / (1 + tau)) into Xn and the extrapolation
// Zn = Xn + theta (Xn - X).  *changed = 1 when Xn differs from X anywhere.
struct CcpArgs {
    const double *Y, *X, *Zold, *U1o, *U2o;
    double *Xn, *Zn, *U1n, *U2n;
    long M, N;
    double tau, theta, sigma, lambda;
    int alg;
    const int *gate;
    int *changed;
};
void ccp_step(const CcpArgs &a, bool first, hipStream_t s);

}  // namespace ptv
