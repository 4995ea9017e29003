// solvers.hpp -- device-resident splitting loops (solvers.hip).  All pointers are HBM pointers.
#pragma once

#include "common.hpp"

namespace ptv {

struct SolveInfo {
    double iters = 0;
    double gap = 0;
    bool gap_set = false;  // DR / Yang leave info[INFO_GAP] untouched, like the reference
    int rc = RC_OK;
};

// out = prox along dimension `dim` of every fibre of `in` (uniform lam, or per-edge `weights`)
void tv1_fibres(const double *in, double *out, const int *ns, int nds, int dim, double lam, const double *weights,
                hipStream_t s);

// the certificate of such a sweep: fibres of `out` that are NOT the prox of the same fibre of `in` (-1: cannot be checked: lam <= 0, in == out)
long certify_fibres(const double *in, const double *out, const int *ns, int nds, int dim, double lam, const double *weights, hipStream_t s);

// out = prox along `dim` with the TV-L1 (norm 1: the sweep kernels) or TV-L2 (norm 2: tv2.hip) penalty; in != out
void prox_fibres(const double *in, double *out, const int *ns, int nds, int dim, double lam, double norm, hipStream_t s);

// Douglas-Rachford on B stacked MxN images (B = 1: DR2_TV / DR2L1W_TV).  W1m/W2m != nullptr selects the weighted
// solver (per-edge penalties, (M-1)xN and Mx(N-1) per image); otherwise scalars W1 (columns) / W2 (rows).
SolveInfo dr2(size_t M, size_t N, size_t B, const double *unary, double W1, double W2, const double *W1m,
              const double *W2m, double *out, int maxit, hipStream_t s);
// the same loop with a TV-L2 penalty in at least one direction (norms in {1, 2}): unfused steps
SolveInfo dr2_norms(size_t M, size_t N, const double *unary, double W1, double W2, double norm1, double norm2, double *out,
                    int maxit, hipStream_t s);

// `norms` (may be null = all 1): one of {1, 2} per penalty term
SolveInfo pd2(const double *y, const double *lambdas, const double *dims, double *x, const int *ns, int nds, int npen,
              int maxIters, hipStream_t s, const double *norms = nullptr);
// lambdas already scaled by npen (the C-ABI wrapper does the in-place scaling the reference does)
SolveInfo pd(const double *y, const double *lambdas, const double *dims, double *x, const int *ns, int nds, int npen,
             int maxIters, hipStream_t s, const double *norms = nullptr);
SolveInfo pdr(const double *y, const double *lambdas, const double *dims, double *x, const int *ns, int nds, int npen,
              int maxIters, hipStream_t s, const double *norms = nullptr);
// order[k] = 0-based dimension of the k-th (Z_k, U_k) pair, lambdas[k] its penalty
SolveInfo yang(const int *ns, int nds, const int *order, const double *lambdas, const double *Y, double *X, int maxit,
               hipStream_t s);

// Kolmogorov et al.'s primal-dual splitting, 2-D (src/TV2Dopt.cpp:907-1024)
SolveInfo kolmogorov2(size_t M, size_t N, const double *Y, double lambda, double *X, int maxit, hipStream_t s);
// Condat / Chambolle-Pock / accelerated Chambolle-Pock, 2-D (src/TV2Dopt.cpp:587-760); alg = 0 / 1 / 2
SolveInfo ccp2(size_t M, size_t N, const double *Y, double lambda, double *X, int alg, int maxit, hipStream_t s);

}  // namespace ptv
