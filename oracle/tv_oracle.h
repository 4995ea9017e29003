/*
 * tv_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's algorithms for the hot path
 * (batched 1-D TV-L1 prox wrapped by the DR / PD / Yang / PDR splitting loops).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The shipped library (proxtv_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked against
 *   (a) oracle/_ref/libproxtv_ref.so  -- the unmodified reference sources
 *       compiled where they lie (recipe: oracle/Makefile, target `ref`), and
 *   (b) the golden vectors in tests/golden/ generated from (a) by
 *       oracle/gen_golden.py.
 *
 * All reference citations are relative to /root/reference.
 */
#ifndef TV_ORACLE_H
#define TV_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* info[] layout and return codes: src/general.h:58-73 */
#define ORC_INFO_ITERS 0
#define ORC_INFO_GAP   1
#define ORC_INFO_RC    2
#define ORC_RC_OK    0
#define ORC_RC_ITERS 1
#define ORC_RC_STUCK 2
#define ORC_RC_ERROR 3

/* iteration caps / tolerances: src/TVopt.h:70-85 */
#define ORC_STOP_PD        1e-6
#define ORC_MAX_ITERS_PD   35
#define ORC_MAX_ITERS_DR   35
#define ORC_MAX_ITERS_YANG 35
#define ORC_MAX_ITERS_CONDAT 2500      /* src/TVopt.h:75 */
#define ORC_STOP_CONDAT 0              /* src/TVopt.h:77 */
#define ORC_MAX_ITERS_KOLMOGOROV 2500  /* src/TVopt.h:79 */
#define ORC_STOP_KOLMOGOROV 0          /* src/TVopt.h:81 */
/* absolute comparison tolerance: src/general.h:64-67 */
#define ORC_EPSILON 1e-10

/* ---- 1-D solvers (oracle/tv1d_oracle.c) ---- */
/* src/TVL1opt.cpp:359-564 */
int  orc_linearizedTautString_TV1(const double *y, double lambda, double *x, int n);
/* src/TVL1opt_hybridtautstring.cpp:56-239 */
void orc_hybridTautString_TV1(const double *y, int n, double lambda, double *x);
void orc_hybridTautString_TV1_custom(const double *y, int n, double lambda, double *x, double backtracksexp);
/* src/TVL1opt_tautstring.cpp:256-357 */
int  orc_classicTautString_TV1(const double *signal, int n, double lam, double *prox);
int  orc_classicTautString_TV1_offset(const double *signal, int n, double lam, double *prox, double offset);
/* src/TVL1Wopt.cpp:364-567 */
int  orc_tautString_TV1_Weighted(const double *y, const double *lambda, double *x, int n);
/* src/condat_fast_tv.cpp:78-121 */
void orc_TV1D_denoise(const double *input, double *output, int width, double lambda);
/* src/johnsonRyanTV.cpp:9-116 -- Johnson's dynamic programme (tv1_1d method 'dp'), an independent exact algorithm */
void orc_dp(int n, const double *y, double lam, double *beta);
/* src/TVL2opt.cpp:190-445 (morePG_TV2), solved to convergence from mu = 0 for every fibre: no projected-gradient
   prelude, no 1e-5 gap stop, no warm start across fibres (see tv1d_oracle.c) */
int  orc_TV2_exact(const double *y, double lambda, double *x, double *info, int n);
/* src/TVgenopt.cpp:30-57: p == 1 (bit-identical) and p == 2 (exact TV-L2 prox); other p -> RC_ERROR */
int  orc_TV(const double *y, double lambda, double *x, double *info, int n, double p);

/* ---- combiners (oracle/tvnd_oracle.c) ---- */
/* src/TV2Dopt.cpp:352-547 */
int orc_DR2_TV(size_t M, size_t N, const double *unary, double W1, double W2, double norm1, double norm2,
               double *s, int nThreads, int maxit, double *info);
/* src/TV2DWopt.cpp:46-240 */
int orc_DR2L1W_TV(size_t M, size_t N, const double *unary, const double *W1, const double *W2,
                  double *s, int nThreads, int maxit, double *info);
/* src/TV2Dopt.cpp:59-302 */
int orc_PD2_TV(const double *y, const double *lambdas, const double *norms, const double *dims, double *x,
               double *info, const int *ns, int nds, int npen, int ncores, int maxIters);
/* src/TVNDopt.cpp:48-252 -- scales lambdas[] in caller memory, like the reference */
int orc_PD_TV(const double *y, double *lambdas, const double *norms, const double *dims, double *x,
              double *info, const int *ns, int nds, int npen, int ncores, int maxIters);
/* src/TVNDopt.cpp:280-500 -- scales lambdas[] in caller memory, like the reference */
int orc_PDR_TV(const double *y, double *lambdas, const double *norms, const double *dims, double *x,
               double *info, const int *ns, int nds, int npen, int ncores, int maxIters);
/* src/TV2Dopt.cpp:787-877 */
int orc_Yang2_TV(size_t M, size_t N, const double *Y, double lambda, double *X, int maxit, double *info);
/* src/TVNDopt.cpp:678-803 */
int orc_Yang3_TV(size_t M, size_t N, size_t O, const double *Y, double lambda, double *X, int maxit, double *info);
/* extension used only to check the per-dimension-lambda Yang variant of the product */
/* 2-D primal-dual baselines of the same problem (src/TVopt.h:129-131) */
int orc_Kolmogorov2_TV(size_t M, size_t N, const double *Y, double lambda, double *X, int maxit, double *info);
int orc_CondatChambollePock2_TV(size_t M, size_t N, const double *Y, double lambda, double *X, short alg, int maxit,
                                double *info);

int orc_Yang3_TV_perdim(size_t M, size_t N, size_t O, const double *Y, const double *lambda3, double *X,
                        int maxit, double *info);

#ifdef __cplusplus
}
#endif
#endif
