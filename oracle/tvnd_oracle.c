/*
 * tvnd_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's 2-D / N-D anisotropic TV-L1
 * splitting loops (Douglas-Rachford, proximal Dykstra, parallel proximal
 * Dykstra, parallel Douglas-Rachford, Yang ADMM).  Arrays are column-major
 * (dimension 0 fastest), exactly like the reference.  Norms p == 1 (the hot
 * path, bit-identical to the reference) and p == 2 (fibres go through the EXACT
 * TV-L2 prox orc_TV2_exact, see tv1d_oracle.c: the reference's own p == 2 solver
 * stops at a duality gap of 1e-5 and warm-starts across fibres) are covered;
 * any other norm returns RC_ERROR.
 * See tv_oracle.h for the parity status.
 */
#include "tv_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static void set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n < 1 ? 1 : n);
#else
    (void)n;
#endif
}

static int fail(const char *who, const char *what, double *info)
{
    printf("%s: %s\n", who, what);
    if (info) info[ORC_INFO_RC] = ORC_RC_ERROR;
    return 0;
}

/* ---- strided fibre helpers ------------------------------------------------ */

/* Walks every 1-D fibre of an N-D column-major array along dimension d:
   fibre j starts at (j / inc) * inc * len + (j % inc) and has stride inc
   (src/TV2Dopt.cpp:140-145,184 ; src/TVNDopt.cpp:133-138,184). */
typedef struct { long inc, len, count; } fibres_t;

static fibres_t fibres_along(const int *ns, int nds, int d)
{
    fibres_t f;
    long n = 1, inc = 1;
    for (int i = 0; i < nds; i++) n *= ns[i];
    for (int i = 0; i < d; i++) inc *= ns[i];
    f.inc = inc; f.len = ns[d]; f.count = n / ns[d];
    return f;
}

static inline long fibre_start(const fibres_t *f, long j)
{
    return (j / f->inc) * f->inc * f->len + (j % f->inc);
}

/* out_fibre = prox_lambda( a[.] + sb * b[.] ) along dimension d, for every fibre.
   b may be NULL.  solver: 0 = hybrid taut string via TV() ; 1 = Condat. */
static void sweep_prox_p(const double *a, const double *b, double sb, double *out,
                         const int *ns, int nds, int d, double lambda, int solver, double pnorm)
{
    const fibres_t f = fibres_along(ns, nds, d);
    #pragma omp parallel
    {
        double *in = (double *)malloc(sizeof(double) * (size_t)f.len);
        double *res = (double *)malloc(sizeof(double) * (size_t)f.len);
        #pragma omp for
        for (long j = 0; j < f.count; j++) {
            const long base = fibre_start(&f, j);
            for (long k = 0; k < f.len; k++) {
                const long idx = base + k * f.inc;
                in[k] = b ? a[idx] + sb * b[idx] : a[idx];
            }
            if (solver == 1 && pnorm == 1) orc_TV1D_denoise(in, res, (int)f.len, lambda);
            else                           orc_TV(in, lambda, res, NULL, (int)f.len, pnorm);
            for (long k = 0; k < f.len; k++) out[base + k * f.inc] = res[k];
        }
        free(in);
        free(res);
    }
}

static void sweep_prox(const double *a, const double *b, double sb, double *out,
                       const int *ns, int nds, int d, double lambda, int solver)
{
    sweep_prox_p(a, b, sb, out, ns, nds, d, lambda, solver, 1);
}

/* ------------------------------------------------------------------------- */
/*  DR2_TV : alternating reflections on the two base polytopes                */
/*  src/TV2Dopt.cpp:352-547                                                   */
/* ------------------------------------------------------------------------- */

/* projection onto B_cols: out = in - colprox(in)   (:459-481, :539-547) */
static void dr_cols(size_t M, size_t N, const double *in, double *out, double W, const double *Wmat, double pnorm)
{
    #pragma omp parallel
    {
        double *res = (double *)malloc(sizeof(double) * M);
        #pragma omp for
        for (long j = 0; j < (long)N; j++) {
            const double *col = in + M * (size_t)j;
            if (Wmat) orc_tautString_TV1_Weighted(col, Wmat + (M - 1) * (size_t)j, res, (int)M);
            else      orc_TV(col, W, res, NULL, (int)M, pnorm);
            for (size_t i = 0; i < M; i++) out[M * (size_t)j + i] = col[i] - res[i];
        }
        free(res);
    }
}

/* projection onto B_{-rows*}: (:498-523).  sign = +1: out = ref - (v - rowprox(v)), v = ref - in
   (unweighted) ; sign = -1: out = (v - rowprox(v)) - ref  (weighted, src/TV2DWopt.cpp:191-221) */
static void dr_rows(size_t M, size_t N, const double *in, double *out, const double *ref,
                    double W, const double *Wmat, int sign, double pnorm)
{
    #pragma omp parallel
    {
        double *v = (double *)malloc(sizeof(double) * N);
        double *res = (double *)malloc(sizeof(double) * N);
        double *wl = (double *)malloc(sizeof(double) * (N ? N : 1));
        #pragma omp for
        for (long j = 0; j < (long)M; j++) {
            for (size_t i = 0; i < N; i++) v[i] = ref[(size_t)j + M * i] - in[(size_t)j + M * i];
            if (Wmat) {
                for (size_t i = 0; i + 1 < N; i++) wl[i] = Wmat[(size_t)j + M * i];
                orc_tautString_TV1_Weighted(v, wl, res, (int)N);
            } else {
                orc_TV(v, W, res, NULL, (int)N, pnorm);
            }
            for (size_t i = 0; i < N; i++) {
                const double diff = v[i] - res[i];
                const size_t idx = (size_t)j + M * i;
                out[idx] = (sign > 0) ? ref[idx] - diff : diff - ref[idx];
            }
        }
        free(v); free(res); free(wl);
    }
}

static int dr_generic(const char *who, size_t M, size_t N, const double *unary, double W1, double W2,
                      const double *W1m, const double *W2m, double *s, int nThreads, int maxit, double *info,
                      double norm1, double norm2)
{
    const size_t n = M * N;
    const int weighted = (W1m != NULL);
    set_threads(nThreads);
    double *t = (double *)malloc(sizeof(double) * n);
    double *tb = (double *)malloc(sizeof(double) * n);
    if (!t || !tb) { free(t); free(tb); return fail(who, "out of memory", info); }
    if (maxit <= 0) maxit = ORC_MAX_ITERS_DR;

    /* t = 2 * mean(unary): serial left-to-right sum (:389-395) */
    double sum = 0;
    for (size_t i = 0; i < n; i++) sum += unary[i];
    sum = 2 * sum / n;
    for (size_t i = 0; i < n; i++) t[i] = sum;

    int iter = 0;
    while (iter < maxit) {
        iter++;
        dr_cols(M, N, t, s, W1, W1m, norm1);                                  /* :408 */
        for (size_t i = 0; i < n; i++) s[i] = 2 * s[i] - t[i];         /* :411 */
        dr_rows(M, N, s, tb, unary, W2, W2m, weighted ? -1 : +1, norm2);      /* :417 */
        if (weighted) for (size_t i = 0; i < n; i++) tb[i] = -2 * tb[i] - s[i];   /* TV2DWopt.cpp:117 */
        else          for (size_t i = 0; i < n; i++) tb[i] = 2 * tb[i] - s[i];    /* :419 */
        for (size_t i = 0; i < n; i++) t[i] = 0.5 * (t[i] + tb[i]);    /* :422 */
    }
    /* recovery projection (:427-430 ; TV2DWopt.cpp:124-126) */
    dr_cols(M, N, t, s, W1, W1m, norm1);
    dr_rows(M, N, s, tb, unary, W2, W2m, weighted ? -1 : +1, norm2);
    if (weighted) for (size_t i = 0; i < n; i++) s[i] = -s[i] - tb[i];
    else          for (size_t i = 0; i < n; i++) s[i] = tb[i] - s[i];

    if (info) { info[ORC_INFO_ITERS] = iter; info[ORC_INFO_RC] = ORC_RC_OK; }
    free(t); free(tb);
    return 0;   /* sic: DR returns 0 on success (:440) */
}

int orc_DR2_TV(size_t M, size_t N, const double *unary, double W1, double W2, double norm1, double norm2,
               double *s, int nThreads, int maxit, double *info)
{
    if ((norm1 != 1 && norm1 != 2) || (norm2 != 1 && norm2 != 2))
        return fail("DR2_TV(oracle)", "only p == 1 and p == 2 are covered", info);
    return dr_generic("DR2_TV", M, N, unary, W1, W2, NULL, NULL, s, nThreads, maxit, info, norm1, norm2);
}

int orc_DR2L1W_TV(size_t M, size_t N, const double *unary, const double *W1, const double *W2,
                  double *s, int nThreads, int maxit, double *info)
{
    return dr_generic("DR2L1W_TV", M, N, unary, 0, 0, W1, W2, s, nThreads, maxit, info, 1, 1);
}

/* ------------------------------------------------------------------------- */
/*  PD2_TV : proximal Dykstra with one or two terms, src/TV2Dopt.cpp:59-302   */
/* ------------------------------------------------------------------------- */
static long total_size(const int *ns, int nds)
{
    long n = 1;
    for (int i = 0; i < nds; i++) n *= ns[i];
    return n;
}

static double mean_abs_change(const double *a, const double *b, long n)
{
    double acc = 0;
    #pragma omp parallel for reduction(+:acc)
    for (long k = 0; k < n; k++) acc += fabs(a[k] - b[k]);
    return acc / n;
}

int orc_PD2_TV(const double *y, const double *lambdas, const double *norms, const double *dims, double *x,
               double *info, const int *ns, int nds, int npen, int ncores, int maxIters)
{
    set_threads(ncores);
    if (maxIters <= 0) maxIters = ORC_MAX_ITERS_PD;
    if (npen > 2) return fail("PD2_TV", "this algorithm can not work with more than 2 penalties", info);
    for (int i = 0; i < npen; i++)
        if (norms[i] != 1 && norms[i] != 2) return fail("PD2_TV(oracle)", "only p == 1 and p == 2 are covered", info);

    const long n = total_size(ns, nds);
    double *p = (double *)calloc((size_t)n, sizeof(double));
    double *q = (double *)calloc((size_t)n, sizeof(double));
    double *z = (double *)malloc(sizeof(double) * (size_t)n);
    double *xl = (double *)malloc(sizeof(double) * (size_t)n);
    if (!p || !q || !z || !xl) { free(p); free(q); free(z); free(xl); return fail("PD2_TV", "out of memory", info); }
    memcpy(x, y, sizeof(double) * (size_t)n);                          /* :132-137 */

    double stop = DBL_MAX;
    int iters = 0;
    while (stop > ORC_STOP_PD && (npen > 1 || !iters) && iters < maxIters) {   /* :157 */
        memcpy(xl, x, sizeof(double) * (size_t)n);
        /* z = prox_{d0}(x + p) ; p += x - z   (:169-213) */
        sweep_prox_p(x, p, 1.0, z, ns, nds, (int)(dims[0] - 1), lambdas[0], 0, norms[0]);
        for (long i = 0; i < n; i++) p[i] += x[i] - z[i];
        if (npen >= 2) {
            /* x = prox_{d1}(z + q) ; q += z - x   (:216-263) */
            sweep_prox_p(z, q, 1.0, x, ns, nds, (int)(dims[1] - 1), lambdas[1], 0, norms[1]);
            for (long i = 0; i < n; i++) q[i] += z[i] - x[i];
        } else {
            memcpy(x, z, sizeof(double) * (size_t)n);                  /* :265-270 */
        }
        stop = mean_abs_change(x, xl, n);                              /* :273-277 */
        iters++;
    }
    if (info) {
        info[ORC_INFO_ITERS] = iters;
        info[ORC_INFO_GAP] = stop;
        info[ORC_INFO_RC] = (iters >= ORC_MAX_ITERS_PD) ? ORC_RC_ITERS : ORC_RC_OK;   /* :289, macro not argument */
    }
    free(p); free(q); free(z); free(xl);
    return 1;
}

/* ------------------------------------------------------------------------- */
/*  PD_TV : parallel proximal Dykstra, src/TVNDopt.cpp:48-252                  */
/* ------------------------------------------------------------------------- */
static int alloc_family(double ***fam, int npen, long n)
{
    *fam = (double **)calloc((size_t)npen, sizeof(double *));
    if (!*fam) return 0;
    for (int i = 0; i < npen; i++) {
        (*fam)[i] = (double *)malloc(sizeof(double) * (size_t)n);
        if (!(*fam)[i]) return 0;
    }
    return 1;
}

static void free_family(double **fam, int npen)
{
    if (!fam) return;
    for (int i = 0; i < npen; i++) free(fam[i]);
    free(fam);
}

int orc_PD_TV(const double *y, double *lambdas, const double *norms, const double *dims, double *x,
              double *info, const int *ns, int nds, int npen, int ncores, int maxIters)
{
    set_threads(ncores);
    if (maxIters <= 0) maxIters = ORC_MAX_ITERS_PD;
    for (int i = 0; i < npen; i++)
        if (norms[i] != 1 && norms[i] != 2) return fail("PD_TV(oracle)", "only p == 1 and p == 2 are covered", info);
    const long n = total_size(ns, nds);

    for (int i = 0; i < npen; i++) lambdas[i] *= npen;                 /* :100-101, caller memory */

    double **p = NULL, **z = NULL;
    double *xl = (double *)malloc(sizeof(double) * (size_t)n);
    if (!alloc_family(&p, npen, n) || !alloc_family(&z, npen, n) || !xl) {
        free_family(p, npen); free_family(z, npen); free(xl);
        return fail("PD_TV", "out of memory", info);
    }
    for (long k = 0; k < n; k++) x[k] = 0;                             /* :126-130 */
    for (int i = 0; i < npen; i++) memcpy(z[i], y, sizeof(double) * (size_t)n);

    double stop = DBL_MAX;
    int iters = 0;
    while (stop > ORC_STOP_PD && iters < maxIters) {                   /* :151 */
        for (long k = 0; k < n; k++) { xl[k] = x[k]; x[k] = 0; }
        for (int i = 0; i < npen; i++)                                 /* :164-209 */
            sweep_prox_p(z[i], NULL, 0, p[i], ns, nds, (int)(dims[i] - 1), lambdas[i], 0, norms[i]);
        for (long k = 0; k < n; k++)                                   /* :212-214 */
            for (int i = 0; i < npen; i++) x[k] += p[i][k] / npen;
        for (long k = 0; k < n; k++)                                   /* :217-220 */
            for (int i = 0; i < npen; i++) z[i][k] += x[k] - p[i][k];
        stop = mean_abs_change(x, xl, n);                              /* :223-227 */
        iters++;
    }
    if (info) {
        info[ORC_INFO_ITERS] = iters;
        info[ORC_INFO_GAP] = stop;
        info[ORC_INFO_RC] = (iters >= ORC_MAX_ITERS_PD) ? ORC_RC_ITERS : ORC_RC_OK;   /* :239 */
    }
    free_family(p, npen); free_family(z, npen); free(xl);
    return 1;
}

/* ------------------------------------------------------------------------- */
/*  PDR_TV : parallel Douglas-Rachford with Condat inner solver               */
/*  src/TVNDopt.cpp:280-500                                                   */
/* ------------------------------------------------------------------------- */
int orc_PDR_TV(const double *y, double *lambdas, const double *norms, const double *dims, double *x,
               double *info, const int *ns, int nds, int npen, int ncores, int maxIters)
{
    set_threads(ncores);
    if (maxIters <= 0) maxIters = ORC_MAX_ITERS_DR;
    for (int i = 0; i < npen; i++)
        if (norms[i] != 1 && norms[i] != 2) return fail("PDR_TV(oracle)", "only p == 1 and p == 2 are covered", info);
    const long n = total_size(ns, nds);

    for (int i = 0; i < npen; i++) lambdas[i] *= npen;                 /* :334-335, caller memory */

    double **p = NULL, **z = NULL;
    double *q = (double *)malloc(sizeof(double) * (size_t)n);
    double *xl = (double *)malloc(sizeof(double) * (size_t)n);
    if (!alloc_family(&p, npen, n) || !alloc_family(&z, npen, n) || !xl || !q) {
        free_family(p, npen); free_family(z, npen); free(xl); free(q);
        return fail("PDR_TV", "out of memory", info);
    }
    for (long k = 0; k < n; k++) x[k] = y[k] / npen;                   /* :362-367 */
    for (int i = 0; i < npen; i++) memcpy(z[i], y, sizeof(double) * (size_t)n);

    double stop = 0;
    int iters = 0;
    while (iters < maxIters) {                                         /* :390 */
        for (long k = 0; k < n; k++) { xl[k] = x[k]; x[k] = 0; q[k] = 0; }
        for (int i = 0; i < npen; i++)                                 /* :405-458, Condat inner solver :438-439 */
            sweep_prox_p(z[i], NULL, 0, p[i], ns, nds, (int)(dims[i] - 1), lambdas[i], 1, norms[i]);
        for (long k = 0; k < n; k++)                                   /* :465-470 */
            for (int i = 0; i < npen; i++) { q[k] += p[i][k] / npen; x[k] += z[i][k] / npen; }
        for (long k = 0; k < n; k++)                                   /* :474-477 */
            for (int i = 0; i < npen; i++) z[i][k] += 2 * q[k] - x[k] - p[i][k];
        stop = mean_abs_change(x, xl, n);                              /* :480-484 */
        iters++;
    }
    if (info) {
        info[ORC_INFO_ITERS] = iters;
        info[ORC_INFO_GAP] = stop;
        info[ORC_INFO_RC] = (iters >= ORC_MAX_ITERS_DR) ? ORC_RC_ITERS : ORC_RC_OK;   /* :495 */
    }
    free_family(p, npen); free_family(z, npen); free(xl); free(q);
    return 1;
}

/* ------------------------------------------------------------------------- */
/*  Yang ADMM, rho = 10                                                       */
/*  2-D: src/TV2Dopt.cpp:787-877 ; 3-D: src/TVNDopt.cpp:678-803               */
/* ------------------------------------------------------------------------- */
static int yang_generic(const char *who, const int *ns, int nds, const int *order, const double *lams,
                        const double *Y, double *X, int maxit, double *info)
{
    const double rho = 10;
    const long n = total_size(ns, nds);
    double **U = NULL, **Z = NULL;
    if (!alloc_family(&U, nds, n) || !alloc_family(&Z, nds, n)) {
        free_family(U, nds); free_family(Z, nds);
        return fail(who, "insufficient memory", info);
    }
    for (int k = 0; k < nds; k++) {
        memset(U[k], 0, sizeof(double) * (size_t)n);
        memcpy(Z[k], Y, sizeof(double) * (size_t)n);
    }
    memcpy(X, Y, sizeof(double) * (size_t)n);
    if (maxit <= 0) maxit = ORC_MAX_ITERS_YANG;
    set_threads(1);   /* the reference's Yang loops are serial (one Workspace) */

    int it;
    for (it = 1; it <= maxit; it++) {
        /* X = (Y + sum U_k + rho * sum Z_k) / (1 + D rho)   (2-D :832-833 ; 3-D :729-730) */
        for (long i = 0; i < n; i++) {
            double su = Y[i], sz = 0;
            if (nds == 2) { su = Y[i] + U[0][i] + U[1][i]; sz = Z[0][i] + Z[1][i]; }
            else          { su = Y[i] + U[0][i] + U[1][i] + U[2][i]; sz = Z[0][i] + Z[1][i] + Z[2][i]; }
            X[i] = (su + rho * sz) / (1 + nds * rho);
        }
        /* Z_k = prox_{lambda/rho} along dim order[k] of ( -1/rho * U_k + X )  */
        for (int k = 0; k < nds; k++)
            sweep_prox(X, U[k], -1. / rho, Z[k], ns, nds, order[k], lams[k] / rho, 0);
        /* U_k += rho (Z_k - X) */
        for (int k = 0; k < nds; k++)
            for (long i = 0; i < n; i++) U[k][i] += rho * (Z[k][i] - X[i]);
    }
    if (info) { info[ORC_INFO_ITERS] = it; info[ORC_INFO_RC] = ORC_RC_OK; }   /* it == maxit + 1 */
    free_family(U, nds); free_family(Z, nds);
    return 1;
}

int orc_Yang2_TV(size_t M, size_t N, const double *Y, double lambda, double *X, int maxit, double *info)
{
    /* Z1/U1 act along rows (dim 1), Z2/U2 along columns (dim 0): src/TV2Dopt.cpp:836-855 */
    const int ns[2] = { (int)M, (int)N };
    const int order[2] = { 1, 0 };
    const double lams[2] = { lambda, lambda };
    return yang_generic("Yang2_TV", ns, 2, order, lams, Y, X, maxit, info);
}

int orc_Yang3_TV(size_t M, size_t N, size_t O, const double *Y, double lambda, double *X, int maxit, double *info)
{
    const int ns[3] = { (int)M, (int)N, (int)O };
    const int order[3] = { 0, 1, 2 };                                  /* src/TVNDopt.cpp:733-781 */
    const double lams[3] = { lambda, lambda, lambda };
    return yang_generic("Yang3_TV", ns, 3, order, lams, Y, X, maxit, info);
}

int orc_Yang3_TV_perdim(size_t M, size_t N, size_t O, const double *Y, const double *lambda3, double *X,
                        int maxit, double *info)
{
    const int ns[3] = { (int)M, (int)N, (int)O };
    const int order[3] = { 0, 1, 2 };
    return yang_generic("Yang3_TV_perdim", ns, 3, order, lambda3, Y, X, maxit, info);
}

/* ------------------------------------------------------------------------- */
/*  Kolmogorov et al.'s primal-dual splitting, 2-D        src/TV2Dopt.cpp:907-1024 */
/*  Saddle point  min_X max_U <X,U> + TV_rows(X) + 1/2|X-Y|^2 - TV_cols^*(U); the   */
/*  dual step goes through Moreau's identity, so both steps are 1-D TV-L1 proxes.   */
/*  Step sizes: theta=1, tau=1/2, sigma=1, then theta=1/sqrt(1+tau), tau*=theta,    */
/*  sigma/=theta every iteration (:1002-1004).  Stops when the relative change of X */
/*  is not > STOP_KOLMOGOROV = 0 (i.e. X reached a bitwise fixed point) or after     */
/*  maxit (default 2500) iterations; info[0] = iterations done + 1.                  */
/* ------------------------------------------------------------------------- */
int orc_Kolmogorov2_TV(size_t M, size_t N, const double *Y, double lambda, double *X, int maxit, double *info)
{
    const long n = (long)M * (long)N;
    const int ns[2] = { (int)M, (int)N };
    double *U = (double *)malloc(sizeof(double) * (size_t)n);
    double *Xold = (double *)malloc(sizeof(double) * (size_t)n);
    double *V = (double *)malloc(sizeof(double) * (size_t)n);
    double *P = (double *)malloc(sizeof(double) * (size_t)n);
    if (!U || !Xold || !V || !P) {
        free(U); free(Xold); free(V); free(P);
        return fail("Kolmogorov2_TV", "insufficient memory", info);
    }
    double theta = 1., tau = 1. / 2., sigma = 1.;
    set_threads(1);   /* the reference's loop is serial (one fibre at a time through TV(..., NULL)) */
    memcpy(X, Y, sizeof(double) * (size_t)n);
    memcpy(Xold, Y, sizeof(double) * (size_t)n);
    memcpy(U, Y, sizeof(double) * (size_t)n);
    if (maxit <= 0) maxit = ORC_MAX_ITERS_KOLMOGOROV;

    int it;
    double stop = DBL_MAX;
    for (it = 1; stop > ORC_STOP_KOLMOGOROV && it <= maxit; it++) {
        /* dual step along columns: V = (U + sigma (X + theta (X - Xold))) / sigma ; U = sigma (V - colprox_{lambda/sigma} V)   (:963-973) */
        for (long i = 0; i < n; i++) {
            double v = U[i] + sigma * (X[i] + theta * (X[i] - Xold[i]));
            v /= sigma;
            V[i] = v;
        }
        sweep_prox(V, NULL, 0, P, ns, 2, 0, lambda / sigma, 0);
        for (long i = 0; i < n; i++) U[i] = sigma * (V[i] - P[i]);
        memcpy(Xold, X, sizeof(double) * (size_t)n);                                   /* :976 */
        /* primal step along rows: V = 1/(1+1/tau) (Y + 1/tau (X - tau U)) ; X = rowprox_{lambda/(1+1/tau)} V   (:981-996) */
        for (long i = 0; i < n; i++) {
            double v = X[i] - tau * U[i];
            v = 1 / (1 + 1 / tau) * (Y[i] + 1 / tau * v);
            V[i] = v;
        }
        sweep_prox(V, NULL, 0, X, ns, 2, 1, lambda / (1. + 1. / tau), 0);
        theta = 1. / sqrt(1 + 1 * tau);
        tau *= theta;
        sigma /= theta;
        /* relative change, summed serially in index order (:1007-1013) */
        double num = 0, den = 0;
        for (long i = 0; i < n; i++) {
            den += X[i] * X[i];
            const double d = Xold[i] - X[i];
            num += d * d;
        }
        stop = sqrt(num / den);
    }
    if (info) { info[ORC_INFO_ITERS] = it; info[ORC_INFO_RC] = ORC_RC_OK; }
    free(U); free(Xold); free(V); free(P);
    return 1;
}

/* ------------------------------------------------------------------------- */
/*  Condat's and Chambolle-Pock's primal-dual iterations, 2-D   src/TV2Dopt.cpp:587-760 */
/*  Duals U1 ((M-1) x N, vertical differences) and U2 (M x (N-1)), both column-major.  */
/*  alg 0: Condat (gradient step on the data term), 1: Chambolle-Pock (prox step),      */
/*  2: accelerated Chambolle-Pock (gamma = 1/lambda; tau, sigma, theta updated after     */
/*  the extrapolation and BEFORE the dual updates, :697-701).  sigma=10, tau=.9/(8 sigma),*/
/*  theta=1.  Same stopping rule as above with STOP_CONDAT = 0, default 2500 iterations. */
/* ------------------------------------------------------------------------- */
int orc_CondatChambollePock2_TV(size_t M, size_t N, const double *Y, double lambda, double *X, short alg, int maxit,
                                double *info)
{
    if (alg != 0 && alg != 1 && alg != 2)
        return fail("Condat2_TV", "Algorithm parameter has an invalid value", info);
    if (M < 2 || N < 2)   /* the reference indexes U1[(M-1)*j] / U2[i+M*(N-2)] regardless: out of bounds for a single row / column */
        return fail("Condat2_TV", "needs at least two rows and two columns", info);
    const long m = (long)M, nn = (long)N, n = m * nn;
    double sigma = 10, tau = .9 / (sigma * 8), theta = 1., gamma = (alg == 2) ? 1. / lambda : 0.;
    double *U1 = (double *)malloc(sizeof(double) * (size_t)((m - 1) * nn));
    double *U2 = (double *)malloc(sizeof(double) * (size_t)(m * (nn - 1)));
    double *Xt = (double *)malloc(sizeof(double) * (size_t)n);
    double *Z = (double *)malloc(sizeof(double) * (size_t)n);
    if (!U1 || !U2 || !Xt || !Z) {
        free(U1); free(U2); free(Xt); free(Z);
        return fail("Condat2_TV", "insufficient memory", info);
    }
    memcpy(X, Y, sizeof(double) * (size_t)n);
    for (long j = 0; j < nn; j++)
        for (long i = 0; i < m - 1; i++) U1[i + (m - 1) * j] = Y[i + 1 + m * j] - Y[i + m * j];
    for (long j = 0; j < nn - 1; j++)
        for (long i = 0; i < m; i++) U2[i + m * j] = Y[i + m * (j + 1)] - Y[i + m * j];
    if (maxit <= 0) maxit = ORC_MAX_ITERS_CONDAT;

    int it;
    double stop = DBL_MAX;
    for (it = 1; stop > ORC_STOP_CONDAT && it <= maxit; it++) {
        for (long j = 0; j < nn; j++)
            for (long i = 0; i < m; i++) {
                /* adjoint of the two difference operators at (i, j): vertical part first, then "+=" the horizontal part (:656-675) */
                double g = (i == 0) ? -U1[(m - 1) * j] : (i == m - 1) ? U1[m - 2 + (m - 1) * j]
                                                                       : U1[i - 1 + (m - 1) * j] - U1[i + (m - 1) * j];
                if (j == 0)            g += -U2[i];
                else if (j == nn - 1)  g += U2[i + m * (nn - 2)];
                else                   g += U2[i + m * (j - 1)] - U2[i + m * j];
                const long k = i + m * j;
                if (alg == 0) {
                    Xt[k] = X[k] - tau * (X[k] - Y[k] + g);                       /* :683 */
                } else {
                    const double c = 1. / (1. + tau);
                    Xt[k] = c * (X[k] + tau * (Y[k] - g));                        /* :691-693 */
                }
            }
        for (long k = 0; k < n; k++) Z[k] = Xt[k] + theta * (Xt[k] - X[k]);       /* :697-698 */
        if (alg == 2) {
            tau *= theta;
            sigma /= theta;
            theta = 1. / sqrt(1 + 2 * gamma * tau);
        }
        double num = 0, den = 0;
        for (long k = 0; k < n; k++) {
            den += X[k] * X[k];
            const double d = Xt[k] - X[k];
            num += d * d;
        }
        stop = sqrt(num / den);
        memcpy(X, Xt, sizeof(double) * (size_t)n);
        /* dual ascent + projection onto the l_inf ball of radius lambda (:720-747) */
        for (long j = 0; j < nn; j++)
            for (long i = 1; i < m; i++) {
                double u = U1[i - 1 + (m - 1) * j] + sigma * (Z[i + m * j] - Z[i - 1 + m * j]);
                if (u < -lambda) u = -lambda; else if (u > lambda) u = lambda;
                U1[i - 1 + (m - 1) * j] = u;
            }
        for (long j = 1; j < nn; j++)
            for (long i = 0; i < m; i++) {
                double u = U2[i + m * (j - 1)] + sigma * (Z[i + m * j] - Z[i + m * (j - 1)]);
                if (u < -lambda) u = -lambda; else if (u > lambda) u = lambda;
                U2[i + m * (j - 1)] = u;
            }
    }
    if (info) { info[ORC_INFO_ITERS] = it; info[ORC_INFO_RC] = ORC_RC_OK; }
    free(U1); free(U2); free(Xt); free(Z);
    return 1;
}
