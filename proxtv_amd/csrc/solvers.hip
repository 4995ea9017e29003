// solvers.hip -- the 2-D / N-D splitting loops, resident in HBM.
//
// Each loop is the reference's recurrence (cited per function) with the per-fibre gather/prox/scatter of the
// OpenMP scheduler replaced by one fibre-sweep launch per direction and the serial pointwise loops fused into the
// sweeps (ops.hpp) or into one streaming kernel (pointwise.hip).  Nothing leaves the device inside a solve except,
// for the Dykstra-type loops, the 8-byte stopping value once per iteration.
#include "solvers.hpp"

#include <cfloat>
#include <memory>
#include <utility>

#include "pointwise.hpp"
#include "sweep.hpp"

namespace ptv {

namespace {

long total(const int *ns, int nds) {
    long n = 1;
    for (int i = 0; i < nds; i++) n *= ns[i];
    return n;
}

double fetch(const double *dev, hipStream_t s) {
    double h = 0;
    PTV_HIP(hipMemcpyAsync(&h, dev, sizeof(double), hipMemcpyDeviceToHost, s));
    PTV_HIP(hipStreamSynchronize(s));
    return h;
}

int fam_of_dim(int d) { return d == 0 ? FAM_COL : d == 1 ? FAM_ROW : FAM_OTHER; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
void tv1_fibres(const double *in, double *out, const int *ns, int nds, int dim, double lam, const double *weights,
                hipStream_t s) {
    SweepArgs a;
    a.a = in;
    a.o0 = out;
    a.lam = lam;
    a.w = weights;
    launch_sweep(OP_PROX, weights != nullptr, a, fibres_along(ns, nds, dim), s, fam_of_dim(dim), in != out);
}

// ---------------------------------------------------------------------------------------------------------------------
// Douglas-Rachford / alternating reflections, reference: src/TV2Dopt.cpp:352-444 (weighted: src/TV2DWopt.cpp:46-140).
//   t <- 2 mean(U)
//   repeat maxit:  s' = 2 (t - colprox(t)) - t ;  t <- 1/2 (t + 2 [U - ((U - s') - rowprox(U - s'))] - s')
//   s = t - colprox(t) ; out = [U - ((U - s) - rowprox(U - s))] - s
// Two launches per iteration, six array passes (R t, W s' | R s', R U, R t, W t); t ping-pongs between two
// buffers so that the row sweep never writes an array another workgroup is still reading.
SolveInfo dr2(size_t M, size_t N, size_t B, const double *unary, double W1, double W2, const double *W1m,
              const double *W2m, double *out, int maxit, hipStream_t s) {
    SolveInfo info;
    const long n1 = (long)(M * N);
    const long n = n1 * (long)B;
    const bool weighted = (W1m != nullptr);
    if (maxit <= 0) maxit = MAX_ITERS_DR;
    info.iters = maxit;
    if (n == 0) return info;

    const int ns[3] = {(int)M, (int)N, (int)B};
    const FibreGeom cols = fibres_along(ns, 3, 0), rows = fibres_along(ns, 3, 1);

    Scratch t0(sizeof(double) * n), t1(sizeof(double) * n), sp(sizeof(double) * n);
    Scratch partials(sizeof(double) * kReduceBlocks * B), sums(sizeof(double) * B);
    double *t = t0.d(), *tn = t1.d();

    sum_to(unary, n1, (long)B, partials.d(), sums.d(), s);
    dr_fill(t, n1, (long)B, sums.d(), 1.0, s);

    SweepArgs col, row;
    col.lam = W1; col.w = W1m; col.o0 = sp.d();
    row.lam = W2; row.w = W2m; row.a = sp.d(); row.b = unary;
    for (int it = 0; it < maxit; it++) {
        col.a = t;
        if (it == 0) {
            // t is constant per image, and the prox of a constant fibre is that constant: s = t - prox(t) = 0,
            // s' = 2 s - t = -t.  (One flat piece per fibre is also the worst case for any taut-string walk.)
            FamilyTimer tm(FAM_OTHER, s);
            dr_fill(sp.d(), n1, (long)B, sums.d(), -1.0, s);
        } else {
            launch_sweep(OP_DR_COL, weighted, col, cols, s, FAM_COL, true);
        }
        row.c = t; row.o0 = tn;
        launch_sweep(OP_DR_ROW, weighted, row, rows, s, FAM_ROW, true);
        std::swap(t, tn);
    }
    col.a = t;
    launch_sweep(OP_DR_COL_FINAL, weighted, col, cols, s, FAM_COL, true);
    row.c = nullptr; row.o0 = out;
    launch_sweep(weighted ? OP_DRW_ROW_FINAL : OP_DR_ROW_FINAL, weighted, row, rows, s, FAM_ROW, true);
    return info;
}

// ---------------------------------------------------------------------------------------------------------------------
// Proximal Dykstra with one or two terms, reference: src/TV2Dopt.cpp:59-302.
//   x = y, p = q = 0 ; repeat: z = prox_d0(x + p), p += x - z ; x' = prox_d1(z + q), q += z - x' ; stop = mean|x' - x|
SolveInfo pd2(const double *y, const double *lambdas, const double *dims, double *x, const int *ns, int nds, int npen,
              int maxIters, hipStream_t s) {
    SolveInfo info;
    info.gap_set = true;
    if (maxIters <= 0) maxIters = MAX_ITERS_PD;
    const long n = total(ns, nds);
    const size_t bytes = sizeof(double) * (size_t)n;

    Scratch xa(bytes), xb(bytes), z(bytes), pa(bytes), pb(bytes), qa(bytes), qb(bytes);
    Scratch partials(sizeof(double) * kReduceBlocks), acc(sizeof(double));
    double *xc = xa.d(), *xn = xb.d(), *pi = pa.d(), *po = pb.d(), *qi = qa.d(), *qo = qb.d();
    PTV_HIP(hipMemcpyAsync(xc, y, bytes, hipMemcpyDeviceToDevice, s));
    PTV_HIP(hipMemsetAsync(pi, 0, bytes, s));
    PTV_HIP(hipMemsetAsync(qi, 0, bytes, s));

    const int d0 = (int)(dims[0] - 1);
    const int d1 = npen >= 2 ? (int)(dims[1] - 1) : 0;
    const FibreGeom g0 = fibres_along(ns, nds, d0);
    const FibreGeom g1 = fibres_along(ns, nds, d1);

    double stop = DBL_MAX;
    int iters = 0;
    while (stop > STOP_PD && (npen > 1 || !iters) && iters < maxIters) {   // :157
        SweepArgs a;
        a.a = xc; a.b = pi; a.o0 = z.d(); a.o1 = po; a.lam = lambdas[0];
        launch_sweep(OP_PD2_A, false, a, g0, s, fam_of_dim(d0), true);
        std::swap(pi, po);
        const double *xnew;
        if (npen >= 2) {
            SweepArgs b;
            b.a = z.d(); b.b = qi; b.o0 = xn; b.o1 = qo; b.lam = lambdas[1];
            launch_sweep(OP_PD2_B, false, b, g1, s, fam_of_dim(d1), true);
            std::swap(qi, qo);
            xnew = xn;
        } else {
            xnew = z.d();   // x = z (:265-270)
        }
        {
            FamilyTimer tm(FAM_OTHER, s);
            absdiff_to(xnew, xc, n, partials.d(), acc.d(), s);
        }
        stop = fetch(acc.d(), s) / n;   // one 8-byte read-back per iteration
        if (npen >= 2) std::swap(xc, xn);
        else PTV_HIP(hipMemcpyAsync(xc, z.d(), bytes, hipMemcpyDeviceToDevice, s));
        iters++;
    }
    PTV_HIP(hipMemcpyAsync(x, xc, bytes, hipMemcpyDeviceToDevice, s));
    info.iters = iters;
    info.gap = stop;
    info.rc = (iters >= MAX_ITERS_PD) ? RC_ITERS : RC_OK;   // compares with the macro, not maxIters (:289)
    return info;
}

// ---------------------------------------------------------------------------------------------------------------------
// Parallel proximal Dykstra, reference: src/TVNDopt.cpp:48-252 ; parallel Douglas-Rachford: src/TVNDopt.cpp:280-500.
namespace {

struct Family {
    std::vector<std::unique_ptr<Scratch>> blocks;
    PtrPack pack{};
    Family(int count, size_t bytes) {
        for (int i = 0; i < count; i++) {
            blocks.emplace_back(new Scratch(bytes));
            pack.v[i] = blocks.back()->d();
        }
    }
};

SolveInfo pd_like(bool dr_variant, const double *y, const double *lambdas, const double *dims, double *x, const int *ns,
                  int nds, int npen, int maxIters, hipStream_t s) {
    SolveInfo info;
    info.gap_set = true;
    if (npen > kMaxTerms) {
        set_error("at most %d penalty terms are supported per call (got %d)", kMaxTerms, npen);
        throw HipFailure{hipErrorInvalidValue};
    }
    if (maxIters <= 0) maxIters = dr_variant ? MAX_ITERS_DR : MAX_ITERS_PD;
    const long n = total(ns, nds);
    const size_t bytes = sizeof(double) * (size_t)n;

    Family p(npen, bytes), z(npen, bytes);
    Scratch partials(sizeof(double) * kReduceBlocks), acc(sizeof(double));
    if (dr_variant) scale_to(y, x, (double)npen, n, s);               // x = y / npen        (:362-367)
    else            PTV_HIP(hipMemsetAsync(x, 0, bytes, s));          // x = 0               (:126-130)
    for (int i = 0; i < npen; i++) PTV_HIP(hipMemcpyAsync(z.pack.v[i], y, bytes, hipMemcpyDeviceToDevice, s));

    double stop = dr_variant ? 0.0 : DBL_MAX;
    int iters = 0;
    while ((dr_variant || stop > STOP_PD) && iters < maxIters) {
        for (int i = 0; i < npen; i++) {
            const int d = (int)(dims[i] - 1);
            SweepArgs a;
            a.a = z.pack.v[i]; a.o0 = p.pack.v[i]; a.lam = lambdas[i];
            launch_sweep(OP_PROX, false, a, fibres_along(ns, nds, d), s, fam_of_dim(d), true);
        }
        {
            FamilyTimer tm(FAM_OTHER, s);
            if (dr_variant) pdr_combine(p.pack, z.pack, x, x, npen, n, partials.d(), acc.d(), s);
            else            pd_combine(p.pack, z.pack, x, x, npen, n, partials.d(), acc.d(), s);
        }
        if (!dr_variant || iters == maxIters - 1) stop = fetch(acc.d(), s) / n;
        iters++;
    }
    info.iters = iters;
    info.gap = stop;
    info.rc = (iters >= (dr_variant ? MAX_ITERS_DR : MAX_ITERS_PD)) ? RC_ITERS : RC_OK;   // :239 / :495
    return info;
}

}  // namespace

SolveInfo pd(const double *y, const double *lambdas, const double *dims, double *x, const int *ns, int nds, int npen,
             int maxIters, hipStream_t s) {
    return pd_like(false, y, lambdas, dims, x, ns, nds, npen, maxIters, s);
}

SolveInfo pdr(const double *y, const double *lambdas, const double *dims, double *x, const int *ns, int nds, int npen,
              int maxIters, hipStream_t s) {
    return pd_like(true, y, lambdas, dims, x, ns, nds, npen, maxIters, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Yang's ADMM, rho = 10: reference src/TV2Dopt.cpp:787-877 (2-D) and src/TVNDopt.cpp:678-803 (3-D).
//   U_k = 0, Z_k = X = Y ; repeat maxit: X = (Y + sum U_k + rho sum Z_k) / (1 + D rho) ;
//   Z_k = prox^{order[k]}_{lambda_k / rho}(X - U_k / rho) ; U_k += rho (Z_k - X)     (the U update rides in the sweep)
SolveInfo yang(const int *ns, int nds, const int *order, const double *lambdas, const double *Y, double *X, int maxit,
               hipStream_t s) {
    SolveInfo info;
    const double rho = 10;
    if (maxit <= 0) maxit = MAX_ITERS_YANG;
    const long n = total(ns, nds);
    const size_t bytes = sizeof(double) * (size_t)n;

    Family Ua(nds, bytes), Ub(nds, bytes), Z(nds, bytes);
    PtrPack Uin = Ua.pack, Uout = Ub.pack;
    for (int k = 0; k < nds; k++) {
        PTV_HIP(hipMemsetAsync(Uin.v[k], 0, bytes, s));
        PTV_HIP(hipMemcpyAsync(Z.pack.v[k], Y, bytes, hipMemcpyDeviceToDevice, s));
    }
    PTV_HIP(hipMemcpyAsync(X, Y, bytes, hipMemcpyDeviceToDevice, s));

    for (int it = 1; it <= maxit; it++) {
        {
            FamilyTimer tm(FAM_OTHER, s);
            yang_x(Y, Uin, Z.pack, X, nds, rho, n, s);
        }
        for (int k = 0; k < nds; k++) {
            SweepArgs a;
            a.a = X; a.b = Uin.v[k]; a.o0 = Z.pack.v[k]; a.o1 = Uout.v[k];
            a.s0 = rho; a.lam = lambdas[k] / rho;
            launch_sweep(OP_YANG, false, a, fibres_along(ns, nds, order[k]), s, fam_of_dim(order[k]), true);
        }
        std::swap(Uin, Uout);
    }
    info.iters = maxit + 1;   // the reference reports its loop counter after exit (:867 / :793)
    return info;
}

}  // namespace ptv
