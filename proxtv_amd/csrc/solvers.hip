// solvers.hip -- the 2-D / N-D splitting loops, resident in HBM.
//
// Each loop is the reference's recurrence (cited per function) with the per-fibre gather/prox/scatter of the
// OpenMP scheduler replaced by one fibre-sweep launch per direction and the serial pointwise loops fused into the
// sweeps (ops.hpp) or into one streaming kernel (pointwise.hip).  Nothing leaves the device inside a solve except,
// for the Dykstra-type loops, the 8-byte stopping value once per iteration.
#include "solvers.hpp"

#include "transposed.hpp"

#include <cfloat>
#include <cmath>
#include <memory>
#include <utility>
#include <vector>

#include "pointwise.hpp"
#include "policy.hpp"
#include "sweep.hpp"
#include "tv2.hpp"

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
    if (in != out) policy_probe(in, &weights, ns, nds, &dim, 1, s);
    launch_sweep(OP_PROX, weights != nullptr, a, fibres_along(ns, nds, dim), s, fam_of_dim(dim), in != out);
}

long certify_fibres(const double *in, const double *out, const int *ns, int nds, int dim, double lam, const double *weights, hipStream_t s) {
    SweepArgs a;
    a.a = in;
    a.o0 = const_cast<double *>(out);   // (read only: the certifier never writes an operand)
    a.lam = lam;
    a.w = weights;
    return certify_sweep(OP_PROX, weights != nullptr, a, fibres_along(ns, nds, dim), s);
}

void prox_fibres(const double *in, double *out, const int *ns, int nds, int dim, double lam, double norm, hipStream_t s) {
    if (norm == 2) tv2_fibres(in, out, ns, nds, dim, lam, s);
    else           tv1_fibres(in, out, ns, nds, dim, lam, nullptr, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Douglas-Rachford / alternating reflections, reference: src/TV2Dopt.cpp:352-444 (weighted: src/TV2DWopt.cpp:46-140).
//   t <- 2 mean(U)
//   repeat maxit:  s' = 2 (t - colprox(t)) - t ;  t <- 1/2 (t + 2 [U - ((U - s') - rowprox(U - s'))] - s')
//   s = t - colprox(t) ; out = [U - ((U - s) - rowprox(U - s))] - s
// Two launches per iteration, six array passes (R t, W s' | R s', R U, R t, W t); t ping-pongs between two
// buffers so that the row sweep never writes an array another workgroup is still reading.
static SolveInfo dr2_run(size_t M, size_t N, size_t B, const double *unary, double W1, double W2, const double *W1m,
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

    {   // seed of the geometry policy: the image's edge statistics along both directions (the iterates' are close to them)
        const int both[2] = {0, 1};
        const double *const wts[2] = {W1m, W2m};
        policy_probe(unary, wts, ns, 3, both, 2, s);
    }
    sum_to(unary, n1, (long)B, partials.d(), sums.d(), s);
    dr_fill(t, n1, (long)B, sums.d(), 1.0, s);

    SweepArgs col, row;
    col.lam = W1; col.w = W1m; col.o0 = sp.d();
    row.lam = W2; row.w = W2m; row.a = sp.d(); row.b = unary;
    // every array of this loop is written by sweeps only (t, s' once by dr_fill, before any row sweep): row sweeps that
    // run on transposed copies may keep them (transposed.hpp)
    TransposeScope keep_transposed;
    double row_f = -1.0;
    const int tile_rung = options().dr_form ? strided_tile_rung(rows, W2, weighted, &row_f) : -1;
    if (options().dr_form == 2 ? tile_rung >= 0 : (tile_rung == 1 && (!weighted || (row_f >= 0.0 && row_f < kSeedDrFormWeighted)))) {
        // The row sweep runs on the robust 64-fibre tile: the column sweep does all the pointwise work (ops.hpp, OP_DR_COL_V):
        // R t, R U, W v, W s | R v, R s, W t.  Measured on 4096^2, unit noise: the column sweep, bound by its walk there, takes
        // the two extra passes for nothing (lambda = 0.5: 150 -> 153 us) and the row sweep loses a staged operand and a fetch
        // (235 -> 200 us); on rung 0 the column sweep is short enough to become bandwidth-bound (84 -> 120 us against
        // 134 -> 110 for the rows), so the reference's split stays there (option dr_form = 2 forces this form on rung 0 too).
        // No array is read and written by the same sweep, so t needs no second buffer: `tn` holds s, `sp` holds v = U - s'.
        double *v = sp.d(), *sarr = tn;
        for (int it = 0; it < maxit; it++) {
            if (it == 0) {
                // s = t - prox(t) = 0 for the constant t, s' = -t, v = U - s'
                FamilyTimer tm(FAM_OTHER, s);
                dr_fill(v, n1, (long)B, sums.d(), -1.0, s);
                lincomb(v, unary, 1.0, v, -1.0, nullptr, 0.0, nullptr, 0.0, n, s);
                PTV_HIP(hipMemsetAsync(sarr, 0, sizeof(double) * n, s));
            } else {
                col.a = t; col.b = unary; col.o0 = v; col.o1 = sarr;
                launch_sweep(OP_DR_COL_V, weighted, col, cols, s, FAM_COL, true);
            }
            row.a = v; row.b = sarr; row.c = nullptr; row.o0 = t;
            launch_sweep(OP_DR_ROW_V, weighted, row, rows, s, FAM_ROW, true);
        }
        col.b = nullptr; col.o0 = sp.d(); col.o1 = nullptr;
        row.a = sp.d(); row.b = unary;
    } else
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

// The solve as the entry points see it.  Where the sampled statistics put every sweep on rung 0 the solve first runs optimistically --
// no repair launch behind its 72 sweeps, a sticky word instead (sweep.hpp) -- and is run again, with the repairs, if any sweep left
// anything: the second run is the reference-exact one, the first is identical to it whenever the word stays clear (the sweeps
// themselves are the same kernels on the same data).
SolveInfo dr2(size_t M, size_t N, size_t B, const double *unary, double W1, double W2, const double *W1m,
              const double *W2m, double *out, int maxit, hipStream_t s) {
    if (M * N * B == 0) return dr2_run(M, N, B, unary, W1, W2, W1m, W2m, out, maxit, s);
    const int ns[3] = {(int)M, (int)N, (int)B};
    {   // (the seed of the policy: dr2_run finds it taken)
        const int both[2] = {0, 1};
        const double *const wts[2] = {W1m, W2m};
        policy_probe(unary, wts, ns, 3, both, 2, s);
    }
    const FibreGeom geoms[2] = {fibres_along(ns, 3, 0), fibres_along(ns, 3, 1)};
    const double lams[2] = {W1, W2};
    OptimisticScope opt(s, optimistic_eligible(geoms, lams, 2, W1m != nullptr));
    const bool tried = opt.on;
    SolveInfo info = dr2_run(M, N, B, unary, W1, W2, W1m, W2m, out, maxit, s);
    if (tried) count_event(CNT_OPTIMISTIC_SOLVES);
    if (!opt.clean()) {
        count_event(CNT_OPTIMISTIC_REDONE);
        info = dr2_run(M, N, B, unary, W1, W2, W1m, W2m, out, maxit, s);
    }
    return info;
}

// The same recurrence with the fibre prox of either norm as a separate step (src/TV2Dopt.cpp:539-547 applies
// TV(..., norm) per fibre): s = t - prox_c(t) ; s' = 2 s - t ; v = U - s' ; tb = U - (v - prox_r(v)) ; t <- 1/2 (t + 2 tb - s').
SolveInfo dr2_norms(size_t M, size_t N, const double *unary, double W1, double W2, double norm1, double norm2, double *out,
                    int maxit, hipStream_t s) {
    SolveInfo info;
    const long n = (long)(M * N);
    if (maxit <= 0) maxit = MAX_ITERS_DR;
    info.iters = maxit;
    if (n == 0) return info;
    const int ns[2] = {(int)M, (int)N};
    const size_t bytes = sizeof(double) * (size_t)n;
    Scratch t(bytes), x(bytes), sp(bytes), v(bytes);
    Scratch partials(sizeof(double) * kReduceBlocks), sums(sizeof(double));
    {   // seed of the geometry policy: the image's edge statistics, like dr2 -- probes are matched by geometry, and the first array a
        // TV-L1 sweep of this loop sees is the constant t0 (every edge 0: it would pin the whole solve to the long-piece rung)
        const int both[2] = {0, 1};
        policy_probe(unary, nullptr, ns, 2, both, 2, s);
    }
    sum_to(unary, n, 1, partials.d(), sums.d(), s);
    dr_fill(t.d(), n, 1, sums.d(), 1.0, s);
    auto cols = [&](const double *tin, double *sout, double c_s, double c_t) {   // sout = c_s (t - prox_c(t)) + c_t t
        prox_fibres(tin, x.d(), ns, 2, 0, W1, norm1, s);
        lincomb(sout, tin, c_s + c_t, x.d(), -c_s, nullptr, 0, nullptr, 0, n, s);
    };
    for (int it = 0; it < maxit; it++) {
        cols(t.d(), sp.d(), 2.0, -1.0);                                                  // s' = 2 (t - x) - t
        lincomb(v.d(), unary, 1.0, sp.d(), -1.0, nullptr, 0, nullptr, 0, n, s);          // v = U - s'
        prox_fibres(v.d(), x.d(), ns, 2, 1, W2, norm2, s);
        // tb = U - (v - x) ; t = 1/2 (t + 2 tb - s') = 1/2 t + U - v + x - 1/2 s' = 1/2 t + x + 1/2 s'   (U - v = s')
        lincomb(t.d(), t.d(), 0.5, x.d(), 1.0, sp.d(), 0.5, nullptr, 0, n, s);
    }
    cols(t.d(), sp.d(), 1.0, 0.0);                                                       // s = t - prox_c(t)
    lincomb(v.d(), unary, 1.0, sp.d(), -1.0, nullptr, 0, nullptr, 0, n, s);
    prox_fibres(v.d(), x.d(), ns, 2, 1, W2, norm2, s);
    // out = [U - (v - x)] - s = (U - v) + x - s = x        (U - v = s)
    PTV_HIP(hipMemcpyAsync(out, x.d(), bytes, hipMemcpyDeviceToDevice, s));
    return info;
}

// ---------------------------------------------------------------------------------------------------------------------
// Proximal Dykstra with one or two terms, reference: src/TV2Dopt.cpp:59-302.
//   x = y, p = q = 0 ; repeat: z = prox_d0(x + p), p += x - z ; x' = prox_d1(z + q), q += z - x' ; stop = mean|x' - x|
SolveInfo pd2(const double *y, const double *lambdas, const double *dims, double *x, const int *ns, int nds, int npen,
              int maxIters, hipStream_t s, const double *norms) {
    SolveInfo info;
    const bool l2a = norms && norms[0] == 2, l2b = norms && npen >= 2 && norms[1] == 2;
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
    {
        const int swept[2] = {d0, d1};
        policy_probe(y, nullptr, ns, nds, swept, npen >= 2 ? 2 : 1, s);
    }

    double stop = DBL_MAX;
    int iters = 0;
    bool settled0 = false, settled1 = false;   // (that term's sweeps are on the pinning rung, or its operand has stopped moving: no more samples of it)
    while (stop > STOP_PD && (npen > 1 || !iters) && iters < maxIters) {   // :157
        if (l2a) {   // z = prox2(x + p) ; p += x - z     (unfused: the TV-L2 prox is its own kernel)
            lincomb(po, xc, 1.0, pi, 1.0, nullptr, 0, nullptr, 0, n, s);
            tv2_fibres(po, z.d(), ns, nds, d0, lambdas[0], s);
            lincomb(po, pi, 1.0, xc, 1.0, z.d(), -1.0, nullptr, 0, n, s);
        } else {
            if (!settled0 && reprobe_at(iters + 1)) {   // (the operand x + p is not the solve's input; the loop reads a value back every iteration anyway)
                const double *as[1] = {xc}, *bs[1] = {pi};
                const double cs[1] = {1.0};
                settled0 = policy_reprobe(1, 1, as, bs, cs, &lambdas[0], ns, nds, &d0, s) != kReprobeAskAgain;
            }
            SweepArgs a;
            a.a = xc; a.b = pi; a.o0 = z.d(); a.o1 = po; a.lam = lambdas[0];
            launch_sweep(OP_PD2_A, false, a, g0, s, fam_of_dim(d0), true);
        }
        std::swap(pi, po);
        const double *xnew;
        if (npen >= 2) {
            if (l2b) {
                lincomb(qo, z.d(), 1.0, qi, 1.0, nullptr, 0, nullptr, 0, n, s);
                tv2_fibres(qo, xn, ns, nds, d1, lambdas[1], s);
                lincomb(qo, qi, 1.0, z.d(), 1.0, xn, -1.0, nullptr, 0, n, s);
            } else {
                if (!settled1 && reprobe_at(iters + 1)) {
                    const double *as[1] = {z.d()}, *bs[1] = {qi};
                    const double cs[1] = {1.0};
                    settled1 = policy_reprobe(1, 1, as, bs, cs, &lambdas[1], ns, nds, &d1, s) != kReprobeAskAgain;
                }
                SweepArgs b;
                b.a = z.d(); b.b = qi; b.o0 = xn; b.o1 = qo; b.lam = lambdas[1];
                launch_sweep(OP_PD2_B, false, b, g1, s, fam_of_dim(d1), true);
            }
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
    std::vector<double *> ptr;   // every block's address
    PtrPack pack{};              // ... and the first kMaxTerms of them as a kernel argument
    Family(int count, size_t bytes) {
        for (int i = 0; i < count; i++) {
            blocks.emplace_back(new Scratch(bytes));
            ptr.push_back(blocks.back()->d());
            if (i < kMaxTerms) pack.v[i] = blocks.back()->d();
        }
    }
};

SolveInfo pd_like(bool dr_variant, const double *y, const double *lambdas, const double *dims, double *x, const int *ns,
                  int nds, int npen, int maxIters, hipStream_t s, const double *norms) {
    SolveInfo info;
    info.gap_set = true;
    if (maxIters <= 0) maxIters = dr_variant ? MAX_ITERS_DR : MAX_ITERS_PD;
    const long n = total(ns, nds);
    const size_t bytes = sizeof(double) * (size_t)n;

    Family p(npen, bytes), z(npen, bytes);
    Scratch partials(sizeof(double) * kReduceBlocks), acc(sizeof(double));
    // more terms than a kernel-argument pack holds (the reference takes any number: :48-110): the combine kernel reads the array
    // addresses from a table in HBM
    std::unique_ptr<Scratch> table;
    if (npen > kMaxTerms) {
        std::vector<double *> host(p.ptr);
        host.insert(host.end(), z.ptr.begin(), z.ptr.end());
        table.reset(new Scratch(sizeof(double *) * host.size()));
        PTV_HIP(hipMemcpyAsync(table->as<double *>(), host.data(), sizeof(double *) * host.size(), hipMemcpyHostToDevice, s));
        PTV_HIP(hipStreamSynchronize(s));   // (`host` lives on this frame)
    }
    if (dr_variant) scale_to(y, x, (double)npen, n, s);               // x = y / npen        (:362-367)
    else            PTV_HIP(hipMemsetAsync(x, 0, bytes, s));          // x = 0               (:126-130)
    for (int i = 0; i < npen; i++) PTV_HIP(hipMemcpyAsync(z.ptr[(size_t)i], y, bytes, hipMemcpyDeviceToDevice, s));

    {
        std::vector<int> swept((size_t)npen);
        for (int i = 0; i < npen; i++) swept[(size_t)i] = (int)(dims[i] - 1);
        policy_probe(y, nullptr, ns, nds, swept.data(), npen, s);
    }
    double stop = dr_variant ? 0.0 : DBL_MAX;
    int iters = 0;
    bool settled = false;   // (every term's sweeps are on the pinning rung, or the operands have stopped moving: no more samples of them)
    while ((dr_variant || stop > STOP_PD) && iters < maxIters) {
        if (!settled && reprobe_at(iters + 1) && npen <= 8) {   // the operands z_i drift away from the solve's input
            std::vector<const double *> as((size_t)npen);
            std::vector<int> ds((size_t)npen);
            std::vector<double> ls((size_t)npen);
            int m = 0;
            for (int i = 0; i < npen; i++) {
                if (norms && norms[i] == 2) continue;
                bool dup = false;   // (several terms along one dimension share a record: the first one speaks for it)
                for (int j = 0; j < m; j++) dup = dup || ds[(size_t)j] == (int)(dims[i] - 1);
                if (dup) continue;
                as[(size_t)m] = z.ptr[(size_t)i];
                ls[(size_t)m] = lambdas[i];
                ds[(size_t)m++] = (int)(dims[i] - 1);
            }
            if (m > 0) settled = policy_reprobe(1, m, as.data(), nullptr, nullptr, ls.data(), ns, nds, ds.data(), s) != kReprobeAskAgain;
        }
        for (int i = 0; i < npen; i++) {
            const int d = (int)(dims[i] - 1);
            if (norms && norms[i] == 2) {
                tv2_fibres(z.ptr[(size_t)i], p.ptr[(size_t)i], ns, nds, d, lambdas[i], s);
                continue;
            }
            SweepArgs a;
            a.a = z.ptr[(size_t)i]; a.o0 = p.ptr[(size_t)i]; a.lam = lambdas[i];
            launch_sweep(OP_PROX, false, a, fibres_along(ns, nds, d), s, fam_of_dim(d), true);
        }
        {
            FamilyTimer tm(FAM_OTHER, s);
            if (table)           pd_combine_many(table->as<double *>(), x, x, npen, n, partials.d(), acc.d(), dr_variant, s);
            else if (dr_variant) pdr_combine(p.pack, z.pack, x, x, npen, n, partials.d(), acc.d(), s);
            else                 pd_combine(p.pack, z.pack, x, x, npen, n, partials.d(), acc.d(), s);
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
             int maxIters, hipStream_t s, const double *norms) {
    return pd_like(false, y, lambdas, dims, x, ns, nds, npen, maxIters, s, norms);
}

SolveInfo pdr(const double *y, const double *lambdas, const double *dims, double *x, const int *ns, int nds, int npen,
              int maxIters, hipStream_t s, const double *norms) {
    return pd_like(true, y, lambdas, dims, x, ns, nds, npen, maxIters, s, norms);
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
    policy_probe(Y, nullptr, ns, nds, order, nds, s);
    bool settled = false;   // (every dimension on the pinning rung: no more samples)
    bool steady = true;     // (sample every fourth iteration from the ninth on, not only at 17 and 33: unless the last sample was calm)

    for (int it = 1; it <= maxit; it++) {
        {
            FamilyTimer tm(FAM_OTHER, s);
            yang_x(Y, Uin, Z.pack, X, nds, rho, n, s);
        }
        if (!settled && reprobe_at(it, steady)) {   // the sweeps' operands, X - U_k / rho, are not the solve's input: which rung they want is theirs to say
            const double *as[kMaxTerms], *bs[kMaxTerms];
            double cs[kMaxTerms], ls[kMaxTerms];
            for (int k = 0; k < nds; k++) {
                as[k] = X;
                bs[k] = Uin.v[k];
                cs[k] = -1.0 / rho;
                ls[k] = lambdas[k] / rho;
            }
            const int status = policy_reprobe(2, nds, as, bs, cs, ls, ns, nds, order, s);
            settled = status == kReprobeSettled;
            steady = status != kReprobeCalm;
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

namespace {

// Loops whose exit test is "X changed at all" (STOP = 0 in the reference): the test lives on the device.  flag[k] says
// "iteration k changed X" (flag[0] = 1); the kernels of iteration k + 1 are no-ops when flag[k] == 0, so the host
// enqueues all iterations without a round trip and reads the flags once at the end.
struct ChangeFlags {
    Scratch buf;
    int count;
    ChangeFlags(int n, hipStream_t s) : buf(sizeof(int) * (size_t)(n + 2)), count(n + 2) {
        PTV_HIP(hipMemsetAsync(buf.as<int>(), 0, sizeof(int) * (size_t)count, s));
        const int one = 1;
        PTV_HIP(hipMemcpyAsync(buf.as<int>(), &one, sizeof(int), hipMemcpyHostToDevice, s));
        PTV_HIP(hipStreamSynchronize(s));   // `one` lives on this stack frame
    }
    int *at(int k) const { return buf.as<int>() + k; }
    // the reference's loop counter at exit: first k in [1, last] with flag[k] == 0 -> k + 1 ; none -> maxit + 1
    int exit_counter(int last, int maxit, hipStream_t s) const {
        std::vector<int> h((size_t)count, 0);
        PTV_HIP(hipMemcpyAsync(h.data(), buf.as<int>(), sizeof(int) * (size_t)count, hipMemcpyDeviceToHost, s));
        PTV_HIP(hipStreamSynchronize(s));
        for (int k = 1; k <= last; k++)
            if (!h[(size_t)k]) return k + 1;
        return maxit + 1;
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Kolmogorov et al., "Total variation on a tree", p. 19: reference src/TV2Dopt.cpp:907-1024.
//   X = Xold = U = Y ; theta = 1, tau = 1/2, sigma = 1 ; repeat:
//     V = (U + sigma (X + theta (X - Xold))) / sigma ; U = sigma (V - colprox_{lambda/sigma} V)       (Moreau)
//     Xold = X ; V = (Y + (X - tau U)/tau) / (1 + 1/tau) ; X = rowprox_{lambda/(1+1/tau)} V
//     theta = 1/sqrt(1+tau) ; tau *= theta ; sigma /= theta ; stop = |X - Xold| / |X|
// The dual is stored unscaled (D = V - colprox V, the fused output of the column sweep; U = su D is formed where it
// is used, the same single rounding), X and Xold trade places instead of being copied.
SolveInfo kolmogorov2(size_t M, size_t N, const double *Y, double lambda, double *X, int maxit, hipStream_t s) {
    SolveInfo info;
    if (maxit <= 0) maxit = MAX_ITERS_KOLMOGOROV;
    const long n = (long)M * (long)N;
    const size_t bytes = sizeof(double) * (size_t)(n ? n : 1);
    const int ns[2] = {(int)M, (int)N};
    Scratch D(bytes), Xa(bytes), Xb(bytes), V(bytes);
    ChangeFlags flags(maxit, s);
    double *xcur = Xa.d(), *xold = Xb.d();
    PTV_HIP(hipMemcpyAsync(xcur, Y, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    PTV_HIP(hipMemcpyAsync(xold, Y, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    PTV_HIP(hipMemcpyAsync(D.d(), Y, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    double theta = 1., tau = 1. / 2., sigma = 1., su = 1.;
    {
        const int both[2] = {0, 1};
        policy_probe(Y, nullptr, ns, 2, both, 2, s);
    }

    for (int it = 1; it <= maxit; it++) {
        const int *gate = flags.at(it - 1);
        {
            // also settles flag[it - 1] (did iteration it - 1 change X?); itself gated on iteration it - 1 having run
            FamilyTimer tm(FAM_OTHER, s);
            kolmo_dual_in(D.d(), su, xcur, xold, sigma, theta, V.d(), n, it > 1 ? flags.at(it - 2) : nullptr,
                          it > 1 ? flags.at(it - 1) : nullptr, s);
        }
        SweepArgs col;
        col.a = V.d(); col.o0 = D.d(); col.lam = lambda / sigma; col.gate = gate;
        launch_sweep(OP_DR_COL_FINAL, false, col, fibres_along(ns, 2, 0), s, FAM_COL, true);   // D = V - prox(V)
        su = sigma;
        {
            FamilyTimer tm(FAM_OTHER, s);
            kolmo_primal_in(xcur, D.d(), su, Y, tau, 1 / (1 + 1 / tau), 1 / tau, V.d(), n, gate, s);
        }
        SweepArgs row;
        row.a = V.d(); row.o0 = xold; row.lam = lambda / (1. + 1. / tau); row.gate = gate;
        launch_sweep(OP_PROX, false, row, fibres_along(ns, 2, 1), s, FAM_ROW, true);
        std::swap(xcur, xold);   // the buffer just written is X, the other one Xold
        theta = 1. / sqrt(1 + 1 * tau);
        tau *= theta;
        sigma /= theta;
    }
    // flag[maxit - 1] is the last one the loop evaluated (iteration maxit's own change is never tested: it <= maxit ends it)
    info.iters = flags.exit_counter(maxit - 1, maxit, s);
    PTV_HIP(hipMemcpyAsync(X, xcur, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    return info;
}

// ---------------------------------------------------------------------------------------------------------------------
// Condat / Chambolle-Pock primal-dual iterations: reference src/TV2Dopt.cpp:587-760.  sigma = 10, tau = .9/(8 sigma),
// theta = 1; the accelerated variant (gamma = 1/lambda) rescales tau, sigma, theta after the extrapolation and before
// the dual update (:697-701).  One fused kernel per iteration; every array is double-buffered (neighbours read the
// previous iterate of Z and of the duals).  The final dual update of the reference does not touch X and is skipped.
SolveInfo ccp2(size_t M, size_t N, const double *Y, double lambda, double *X, int alg, int maxit, hipStream_t s) {
    SolveInfo info;
    if (alg != 0 && alg != 1 && alg != 2) {
        set_error("Algorithm parameter has an invalid value");
        throw HipFailure{hipErrorInvalidValue};
    }
    if (M < 2 || N < 2) {
        set_error("needs at least two rows and two columns (the reference indexes out of bounds otherwise)");
        throw HipFailure{hipErrorInvalidValue};
    }
    if (maxit <= 0) maxit = MAX_ITERS_CONDAT;
    const long m = (long)M, nn = (long)N, n = m * nn;
    const size_t bytes = sizeof(double) * (size_t)n;
    Scratch X0(bytes), X1(bytes), Z0(bytes), Z1(bytes), U10(bytes), U11(bytes), U20(bytes), U21(bytes);
    ChangeFlags flags(maxit, s);
    double *xc = X0.d(), *xn = X1.d(), *zc = Z0.d(), *zn = Z1.d();
    double *u1c = U10.d(), *u1n = U11.d(), *u2c = U20.d(), *u2n = U21.d();
    PTV_HIP(hipMemcpyAsync(xc, Y, bytes, hipMemcpyDeviceToDevice, s));
    ccp_init(Y, u1c, u2c, m, nn, s);
    double sigma = 10, tau = .9 / (sigma * 8), theta = 1.;
    const double gamma = (alg == 2) ? 1. / lambda : 0.;

    for (int it = 1; it <= maxit; it++) {
        FamilyTimer tm(FAM_OTHER, s);
        CcpArgs a{Y, xc, zc, u1c, u2c, xn, zn, u1n, u2n, m, nn, tau, theta, sigma, lambda, alg, flags.at(it - 1), flags.at(it)};
        ccp_step(a, it == 1, s);
        if (it > 1) {
            std::swap(u1c, u1n);
            std::swap(u2c, u2n);
        }
        std::swap(xc, xn);
        std::swap(zc, zn);
        if (alg == 2) {
            tau *= theta;
            sigma /= theta;
            theta = 1. / sqrt(1 + 2 * gamma * tau);
        }
    }
    info.iters = flags.exit_counter(maxit, maxit, s);
    PTV_HIP(hipMemcpyAsync(X, xc, bytes, hipMemcpyDeviceToDevice, s));
    return info;
}

}  // namespace ptv
