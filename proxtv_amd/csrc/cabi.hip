// cabi.hip -- the extern "C" boundary of libproxtv_amd.so (declared in include/proxtv_amd.h).
//
// Part 1 mirrors the reference's C entry points (src/TVopt.h:88-141): host pointers in, host pointers out,
// same return / info conventions, never throws.  Each call stages its arrays into HBM scratch, runs the device
// solver of solvers.hip on the calling thread's stream and copies the result back.
// Part 2 exposes the same solvers on device pointers.
#include <exception>
#include <initializer_list>
#include <memory>
#include <vector>

#include "solvers.hpp"
#include "pointwise.hpp"
#include "sweep.hpp"

using namespace ptv;

namespace {

constexpr const char *kVersion = "proxtv_amd 0.1 (gfx950)";

// Runs `body`; maps any failure to the reference's CANCEL convention: print "<who>: <why>", info[RC]=RC_ERROR, return 0.
template <class F>
int guarded(const char *who, double *info, int ok_ret, F &&body) {
    try {
        set_error("");
        (void)hipGetLastError();   // an error another HIP user of this thread left behind is not ours to report
        ensure_device();
        body();
        return ok_ret;
    } catch (const HipFailure &) {
        // message already recorded
    } catch (const std::exception &e) {
        set_error("%s", e.what());
    } catch (...) {
        set_error("unknown failure");
    }
    printf("%s: %s\n", who, last_error());
    fflush(stdout);
    if (info) info[INFO_RC] = RC_ERROR;
    return 0;
}

[[noreturn]] void reject(const char *why) {
    set_error("%s", why);
    throw HipFailure{hipErrorInvalidValue};
}

// (Page-locking the caller's arrays around the transfers -- hipHostRegister -- was measured in round 4 and removed: the pageable
//  copies already run at the link's rate and locking 134 MB costs more than it saves.)
// host array staged into HBM scratch
struct Staged {
    Scratch buf;
    Staged(const double *host, size_t count, hipStream_t s) : buf(sizeof(double) * (count ? count : 1)) {
        if (!count) return;
        PTV_HIP(hipMemcpyAsync(buf.d(), host, sizeof(double) * count, hipMemcpyHostToDevice, s));
    }
    double *d() const { return buf.d(); }
};

void download(double *host, const double *dev, size_t count, hipStream_t s) {
    if (count) PTV_HIP(hipMemcpyAsync(host, dev, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    PTV_HIP(hipStreamSynchronize(s));
}

void put_info(double *info, const SolveInfo &si) {
    if (!info) return;
    info[INFO_ITERS] = si.iters;
    if (si.gap_set) info[INFO_GAP] = si.gap;
    info[INFO_RC] = si.rc;
}

hipStream_t pick(void *stream) { return stream ? (hipStream_t)stream : thread_stream(); }

// Device entry points: an output array that overlaps an input.  The chunked sweeps read their operands through windows
// that reach into rows other workgroups own, and the splitting loops read the input every iteration while the iterate
// is written, so in-place use would race.  Overlap is therefore detected here and the solve goes to a scratch array
// that is copied to the caller's at the end (one extra pass); distinct arrays cost nothing.
struct OutGuard {
    double *user;
    size_t count;
    std::unique_ptr<Scratch> tmp;
    struct In { const double *p; size_t n; };
    OutGuard(double *out, size_t n, std::initializer_list<In> inputs) : user(out), count(n) {
        const char *o0 = reinterpret_cast<const char *>(out), *o1 = o0 + sizeof(double) * n;
        for (const In &in : inputs) {
            if (!in.p || !in.n) continue;
            const char *i0 = reinterpret_cast<const char *>(in.p), *i1 = i0 + sizeof(double) * in.n;
            if (i0 < o1 && o0 < i1) {
                tmp.reset(new Scratch(sizeof(double) * (n ? n : 1)));
                break;
            }
        }
    }
    double *ptr() const { return tmp ? tmp->d() : user; }
    void commit(hipStream_t s) {
        if (tmp && count) PTV_HIP(hipMemcpyAsync(user, tmp->d(), sizeof(double) * count, hipMemcpyDeviceToDevice, s));
    }
};

struct SolveScope {
    hipStream_t s;
    explicit SolveScope(hipStream_t st) : s(st) {
        if (options().profile) timing_reset();
        chunk_stats_reset(st);
    }
    void finish() {
        if (options().profile) {
            PTV_HIP(hipStreamSynchronize(s));
            timing_collect();
        }
    }
};

// exact 1-D prox of one host signal (all unweighted 1-D entry points land here)
void prox1d_host(const double *y, const double *w, double lam, double *x, int n, double first_offset, double norm = 1) {
    if (n <= 0) return;
    hipStream_t s = thread_stream();
    SolveScope scope(s);   // per-call counters, profile option, the policy's across-call exploration
    Staged in(y, (size_t)n, s);
    if (first_offset != 0.0) {
        // a start `offset` above the tube centre == the same tube with its first increment lowered by `offset`
        const double y0 = y[0] - first_offset;
        PTV_HIP(hipMemcpyAsync(in.d(), &y0, sizeof(double), hipMemcpyHostToDevice, s));
    }
    Scratch out(sizeof(double) * (size_t)n);
    std::unique_ptr<Staged> wd;
    if (w && n > 1) wd.reset(new Staged(w, (size_t)n - 1, s));
    const int ns[1] = {n};
    if (norm == 2) prox_fibres(in.d(), out.d(), ns, 1, 0, lam, 2, s);
    else           tv1_fibres(in.d(), out.d(), ns, 1, 0, lam, wd ? wd->d() : nullptr, s);
    download(x, out.d(), (size_t)n, s);
    scope.finish();
}

bool check_norms(const double *norms, int npen) {   // returns whether any term is TV-L2
    bool l2 = false;
    for (int i = 0; i < npen; i++) {
        if (norms[i] != 1 && norms[i] != 2) reject("only the p = 1 (TV-L1) and p = 2 (TV-L2) norms are implemented on the HIP path");
        l2 = l2 || norms[i] == 2;
    }
    return l2;
}

long total(const int *ns, int nds) {
    long n = 1;
    for (int i = 0; i < nds; i++) n *= ns[i];
    return n;
}

void check_dims(const double *dims, int npen, int nds) {
    for (int i = 0; i < npen; i++) {
        const int d = (int)(dims[i] - 1);
        if (d < 0 || d >= nds) reject("penalty dimension out of range");
    }
}

}  // namespace

extern "C" {

// ====================================================== PART 1 ======================================================

int TV(double *y, double lambda, double *x, double *info, int n, double p, Workspace *) {
    if (p < 1) {   // reference message, src/TVgenopt.cpp:37-38
        printf("TVopt: %s\n", "TV only works for norms p >= 1");
        if (info) info[INFO_RC] = RC_ERROR;
        return 0;
    }
    return guarded("TVopt", info, 1, [&] {
        if (p != 1 && p != 2) reject("only the p = 1 (TV-L1) and p = 2 (TV-L2) norms are implemented on the HIP path");
        prox1d_host(y, nullptr, lambda, x, n, 0.0, p);
        if (info) { info[INFO_RC] = RC_OK; info[INFO_ITERS] = 0; info[INFO_GAP] = 0; }
    });
}

// TV-L2 (src/TVL2opt.cpp: more_TV2 :35, morePG_TV2 :190, PG_TV2 :446 -- three iteration schemes for the same prox; all are
// served by the exact trust-region solve of tv2.hip; info reports 0 iterations and a gap of 0)
static int tv2_host(const char *who, double *y, double lambda, double *x, double *info, int n) {
    return guarded(who, info, 1, [&] {
        prox1d_host(y, nullptr, lambda, x, n, 0.0, 2);
        if (info) { info[INFO_RC] = RC_OK; info[INFO_ITERS] = 0; info[INFO_GAP] = 0; }
    });
}
int more_TV2(double *y, double lambda, double *x, double *info, int n) { return tv2_host("more_TV2", y, lambda, x, info, n); }
int morePG_TV2(double *y, double lambda, double *x, double *info, int n, Workspace *) {
    return tv2_host("morePG_TV2", y, lambda, x, info, n);
}
int PG_TV2(double *y, double lambda, double *x, double *info, int n) { return tv2_host("PG_TV2", y, lambda, x, info, n); }

int linearizedTautString_TV1(double *y, double lambda, double *x, int n) {
    return guarded("linearizedTautString_TV1", nullptr, 1, [&] { prox1d_host(y, nullptr, lambda, x, n, 0.0); });
}

void hybridTautString_TV1(double *y, int n, double lambda, double *x) {
    guarded("hybridTautString_TV1", nullptr, 1, [&] { prox1d_host(y, nullptr, lambda, x, n, 0.0); });
}

void hybridTautString_TV1_custom(double *y, int n, double lambda, double *x, double) {
    guarded("hybridTautString_TV1_custom", nullptr, 1, [&] { prox1d_host(y, nullptr, lambda, x, n, 0.0); });
}

int classicTautString_TV1_offset(double *signal, int n, double lam, double *prox, double offset) {
    if (n <= 0) return 1;                                        // src/TVL1opt_tautstring.cpp:258-259
    if (lam <= 0 || n == 1) {                                    // :260-263
        memcpy(prox, signal, sizeof(double) * (size_t)n);
        return 1;
    }
    return guarded("classicTautString_TV1", nullptr, 1, [&] { prox1d_host(signal, nullptr, lam, prox, n, offset); });
}

int classicTautString_TV1(double *signal, int n, double lam, double *prox) {
    return classicTautString_TV1_offset(signal, n, lam, prox, 0.0);
}

int tautString_TV1_Weighted(double *y, double *lambda, double *x, int n) {
    return guarded("tautString_TV1_Weighted", nullptr, 1, [&] { prox1d_host(y, lambda, 0.0, x, n, 0.0); });
}

void TV1D_denoise(double *input, double *output, const int width, const double lambda) {
    if (!(width > 0 && lambda >= 0)) return;                     // src/condat_fast_tv.cpp:79
    guarded("TV1D_denoise", nullptr, 1, [&] { prox1d_host(input, nullptr, lambda, output, width, 0.0); });
}

// ---- the remaining 1-D entry points of the reference's cdef (prox_tv/prox_tv_build.py:13-76) -------------------------------
// PN_TV1 (src/TVL1opt.cpp:37), PN_TV1_Weighted (src/TVL1Wopt.cpp:37), SolveTVConvexQuadratic_a1[_nw]
// (src/TVL1opt_kolmogorov.cpp:38,133), TV1D_denoise_tautstring (src/condat_fast_tv.cpp:133) and dp (src/johnsonRyanTV.cpp:9)
// are alternative CPU algorithms for the SAME minimiser the exact HIP solver computes: each keeps its reference argument
// order, trivial-case behaviour and info convention and lands on prox1d_host.  (sigma, the projected-Newton descent
// tolerance, has no counterpart in an exact solve.)
int PN_TV1(double *y, double lambda, double *x, double *info, int n, double, Workspace *) {
    return guarded("PN_TV1", info, 1, [&] {
        prox1d_host(y, nullptr, lambda, x, n, 0.0);
        if (info) { info[INFO_ITERS] = 0; info[INFO_GAP] = 0; info[INFO_RC] = RC_OK; }
    });
}

int PN_TV1_Weighted(double *Y, double *W, double *X, double *info, int n, double, Workspace *) {
    return guarded("PN_TV1_Weighted", info, 1, [&] {
        prox1d_host(Y, W, 0.0, X, n, 0.0);
        if (info) { info[INFO_ITERS] = 0; info[INFO_GAP] = 0; info[INFO_RC] = RC_OK; }
    });
}

void SolveTVConvexQuadratic_a1_nw(int n, double *b, double w, double *solution) {
    if (n <= 1) {                                                 // src/TVL1opt_kolmogorov.cpp:135-139
        if (n == 1) solution[0] = b[0];
        return;
    }
    guarded("SolveTVConvexQuadratic_a1_nw", nullptr, 1, [&] { prox1d_host(b, nullptr, w, solution, n, 0.0); });
}

void SolveTVConvexQuadratic_a1(int n, double *b, double *w, double *solution) {
    if (n <= 1) {                                                 // src/TVL1opt_kolmogorov.cpp:40-44
        if (n == 1) solution[0] = b[0];
        return;
    }
    guarded("SolveTVConvexQuadratic_a1", nullptr, 1, [&] { prox1d_host(b, w, 0.0, solution, n, 0.0); });
}

void TV1D_denoise_tautstring(double *input, double *output, int width, const double lambda) {
    if (width <= 0) return;   // (the reference allocates width + 1 work arrays and walks them: nothing to do for an empty signal)
    guarded("TV1D_denoise_tautstring", nullptr, 1, [&] { prox1d_host(input, nullptr, lambda, output, width, 0.0); });
}

void dp(int n, double *y, double lam, double *beta) {
    if (n <= 0) return;                                           // src/johnsonRyanTV.cpp:11
    if (n == 1 || lam == 0) {                                     // :12-15
        for (int i = 0; i < n; i++) beta[i] = y[i];
        return;
    }
    guarded("dp", nullptr, 1, [&] { prox1d_host(y, nullptr, lam, beta, n, 0.0); });
}

// TV-Lp for general p (src/TVLPopt.cpp: GP_TVp :37, OGP_TVp :295, FISTA_TVp :583, FW_TVp :871, GPFW_TVp :1111): five
// first-order schemes for min 1/2 ||x-y||^2 + lambda ||Dx||_p.  Out of scope for general p (SURVEY section 2 allows the
// RC_ERROR stub); p = 1 and p = 2 have exact device solvers and are served by them, like TV().
static int tvp_host(const char *who, double *y, double lambda, double *x, double *info, int n, double p) {
    return guarded(who, info, 1, [&] {
        if (p != 1 && p != 2) reject("only the p = 1 (TV-L1) and p = 2 (TV-L2) norms are implemented on the HIP path");
        prox1d_host(y, nullptr, lambda, x, n, 0.0, p);
        if (info) { info[INFO_ITERS] = 0; info[INFO_GAP] = 0; info[INFO_RC] = RC_OK; }
    });
}
int GP_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *) { return tvp_host("GP_TVp", y, lambda, x, info, n, p); }
int OGP_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *) { return tvp_host("OGP_TVp", y, lambda, x, info, n, p); }
int FISTA_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *) { return tvp_host("FISTA_TVp", y, lambda, x, info, n, p); }
int FW_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *) { return tvp_host("FW_TVp", y, lambda, x, info, n, p); }
int GPFW_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *) { return tvp_host("GPFW_TVp", y, lambda, x, info, n, p); }

int DR2_TV(size_t M, size_t N, double *unary, double W1, double W2, double norm1, double norm2, double *s, int,
           int maxit, double *info) {
    // returns 0 on success like the reference (src/TV2Dopt.cpp:440); failures also return 0 with info[RC]=RC_ERROR
    return guarded("DR2_TV", info, 0, [&] {
        const double nrm[2] = {norm1, norm2};
        const bool l2 = check_norms(nrm, 2);
        hipStream_t st = thread_stream();
        SolveScope scope(st);
        Staged u(unary, M * N, st);
        Scratch out(sizeof(double) * M * N);
        const SolveInfo si = l2 ? dr2_norms(M, N, u.d(), W1, W2, norm1, norm2, out.d(), maxit, st)
                                : dr2(M, N, 1, u.d(), W1, W2, nullptr, nullptr, out.d(), maxit, st);
        download(s, out.d(), M * N, st);
        scope.finish();
        put_info(info, si);
    });
}

int DR2L1W_TV(size_t M, size_t N, double *unary, double *W1, double *W2, double *s, int, int maxit, double *info) {
    return guarded("DR2L1W_TV", info, 0, [&] {
        hipStream_t st = thread_stream();
        SolveScope scope(st);
        Staged u(unary, M * N, st);
        Staged w1(W1, M > 0 ? (M - 1) * N : 0, st), w2(W2, N > 0 ? M * (N - 1) : 0, st);
        Scratch out(sizeof(double) * M * N);
        const SolveInfo si = dr2(M, N, 1, u.d(), 0, 0, w1.d(), w2.d(), out.d(), maxit, st);
        download(s, out.d(), M * N, st);
        scope.finish();
        put_info(info, si);
    });
}

int PD2_TV(double *y, double *lambdas, double *norms, double *dims, double *x, double *info, int *ns, int nds,
           int npen, int, int maxIters) {
    return guarded("PD2_TV", info, 1, [&] {
        if (npen > 2) reject("this algorithm can not work with more than 2 penalties");   // src/TV2Dopt.cpp:95-96
        if (npen < 1) reject("at least one penalty term is required");
        check_norms(norms, npen);
        check_dims(dims, npen, nds);
        hipStream_t st = thread_stream();
        SolveScope scope(st);
        const size_t n = (size_t)total(ns, nds);
        Staged in(y, n, st);
        Scratch out(sizeof(double) * n);
        const SolveInfo si = pd2(in.d(), lambdas, dims, out.d(), ns, nds, npen, maxIters, st, norms);
        download(x, out.d(), n, st);
        scope.finish();
        put_info(info, si);
    });
}

static int pd_family_host(const char *who, bool dr_variant, double *y, double *lambdas, double *norms, double *dims,
                          double *x, double *info, int *ns, int nds, int npen, int maxIters) {
    return guarded(who, info, 1, [&] {
        if (npen < 1) reject("at least one penalty term is required");
        check_norms(norms, npen);
        check_dims(dims, npen, nds);
        for (int i = 0; i < npen; i++) lambdas[i] *= npen;   // caller memory, like src/TVNDopt.cpp:100-101 / :334-335
        hipStream_t st = thread_stream();
        SolveScope scope(st);
        const size_t n = (size_t)total(ns, nds);
        Staged in(y, n, st);
        Scratch out(sizeof(double) * n);
        const SolveInfo si = dr_variant ? pdr(in.d(), lambdas, dims, out.d(), ns, nds, npen, maxIters, st, norms)
                                        : pd(in.d(), lambdas, dims, out.d(), ns, nds, npen, maxIters, st, norms);
        download(x, out.d(), n, st);
        scope.finish();
        put_info(info, si);
    });
}

int PD_TV(double *y, double *lambdas, double *norms, double *dims, double *x, double *info, int *ns, int nds, int npen,
          int, int maxIters) {
    return pd_family_host("PD_TV", false, y, lambdas, norms, dims, x, info, ns, nds, npen, maxIters);
}

int PDR_TV(double *y, double *lambdas, double *norms, double *dims, double *x, double *info, int *ns, int nds, int npen,
           int, int maxIters) {
    return pd_family_host("PDR_TV", true, y, lambdas, norms, dims, x, info, ns, nds, npen, maxIters);
}

static int yang_host(const char *who, const int *ns, int nds, const int *order, double lambda, double *Y, double *X,
                     int maxit, double *info) {
    return guarded(who, info, 1, [&] {
        hipStream_t st = thread_stream();
        SolveScope scope(st);
        const size_t n = (size_t)total(ns, nds);
        Staged in(Y, n, st);
        Scratch out(sizeof(double) * (n ? n : 1));
        const double lams[3] = {lambda, lambda, lambda};
        const SolveInfo si = yang(ns, nds, order, lams, in.d(), out.d(), maxit, st);
        download(X, out.d(), n, st);
        scope.finish();
        put_info(info, si);
    });
}

int Yang2_TV(size_t M, size_t N, double *Y, double lambda, double *X, int maxit, double *info) {
    const int ns[2] = {(int)M, (int)N};
    const int order[2] = {1, 0};   // (Z1,U1) along rows, (Z2,U2) along columns: src/TV2Dopt.cpp:836-855
    return yang_host("Yang2_TV", ns, 2, order, lambda, Y, X, maxit, info);
}

int Yang3_TV(size_t M, size_t N, size_t O, double *Y, double lambda, double *X, int maxit, double *info) {
    const int ns[3] = {(int)M, (int)N, (int)O};
    const int order[3] = {0, 1, 2};   // src/TVNDopt.cpp:733-781
    return yang_host("Yang3_TV", ns, 3, order, lambda, Y, X, maxit, info);
}

int Kolmogorov2_TV(size_t M, size_t N, double *Y, double lambda, double *X, int maxit, double *info) {
    return guarded("Kolmogorov2_TV", info, 1, [&] {
        hipStream_t st = thread_stream();
        SolveScope scope(st);
        Staged in(Y, M * N, st);
        Scratch out(sizeof(double) * (M * N ? M * N : 1));
        const SolveInfo si = kolmogorov2(M, N, in.d(), lambda, out.d(), maxit, st);
        download(X, out.d(), M * N, st);
        scope.finish();
        put_info(info, si);
    });
}

int CondatChambollePock2_TV(size_t M, size_t N, double *Y, double lambda, double *X, short alg, int maxit, double *info) {
    return guarded("Condat2_TV", info, 1, [&] {   // the reference's messages say "Condat2_TV" (src/TV2Dopt.cpp:598)
        hipStream_t st = thread_stream();
        SolveScope scope(st);
        Staged in(Y, M * N, st);
        Scratch out(sizeof(double) * (M * N ? M * N : 1));
        const SolveInfo si = ccp2(M, N, in.d(), lambda, out.d(), alg, maxit, st);
        download(X, out.d(), M * N, st);
        scope.finish();
        put_info(info, si);
    });
}

// ---- Workspace shims (ABI only) ----
struct Workspace {
    int n;
};
Workspace *newWorkspace(int n) {
    Workspace *w = (Workspace *)calloc(1, sizeof(Workspace));
    if (w) w->n = n;
    return w;
}
void resetWorkspace(Workspace *) {}
void freeWorkspace(Workspace *ws) { free(ws); }
Workspace **newWorkspaces(int n, int p) {
    Workspace **wa = (Workspace **)calloc((size_t)(p > 0 ? p : 1), sizeof(Workspace *));
    if (!wa) return nullptr;
    for (int i = 0; i < p; i++) wa[i] = newWorkspace(n);
    return wa;
}
void freeWorkspaces(Workspace **wa, int p) {
    if (!wa) return;
    for (int i = 0; i < p; i++) freeWorkspace(wa[i]);
    free(wa);
}

// ====================================================== PART 2 ======================================================

int proxtv_init(int device) {
    try {
        set_error("");
        if (device >= 0) PTV_HIP(hipSetDevice(device));
        ensure_device();
        (void)thread_stream();
        return 0;
    } catch (...) {
        (void)hipGetLastError();   // (a failed hipSetDevice must not resurface after this thread's next launch)
        printf("proxtv_init: %s\n", last_error());
        fflush(stdout);
        return 1;
    }
}

const char *proxtv_version(void) { return kVersion; }
const char *proxtv_last_error(void) { return last_error(); }
void proxtv_release_scratch(void) { release_scratch(); }

int proxtv_set_option(const char *key, int value) {
    int *slot = option_slot(key);   // (common.hip: the one table of knobs, shared with the PROXTV_<KEY> environment variables)
    if (!slot) return -1;
    const int old = *slot;
    *slot = value;
    if (!strcmp(key, "optimistic")) {
        try { optimistic_forget(); } catch (...) {}   // (per-device state: nothing to forget where there is no device)
    }
    if (value != 0 && (!strcmp(key, "ablate") || !strcmp(key, "debug_legacy_rebuild")))
        fprintf(stderr, "[proxtv_amd] WARNING: option \"%s\" = %d -- a profiling / test aid: results are WRONG while it is non-zero\n", key, value);
    return old;
}

int proxtv_chunk_mode(void) { return chunk_stats_mode(); }

long proxtv_last_fixups(void) {
    try { return chunk_stats_fixups(thread_stream()); } catch (...) { return -1; }
}

double proxtv_last_kernel_ms(int which) { return timing_ms(which); }
long proxtv_debug_trace(unsigned long long *dst, long max_wgs) {
    try { return chunk_trace_fetch(dst, max_wgs, thread_stream()); } catch (...) { return -1; }
}
long proxtv_last_kernel_launches(int which) { return timing_launches(which); }
long proxtv_debug_counter(const char *name) { return counter_value(name); }
int proxtv_debug_why(unsigned *dst) {
    try { return chunk_why_fetch(dst, thread_stream()); } catch (...) { return -1; }
}

int proxtv_DR2_TV_batch_dev(size_t M, size_t N, size_t B, const double *unary, double W1, double W2, double *s,
                            int maxit, double *info, void *stream) {
    return guarded("proxtv_DR2_TV_batch_dev", info, 0, [&] {
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        OutGuard out(s, M * N * B, {{unary, M * N * B}});
        const SolveInfo si = dr2(M, N, B, unary, W1, W2, nullptr, nullptr, out.ptr(), maxit, st);
        out.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
        put_info(info, si);
    });
}

int proxtv_DR2_TV_batch(size_t M, size_t N, size_t B, const double *unary, double W1, double W2, double *s, int maxit,
                        double *info) {
    return guarded("proxtv_DR2_TV_batch", info, 0, [&] {
        hipStream_t st = thread_stream();
        SolveScope scope(st);
        const size_t n = M * N * B;
        Staged u(unary, n, st);
        Scratch out(sizeof(double) * (n ? n : 1));
        const SolveInfo si = dr2(M, N, B, u.d(), W1, W2, nullptr, nullptr, out.d(), maxit, st);
        download(s, out.d(), n, st);
        scope.finish();
        put_info(info, si);
    });
}

int proxtv_DR2_TV_dev(size_t M, size_t N, const double *unary, double W1, double W2, double *s, int maxit, double *info,
                      void *stream) {
    return proxtv_DR2_TV_batch_dev(M, N, 1, unary, W1, W2, s, maxit, info, stream);
}

int proxtv_DR2L1W_TV_dev(size_t M, size_t N, const double *unary, const double *W1, const double *W2, double *s,
                         int maxit, double *info, void *stream) {
    return guarded("proxtv_DR2L1W_TV_dev", info, 0, [&] {
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        OutGuard out(s, M * N, {{unary, M * N}, {W1, M > 0 ? (M - 1) * N : 0}, {W2, N > 0 ? M * (N - 1) : 0}});
        const SolveInfo si = dr2(M, N, 1, unary, 0, 0, W1, W2, out.ptr(), maxit, st);
        out.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
        put_info(info, si);
    });
}

int proxtv_PD2_TV_dev(const double *y, const double *lambdas, const double *dims, double *x, double *info, const int *ns,
                      int nds, int npen, int maxIters, void *stream) {
    return guarded("proxtv_PD2_TV_dev", info, 1, [&] {
        if (npen > 2 || npen < 1) reject("PD2 works with 1 or 2 penalties");
        check_dims(dims, npen, nds);
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        OutGuard out(x, (size_t)total(ns, nds), {{y, (size_t)total(ns, nds)}});
        const SolveInfo si = pd2(y, lambdas, dims, out.ptr(), ns, nds, npen, maxIters, st);
        out.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
        put_info(info, si);
    });
}

static int pd_family_dev(const char *who, bool dr_variant, const double *y, const double *lambdas, const double *dims,
                         double *x, double *info, const int *ns, int nds, int npen, int maxIters, void *stream) {
    return guarded(who, info, 1, [&] {
        if (npen < 1) reject("at least one penalty term is required");
        check_dims(dims, npen, nds);
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        OutGuard out(x, (size_t)total(ns, nds), {{y, (size_t)total(ns, nds)}});
        const SolveInfo si = dr_variant ? pdr(y, lambdas, dims, out.ptr(), ns, nds, npen, maxIters, st)
                                        : pd(y, lambdas, dims, out.ptr(), ns, nds, npen, maxIters, st);
        out.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
        put_info(info, si);
    });
}

int proxtv_PD_TV_dev(const double *y, const double *lambdas_scaled, const double *dims, double *x, double *info,
                     const int *ns, int nds, int npen, int maxIters, void *stream) {
    return pd_family_dev("proxtv_PD_TV_dev", false, y, lambdas_scaled, dims, x, info, ns, nds, npen, maxIters, stream);
}

int proxtv_PDR_TV_dev(const double *y, const double *lambdas_scaled, const double *dims, double *x, double *info,
                      const int *ns, int nds, int npen, int maxIters, void *stream) {
    return pd_family_dev("proxtv_PDR_TV_dev", true, y, lambdas_scaled, dims, x, info, ns, nds, npen, maxIters, stream);
}

int proxtv_Yang_TV_dev(const int *ns, int nds, const double *Y, const double *lambdas, double *X, int maxit, double *info,
                       void *stream) {
    return guarded("proxtv_Yang_TV_dev", info, 1, [&] {
        if (nds != 2 && nds != 3) reject("Yang's ADMM is implemented for 2-D and 3-D arrays");
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        int order[3] = {0, 1, 2};
        double lams[3] = {lambdas[0], lambdas[1], nds == 3 ? lambdas[2] : 0.0};
        if (nds == 2) { order[0] = 1; order[1] = 0; std::swap(lams[0], lams[1]); }   // (Z1,U1) = rows = dim 2
        OutGuard out(X, (size_t)total(ns, nds), {{Y, (size_t)total(ns, nds)}});
        const SolveInfo si = yang(ns, nds, order, lams, Y, out.ptr(), maxit, st);
        out.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
        put_info(info, si);
    });
}

int proxtv_Kolmogorov2_TV_dev(size_t M, size_t N, const double *Y, double lambda, double *X, int maxit, double *info,
                              void *stream) {
    return guarded("proxtv_Kolmogorov2_TV_dev", info, 1, [&] {
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        OutGuard out(X, M * N, {{Y, M * N}});
        const SolveInfo si = kolmogorov2(M, N, Y, lambda, out.ptr(), maxit, st);
        out.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
        put_info(info, si);
    });
}

int proxtv_CondatChambollePock2_TV_dev(size_t M, size_t N, const double *Y, double lambda, double *X, short alg,
                                       int maxit, double *info, void *stream) {
    return guarded("proxtv_CondatChambollePock2_TV_dev", info, 1, [&] {
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        OutGuard out(X, M * N, {{Y, M * N}});
        const SolveInfo si = ccp2(M, N, Y, lambda, out.ptr(), alg, maxit, st);
        out.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
        put_info(info, si);
    });
}

int proxtv_tv1_fibres_dev(const double *in, double *out, const int *ns, int nds, int dim, double lambda,
                          const double *weights, void *stream) {
    return guarded("proxtv_tv1_fibres_dev", nullptr, 1, [&] {
        if (dim < 0 || dim >= nds) reject("dimension out of range");
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        const size_t n = (size_t)total(ns, nds);
        OutGuard guard(out, n, {{in, n}, {weights, weights ? n : 0}});
        tv1_fibres(in, guard.ptr(), ns, nds, dim, lambda, weights, st);
        guard.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
    });
}

long proxtv_certify_fibres_dev(const double *in, const double *out, const int *ns, int nds, int dim, double lambda, const double *weights,
                               void *stream) {
    long failed = -2;
    guarded("proxtv_certify_fibres_dev", nullptr, 1, [&] {
        if (dim < 0 || dim >= nds) reject("dimension out of range");
        hipStream_t st = pick(stream);
        failed = certify_fibres(in, out, ns, nds, dim, lambda, weights, st);
    });
    return failed;
}

int proxtv_tvp_fibres_dev(const double *in, double *out, const int *ns, int nds, int dim, double lambda, double p,
                          void *stream) {
    return guarded("proxtv_tvp_fibres_dev", nullptr, 1, [&] {
        if (dim < 0 || dim >= nds) reject("dimension out of range");
        if (p != 1 && p != 2) reject("only the p = 1 (TV-L1) and p = 2 (TV-L2) norms are implemented on the HIP path");
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        const size_t n = (size_t)total(ns, nds);
        OutGuard guard(out, n, {{in, n}});
        prox_fibres(in, guard.ptr(), ns, nds, dim, lambda, p, st);
        guard.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
    });
}

int proxtv_DR2_TVp_dev(size_t M, size_t N, const double *unary, double W1, double W2, double norm1, double norm2, double *s,
                       int maxit, double *info, void *stream) {
    return guarded("proxtv_DR2_TVp_dev", info, 0, [&] {
        const double nrm[2] = {norm1, norm2};
        const bool l2 = check_norms(nrm, 2);
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        OutGuard out(s, M * N, {{unary, M * N}});
        const SolveInfo si = l2 ? dr2_norms(M, N, unary, W1, W2, norm1, norm2, out.ptr(), maxit, st)
                                : dr2(M, N, 1, unary, W1, W2, nullptr, nullptr, out.ptr(), maxit, st);
        out.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
        put_info(info, si);
    });
}

int proxtv_PD_TVp_dev(int which, const double *y, const double *lambdas_scaled, const double *norms, const double *dims,
                      double *x, double *info, const int *ns, int nds, int npen, int maxIters, void *stream) {
    return guarded("proxtv_PD_TVp_dev", info, 1, [&] {
        if (npen < 1) reject("at least one penalty term is required");
        if (which == 0 && npen > 2) reject("PD2 works with 1 or 2 penalties");
        check_norms(norms, npen);
        check_dims(dims, npen, nds);
        hipStream_t st = pick(stream);
        SolveScope scope(st);
        OutGuard out(x, (size_t)total(ns, nds), {{y, (size_t)total(ns, nds)}});
        const SolveInfo si = which == 0 ? pd2(y, lambdas_scaled, dims, out.ptr(), ns, nds, npen, maxIters, st, norms)
                           : which == 1 ? pd(y, lambdas_scaled, dims, out.ptr(), ns, nds, npen, maxIters, st, norms)
                                        : pdr(y, lambdas_scaled, dims, out.ptr(), ns, nds, npen, maxIters, st, norms);
        out.commit(st);
        PTV_HIP(hipStreamSynchronize(st));
        scope.finish();
        put_info(info, si);
    });
}

int proxtv_calib_copy_dev(const double *src, double *dst, long n, void *stream) {
    return guarded("proxtv_calib_copy_dev", nullptr, 1, [&] {
        hipStream_t st = pick(stream);
        calib_copy(src, dst, n, st);
        PTV_HIP(hipStreamSynchronize(st));
    });
}

}  // extern "C"
