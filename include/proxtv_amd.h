/*
 * proxtv_amd.h -- C-ABI of libproxtv_amd.so, the MI355X (gfx950) TV-proximity library.
 *
 * PART 1 is the drop-in boundary: the same unmangled symbols, argument meaning, ownership and
 * return/info conventions as the reference's `extern "C"` block (reference: src/TVopt.h:88-141,
 * src/condat_fast_tv.h:76-81).  All pointers in part 1 are HOST pointers; the library stages them
 * to HBM, runs the HIP path and copies the result back.  Arrays are column-major (dimension 0
 * fastest), float64.  `info` is caller-owned double[3] or NULL: {iterations, gap/stop, return code}
 * (reference: src/general.h:58-73).  Functions never throw; a HIP failure or an unsupported
 * argument prints "<fn>: <reason>" to stdout like the reference's CANCEL macros, sets
 * info[2]=RC_ERROR and returns 0.  THERE IS NO CPU FALLBACK: without a usable gfx950 device every
 * entry point fails that way.
 *
 * PART 2 is new API (not in the reference): device-pointer entry points used by bench.py and by
 * callers that already hold data in HBM, plus the batch solver that shards independent images.
 *
 * Norms: p = 1 (TV-L1, the hot path of BASELINE.json) and p = 2 (TV-L2) have exact device solvers.  Every function
 * the reference's cffi cdef declares (prox_tv/prox_tv_build.py:13-76) is exported, so that cdef links against this
 * library unmodified: the alternative 1-D TV-L1 algorithms (projected Newton, Kolmogorov's and Johnson's solvers,
 * Condat's taut string) are entry points of the one exact solver -- the minimiser is unique -- and the general-p
 * TV-Lp schemes report RC_ERROR for p outside {1, 2} (DESIGN.md "Out of scope").
 */
#ifndef PROXTV_AMD_H
#define PROXTV_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants shared with the reference ------------------------------------------------------ */
#define N_INFO 3              /* src/general.h:58 */
#define INFO_ITERS 0
#define INFO_GAP 1
#define INFO_RC 2
#define RC_OK 0               /* src/general.h:70-73 */
#define RC_ITERS 1
#define RC_STUCK 2
#define RC_ERROR 3
#define STOP_PD 1e-6          /* src/TVopt.h:71 */
#define MAX_ITERS_PD 35       /* src/TVopt.h:73 */
#define MAX_ITERS_DR 35       /* src/TVopt.h:83 */
#define MAX_ITERS_YANG 35     /* src/TVopt.h:85 */
#define MAX_ITERS_CONDAT 2500      /* src/TVopt.h:75 ; STOP_CONDAT = 0 (:77) */
#define MAX_ITERS_KOLMOGOROV 2500  /* src/TVopt.h:79 ; STOP_KOLMOGOROV = 0 (:81) */
#define ALG_CONDAT 0               /* src/TVopt.h: algorithm selector of CondatChambollePock2_TV */
#define ALG_CHAMBOLLE_POCK 1
#define ALG_CHAMBOLLE_POCK_ACC 2

/* Opaque, ABI-only (reference: src/utils.h:20-34).  The HIP path keeps its scratch in HBM and
   ignores workspaces; callers (like the reference's Python layer) pass NULL. */
typedef struct Workspace Workspace;

/* =============================== PART 1: drop-in entry points ================================== */

/* replaces src/TVgenopt.cpp:30 `TV` -- p == 1 (TV-L1, the hot path) and p == 2 (TV-L2); p < 1 or any other p ->
   RC_ERROR, returns 0 */
int TV(double *y, double lambda, double *x, double *info, int n, double p, Workspace *ws);
/* replace src/TVL2opt.cpp:35, :190, :446 -- the TV-L2 prox  min 1/2 ||x-y||^2 + lambda ||Dx||_2.  The three reference
   entry points are three iteration schemes for the same minimiser (which they reach to a duality gap of 1e-5, the
   hybrid one warm-started from its workspace); all are served by one exact device solve that depends on (y, lambda)
   only (tv2.hip).  info: iterations 0, gap 0, RC_OK. */
int more_TV2(double *y, double lambda, double *x, double *info, int n);
int morePG_TV2(double *y, double lambda, double *x, double *info, int n, Workspace *ws);
int PG_TV2(double *y, double lambda, double *x, double *info, int n);

/* replace src/TVL1opt.cpp:37 and src/TVL1Wopt.cpp:37 (projected Newton; `sigma`, its sufficient-descent tolerance, has no
   counterpart in an exact solve), src/TVL1opt_kolmogorov.cpp:133 and :38 (Kolmogorov et al.'s message passing; n <= 1
   handled like the reference), src/condat_fast_tv.cpp:133 (Condat's taut string) and src/johnsonRyanTV.cpp:9 (Johnson's
   dynamic programme; n == 0 / n == 1 / lam == 0 like the reference): the same unique minimiser as the entry points below,
   served by the same exact HIP solver.  info (where present): iterations 0, gap 0, RC_OK. */
int PN_TV1(double *y, double lambda, double *x, double *info, int n, double sigma, Workspace *ws);
int PN_TV1_Weighted(double *Y, double *W, double *X, double *info, int n, double sigma, Workspace *ws);
void SolveTVConvexQuadratic_a1_nw(int n, double *b, double w, double *solution);
void SolveTVConvexQuadratic_a1(int n, double *b, double *w, double *solution);
void TV1D_denoise_tautstring(double *input, double *output, int width, const double lambda);
void dp(int n, double *y, double lam, double *beta);
/* replace src/TVLPopt.cpp:37, :295, :583, :871, :1111 -- TV-Lp first-order schemes.  p == 1 and p == 2 are served by the
   exact solvers (like TV); any other p -> prints the reason, info[2] = RC_ERROR, returns 0. */
int GP_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *ws);
int OGP_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *ws);
int FISTA_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *ws);
int FW_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *ws);
int GPFW_TVp(double *y, double lambda, double *x, double *info, int n, double p, Workspace *ws);

/* replaces src/TVL1opt.cpp:359 */
int linearizedTautString_TV1(double *y, double lambda, double *x, int n);
/* replaces src/TVL1opt_tautstring.cpp:355 and :256 */
int classicTautString_TV1(double *signal, int n, double lam, double *prox);
int classicTautString_TV1_offset(double *signal, int n, double lam, double *prox, double offset);
/* replaces src/TVL1opt_hybridtautstring.cpp:237 and :56.  The 1-D TV-L1 prox is unique, so all
   five unweighted entry points run the same exact HIP solver; `backtracksexp` is accepted and has
   no observable effect on the result. */
void hybridTautString_TV1(double *y, int n, double lambda, double *x);
void hybridTautString_TV1_custom(double *y, int n, double lambda, double *x, double backtracksexp);
/* replaces src/TVL1Wopt.cpp:364 -- lambda has n-1 entries */
int tautString_TV1_Weighted(double *y, double *lambda, double *x, int n);
/* replaces src/condat_fast_tv.cpp:78 -- no-op when width <= 0 or lambda < 0; in-place allowed */
void TV1D_denoise(double *input, double *output, const int width, const double lambda);

/* replaces src/TV2Dopt.cpp:352 -- returns 0 on success (sic); norm1 / norm2 in {1, 2} (2: TV-L2 fibres, exact solve) */
int DR2_TV(size_t M, size_t N, double *unary, double W1, double W2, double norm1, double norm2,
           double *s, int nThreads, int maxit, double *info);
/* replaces src/TV2DWopt.cpp:46 -- W1 is (M-1)xN, W2 is Mx(N-1), column-major; returns 0 on success (sic) */
int DR2L1W_TV(size_t M, size_t N, double *unary, double *W1, double *W2, double *s, int nThreads,
              int maxit, double *info);
/* replaces src/TV2Dopt.cpp:59 -- npen in {1,2}; dims are 1-based doubles; norms in {1, 2} (here and below) */
int PD2_TV(double *y, double *lambdas, double *norms, double *dims, double *x, double *info, int *ns,
           int nds, int npen, int ncores, int maxIters);
/* replaces src/TVNDopt.cpp:48 -- multiplies lambdas[] by npen in caller memory, like the reference */
int PD_TV(double *y, double *lambdas, double *norms, double *dims, double *x, double *info, int *ns,
          int nds, int npen, int ncores, int maxIters);
/* replaces src/TVNDopt.cpp:280 -- multiplies lambdas[] by npen in caller memory, like the reference */
int PDR_TV(double *y, double *lambdas, double *norms, double *dims, double *x, double *info, int *ns,
           int nds, int npen, int ncores, int maxIters);
/* replaces src/TV2Dopt.cpp:787 */
int Yang2_TV(size_t M, size_t N, double *Y, double lambda, double *X, int maxit, double *info);
/* replaces src/TVNDopt.cpp:678 */
int Yang3_TV(size_t M, size_t N, size_t O, double *Y, double lambda, double *X, int maxit, double *info);
/* replaces src/TV2Dopt.cpp:907 -- Kolmogorov et al.'s primal-dual splitting of the same 2-D TV-L1 problem; both of
   its steps are 1-D proxes (columns through Moreau's identity, rows directly) and run on the sweep kernels.
   maxit <= 0 -> 2500; stops earlier only when X reaches a bitwise fixed point (STOP = 0); info[0] = iterations + 1 */
int Kolmogorov2_TV(size_t M, size_t N, double *Y, double lambda, double *X, int maxit, double *info);
/* replaces src/TV2Dopt.cpp:587 -- Condat (alg 0) / Chambolle-Pock (1) / accelerated Chambolle-Pock (2) primal-dual
   iterations: pointwise stencils, one fused kernel per iteration.  Images with a single row or column are rejected
   (the reference indexes out of bounds there). */
int CondatChambollePock2_TV(size_t M, size_t N, double *Y, double lambda, double *X, short alg, int maxit,
                            double *info);

/* Workspace allocator shims (reference: src/utils.cpp:79-237).  ABI-only: they return / accept
   small host objects so reference callers that create workspaces keep linking. */
Workspace *newWorkspace(int n);
void resetWorkspace(Workspace *ws);
void freeWorkspace(Workspace *ws);
Workspace **newWorkspaces(int n, int p);
void freeWorkspaces(Workspace **wa, int p);

/* =============================== PART 2: MI355X-native extensions ============================== */

/* Library / device state.  proxtv_init returns 0 when a gfx950 device is usable, else non-zero and
   prints the reason.  device < 0 keeps the current HIP device; device >= 0 makes it current (hipSetDevice).
   Everything the library keeps between calls (stream, scratch pool, chunk-kernel buffers, geometry policy) is
   per host thread AND per device: a thread may move between GPUs freely (hipSetDevice / proxtv_init), every entry
   point works on the device that is current when it is called, and pointers must belong to that device.
   Host threads never share state: concurrent calls from several threads are safe (options below are process-wide:
   set them before the threads start solving -- a change mid-solve takes effect at that solve's next sweep). */
int  proxtv_init(int device);
const char *proxtv_version(void);
const char *proxtv_last_error(void);
/* free every cached HBM scratch block held by the calling thread's pool on the current device (the pool is capped:
   PROXTV_POOL_CAP_MB, default 65536) */
void proxtv_release_scratch(void);

/* Knobs (process-wide; returns the previous value, -1 for an unknown key).  Each also has an environment variable
   PROXTV_<KEY> read at load time.  Twenty in all; none is needed for correct results.
     "runs"           1 (default): on rung 0, dimension-0 sweeps over data most of whose edges are bends known a priori (|dy| > 4 lambda)
                      cut the interior segments of their fibres at those bends and solve them run by run -- runs of one and two samples by
                      rule, longer ones walked one per lane: exact by construction, nothing speculated ; 0: speculative chunks everywhere
     "chunk_mode"     -1: the geometry policy chooses per sweep (default) ; 0..5: pin a rung of the ladder (see proxtv_chunk_mode)
     "deterministic"  1 (default): the rung of a sweep is a function of sampled statistics of its input and of lambda alone --
                      the same call gives the same bits whatever ran before ; 0: hill climb on measured sweep times, seeded by
                      the same statistics (results then agree to ~1e-13 between calls, not bit for bit)
     "chunk_min_len"  fibres shorter than this do not take the multi-block chunk kernels (default 96)
     "whole"          fibres of 16 .. chunk_min_len samples: 1 (default) by length and data, 2 whole fibres in LDS, 0 sequential
     "along"          1 (default): dimension-0 sweeps by chunks along the fibre ; 0: through the transposing 64-fibre tile
     "xlink"          1 (default): chunk kernels check the links across their workgroups themselves ; 0: the repair kernel does
     "dr_form"        DR2_TV / DR2L1W_TV: which sweep does the pointwise work of an iteration.  1 (default): the column sweep leaves
                      the row sweep's input and epilogue operand when the row sweep will run on the robust 64-fibre tile (decided
                      from the same sampled statistics as the rung) ; 2: on the plain tile too ; 0: never (the reference's split).
                      Same iterates either way, to a few ulps
     "pin"            1 (default): rung 3 is the pinning solver ; 0: the global-memory chunk kernel
     "pin_seed"       the pinning solver (rung 3) starts from the knots known a priori: 2 (default) jumps above 4 lambda and the
                      deepest knots of windows of 4 / 16 / 64 knots, 1: the jumps alone, 0: from the fibre ends alone
     "repair_jobs"    failed links across the workgroups of a chunked sweep are first repaired one lane per failure, four to a
                      fibre; what that leaves goes to the sequential repair: 1 (default) where the sampled statistic of the sweep's
                      input says such links fail in numbers (an unsampled input counts as such), 2 always ; 0: the sequential repair
                      alone (same results, bit for bit)
     "tile"           strided sweeps on rungs 0 and 1 (the robust instantiation follows the same knob): 1 (default) tiles of 32 fibres
                      x 8 chunks in 4 waves, four workgroups per CU ; 0: the 64-fibre x 8-wave tile, two per CU
     "optimistic"     1 (default): a DR solve whose every sweep will run on rung 0 launches no repair kernel behind its sweeps; a sweep
                      that leaves anything marks one word, read at the end, and such a solve is run again with the repairs (counters
                      optimistic_solves / optimistic_redone) ; 0: a repair launch behind every chunked sweep.  Same results, bit for bit
     "certify"        1: behind every fibre sweep a second kernel checks the optimality conditions of the prox on what the sweep wrote,
                      fibre by fibre (u = cumsum(y - x): |u_k| <= lambda_k ; u_k = -+lambda_k where x steps up / down ; u_{n-1} = 0),
                      re-solves a fibre that fails with the sequential walk and counts it (proxtv_debug_counter: certify_failures,
                      certify_sweeps, certify_skipped) ; 0 (default): sweeps are exact by their own argument and
                      the check costs a pass over the sweep's arrays
     "verbose"        1: log every decision of the geometry policy to stderr
     "profile"        1: hipEvent pair around every sweep launch (proxtv_last_kernel_ms / _launches)
     "why", "trace"   tuning / profiling aids (proxtv_debug_why, proxtv_debug_trace)
     "ablate"         profiling aid: skips phases of the chunk kernels -- RESULTS ARE WRONG while it is non-zero (a warning is printed)
     "debug_legacy_rebuild"   test aid: 1 plants the rebuild semantics of rounds 1-4 in the along-fibre kernel (an unproven chunk values
                      its first piece from its own first row: the hole of DESIGN.md 6 (ii')) so that the suite can watch the certifier catch
                      it -- RESULTS MAY BE WRONG while it is non-zero (a warning is printed) */
int proxtv_set_option(const char *key, int value);

/* Device-pointer solvers: every double* is an HBM pointer valid on the current device, `stream` is
   a hipStream_t (NULL = the library's per-thread stream), `info` is a HOST double[3] or NULL.
   Work is enqueued AND completed (the call synchronises `stream`) unless `info` is NULL and
   PD-type stopping is not involved; see DESIGN.md.  Same return conventions as part 1.
   Aliasing: an output array may overlap an input (in-place use); the solve then runs into a scratch array that is
   copied over the caller's at the end -- one extra pass; distinct arrays cost nothing. */
int proxtv_DR2_TV_dev(size_t M, size_t N, const double *unary, double W1, double W2, double *s,
                      int maxit, double *info, void *stream);
int proxtv_DR2L1W_TV_dev(size_t M, size_t N, const double *unary, const double *W1, const double *W2,
                         double *s, int maxit, double *info, void *stream);
int proxtv_PD2_TV_dev(const double *y, const double *lambdas /*host*/, const double *dims /*host*/,
                      double *x, double *info, const int *ns /*host*/, int nds, int npen, int maxIters,
                      void *stream);
int proxtv_PD_TV_dev(const double *y, const double *lambdas_scaled /*host, already * npen*/,
                     const double *dims /*host*/, double *x, double *info, const int *ns /*host*/,
                     int nds, int npen, int maxIters, void *stream);
int proxtv_PDR_TV_dev(const double *y, const double *lambdas_scaled /*host*/, const double *dims /*host*/,
                      double *x, double *info, const int *ns /*host*/, int nds, int npen, int maxIters,
                      void *stream);
/* Yang ADMM on a 2-D or 3-D array with one lambda per dimension (scalar-lambda reference behaviour
   when all entries are equal).  order follows the reference: 2-D rows then columns, 3-D dims 1,2,3. */
int proxtv_Yang_TV_dev(const int *ns /*host*/, int nds, const double *Y, const double *lambdas /*host, nds*/,
                       double *X, int maxit, double *info, void *stream);
int proxtv_Kolmogorov2_TV_dev(size_t M, size_t N, const double *Y, double lambda, double *X, int maxit, double *info,
                              void *stream);
int proxtv_CondatChambollePock2_TV_dev(size_t M, size_t N, const double *Y, double lambda, double *X, short alg,
                                       int maxit, double *info, void *stream);

/* Batched exact 1-D TV-L1 prox along one dimension of an N-D column-major array (the per-sweep
   kernel of every solver above): out = prox_{lambda}(in) on every fibre along dimension `dim`
   (0-based).  weights == NULL -> uniform lambda; otherwise `weights` has the shape of `in` with
   ns[dim] reduced by one and holds per-edge penalties. */
int proxtv_tv1_fibres_dev(const double *in, double *out, const int *ns /*host*/, int nds, int dim,
                          double lambda, const double *weights, void *stream);

/* The certificate of that prox, on its own: the number of fibres along `dim` for which `out` is NOT the exact TV-L1 prox of `in` --
   the optimality conditions of the 1-D problem checked fibre by fibre (u = cumsum(in - out): |u_k| <= lambda_k, u_k = -+lambda_k where
   out steps up / down, u_{n-1} = 0; reference: what src/TVL1opt.cpp:359-564 solves), within rounding (64 n ulps of the largest sample)
   plus the 4e-10 the reference's own EPSILON tests at a fibre's last sample leave in those sums.
   0 = `out` is the prox; -1 = nothing to check against (lambda <= 0 without weights, in == out); -2 = the call failed
   (proxtv_last_error).  Read-only on both arrays; synchronises the stream.  Option "certify" runs the same check behind every sweep
   of every solver and repairs what fails. */
long proxtv_certify_fibres_dev(const double *in, const double *out, const int *ns /*host*/, int nds, int dim, double lambda,
                               const double *weights, void *stream);

/* Batch of B independent MxN images stored back to back (image b at unary + b*M*N), each solved
   with DR2_TV semantics (oracle = loop of DR2_TV).  All B images advance together, one
   launch per sweep over B*N (columns) / B*M (rows) fibres. */
int proxtv_DR2_TV_batch_dev(size_t M, size_t N, size_t B, const double *unary, double W1, double W2,
                            double *s, int maxit, double *info, void *stream);

/* Same on HOST pointers (stages through HBM like part 1). */
int proxtv_DR2_TV_batch(size_t M, size_t N, size_t B, const double *unary, double W1, double W2, double *s,
                        int maxit, double *info);

/* The same solvers with a TV-L2 penalty (p = 2: lambda ||Dx||_2 per fibre, exact trust-region solve, tv2.hip) allowed
   per dimension / per term; norms in {1, 2}.  proxtv_PD_TVp_dev: which = 0 PD2_TV, 1 PD_TV, 2 PDR_TV (lambdas already
   scaled by npen for 1 and 2, like proxtv_PD_TV_dev). */
int proxtv_tvp_fibres_dev(const double *in, double *out, const int *ns, int nds, int dim, double lambda, double p,
                          void *stream);
int proxtv_DR2_TVp_dev(size_t M, size_t N, const double *unary, double W1, double W2, double norm1, double norm2,
                       double *s, int maxit, double *info, void *stream);
int proxtv_PD_TVp_dev(int which, const double *y, const double *lambdas_scaled, const double *norms, const double *dims,
                      double *x, double *info, const int *ns, int nds, int npen, int maxIters, void *stream);

/* Timing hook for bench.py: average device time in milliseconds of the `which`-th kernel family
   over the last solve (0 = column sweep, 1 = row sweep, 2 = everything else), measured with
   hipEvents on the solve's own stream when option "profile" is 1. */
double proxtv_last_kernel_ms(int which);
/* Fibres the speculative-chunk kernel could not prove and re-solved sequentially during the last solve on this
   thread's stream (0 for noisy data; large when lambda dwarfs the noise).  Diagnostic only: results are exact
   either way. */
long   proxtv_last_fixups(void);
/* Profiling aid (option "trace" = 1): per-workgroup record of the last speculative-chunk kernel launched by this thread,
   8 words each: [0] = XCC_ID << 32 | HW_ID (where it ran), [1..5] = 100 MHz timestamps at start, window staged, walk
   done, rebuild done, end.  Returns the number of workgroups copied to `dst` (at most max_wgs). */
long   proxtv_debug_trace(unsigned long long *dst, long max_wgs);
/* Tuning aid (option "why" = 1): what left work to the repair kernel since the last call, 8 counters: [0] walks that ran off
   their window, [1] links inside a workgroup / wave that stayed unproven, [2] links across workgroups / segments whose codes
   differ, [3] ... that were not published in time, [4] second chances taken across workgroups.  Returns 8, or <= 0. */
int    proxtv_debug_why(unsigned *dst);
/* What ran (process-wide, cumulative since load; tests and tools take differences): "sweep_launches" (fibre-sweep kernels),
   "repair_launches" (sweep_repair_kernel behind them), "repair_jobs_launches" (option repair_jobs), "pin_sweeps" (sweeps the
   pinning solver took), "pin_cap_next_rung" (sweeps its grid-wide variant handed on after the level cap), "tv2_long_fibres"
   (TV-L2 fibres solved parallel inside the fibre), "optimistic_solves" / "optimistic_redone" (option optimistic), "certify_sweeps" / "certify_failures" / "certify_skipped" (option certify:
   sweeps checked, fibres that failed and were re-solved, sweeps that could not be checked).  -1 for an unknown name. */
long   proxtv_debug_counter(const char *name);
/* Geometry policy the adaptive chunk kernel currently uses on this thread (the highest over the sweep families):
   0 = 16-sample warm-up zones (noisy data, small lambda), 1 = the same, robust instantiation (walks may run past
   the window, second-chance rounds inside a block: pieces of ~5 samples), 2 = 64-sample zones (pieces of ~10 samples),
   3 = the pinning solver (exact and data-parallel inside the fibre, any piece length; where it does not apply --
   lambda <= 0, fibres beyond what the device holds -- chunks walked from global memory with 256-sample zones),
   4 = chunks walked from global memory with 1024-sample zones, 5 = one sequential walk per fibre. */
int    proxtv_chunk_mode(void);
/* dst = src with the 8-bytes-per-lane access width of the sweep kernels: a known byte count against which the
   rocprofv3 FETCH_SIZE / WRITE_SIZE counters are calibrated (tools/pmc_traffic.py).  Device pointers. */
int    proxtv_calib_copy_dev(const double *src, double *dst, long n, void *stream);
long   proxtv_last_kernel_launches(int which);

#ifdef __cplusplus
}
#endif
#endif /* PROXTV_AMD_H */
