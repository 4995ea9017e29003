// sweep.hpp -- host interface of the fibre-sweep kernels (sweep.hip).
#pragma once

#include "common.hpp"
#include "ops.hpp"

namespace ptv {

// One sweep: for every fibre of `g`, y = Op::load_y, x = prox(y), Op::store.
// `weighted` selects per-edge penalties args.w (length len-1 per fibre, laid out like the data array with the
// fibre dimension shortened by one); otherwise the uniform penalty args.lam.
// `fam` tags the launch for the per-family timers (FAM_COL / FAM_ROW / FAM_OTHER).
//
// Aliasing contract: outputs may alias operands element-for-element ONLY when
// `allow_chunked` is false (sequential kernel: each element is read and written by the one lane that owns the
// fibre, and never re-read after it was written).  The chunked kernels read operand rows that other workgroups
// write, so their callers pass distinct in/out arrays (ping-pong) -- see solvers.hip.
void launch_sweep(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam,
                  bool allow_chunked);

// The certificate alone (option certify runs it behind every sweep): the number of fibres for which what a sweep of `op` wrote is NOT
// the prox of what it read -- the optimality conditions of the 1-D problem, checked per fibre (kernel_certify.hpp) -- or -1 when
// the sweep cannot be checked (lambda <= 0, an output aliasing an operand).  Synchronises `stream`.
long certify_sweep(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream);

// Seed of the geometry policy for the solve that is starting: samples the edge statistics of `y` along each of the `ndims`
// dimensions `dims` (0-based; weights[k], when non-null, are the per-edge penalties of a weighted sweep along dims[k]) and
// reads them back (one small copy + stream synchronisation per call).  Dimensions whose fibres never reach the chunked
// kernels, or that this solve has already sampled, are skipped.  Call after chunk_stats_reset(), before the first sweep.
void policy_probe(const double *y, const double *const *weights, const int *ns, int nds, const int *dims, int ndims, hipStream_t s);
// mid-solve: the operands of the next sweeps along dims[k] are a[k] + c[k] b[k] (b / c may be null): sample them in place of the solve's input
// (kind: 1 = a Dykstra loop's x + p, 2 = an ADMM loop's X - U / rho: policy.hpp reads their certain fractions differently)
// lams[k]: the penalty of the sweeps along dims[k].  Returns kReprobeSettled when there is nothing left to ask -- every sampled dimension now
// seeds the pinning rung (or none was sampled, or the rung is pinned) --, kReprobeCalm when the operands have stopped moving (Dykstra: a loop
// may stop sampling) or sit deep in rung-0 territory (ADMM: the sparse schedule will do), kReprobeAskAgain otherwise.
constexpr int kReprobeAskAgain = 0, kReprobeSettled = 1, kReprobeCalm = 2;
constexpr double kReprobeCalmAdmm = 0.7;
int policy_reprobe(int kind, int count, const double *const *a, const double *const *b, const double *c, const double *lams, const int *ns, int nds,
                   const int *dims, hipStream_t s);
// (the schedule of the samples: policy.hpp, reprobe_at)

// Fibres that the chunked path had to re-solve sequentially (unproven chunk links) since the last reset, on this thread.
void chunk_stats_reset(hipStream_t s);
long chunk_stats_fixups(hipStream_t s);
// the rung (0 / 1) a strided sweep of this geometry will take on the 64-fibre tile, -1 if it will not run there.  Call after policy_probe.
// (*certain_fraction: the seed statistic of that sweep's input, -1 when it was not sampled)
int strided_tile_rung(const FibreGeom &g, double lam, bool weighted, double *certain_fraction = nullptr);
// Optimistic solves (option "optimistic"; solvers.hip: dr2).  A chunked sweep on rung 0 hardly ever leaves anything to the repair kernel
// (profiles/r06_dirty_rate.txt: none in 2 130 sweeps of the headline), yet the empty repair launch behind it costs a dependent launch.  A
// solve whose every sweep will run on rung 0 -- `optimistic_eligible`: decided from the sampled statistics, like the rung itself -- may
// run between optimistic_begin / optimistic_end: no repair launches, a sticky word instead; optimistic_end (synchronises) says whether
// every sweep was clean.  If not, the caller runs the solve again outside the bracket: that run is the exact one.
bool optimistic_eligible(const FibreGeom *geoms, const double *lams, int n, bool weighted);
void optimistic_forget();   // (the calling thread's back-off history)
constexpr int kOptimisticBackoff = 64;
struct OptimisticScope {
    explicit OptimisticScope(hipStream_t s, bool on);
    ~OptimisticScope();
    bool clean();   // ends the bracket (synchronises when it was on); true: nothing was left anywhere (always true when it was off)
    hipStream_t s;
    bool on;
};
// current geometry policy of this thread (highest over the sweep families the last solve used): 0 / 1 / 2 = LDS windows
// (16-sample zones, the same with second-chance rounds, 64-sample zones), 3 = the pinning solver (or global-memory chunks
// where it does not apply), 4 = global-memory chunks, 5 = sequential
int chunk_stats_mode();
long chunk_trace_fetch(unsigned long long *dst, long max_wgs, hipStream_t s);
int chunk_why_fetch(unsigned *dst /*[8]*/, hipStream_t s);

}  // namespace ptv
