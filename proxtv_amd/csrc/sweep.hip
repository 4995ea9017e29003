// sweep.hip -- batched exact 1-D TV-L1 prox over the fibres of an N-D array, gfx950.
//
// Mapping: one wavefront lane per fibre, 64 adjacent fibres per wavefront.  For every dimension but the first,
// adjacent fibres are adjacent in memory, so sample k of the 64 lanes is one contiguous 512-byte row: every
// global access of the wave is fully coalesced.  For the first dimension (fibres contiguous in memory) tiles are
// transposed through LDS so that HBM still sees 512-byte rows.  (The reference instead parallelises fibres over
// OpenMP threads with per-thread gather/scatter copies: src/TV2Dopt.cpp:459-523, src/TVNDopt.cpp:164-209.)
//
// Kernel 1, `sweep_seq_kernel`: the sequential lane-per-fibre walk straight from global memory.  Always exact, any
// fibre length, operands may alias outputs.  One image gives only (#fibres / 64) wavefronts, and every step is a
// dependent memory access, so it is latency-bound; it is the fallback of kernel 2 and the path for short fibres.
//
// Kernel 2, `sweep_chunk_kernel`: intra-fibre parallelism by SPECULATIVE CHUNKS.  The walker forgets everything at
// a bend: its state right after one depends only on (restart index, bend type).  So a walk started anywhere from a
// guessed state coincides with the true walk from the first bend they have in common.  Each fibre is cut into
// chunks of C samples; the lane owning chunk c starts H samples early from a free-end state, and owns the outputs
// of [cC, (c+1)C).  Every lane remembers its last bend before its chunk start and before its chunk end; chunk c is
// proven exact iff its last bend at-or-before cC equals that of chunk c-1's lane (which walks through the same zone
// on its way to closing its last piece): two walks share their last bend or none.  Fibres with an unproven link
// (long flat pieces: lambda large against the noise) get their unproven stretches re-walked by `sweep_repair_kernel`
// from the last proven bend, so the result is exact for every input; for noisy data the two walks coincide within a
// handful of samples and nothing needs repair.
// A workgroup = NW wavefronts = NW consecutive chunks of the same 64 fibres, sharing one LDS window
// [first chunk - H, last chunk + T) of the fibre samples; the walk itself touches only LDS.  A lane that finds a bend
// known a priori (|dy| > 4 lambda) just before its chunk starts there instead: exact by construction, nothing to prove.
// A second, "robust" instantiation lets walks run past the window and gives failed links second chances in the block.
//
// Kernel 2a, `sweep_along_kernel`: the same scheme for contiguous fibres -- 64 consecutive chunks of ONE fibre per wave, no
// workgroup barrier; its robust instantiation settles failed links inside the wave and across the waves of a workgroup.
// Short fibres (one block) run as a single-block instantiation of kernel 2 (SHORT) or whole in LDS (kernel 1b).
// Kernel 2b, `sweep_gchunk_kernel`: the same scheme straight from global memory with run-time chunk / zone sizes, for
// pieces of tens to hundreds of samples.  Kernel 3, `sweep_repair_kernel`, finishes what kernels 2 / 2a / 2b left unproven;
// kernels 2 / 2a check the links across their workgroups themselves and tell it in one word whether anything is left.
// Which of these a sweep runs is the geometry policy's business (policy.hpp; plumbing in ChunkScratch below): by default
// a function of sampled statistics of the sweep's input (policy_probe) and of lambda alone.
#include "sweep_kernels.hpp"

namespace ptv {

using namespace swp;

namespace swp {
static thread_local ChunkScratch g_chunks[kMaxDevices];   // per host thread and per device, like the stream and the pool
ChunkScratch &chunk_state() { return g_chunks[current_device()]; }
}  // namespace swp


// Called at the start of every solve on this thread's stream.
void chunk_stats_reset(hipStream_t s) {
    ChunkScratch &st = chunk_state();
    st.optimistic = false;
    const bool adaptive = options().chunk_mode < 0;
    if (st.h_counts) st.poll(true);
    for (int f = 0; f < FAM_COUNT; f++) {
        ChunkScratch::Policy &pl = st.pol[f];
        // what the previous solve left behind: for one-sweep solves (batched 1-D prox calls) this is where the
        // exploration advances
        if (pl.meas) st.settle(f, adaptive);
        pl.begin_solve();
        pl.chunks_done = pl.chunks_seen = 0;
        pl.rewritten_seen = 0;
        st.latest_rewritten[f] = 0;
        st.latest_chunks[f] = 0;
    }
    if (st.failcount) PTV_HIP(hipMemsetAsync(st.failcount, 0, sizeof(int) * ChunkScratch::kCounters, s));
    st.nprobes = 0;
}

// Elements below which a dimension is not sampled: the rung hardly matters there, a stream synchronisation does.
constexpr long kProbeMinElements = 4096;

void policy_probe(const double *y, const double *const *weights, const int *ns, int nds, const int *dims, int ndims, hipStream_t s) {
    if (options().chunk_mode >= 0) return;   // a pinned rung: nothing to decide
    ChunkScratch &st = chunk_state();
    constexpr size_t kWords = kProbeWords;
    int first = st.nprobes;
    for (int k = 0; k < ndims && st.nprobes < ChunkScratch::kMaxProbes; k++) {
        const FibreGeom g = fibres_along(ns, nds, dims[k]);
        const double *w = weights ? weights[k] : nullptr;
        if (g.len < 16 || g.count < 1 || (long)g.len * g.count < kProbeMinElements) continue;   // (below 16 samples: the sequential kernel, always)
        if (g.len < options().chunk_min_len && options().whole != 1) continue;   // (short fibres with the kernel forced: nothing to decide)
        if (st.find_probe(g, w != nullptr)) continue;
        if (!st.probe_dev) st.probe_dev.reset(new Scratch(sizeof(unsigned) * kWords * ChunkScratch::kMaxProbes));
        unsigned *dev = st.probe_dev->as<unsigned>() + kWords * (size_t)st.nprobes;
        if (st.nprobes == first) PTV_HIP(hipMemsetAsync(dev, 0, sizeof(unsigned) * kWords * (size_t)(ChunkScratch::kMaxProbes - first), s));
        edge_histogram(y, w, (long)g.len * g.count, g.inc, g.len, dev, s);
        ChunkScratch::Probe &p = st.probes[st.nprobes++];
        p.inc = g.inc; p.len = g.len; p.count = g.count; p.weighted = (w != nullptr);
        p.iterate = 0;
    }
    if (st.nprobes == first) return;
    // one copy for all the dimensions sampled by this call (the records are contiguous on both sides)
    static_assert(offsetof(ChunkScratch::Probe, hist) % sizeof(unsigned) == 0, "layout");
    for (int k = first; k < st.nprobes; k++)
        PTV_HIP(hipMemcpyAsync(st.probes[k].hist, st.probe_dev->as<unsigned>() + kWords * (size_t)k, sizeof(unsigned) * kWords,
                               hipMemcpyDeviceToHost, s));
    PTV_HIP(hipStreamSynchronize(s));
}

// Mid-solve: the operands of the next sweeps along `dims` are a[k] + c[k] b[k] (b[k] may be null) -- sample THEM, in place of what
// policy_probe recorded for these dimensions.  The iterates of Dykstra / ADMM loops drift away from the solve's input: Yang's sweeps
// run at lambda / rho on averages of smoothed copies (4096^2 unit noise at lambda = 1 on the rung the input asks for: 129 ms a solve,
// at lambda = 3: seven seconds; on the rung the operands ask for: 29 ms).  A function of the data alone, like the first probe: the
// same solve takes the same kernels every time.  One read-back for all the dimensions given.
int policy_reprobe(int kind, int count, const double *const *a, const double *const *b, const double *c, const double *lams, const int *ns, int nds,
                   const int *dims, hipStream_t s) {
    if (options().chunk_mode >= 0) return kReprobeSettled;   // (a pinned rung: nothing to decide, now or later)
    ChunkScratch &st = chunk_state();
    constexpr size_t kWords = kProbeWords;
    bool any = false;
    double f_before[ChunkScratch::kMaxProbes];
    int seed_before[ChunkScratch::kMaxProbes];
    for (int k = 0; k < count && k < ChunkScratch::kMaxProbes; k++) {
        const FibreGeom g = fibres_along(ns, nds, dims[k]);
        f_before[k] = st.certain_fraction(g, lams[k], false);
        seed_before[k] = st.seed(g, lams[k], false);
        const ChunkScratch::Probe *found = st.find_probe(g, false);
        if (!found) continue;   // (not sampled at the start either: too small, or the kernel is not a matter of choice)
        const size_t slot = (size_t)(found - st.probes);
        unsigned *dev = st.probe_dev->as<unsigned>() + kWords * slot;
        PTV_HIP(hipMemsetAsync(dev, 0, sizeof(unsigned) * kWords, s));
        edge_histogram(a[k], nullptr, (long)g.len * g.count, g.inc, g.len, dev, s, b ? b[k] : nullptr, c ? c[k] : 0.0);
        PTV_HIP(hipMemcpyAsync(st.probes[slot].hist, dev, sizeof(unsigned) * kWords, hipMemcpyDeviceToHost, s));
        st.probes[slot].iterate = kind;
        any = true;
    }
    if (!any) return kReprobeSettled;
    PTV_HIP(hipStreamSynchronize(s));
    count_event(CNT_REPROBES);
    // settled: every sampled dimension on the rung whose cost does not depend on the data (the iterates only get smoother).
    // calm: Dykstra operands that stopped moving (same rung as before this sample, certain fraction within 15 %) ; ADMM operands
    // deep in rung-0 territory (their certain fraction falls slowly from there: lambda / rho = 0.04-0.07 on unit noise: 0.90 -> 0.52 .. 0.84 -> 0.14
    // over 32 iterations).
    bool top = true, calm = true;
    for (int k = 0; k < count && k < ChunkScratch::kMaxProbes; k++) {
        const FibreGeom g = fibres_along(ns, nds, dims[k]);
        if (!st.find_probe(g, false)) continue;
        const int seed = st.seed(g, lams[k], false);
        const double f = st.certain_fraction(g, lams[k], false);
        top = top && seed >= 3;
        if (kind == 2) calm = calm && f >= kReprobeCalmAdmm;
        else calm = calm && seed == seed_before[k] && f_before[k] >= 0.0 && fabs(f - f_before[k]) <= 0.15 * fmax(f_before[k], 1e-3);
    }
    return top ? kReprobeSettled : (calm ? kReprobeCalm : kReprobeAskAgain);
}

// The rung (0 or 1) a strided sweep of this geometry will take on the 64-fibre tile, -1 when it will not run there (transposed
// copies, the pinning solver, sequential kernels).  dr2 asks before it chooses the form of its iteration: a function of the
// seed alone, like the rung itself under the default policy.  Call after policy_probe.
int strided_tile_rung(const FibreGeom &g, double lam, bool weighted, double *certain_fraction) {
    if (certain_fraction) *certain_fraction = -1.0;
    if (g.inc == 1 || g.len < options().chunk_min_len) return -1;
    ChunkScratch &st = chunk_state();
    int mode = options().chunk_mode;
    if (mode < 0) mode = st.seed(g, lam, weighted);   // (the hill climb starts from the seed too)
    if (certain_fraction) *certain_fraction = st.certain_fraction(g, lam, weighted);
    if (mode == 0) return 0;
    if (mode != 1) return -1;                         // (unsampled: the pinning rung)
    if (weighted || !(options().along && g.len >= kAlongMinLen)) return 1;
    return 1;
}

bool optimistic_eligible(const FibreGeom *geoms, const double *lams, int n, bool weighted) {
    const Options &o = options();
    if (!o.optimistic || o.chunk_mode >= 0 || !o.deterministic || !o.xlink || o.ablate || o.certify) return false;   // (certify: every sweep is
    // judged as it stands -- a sweep still waiting for its repair would count as a failure)
    ChunkScratch &st = chunk_state();
    for (int k = 0; k < n; k++) {
        if (geoms[k].len < o.chunk_min_len) continue;               // (sequential / whole-fibre kernels: nothing to repair)
        if (st.seed(geoms[k], lams[k], weighted) != 0) return false;   // (rung 1 and up -- or unsampled: repairs are part of the plan there)
    }
    // Rung 0 by the statistics is not "no repairs": unit noise at lambda = 0.2 leaves four fibres per 4096^2 solve to the repair kernel,
    // an image with a few constant rows some in every sweep -- and a solve run twice costs what seventy solves save.  So the bracket
    // backs off: after a solve that had to be run again, the next kOptimisticBackoff eligible solves of this thread run with their
    // repairs (same bits either way; proxtv_set_option("optimistic", ...) forgets the history).
    if (st.optimistic_backoff > 0) {
        st.optimistic_backoff--;
        return false;
    }
    return true;
}
void optimistic_forget() { chunk_state().optimistic_backoff = 0; }
OptimisticScope::OptimisticScope(hipStream_t s_, bool on_) : s(s_), on(on_) {
    if (on) chunk_state().begin_optimistic(s);
}
OptimisticScope::~OptimisticScope() { chunk_state().optimistic = false; }
bool OptimisticScope::clean() {
    if (!on) return true;
    on = false;
    const bool clean = chunk_state().end_optimistic(s);
    if (!clean) chunk_state().optimistic_backoff = kOptimisticBackoff;
    return clean;
}

long chunk_stats_fixups(hipStream_t s) {
    if (!chunk_state().failcount) return 0;
    int h[ChunkScratch::kCounters] = {};
    PTV_HIP(hipMemcpyAsync(h, chunk_state().failcount, sizeof(h), hipMemcpyDeviceToHost, s));
    PTV_HIP(hipStreamSynchronize(s));
    long total = 0;
    for (int f = 0; f < FAM_COUNT; f++) total += h[2 * f];
    return total;
}

// option "trace": copy the phase timestamps of the last chunk-kernel launch of this thread (8 words per workgroup) to the host
long chunk_trace_fetch(unsigned long long *dst, long max_wgs, hipStream_t s) {
    if (!chunk_state().trace || max_wgs <= 0) return 0;
    const long n = (long)chunk_state().trace_wgs < max_wgs ? (long)chunk_state().trace_wgs : max_wgs;
    PTV_HIP(hipMemcpyAsync(dst, chunk_state().trace->as<unsigned long long>(), (size_t)n * 64, hipMemcpyDeviceToHost, s));
    PTV_HIP(hipStreamSynchronize(s));
    return n;
}

// option "why": read (and clear) the counters of what marked sweeps dirty on this thread
int chunk_why_fetch(unsigned *dst, hipStream_t s) {
    ChunkScratch &st = chunk_state();
    if (!st.dirty_word) return 0;
    PTV_HIP(hipMemcpyAsync(dst, st.dirty_word->as<unsigned>() + 1, sizeof(unsigned) * 8, hipMemcpyDeviceToHost, s));
    PTV_HIP(hipMemsetAsync(st.dirty_word->as<unsigned>() + 1, 0, sizeof(unsigned) * 8, s));
    PTV_HIP(hipStreamSynchronize(s));
    return 8;
}

int chunk_stats_mode() {
    int m = 0;
    for (int f = 0; f < FAM_COUNT; f++)   // (families the last solve did not use keep whatever an earlier workload left them)
        if (chunk_state().pol[f].sweeps > 0 && chunk_state().pol[f].mode > m) m = chunk_state().pol[f].mode;
    return m;
}

void launch_seq_gated(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *flags) {
#define PTV_GATED(ID, W) \
    if (op == ID && weighted == W) return unit_gated<ID, W>(args, g, stream, flags);
    PTV_SWEEP_UNITS(PTV_GATED)
#undef PTV_GATED
    set_error("launch_seq_gated: no %s sweep for op %d", weighted ? "weighted" : "unweighted", (int)op);
    throw HipFailure{hipErrorInvalidValue};
}

void launch_sweep(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam,
                  bool allow_chunked) {
    if (g.count <= 0 || g.len <= 0) return;
    if (transpose_cache().active) {   // transposed copies of what this sweep is about to write are stale
        transpose_cache().forget(args.o0);
        transpose_cache().forget(args.o1);
    }
    FamilyTimer timer(fam, stream);
    // one translation unit per (op, weighted) pair: sweep_unit.hip
#define PTV_CASE(ID, W) \
    if (op == ID && weighted == W) return unit_launch<ID, W>(args, g, stream, allow_chunked, fam);
    PTV_SWEEP_UNITS(PTV_CASE)
#undef PTV_CASE
    set_error("launch_sweep: no %s sweep for op %d", weighted ? "weighted" : "unweighted", (int)op);
    throw HipFailure{hipErrorInvalidValue};
}


long certify_sweep(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream) {
#define PTV_CERT(ID, W) \
    if (op == ID && weighted == W) return unit_certify<ID, W>(args, g, stream);
    PTV_SWEEP_UNITS(PTV_CERT)
#undef PTV_CERT
    set_error("certify_sweep: no %s sweep for op %d", weighted ? "weighted" : "unweighted", (int)op);
    throw HipFailure{hipErrorInvalidValue};
}

void warm_sweep() {
#define PTV_WARM(ID, W) unit_warm<ID, W>();
    PTV_SWEEP_UNITS(PTV_WARM)
#undef PTV_WARM
}

}  // namespace ptv
