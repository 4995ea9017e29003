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
// of [cC, (c+1)C).  Every lane records (as bit masks over the H-sample zone before a chunk boundary) where its walk
// bent; chunk c is proven exact iff its own bends and those of chunk c-1's lane (which walks through the same zone
// on its way to closing its last piece) share one bend at or before the boundary.  Fibres with an unproven link
// (long flat pieces: lambda large against the noise) are re-solved by kernel 1 in `sweep_fix_kernel`, so the result
// is exact for every input; for noisy data the two walks coincide within a handful of samples.
// A workgroup = NW wavefronts = NW consecutive chunks of the same 64 fibres, sharing one LDS window
// [first chunk - H, last chunk + T) of the fibre samples; the walk itself touches only LDS.
#include "sweep.hpp"

#include <memory>

#include "walker.hpp"

namespace ptv {

namespace {

using link_t = unsigned;                 // (restart << 1 | bend type) of a walk's last bend before a chunk boundary
constexpr link_t kLinkAlwaysOk = 0xffffffffu;   // the chunk's walk began at sample 0: exact by construction

// ---- kernel 1: sequential walk straight from / to global memory --------------------------------------------------
template <int OP, bool WEIGHTED>
struct SeqSource {
    const SweepArgs &p;
    long base, inc, wbase;
    __device__ __forceinline__ double y(int i) const { return Op<OP>::load_y(p, base + (long)i * inc); }
    __device__ __forceinline__ double r(int i) const { return p.w[wbase + (long)i * inc]; }
    __device__ __forceinline__ void piece(int from, int to, double v) const {
        for (int j = from; j <= to; j++) {
            const long idx = base + (long)j * inc;
            Op<OP>::finish(p, idx, Op<OP>::fetch(p, idx), Op<OP>::load_y(p, idx), v);
        }
    }
    __device__ __forceinline__ void bend(int, int) const {}
    __device__ __forceinline__ bool keep_going(int) const { return true; }
};

template <int OP, bool WEIGHTED>
__device__ __forceinline__ void solve_fibre_seq(const SweepArgs &p, const FibreGeom &g, long j) {
    const long blk = j / g.inc, off = j % g.inc;
    SeqSource<OP, WEIGHTED> src{p, blk * g.inc * g.len + off, g.inc, blk * g.inc * (g.len - 1) + off};
    if (WEIGHTED && g.len == 1) {  // no edge at all: prox is the identity (the reference reads lambda[0] out of bounds here)
        const double y0 = src.y(0);
        Op<OP>::finish(p, src.base, Op<OP>::fetch(p, src.base), y0, y0);
        return;
    }
    Walker w;
    walker_start<WEIGHTED>(w, src, 0, p.lam);
    walker_run<WEIGHTED>(w, src, g.len, p.lam);
}

template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void sweep_seq_kernel(SweepArgs p, FibreGeom g) {
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    if (j >= g.count || g.len <= 0) return;
    solve_fibre_seq<OP, WEIGHTED>(p, g, j);
}

// ---- kernel 2: speculative chunks over an LDS window -----------------------------------------------------------------
struct ChunkPlan {
    int H;      // warm-up / synchronisation zone, samples (<= 32)
    int T;      // look-ahead rows kept in LDS past the workgroup's last chunk
    int Q;      // workgroups (chunk blocks) per fibre
    int rows;   // LDS window rows = H + NW*C + T
    int ablate; // profiling aid (option "ablate"): 1 = skip the walk, 2 = skip the epilogue, 4 = skip the window loads
};

template <int OP, bool WEIGHTED, int PITCH>
struct ChunkSource {
    const SweepArgs &p;
    long base, inc, wbase;     // this lane's fibre in global memory
    const double *Y;           // LDS window, already offset to this lane's column; row r of the fibre at Y[(r - lo) * PITCH]
    const double *Wt;          // LDS per-edge penalties, same addressing (weighted sweeps)
    double *V;                 // LDS piece values: V[(brk - vrow0) * PITCH] = value of the piece ending at brk
    int lo, hi;                // window rows present in LDS
    int cs, ce;                // samples owned by this lane: [cs, ce)
    int vrow0;                 // first row of the workgroup's V plane
    int len;                   // fibre length
    unsigned ends = 0;         // bit k: a piece ends at sample cs + k (k < ce - 1 - cs; chunks are at most 32 samples)
    // Link proof.  Two walks that share one bend are identical from it on, so inside the zone before a chunk boundary
    // they either share their LAST bend or share none: it is enough to remember, as (restart << 1 | type), the last
    // bend at or before the own chunk start (`mine`) and at or before the own chunk end (`next`, read by the lane of
    // the following chunk, whose walk started H samples before that boundary).  0 = none.
    unsigned mine = 0, next = 0;
    double vclose = 0.0;       // value of the piece covering sample ce - 1
    bool done = false;         // the piece covering ce - 1 is closed
    bool failed = false;       // the walk ran off the LDS window (a piece much longer than a chunk): give the fibre up

    // The speculative walk never leaves the LDS window: a lane that would (a long flat piece) marks its fibre for the
    // sequential kernel instead -- this bounds the cost of a chunk by its window whatever the data.
    __device__ __forceinline__ double y(int i) const { return Y[(min(i, hi - 1) - lo) * PITCH]; }
    __device__ __forceinline__ double r(int i) const { return Wt[(min(i, hi - 1) - lo) * PITCH]; }
    __device__ __forceinline__ void piece(int, int to, double v) {
        if (to >= ce - 1) {
            vclose = v;
            done = true;
        } else if (to >= cs) {
            V[(to - vrow0) * PITCH] = v;
            ends |= 1u << (to - cs);
        }
    }
    __device__ __forceinline__ void bend(int at, int type) {
        const unsigned code = ((unsigned)at << 1) | (unsigned)type;
        mine = (at <= cs) ? code : mine;
        next = (at <= ce) ? code : next;
    }
    __device__ __forceinline__ bool keep_going(int i) {
        if (done) return false;
        if (i >= hi) {   // hi == len for the windows that reach the fibre end, so this is never the true end
            failed = true;
            return false;
        }
        return true;
    }
};

// a / s for a small positive integer s held as a double: reciprocal (v_rcp_f64) + one Newton step, then one
// residual correction of the quotient.  Agrees with IEEE division to the last bit in all but rare ties (at most
// one ulp off) at a third of the instructions of the full v_div_* sequence; both tube pieces share `inv`.
__device__ __forceinline__ double refined_rcp(double s) {
    double inv = __builtin_amdgcn_rcp(s);
    inv = __builtin_fma(__builtin_fma(-s, inv, 1.0), inv, inv);
    return inv;
}
__device__ __forceinline__ double div_by(double a, double s, double inv) {
    const double q = a * inv;
    return __builtin_fma(__builtin_fma(-q, s, a), inv, q);
}

// Hot loop of the chunked kernel: the interior steps of the walk (sample index below the last sample of the fibre
// and inside the LDS window), same state machine and arithmetic order as walker_run, hand-shaped for the wave:
//   * y of the next sample is requested before the current one is processed (the dependent LDS latency hides
//     behind the step); a bend that rewinds re-reads;
//   * both "pull back inside the tube" updates are branch-free selects sharing one reciprocal;
//   * only the bend path is a divergent region.
// Leaves the walker at the first sample it does not handle (i == len - 1, window exhausted, or done).
template <bool WEIGHTED, class S>
__device__ __forceinline__ void walker_run_interior(Walker &w, S &src, int n, double lam) {
    const int last = n - 1;
    const int lim = min(last, src.hi);   // handle i < lim only
    if (w.i >= lim || src.done) return;
    double yi = src.y(w.i);
    while (true) {
        const int i = w.i;
        const bool live = !src.done && i < lim;
        if (!live) break;
        const double ynext = src.y(i + 1);               // speculative: most steps advance by one
        const double r = WEIGHTED ? src.r(i) : lam;
        const double h1 = w.hlo + (w.lo - yi);
        const bool cv = r < h1;
        const double h2 = w.hhi + (w.hi - yi);
        const bool fv = !cv && (-r > h2);
        if (cv || fv) {
            const int brk = cv ? w.klo : w.khi;
            src.piece(w.k0 + 1, brk, cv ? w.lo : w.hi);
            const int at = brk + 1;                       // at <= i < last: the restart is an interior sample
            src.bend(at, cv ? BEND_CEIL : BEND_FLOOR);
            const double yn = (at == i) ? yi : src.y(at);
            if (WEIGHTED) {
                const double wp = src.r(at - 1), wc = (at == i) ? r : src.r(at);
                if (cv) { w.lo = yn + wp - wc; w.hi = yn + wp + wc; }
                else    { w.hi = yn - wp + wc; w.lo = yn - wp - wc; }
                w.hhi = wc;
                w.hlo = -wc;
            } else {
                if (cv) { w.lo = yn; w.hi = 2 * lam + yn; }
                else    { w.hi = yn; w.lo = 2 * (-lam) + yn; }
                w.hhi = lam;
                w.hlo = -lam;
            }
            w.k0 = brk;
            w.klo = w.khi = at;
            w.i = at + 1;
            yi = (at == i) ? ynext : src.y(at + 1);
        } else {
            const double s = (double)(i - w.k0);
            const double inv = refined_rcp(s);
            const bool th = h2 >= r, tl = h1 <= -r;
            const double nhi = w.hi + div_by(r - h2, s, inv);
            const double nlo = w.lo + div_by(-r - h1, s, inv);
            w.hi = th ? nhi : w.hi;
            w.hhi = th ? r : h2;
            w.khi = th ? i : w.khi;
            w.lo = tl ? nlo : w.lo;
            w.hlo = tl ? -r : h1;
            w.klo = tl ? i : w.klo;
            w.i = i + 1;
            yi = ynext;
        }
    }
}

// LDS carve (dynamic, 16-byte aligned base): Y | Wt (weighted) | V | next-masks
template <int OP, bool WEIGHTED, bool TRANSPOSED, int C, int NW>
__global__ __launch_bounds__(64 * NW) void sweep_chunk_kernel(SweepArgs p, FibreGeom g, ChunkPlan plan, link_t *link_in,
                                                               link_t *link_out, int *failflags) {
    constexpr int PITCH = TRANSPOSED ? 65 : 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Yp = reinterpret_cast<double *>(smem);
    double *Wp = Yp + (WEIGHTED ? (size_t)plan.rows * PITCH : 0);
    double *Vp = Wp + (size_t)plan.rows * PITCH;
    link_t *nextmask = reinterpret_cast<link_t *>(Vp + (size_t)NW * C * PITCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = g.len;
    const int q = blockIdx.y;
    const int cs_wg = q * NW * C;
    const int lo = max(0, cs_wg - plan.H);
    const int hi = min(len, cs_wg + NW * C + plan.T);
    const long j0 = (long)blockIdx.x * 64;
    const long j = j0 + lane;
    const bool active = j < g.count;
    long base = 0, wbase = 0;
    if (active) {
        const long blk = j / g.inc, off = j % g.inc;
        base = blk * g.inc * len + off;
        wbase = blk * g.inc * (len - 1) + off;
    }

    // ---- stage the window: every global read is a coalesced 512-byte row -------------------------------------------
    // Loads are issued in batches of UL independent requests per lane before anything waits on them: the window is
    // small, so memory-level parallelism inside the wave is what hides the HBM latency.
    constexpr int UL = 8;
    if (plan.ablate & 4) {
        for (int e = tid; e < plan.rows * PITCH; e += 64 * NW) Yp[e] = (double)((e * 2654435761u) >> 20) * 1e-3;
    } else if (!TRANSPOSED) {
        if (active) {
            for (int r0 = lo + wave * UL; r0 < hi; r0 += NW * UL) {
                double ty[UL], tw[UL];
#pragma unroll
                for (int u = 0; u < UL; u++) {
                    const int r = r0 + u;
                    ty[u] = (r < hi) ? Op<OP>::load_y(p, base + (long)r * g.inc) : 0.0;
                    if (WEIGHTED) tw[u] = (r < len - 1) ? p.w[wbase + (long)r * g.inc] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UL; u++) {
                    const int r = r0 + u;
                    if (r < hi) {
                        Yp[(r - lo) * PITCH + lane] = ty[u];
                        if (WEIGHTED) Wp[(r - lo) * PITCH + lane] = tw[u];
                    }
                }
            }
        }
    } else {
        // fibres are contiguous (inc == 1): lanes run along the fibre, the tile is transposed on its way into LDS.
        // Work item t of a wave = (fibre wave + NW * (t / RB), row block t % RB); UL items in flight.
        const int nrows = hi - lo;
        const int RB = (nrows + 63) / 64;
        const int items = (64 / NW) * RB;
        for (int t0 = 0; t0 < items; t0 += UL) {
            double ty[UL], tw[UL];
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const int t = t0 + u;
                const int f = wave + NW * (t / RB);
                const int r = lo + (t % RB) * 64 + lane;
                const long jf = j0 + f;
                const bool ok = t < items && jf < g.count && r < hi;
                ty[u] = ok ? Op<OP>::load_y(p, jf * len + r) : 0.0;
                if (WEIGHTED) tw[u] = (ok && r < len - 1) ? p.w[jf * (len - 1) + r] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const int t = t0 + u;
                const int f = wave + NW * (t / RB);
                const int r = lo + (t % RB) * 64 + lane;
                if (t < items && j0 + f < g.count && r < hi) {
                    Yp[(r - lo) * PITCH + f] = ty[u];
                    if (WEIGHTED) Wp[(r - lo) * PITCH + f] = tw[u];
                }
            }
        }
    }
    __syncthreads();

    // ---- speculative walk of this wave's chunk ------------------------------------------------------------------------
    const int cs = cs_wg + wave * C;
    const int ce = min(cs + C, len);
    const bool has_chunk = active && cs < len;
    const int start = max(0, cs - plan.H);
    ChunkSource<OP, WEIGHTED, PITCH> src{p, base, g.inc, wbase, Yp + lane, Wp + lane, Vp + lane,
                                         lo, hi, cs, ce, cs_wg, len};
    if (has_chunk && !(plan.ablate & 1)) {
        Walker w;
        walker_start<WEIGHTED>(w, src, start, p.lam);
        walker_run_interior<WEIGHTED>(w, src, len, p.lam);   // the hot part
        walker_run<WEIGHTED>(w, src, len, p.lam);            // fibre end / window end (no-op for most lanes)
        if (src.failed) failflags[j] = 1;
    }

    // ---- prove the links between consecutive chunks ----------------------------------------------------------------------
    nextmask[wave * 64 + lane] = src.next;
    __syncthreads();
    if (has_chunk) {
        const bool true_start = (start == 0);
        if (wave > 0) {
            if (!true_start && (src.mine == 0 || src.mine != nextmask[(wave - 1) * 64 + lane])) failflags[j] = 1;
        } else if (q > 0) {
            link_in[(long)q * g.count + j] = true_start ? kLinkAlwaysOk : src.mine;
        }
        // the lane owning the workgroup's last chunk publishes its bends for the next workgroup's first chunk
        if (q + 1 < plan.Q && (wave == NW - 1)) link_out[(long)q * g.count + j] = src.next;
    }

    // ---- write the outputs of [cs, ce): backward fill from the piece ends ---------------------------------------------------
    // Step 1 (LDS only): expand the piece ends into one prox value per owned sample, in place in the V plane.
    if (has_chunk) {
        double cur = src.vclose;
        for (int k = ce - 1; k >= cs; k--) {
            if (k < ce - 1 && ((src.ends >> (k - cs)) & 1)) cur = Vp[(k - cs_wg) * PITCH + lane];
            Vp[(k - cs_wg) * PITCH + lane] = cur;
        }
    }
    __syncthreads();
    // Step 2: the whole workgroup streams its NW*C rows out, every global access a coalesced 512-byte row, the
    // operand fetches of UL rows in flight before the first dependent store.
    const int ce_wg = min(len, cs_wg + NW * C);
    if (plan.ablate & 2) return;
    if (!TRANSPOSED) {
        if (active) {
            for (int k0 = cs_wg + wave * UL; k0 < ce_wg; k0 += NW * UL) {
                Ext ex[UL];
#pragma unroll
                for (int u = 0; u < UL; u++) {
                    const int k = min(k0 + u, ce_wg - 1);
                    ex[u] = Op<OP>::fetch(p, base + (long)k * g.inc);
                }
#pragma unroll
                for (int u = 0; u < UL; u++) {
                    const int k = k0 + u;
                    if (k < ce_wg)
                        Op<OP>::finish(p, base + (long)k * g.inc, ex[u], Yp[(k - lo) * PITCH + lane],
                                       Vp[(k - cs_wg) * PITCH + lane]);
                }
            }
        }
    } else {
        const int nrows = ce_wg - cs_wg;
        const int RB = (nrows + 63) / 64;
        const int items = (64 / NW) * RB;
        for (int t0 = 0; t0 < items; t0 += UL) {
            Ext ex[UL];
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const int t = t0 + u;
                const int f = wave + NW * (t / RB);
                const int k = cs_wg + (t % RB) * 64 + lane;
                const long jf = j0 + f;
                const bool ok = t < items && jf < g.count && k < ce_wg;
                ex[u] = ok ? Op<OP>::fetch(p, jf * len + k) : Ext{0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const int t = t0 + u;
                const int f = wave + NW * (t / RB);
                const int k = cs_wg + (t % RB) * 64 + lane;
                const long jf = j0 + f;
                if (t < items && jf < g.count && k < ce_wg)
                    Op<OP>::finish(p, jf * len + k, ex[u], Yp[(k - lo) * PITCH + f], Vp[(k - cs_wg) * PITCH + f]);
            }
        }
    }
}

// Re-solves, sequentially and from the untouched operands, every fibre with an unproven link.
template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void sweep_fix_kernel(SweepArgs p, FibreGeom g, int Q, const link_t *link_in,
                                                        const link_t *link_out, int *failflags, int *failcount) {
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    if (j >= g.count) return;
    int bad = failflags[j];
    for (int q = 1; q < Q; q++)
    {
        const link_t in = link_in[(long)q * g.count + j];
        if (in != kLinkAlwaysOk && (in == 0 || in != link_out[(long)(q - 1) * g.count + j])) bad = 1;
    }
    if (!bad) return;
    failflags[j] = 0;
    atomicAdd(failcount, 1);
    solve_fibre_seq<OP, WEIGHTED>(p, g, j);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
template <int OP, bool WEIGHTED>
void launch_seq(const SweepArgs &args, const FibreGeom &g, hipStream_t stream) {
    const unsigned blocks = (unsigned)((g.count + 63) / 64);
    if (blocks == 0) return;
    hipLaunchKernelGGL((sweep_seq_kernel<OP, WEIGHTED>), dim3(blocks), dim3(64), 0, stream, args, g);
    PTV_HIP(hipGetLastError());
}

// persistent per-thread scratch of the chunked path: link masks, fail flags, fail counter
struct ChunkScratch {
    std::unique_ptr<Scratch> links, flags;
    size_t link_bytes = 0, flag_count = 0;
    link_t *link_in = nullptr, *link_out = nullptr;
    int *failflags = nullptr, *failcount = nullptr;
    void ensure(long count, int Q, hipStream_t s) {
        const size_t need = sizeof(link_t) * (size_t)count * (size_t)Q * 2;
        if (need > link_bytes) {
            links.reset(new Scratch(need));
            link_bytes = need;
        }
        link_in = links->as<link_t>();
        link_out = link_in + (size_t)count * (size_t)Q;
        if ((size_t)count + 1 > flag_count) {
            flags.reset(new Scratch(sizeof(int) * ((size_t)count + 1)));
            flag_count = (size_t)count + 1;
            PTV_HIP(hipMemsetAsync(flags->as<int>(), 0, sizeof(int) * flag_count, s));
        }
        failcount = flags->as<int>();
        failflags = failcount + 1;
    }
};
static thread_local ChunkScratch g_chunk;

template <int OP, bool WEIGHTED, bool TRANSPOSED, int C, int NW>
void launch_chunk_cfg(const SweepArgs &args, const FibreGeom &g, hipStream_t stream) {
    constexpr int PITCH = TRANSPOSED ? 65 : 64;
    ChunkPlan plan;
    plan.H = options().warmup < 1 ? 1 : (options().warmup > 32 ? 32 : options().warmup);
    plan.T = 8;
    plan.Q = (g.len + NW * C - 1) / (NW * C);
    plan.rows = plan.H + NW * C + plan.T;
    plan.ablate = options().ablate;
    const size_t lds = sizeof(double) * PITCH * ((size_t)plan.rows * (WEIGHTED ? 2 : 1) + (size_t)NW * C) +
                       sizeof(link_t) * NW * 64;
    if (lds > 160 * 1024) {
        set_error("chunk geometry needs %zu bytes of LDS", lds);
        throw HipFailure{hipErrorInvalidValue};
    }
    g_chunk.ensure(g.count, plan.Q, stream);
    auto kern = sweep_chunk_kernel<OP, WEIGHTED, TRANSPOSED, C, NW>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr_set = true;
    }
    const dim3 grid((unsigned)((g.count + 63) / 64), (unsigned)plan.Q);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, args, g, plan, g_chunk.link_in, g_chunk.link_out,
                       g_chunk.failflags);
    if (!plan.ablate)
        hipLaunchKernelGGL((sweep_fix_kernel<OP, WEIGHTED>), dim3((unsigned)((g.count + 63) / 64)), dim3(64), 0, stream,
                           args, g, plan.Q, g_chunk.link_in, g_chunk.link_out, g_chunk.failflags, g_chunk.failcount);
    PTV_HIP(hipGetLastError());
}

// Chunk geometries that fit the 160 KiB of LDS per CU (window + piece values [+ penalties]):
//   unweighted  C = 32, NW = 4 : ~150 KiB -> 1 workgroup (4 waves) per CU;  C = 16, NW = 4 : ~79 KiB -> 2 per CU
//   weighted    C = 16, NW = 4 : ~125 KiB -> 1 per CU
template <int OP, bool WEIGHTED, bool TRANSPOSED>
void launch_chunk(const SweepArgs &args, const FibreGeom &g, hipStream_t stream) {
    constexpr int NW = 4;
    if (WEIGHTED || options().chunk <= 16) launch_chunk_cfg<OP, WEIGHTED, TRANSPOSED, 16, NW>(args, g, stream);
    else                                   launch_chunk_cfg<OP, WEIGHTED, TRANSPOSED, 32, NW>(args, g, stream);
}

template <int OP, bool WEIGHTED>
void launch_op_w(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool allow_chunked) {
    const int c = options().chunk;
    // chunking pays once a fibre spans several workgroups' worth of chunks; short fibres stay sequential
    const bool chunked = allow_chunked && c > 0 && g.len >= 256;
    if (!chunked) launch_seq<OP, WEIGHTED>(args, g, stream);
    else if (g.inc == 1) launch_chunk<OP, WEIGHTED, true>(args, g, stream);
    else launch_chunk<OP, WEIGHTED, false>(args, g, stream);
}

}  // namespace

void chunk_stats_reset(hipStream_t s) {
    if (g_chunk.failcount) PTV_HIP(hipMemsetAsync(g_chunk.failcount, 0, sizeof(int), s));
}

long chunk_stats_fixups(hipStream_t s) {
    if (!g_chunk.failcount) return 0;
    int h = 0;
    PTV_HIP(hipMemcpyAsync(&h, g_chunk.failcount, sizeof(int), hipMemcpyDeviceToHost, s));
    PTV_HIP(hipStreamSynchronize(s));
    return h;
}

void launch_sweep(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam,
                  bool allow_chunked) {
    if (g.count <= 0 || g.len <= 0) return;
    FamilyTimer timer(fam, stream);
#define PTV_CASE(ID)                                                                             \
    case ID:                                                                                     \
        if (weighted) launch_op_w<ID, true>(args, g, stream, allow_chunked);                     \
        else          launch_op_w<ID, false>(args, g, stream, allow_chunked);                    \
        break;
#define PTV_CASE_U(ID) case ID: launch_op_w<ID, false>(args, g, stream, allow_chunked); break;
#define PTV_CASE_W(ID) case ID: launch_op_w<ID, true>(args, g, stream, allow_chunked); break;
    switch (op) {
        PTV_CASE(OP_PROX)
        PTV_CASE(OP_DR_COL)
        PTV_CASE(OP_DR_COL_FINAL)
        PTV_CASE_U(OP_DR_ROW)
        PTV_CASE_U(OP_DR_ROW_FINAL)
        PTV_CASE_W(OP_DRW_ROW)
        PTV_CASE_W(OP_DRW_ROW_FINAL)
        PTV_CASE_U(OP_PD2_A)
        PTV_CASE_U(OP_PD2_B)
        PTV_CASE_U(OP_YANG)
        default:
            set_error("launch_sweep: unknown op %d", (int)op);
            throw HipFailure{hipErrorInvalidValue};
    }
#undef PTV_CASE
#undef PTV_CASE_U
#undef PTV_CASE_W
}

}  // namespace ptv
