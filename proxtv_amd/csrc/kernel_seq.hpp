// kernel_seq.hpp -- kernel 1: one sequential walk per fibre, straight from / to global memory.
// (One of the pieces of sweep_kernels.hpp, which includes them in order; not meant to be included on its own.)
#pragma once

namespace ptv {
namespace swp {

// ---- kernel 1: sequential walk straight from / to global memory --------------------------------------------------
// Global-memory walks fetch their samples kGlobalBlock at a time (walker_run_blocked) and write a piece out the same
// way: a batch of independent operand fetches, then the batch of stores -- one memory round trip per batch instead of
// one per sample.
constexpr int kGlobalBlock = 8;

template <int OP>
__device__ __forceinline__ void write_run(const SweepArgs &p, long base, long inc, int from, int to, double v) {
    int k = from;
    for (; k + kGlobalBlock - 1 <= to; k += kGlobalBlock) {
        Ext e[kGlobalBlock];
#pragma unroll
        for (int u = 0; u < kGlobalBlock; u++) e[u] = Op<OP>::fetch(p, base + (long)(k + u) * inc);
#pragma unroll
        for (int u = 0; u < kGlobalBlock; u++) Op<OP>::finish(p, base + (long)(k + u) * inc, e[u], v);
    }
    for (; k <= to; k++) {
        const long idx = base + (long)k * inc;
        Op<OP>::finish(p, idx, Op<OP>::fetch(p, idx), v);
    }
}

// Queue of one piece whose outputs are still to be written: walker_run_blocked drains it kGlobalBlock samples per trip,
// software-pipelined like the walk itself -- pump() stores the batch whose operands it fetched one trip earlier and
// issues the fetches of the next.  (The queued samples all lie before the walk's restart point and are never read
// again, so in-place sweeps stay correct.)
template <int OP>
struct LazyRun {
    int k = 0, to = -1;
    double v = 0.0;
    bool inflight = false;   // e[] holds (or is about to receive) the operands of samples k .. k + kGlobalBlock - 1
    Ext e[kGlobalBlock];
    __device__ __forceinline__ void store_batch(const SweepArgs &p, long base, long inc) {
#pragma unroll
        for (int u = 0; u < kGlobalBlock; u++)
            if (k + u <= to) Op<OP>::finish(p, base + (long)(k + u) * inc, e[u], v);
        k = min(k + kGlobalBlock, to + 1);
    }
    __device__ __forceinline__ void pump(const SweepArgs &p, long base, long inc) {
        if (inflight) store_batch(p, base, inc);
        inflight = (k <= to);
        if (inflight) {
#pragma unroll
            for (int u = 0; u < kGlobalBlock; u++)
                if (k + u <= to) e[u] = Op<OP>::fetch(p, base + (long)(k + u) * inc);
        }
    }
    __device__ __forceinline__ void flush(const SweepArgs &p, long base, long inc) {
        if (inflight) store_batch(p, base, inc);
        inflight = false;
        write_run<OP>(p, base, inc, k, to, v);
        k = to + 1;
    }
    __device__ __forceinline__ void queue(const SweepArgs &p, long base, long inc, int from, int to_, double v_) {
        flush(p, base, inc);
        k = from;
        to = to_;
        v = v_;
    }
};

template <int OP, bool WEIGHTED>
struct SeqSource {
    const SweepArgs &p;
    long base, inc, wbase;
    LazyRun<OP> run;
    __device__ __forceinline__ double y(int i) const { return Op<OP>::load_y(p, base + (long)i * inc); }
    __device__ __forceinline__ double r(int i) const { return p.w[wbase + (long)i * inc]; }
    __device__ __forceinline__ void piece(int from, int to, double v) { run.queue(p, base, inc, from, to, v); }
    __device__ __forceinline__ void bend(int, int) const {}
    __device__ __forceinline__ bool keep_going(int) const { return true; }
    __device__ __forceinline__ int limit() const { return 1 << 30; }
    __device__ __forceinline__ void pump() { run.pump(p, base, inc); }
    __device__ __forceinline__ void flush() { run.flush(p, base, inc); }
};

// PIPELINED picks the walker: walker_run_blocked when the pieces are known to be long (the policy's sequential mode),
// the plain per-sample loop otherwise (short fibres, unknown data: with a bend every few samples the pipelined
// walker's mispredictions cost more than its batching saves).
template <int OP, bool WEIGHTED, bool PIPELINED>
__device__ __forceinline__ void solve_fibre_seq(const SweepArgs &p, const FibreGeom &g, long j) {
    long blk, off;
    divmod_nonneg(j, g.inc, blk, off);
    SeqSource<OP, WEIGHTED> src{p, blk * g.inc * g.len + off, g.inc, blk * g.inc * (g.len - 1) + off, {}};
    if (WEIGHTED && g.len == 1) {  // no edge at all: prox is the identity (the reference reads lambda[0] out of bounds here)
        const double y0 = src.y(0);
        Op<OP>::finish(p, src.base, Op<OP>::fetch(p, src.base), y0);
        return;
    }
    Walker w;
    walker_start<WEIGHTED>(w, src, 0, p.lam);
    if (PIPELINED) {
        walker_run_blocked<WEIGHTED, kGlobalBlock>(w, src, g.len, p.lam);
    } else {
        walker_run<WEIGHTED>(w, src, g.len, p.lam);
        src.flush();
    }
}

// fibre_gate (may be null): only the fibres j with fibre_gate[j] != 0 are walked, and their flags are cleared -- the
// mop-up of a kernel that gave some fibres up (pin.hip's level cap).
template <int OP, bool WEIGHTED, bool PIPELINED>
__global__ __launch_bounds__(64) void sweep_seq_kernel(SweepArgs p, FibreGeom g, int *fibre_gate) {
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    if (j >= g.count || g.len <= 0) return;
    if (p.gate && *p.gate == 0) return;
    if (fibre_gate) {
        if (fibre_gate[j] == 0) return;
        fibre_gate[j] = 0;
    }
    solve_fibre_seq<OP, WEIGHTED, PIPELINED>(p, g, j);
}

}  // namespace swp
}  // namespace ptv
