// kernel_short_global.hpp -- kernels 1b / 2b: short fibres whole in LDS; chunks walked from global memory.
// (One of the pieces of sweep_kernels.hpp, which includes them in order; not meant to be included on its own.)
#pragma once

namespace ptv {
namespace swp {

// ---- kernel 1b: short fibres, whole in LDS ---------------------------------------------------------------------------------
// Fibres shorter than a few chunks (the 64-sample dimension of a 512 x 512 x 64 volume) have no room for speculation and
// do not need it: a wave takes 64 adjacent fibres WHOLE into LDS (all loads in flight together, where the sequential
// kernel pays a dependent global access per sample), every lane walks its own fibre from LDS with the assembly walk,
// 32 samples of outputs at a time (the walk records piece ends in 32-bit masks): the walk of the next 32 restarts at
// the last bend at or before its first sample -- the state after a bend is a function of the bend -- so nothing is
// carried but that bend.  Exact, no links, no repair; outputs may alias inputs (a wave reads all it needs before it
// writes).  Unweighted sweeps.
constexpr int kWholeC = 32;
constexpr int kWholeMax = 96;   // longest fibre this kernel takes (LDS: 512 B per sample per wave)

template <int OP, bool TRANSPOSED>
__global__ __launch_bounds__(64) void sweep_whole_kernel(SweepArgs p, FibreGeom g) {
    constexpr int PITCH = TRANSPOSED ? 65 : 64, C = kWholeC, NB = Op<OP>::NIN > 1 ? 16 : 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Yp = reinterpret_cast<double *>(smem);
    if (p.gate && *p.gate == 0) return;
    const int lane = threadIdx.x;
    const int len = g.len;
    const long j0 = (long)blockIdx.x * 64;
    const long j = j0 + lane;
    const bool active = j < g.count;
    long base = 0;
    if (active) {
        long blk, off;
        divmod_nonneg(j, g.inc, blk, off);
        base = blk * g.inc * len + off;
    }
    // ---- stage ---------------------------------------------------------------------------------------------------------------
    if (!TRANSPOSED) {
        for (int k0 = 0; k0 < len; k0 += NB) {
            double s0[NB], s1[NB];
#pragma unroll
            for (int v = 0; v < NB; v++) {
                s0[v] = s1[v] = 0.0;
                if (active && k0 + v < len) Op<OP>::fetch_in(p, base + (long)(k0 + v) * g.inc, s0[v], s1[v]);
            }
#pragma unroll
            for (int v = 0; v < NB; v++)
                if (active && k0 + v < len) Yp[(k0 + v) * PITCH + lane] = Op<OP>::y_of(p, s0[v], s1[v]);
        }
    } else {
        // 64 contiguous fibres = 64 * len contiguous samples: lanes run along memory, the tile is transposed into LDS
        const long nfib = min((long)64, g.count - j0);
        const long total = nfib * len;
        for (long e0 = 0; e0 < total; e0 += 64 * NB) {
            double s0[NB], s1[NB];
#pragma unroll
            for (int v = 0; v < NB; v++) {
                const long e = e0 + 64 * v + lane;
                s0[v] = s1[v] = 0.0;
                if (e < total) Op<OP>::fetch_in(p, j0 * len + e, s0[v], s1[v]);
            }
#pragma unroll
            for (int v = 0; v < NB; v++) {
                const long e = e0 + 64 * v + lane;
                if (e < total) Yp[(int)(e % len) * PITCH + (int)(e / len)] = Op<OP>::y_of(p, s0[v], s1[v]);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- walk and rebuild, 32 samples of outputs at a time ---------------------------------------------------------------------
    if (active) {
        const LdsWin<false, PITCH> win{(lds_double *)Yp + lane, (lds_double *)Yp + lane, 0};
        const FarFibre<OP> far{p, base, g.inc, 0};
        link_t carry = 0;   // last bend at or before the first sample of the coming 32
        for (int cs = 0; cs < len; cs += C) {
            const int ce = min(cs + C, len);
            ChunkRec rec;
            Walker w;
            int start = 0;
            if (carry != 0) {
                start = (int)(carry >> 1);
                walker_restart_with<false>(w, start, (int)(carry & 1u), len, p.lam, win.y(start), 0.0, 0.0);
                rec.mine = rec.next = rec.last = carry;
            } else {
                walker_start<false>(w, win, 0, p.lam);
            }
            walk_chunk<OP, false, PITCH, false>(w, rec, win, far, len, cs, ce, len, p.lam);
            rebuild_owned<Op<OP>, false, C, 8>(win, rec, cs, ce, len, start, true, 0, ce == len, p.lam);
            carry = rec.next;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- stream out ------------------------------------------------------------------------------------------------------------------
    if (!TRANSPOSED) {
        for (int k0 = 0; k0 < len; k0 += NB) {
            Ext ex[NB];
#pragma unroll
            for (int v = 0; v < NB; v++)
                ex[v] = (active && k0 + v < len && !Op<OP>::FUSED) ? Op<OP>::fetch(p, base + (long)(k0 + v) * g.inc) : Ext{0, 0};
#pragma unroll
            for (int v = 0; v < NB; v++) {
                if (active && k0 + v < len) {
                    const double x = Yp[(k0 + v) * PITCH + lane];
                    if (Op<OP>::FUSED) Op<OP>::store_fused(p, base + (long)(k0 + v) * g.inc, x);
                    else               Op<OP>::finish(p, base + (long)(k0 + v) * g.inc, ex[v], x);
                }
            }
        }
    } else {
        const long nfib = min((long)64, g.count - j0);
        const long total = nfib * len;
        for (long e0 = 0; e0 < total; e0 += 64 * NB) {
            Ext ex[NB];
#pragma unroll
            for (int v = 0; v < NB; v++) {
                const long e = e0 + 64 * v + lane;
                ex[v] = (e < total && !Op<OP>::FUSED) ? Op<OP>::fetch(p, j0 * len + e) : Ext{0, 0};
            }
#pragma unroll
            for (int v = 0; v < NB; v++) {
                const long e = e0 + 64 * v + lane;
                if (e < total) {
                    const double x = Yp[(int)(e % len) * PITCH + (int)(e / len)];
                    if (Op<OP>::FUSED) Op<OP>::store_fused(p, j0 * len + e, x);
                    else               Op<OP>::finish(p, j0 * len + e, ex[v], x);
                }
            }
        }
    }
}

// ---- kernel 2b: speculative chunks straight from global memory (long pieces) ---------------------------------------------------
// Same scheme as kernel 2 -- one lane per (fibre, chunk), warm-up zone, link codes, repairs by kernel 3 -- for data
// whose pieces are tens to hundreds of samples long (lambda several times the noise).  There the zone a walk needs to
// meet the true one is hundreds of samples: no LDS window holds that for 64 fibres, so this variant walks global
// memory like kernel 1 and lets chunk-level parallelism (fibres x chunks lanes instead of fibres) hide the latency.
// Chunk and zone sizes are run-time values; the lane owns, and writes, exactly the outputs of its chunk.
template <int OP, bool WEIGHTED>
struct GlobalChunkSource {
    const SweepArgs &p;
    long base, inc, wbase;
    int cs, ce;                // samples owned by this lane: [cs, ce)
    int hi;                    // the walk gives up at this sample (pieces far longer than the zone); == len near the fibre end
    unsigned mine = 0, next = 0;
    bool done = false, failed = false;
    LazyRun<OP> run;
    __device__ __forceinline__ double y(int i) const { return Op<OP>::load_y(p, base + (long)i * inc); }
    __device__ __forceinline__ double r(int i) const { return p.w[wbase + (long)i * inc]; }
    __device__ __forceinline__ void piece(int from, int to, double v) {
        if (to >= cs) run.queue(p, base, inc, max(from, cs), min(to, ce - 1), v);
        if (to >= ce - 1) done = true;
    }
    __device__ __forceinline__ void pump() { run.pump(p, base, inc); }
    __device__ __forceinline__ void flush() { run.flush(p, base, inc); }
    __device__ __forceinline__ void bend(int at, int type) {
        const unsigned code = ((unsigned)at << 1) | (unsigned)type;
        mine = (at <= cs) ? code : mine;
        next = (at <= ce) ? code : next;
    }
    __device__ __forceinline__ bool keep_going(int i) {
        if (done) return false;
        if (i >= hi) {   // hi == len is never reached by a live walk
            failed = true;
            return false;
        }
        return true;
    }
    __device__ __forceinline__ int limit() const { return hi; }
};

template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void sweep_gchunk_kernel(SweepArgs p, FibreGeom g, int C, int H, link_t *code_mine,
                                                           link_t *code_next, int *failflags) {
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    const int c = blockIdx.y;
    const int len = g.len;
    const int cs = c * C;
    if (j >= g.count || cs >= len) return;
    if (p.gate && *p.gate == 0) return;
    const int ce = min(cs + C, len);
    long blk, off;
    divmod_nonneg(j, g.inc, blk, off);
    GlobalChunkSource<OP, WEIGHTED> src{p, blk * g.inc * len + off, g.inc, blk * g.inc * (len - 1) + off, cs, ce,
                                        min(len, ce + H), 0u, 0u, false, false, {}};
    Walker w;
    walker_start<WEIGHTED>(w, src, max(0, cs - H), p.lam);
    walker_run_blocked<WEIGHTED, kGlobalBlock>(w, src, len, p.lam);
    if (src.failed) {   // nothing this lane recorded may be trusted; the repair walk rewrites its chunk
        flag_chunk(failflags, j, c, (len + C - 1) / C, DirtyMark{nullptr, 0u, nullptr, nullptr});
        src.mine = kLinkBad;
        src.next = 0;
    }
    code_mine[(long)c * g.count + j] = src.mine;
    code_next[(long)c * g.count + j] = src.next;
}

}  // namespace swp
}  // namespace ptv
