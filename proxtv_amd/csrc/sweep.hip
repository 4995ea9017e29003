// sweep.hip -- batched exact 1-D TV-L1 prox over the fibres of an N-D array, gfx950.
//
// Mapping: one wavefront lane per fibre, 64 adjacent fibres per wavefront.  For every dimension but the first,
// adjacent fibres are adjacent in memory, so sample k of the 64 lanes is one contiguous 512-byte row: every
// global access of the wave is a fully coalesced line pair.  (The reference instead parallelises fibres over
// OpenMP threads with per-thread gather/scatter copies: src/TV2Dopt.cpp:459-523, src/TVNDopt.cpp:164-209.)
//
// Kernel 1 (this file, `sweep_seq_kernel`): the sequential lane-per-fibre walk.  Always exact, any fibre length,
// operands may alias outputs.  It is the fallback of the chunked kernel and the path for short fibres.
#include "sweep.hpp"
#include "walker.hpp"

namespace ptv {

namespace {

// ---- sequential walk straight from / to global memory ------------------------------------------------------------
template <int OP, bool WEIGHTED>
struct SeqSource {
    const SweepArgs &p;
    long base, inc, wbase;
    __device__ __forceinline__ double y(int i) const { return Op<OP>::load_y(p, base + (long)i * inc); }
    __device__ __forceinline__ double r(int i) const { return p.w[wbase + (long)i * inc]; }
    __device__ __forceinline__ void piece(int from, int to, double v) const {
        for (int j = from; j <= to; j++) {
            const long idx = base + (long)j * inc;
            Op<OP>::store(p, idx, Op<OP>::load_y(p, idx), v);
        }
    }
    __device__ __forceinline__ void bend(int, int) const {}
    __device__ __forceinline__ bool keep_going(int) const { return true; }
};

template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void sweep_seq_kernel(SweepArgs p, FibreGeom g) {
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    if (j >= g.count || g.len <= 0) return;
    const long blk = j / g.inc, off = j % g.inc;
    SeqSource<OP, WEIGHTED> src{p, blk * g.inc * g.len + off, g.inc, blk * g.inc * (g.len - 1) + off};
    if (WEIGHTED && g.len == 1) {  // no edge at all: prox is the identity (the reference reads lambda[0] out of bounds here)
        const double y0 = src.y(0);
        Op<OP>::store(p, src.base, y0, y0);
        return;
    }
    Walker w;
    walker_start<WEIGHTED>(w, src, 0, p.lam);
    walker_run<WEIGHTED>(w, src, g.len, p.lam);
}

template <int OP, bool WEIGHTED>
void launch_seq(const SweepArgs &args, const FibreGeom &g, hipStream_t stream) {
    const unsigned blocks = (unsigned)((g.count + 63) / 64);
    if (blocks == 0) return;
    hipLaunchKernelGGL((sweep_seq_kernel<OP, WEIGHTED>), dim3(blocks), dim3(64), 0, stream, args, g);
    PTV_HIP(hipGetLastError());
}

template <int OP>
void launch_op(bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool allow_chunked) {
    (void)allow_chunked;
    if (weighted) launch_seq<OP, true>(args, g, stream);
    else          launch_seq<OP, false>(args, g, stream);
}

}  // namespace

void launch_sweep(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam,
                  bool allow_chunked) {
    FamilyTimer timer(fam, stream);
    switch (op) {
        case OP_PROX:          launch_op<OP_PROX>(weighted, args, g, stream, allow_chunked); break;
        case OP_DR_COL:        launch_op<OP_DR_COL>(weighted, args, g, stream, allow_chunked); break;
        case OP_DR_COL_FINAL:  launch_op<OP_DR_COL_FINAL>(weighted, args, g, stream, allow_chunked); break;
        case OP_DR_ROW:        launch_op<OP_DR_ROW>(false, args, g, stream, allow_chunked); break;
        case OP_DR_ROW_FINAL:  launch_op<OP_DR_ROW_FINAL>(false, args, g, stream, allow_chunked); break;
        case OP_DRW_ROW:       launch_op<OP_DRW_ROW>(true, args, g, stream, allow_chunked); break;
        case OP_DRW_ROW_FINAL: launch_op<OP_DRW_ROW_FINAL>(true, args, g, stream, allow_chunked); break;
        case OP_PD2_A:         launch_op<OP_PD2_A>(false, args, g, stream, allow_chunked); break;
        case OP_PD2_B:         launch_op<OP_PD2_B>(false, args, g, stream, allow_chunked); break;
        case OP_YANG:          launch_op<OP_YANG>(false, args, g, stream, allow_chunked); break;
        default:
            set_error("launch_sweep: unknown op %d", (int)op);
            throw HipFailure{hipErrorInvalidValue};
    }
}

}  // namespace ptv
