// sweep_window.hpp -- what the chunk kernels share: the launch plan, the LDS window as a lane sees it, one chunk's walk.
// (One of the pieces of sweep_kernels.hpp, which includes them in order; not meant to be included on its own.)
#pragma once

namespace ptv {
namespace swp {

// ---- kernel 2: speculative chunks over an LDS window -----------------------------------------------------------------
struct ChunkPlan {
    int Q;      // blocks (NW chunks each) per fibre
    int qpw;    // consecutive blocks of one fibre group processed (software-pipelined) by one workgroup
    int ablate; // profiling aid (option "ablate"): 1 = skip the walk, 2 = skip the epilogue, 4 = skip the window loads
    int rounds; // second-chance rounds inside a block (0 = none): see the link-proof step of sweep_chunk_kernel
    unsigned long long *trace;   // option "trace": 8 words per workgroup -- where it ran and when its phases ended (100 MHz clock)
    DirtyMark dirty;             // this launch's "something is left for the repair kernel" word
    unsigned long long *xlink;   // links across workgroups / segments: [boundary][fibre] (tile) or [fibre][segment] (along)
    int legacy;                  // option debug_legacy_rebuild (along-fibre kernel): rebuild_owned with the first-piece semantics of rounds 1-4
};

__device__ __forceinline__ void trace_mark(const ChunkPlan &plan, int slot) {
    if (plan.trace && threadIdx.x == 0)
        plan.trace[8 * (size_t)(blockIdx.x + gridDim.x * blockIdx.y) + slot] = wall_clock64();
}

// The LDS window as chunkcore.hpp sees it from one lane: row i of the lane's fibre at Y[(i - lo) * PITCH] (`lo` may be
// negative at the fibre start: rows below 0 are never touched).  32-bit LDS addressing throughout.
typedef __attribute__((address_space(3))) double lds_double;
template <bool WEIGHTED, int PITCH>
struct LdsWin {
    lds_double *Y;     // already offset to this lane's column
    lds_double *Wt;    // per-edge penalties, same addressing (weighted sweeps)
    int lo;
    __device__ __forceinline__ double y(int i) const { return Y[(i - lo) * PITCH]; }
    __device__ __forceinline__ double r(int i) const { return Wt[(i - lo) * PITCH]; }
    __device__ __forceinline__ void put(int i, double v) const { Y[(i - lo) * PITCH] = v; }
};

// samples beyond the window, one dependent global access each (robust instantiation only)
template <int OP>
struct FarFibre {
    const SweepArgs &p;
    long base, inc, wbase;
    __device__ __forceinline__ double far_y(int i) const { return Op<OP>::load_y(p, base + (long)i * inc); }
    __device__ __forceinline__ double far_r(int i) const { return p.w[wbase + (long)i * inc]; }
};

constexpr int kWarm = 16;       // H: samples a speculative walk starts before its chunk (its synchronisation zone)
constexpr int kWarmLong = 64;   // ... for data whose walks need longer to meet (moderate lambda: pieces of ~10 samples)
constexpr int kTail = 8;    // T: look-ahead rows kept in LDS past the last chunk of a block (short-zone geometry)
// The last chunk of a block must see the end of the piece that covers its last sample: the look-ahead has to scale
// with the piece length the geometry is meant for, like the warm-up zone does.
constexpr int tail_rows(int H) { return H > kWarm ? H : kTail; }
constexpr int kOverflow = 48;   // samples a walk of the robust instantiation may read past its window (global memory)

// One chunk's walk from a given walker state: the branch-free interior loop (chunkcore.hpp), then walker_run for what
// is left -- the fibre's last sample, the window's end (PAST: up to kOverflow samples beyond it from global memory).
template <int OP, bool WEIGHTED, int PITCH, bool PAST, bool TAB = false>
__device__ __forceinline__ void walk_chunk(Walker &w, ChunkRec &rec, const LdsWin<WEIGHTED, PITCH> &win, const FarFibre<OP> &far,
                                           int hi, int cs, int ce, int len, double lam, unsigned rtab = 0u) {
#ifndef PTV_NO_ASM_WALK
    if constexpr (!WEIGHTED && TAB) walk_interior_asm_tab<PITCH, PAST>(w, rec, win, min(len - 1, hi), cs, ce, lam, rtab);   // (spans bounded: see walk_asm.hpp)
    else if constexpr (WEIGHTED && TAB) walk_interior_asm_w_tab<PITCH, PAST>(w, rec, win, min(len - 1, hi), cs, ce, rtab);
    else if constexpr (!WEIGHTED) walk_interior_asm<PITCH>(w, rec, win, min(len - 1, hi), cs, ce, lam);
    else                     walk_interior_asm_w<PITCH>(w, rec, win, min(len - 1, hi), cs, ce);
#else
    walk_interior<WEIGHTED>(w, rec, win, min(len - 1, hi), cs, ce, lam);
#endif
    // (every walking lane of the wave closed the piece that covers its chunk's last sample inside the window -- the common case:
    //  walker_run would turn each of them away at its first test, after some eighty instructions of entry and exit)
    if (__builtin_amdgcn_ballot_w64(!rec.done) == 0ull) return;
    TailSource<WEIGHTED, PAST, kOverflow, LdsWin<WEIGHTED, PITCH>, FarFibre<OP>> tail{win, far, rec, cs, ce, hi, len};
    walker_run<WEIGHTED>(w, tail, len, lam);
    if (rec.failed) rec.next = 0;   // ran off the window: nothing this lane recorded may be trusted
}

// fetch_in without an op's streaming hint where the op has one (ops.hpp: InBminusA::fetch_in_shared)
template <int OP, class = void>
struct HasSharedFetch : std::false_type {};
template <int OP>
struct HasSharedFetch<OP, std::void_t<decltype(&Op<OP>::fetch_in_shared)>> : std::true_type {};
template <int OP>
__device__ __forceinline__ void fetch_in_shared_or_plain(const SweepArgs &p, long idx, double &i0, double &i1) {
    if constexpr (HasSharedFetch<OP>::value) Op<OP>::fetch_in_shared(p, idx, i0, i1);
    else Op<OP>::fetch_in(p, idx, i0, i1);
}

}  // namespace swp
}  // namespace ptv
