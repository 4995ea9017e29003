// sweep_kernels.hpp -- the fibre-sweep kernels and their launch plumbing (templates on the op and on weighted / unweighted).
//
// Included by sweep.hip (policy state, dispatch, the entry points of sweep.hpp) and by sweep_unit.hip, which is compiled once per
// (op, weighted) pair: every pair instantiates its own set of kernels, so the pairs build in parallel (one translation unit took
// 4.5 minutes; seventeen take about one on eight cores) and a variant build touches only what it changes.
#pragma once

#include "sweep.hpp"

#include <cstddef>
#include <memory>

#include "chunkcore.hpp"
#include "walk_asm.hpp"

#ifndef PTV_TILE_UNROLL
#define PTV_TILE_UNROLL 4   // rows of the rebuild passes in flight together in the tile kernels (round 4, with the rebuild in branch form: DR row
                            // sweep 114.9 -> 112.8 us, plain row sweep 83.7 -> 81.7 against 1; 2 in between)
#endif
#include "pin.hpp"
#include "pointwise.hpp"
#include "policy.hpp"
#include "transposed.hpp"
#include "walker.hpp"

#include <cstdio>

namespace ptv {

namespace swp {   // (named, not anonymous: the per-op translation units of sweep_unit.hip share ChunkScratch and the declarations below)

using link_t = unsigned;                 // (restart << 1 | bend type) of a walk's last bend before a chunk boundary
constexpr link_t kLinkBad = 0xfffffffeu;        // the chunk's walk ran off its LDS window: trust nothing it recorded
constexpr link_t kLinkCertain = 0x80000000u;    // flag on a published `mine` code: the chunk's walk began AT a bend known a priori

// Per fibre, two words tell the repair kernel where the chunk kernels left work: the first and the last chunk with an
// unproven link (both as maxima, so that 0 = none: NC - first and last + 1).  Only failing lanes touch them.
// A sweep that leaves anything to the repair kernel says so in ONE word, *dirty = the launch's epoch: the repair kernel's
// common case -- nothing to do -- is then a single load.  (Every launch has its own epoch, so nothing is ever reset.)
struct DirtyMark {
    unsigned *word;   // null: the repair kernel always does its full check (global-memory chunks)
    unsigned epoch;
    unsigned *why;    // option "why" (tuning aid): counters of what marked sweeps dirty -- [0] a walk ran off its window,
                      // [1] a link inside a workgroup / wave stayed unproven, [2] a link across workgroups / segments did not match,
                      // [3] ... was not published in time, [4] second chances taken across workgroups
    unsigned *sticky; // an optimistic solve (solvers.hip: dr2): no repair kernel is launched behind the sweeps; a sweep that leaves anything
                      // says so here, once and for all, and the solve is run again with the repairs (null otherwise)
    __device__ __forceinline__ void set(int reason) const {
        if (word) __hip_atomic_store(word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sticky) __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (why) atomicAdd(why + reason, 1u);
    }
    __device__ __forceinline__ void note(int reason) const {
        if (why) atomicAdd(why + reason, 1u);
    }
};
__device__ __forceinline__ void flag_chunk(int *failflags, long j, int chunk, int NC, const DirtyMark &dirty, bool ran_off = false) {
    atomicMax(failflags + 2 * j, NC - chunk);
    atomicMax(failflags + 2 * j + 1, chunk + 1);
    dirty.set(ran_off ? 0 : 1);
}

// The link between two workgroups (tile kernel) or two segments (along-fibre kernel) is checked by the LATER one at its
// very end, against what the earlier one published for it: (epoch << 32 | its last chunk's `next` code), one 8-byte word
// per fibre and boundary.  The earlier workgroup was dispatched first and publishes half-way through its life, so the
// word is almost always there; when it is not (or the codes differ) the sweep is marked dirty and the repair kernel runs
// its own check of every boundary, from the codes both sides publish in full, as before.
// (A second chance ACROSS workgroups of the tile kernel -- the next workgroup's first chunk waiting for a provisional word
// published right after the walk, and walking again from it -- was built and measured: 33 walks taken per 4096^2 solve at
// lambda = 0.5, fibres left to the repair kernel 104 -> 71, and every row sweep 35 us slower for the wait.  Not kept.)
constexpr link_t kLinkFinal = 0x80000000u;   // marks a published word (restart indices are below 2^30: the bit is free in a `next` code)
__device__ __forceinline__ void xlink_publish(unsigned long long *slot, unsigned epoch, link_t next, bool final_word = true) {
    __hip_atomic_store(slot, ((unsigned long long)epoch << 32) | next | (final_word ? kLinkFinal : 0u), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// 0: the link holds ; 2: the codes differ ; 3: no final word of this launch yet
// The earlier workgroup / wave has a lower linear index: it was dispatched first, is resident or done, and depends on nobody -- so a word
// that is not there yet is on its way, and the later one waits for it, a bounded while (kXlinkPatience sleeps of ~0.4 us), before it gives
// the sweep to the repair kernel.  Round 6: without the wait a weighted 4096^2 DR solve marked EVERY row sweep dirty (its tiles run eight
// blocks per workgroup: both sides of a boundary finish together, 48 000 late words per solve) and lambda = 0.2 one sweep in fifty --
// each a full scan by the repair kernel, and a whole solve again where the repairs are deferred (profiles/r06_dirty_rate.txt).
constexpr int kXlinkPatience = 128;
__device__ __forceinline__ int xlink_check(const unsigned long long *slot, unsigned epoch, link_t mine) {
    unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int patience = kXlinkPatience; patience > 0 && ((unsigned)(v >> 32) != epoch || !((link_t)v & kLinkFinal)); patience--) {
        __builtin_amdgcn_s_sleep(16);
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((unsigned)(v >> 32) != epoch || !((link_t)v & kLinkFinal)) return 3;
    return (mine != 0 && (link_t)v == (mine | kLinkFinal)) ? 0 : 2;
}

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

// One workgroup = NW waves = NW consecutive chunks (a "block" of NW*C samples) of the same 64 fibres; it processes
// plan.qpw consecutive blocks of those fibres.  Per block:
//   1. stage the window [block start - H, block end + T) into LDS through the op's input functor: all loads of a
//      thread are issued before the first is waited for; for dimension-0 sweeps the tile is transposed on the way;
//   2. every wave walks its chunk speculatively (LDS only), recording piece ends, bend types and link codes;
//   3. links between consecutive chunks are proven through LDS (and, across workgroups, by sweep_repair_kernel);
//   4. piece values are rebuilt in place (rebuild_owned: a piece is rewritten by the lane in whose chunk it ends), then
//      the block's rows are streamed out: straight from LDS for fused ops, otherwise through the op's output functor
//      (an operand that was staged for the walk and is needed again stays in registers: Op::KEEP).
// LDS carve (dynamic, 16-byte aligned base): Y window | Wt window (weighted) | link codes.
// SHORT: fibres no longer than one block (len <= NW * C: the 64-sample dimension of a 512 x 512 x 64 volume).  The window is
// the fibre itself -- no zone rows before it, no look-ahead rows after it are allocated (HA = TA = 0: chunks still start
// their walks H samples early, inside the block) -- so a workgroup of NW = 4 waves holds 32 KB of LDS and four or five of
// them share a CU; there are no links between workgroups, and the HBM traffic is exactly the algorithmic one.
// FW (fibres per tile, 64 or 32): with FW = 32 a wave carries TWO consecutive chunks of the same 32 fibres (lanes 0-31 the
// first, 32-63 the second), so a workgroup of NW = 4 waves covers the same 8-chunk block over half the fibres: half the LDS,
// FOUR independent workgroups per CU instead of two -- the stage / stream-out phases of one (memory latency) overlap the walks
// of three others -- and rows of 256 bytes towards HBM (two full 128-byte lines).  Strided plain tiles only.
template <int OP, bool WEIGHTED, bool TRANSPOSED, int C, int NW, int H, bool ROUNDS, int T = tail_rows(H), bool SHORT = false, int FW = 64>
__global__ __launch_bounds__(64 * NW, SHORT ? 16 / NW : (FW < 64 ? ((WEIGHTED ? 8 : 16) / NW) : ((WEIGHTED || H > 16 || NW > 8) ? NW / 4 : NW / 2))) void sweep_chunk_kernel(SweepArgs p, FibreGeom g, ChunkPlan plan,
                                                                                   link_t *code_mine, link_t *code_next,
                                                                                   int *failflags) {
    static_assert(FW == 64 || (FW == 32 && !TRANSPOSED && !SHORT && H <= C), "the 32-fibre tile is a strided short-zone tile");
    constexpr int CPW = 64 / FW;            // chunks per wave
    constexpr int NCH = NW * CPW;           // chunks per block
    constexpr int PITCH = TRANSPOSED ? 65 : FW;
    constexpr int HA = SHORT ? 0 : H, TA = SHORT ? 0 : T;   // zone / look-ahead rows the window really has
    constexpr int ROWS = HA + NCH * C + TA;
    static_assert(!(!TRANSPOSED && Op<OP>::KEEP) || (HA % NCH == 0 && TA % NCH == 0), "Op::KEEP relies on whole staging shares");
    constexpr int RB = (ROWS + 63) / 64;                                      // transposed: 64-row blocks per fibre
    constexpr int FPW = (64 + NW - 1) / NW;                                   // transposed: fibres per wave (the last wave's share may be short)
    constexpr int NST = TRANSPOSED ? FPW * RB : (ROWS + NCH - 1) / NCH;       // staged window elements per thread (the last may fall past the window)
    constexpr int UL = 8;                                                     // epilogue rows in flight per lane
    constexpr bool KEEP = !TRANSPOSED && Op<OP>::KEEP;                        // a staged operand is reused by the epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Yp = reinterpret_cast<double *>(smem);
    double *Wp = Yp + (WEIGHTED ? (size_t)ROWS * PITCH : 0);
    // (the walk's look-ahead read of row `hi` lands in whatever follows the Y window -- allocated LDS, value never used)
    link_t *codes = reinterpret_cast<link_t *>(Wp + (size_t)ROWS * PITCH);   // [NCH + 2][FW]; slots NCH, NCH + 1 carry over blocks
    // (bit 31 of a slot -- never part of a code: restart indices are below 2^30 -- says "this lane's link is proven")
    int *anybad = reinterpret_cast<int *>(codes + (NCH + 2) * FW);           // [2], by round parity: some lane of the block has an unproven link
    // A lane rewrites the rows of a piece that ends in its chunk even where they lie in earlier chunks; its walk reaches
    // back H rows (second-chance walks: anywhere in the block).  With H <= C that is the chunk before at most, and if
    // that chunk's lane is unproven its rows are rewritten by the repair kernel anyway.  Further back there may be
    // PROVEN chunks before an unproven one -- rows the repair kernel will not touch -- so those instantiations stop a
    // lane's writes at the nearest unproven chunk before it (GUARD: one flag per lane through LDS, one more barrier).
    constexpr bool GUARD = ROUNDS || H > C;
    unsigned long long *unproven = reinterpret_cast<unsigned long long *>(anybad + 2);   // [NW] lane masks (GUARD)
    // what the first chunk of the workgroup's first block began with, kept for the check at the kernel's end: an LDS row ([64];
    // the pitch-65 tile has no room left for one at two workgroups per CU and keeps it in a register)
    link_t *stash = reinterpret_cast<link_t *>(unproven + NW);
    constexpr link_t kNoCheck = 0xffffffffu;
    link_t began_reg = kNoCheck;
    // the walk's reciprocal table (walk_asm.hpp: walk_interior_asm_tab), after the stash row: the strided short-zone tiles only
    // (the pitch-65 tile has no LDS left for it at two workgroups per CU)
#ifndef PTV_NO_WALK_TABLE   // (the switch stays for A/B builds: the walk then divides with v_rcp_f64 + Newton + residual)
    constexpr bool TAB = (WEIGHTED || !TRANSPOSED) && !SHORT && H <= kWarm && NW <= 8 && (ROUNDS || H + C + T < kRecipTable);
#else
    constexpr bool TAB = false;
#endif
    constexpr int TS = ROUNDS ? kRecipTableRobust : kRecipTable;
    double *rtab = reinterpret_cast<double *>(stash + (TRANSPOSED ? 0 : FW));   // (the pitch-65 tile has no stash row: launch_chunk_h's LDS size)
    if constexpr (TAB) {
        if (threadIdx.x < TS) rtab[threadIdx.x] = threadIdx.x ? 1.0 / (double)threadIdx.x : 0.0;   // (visible after the staging barrier)
    }

    if (p.gate && *p.gate == 0) return;   // uniform over the grid
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform over the wave: scalar)
    // fl: this lane's fibre within the tile ; ch: its chunk within the block (FW = 64: the lane and the wave)
    const int fl = FW == 64 ? lane : (lane & (FW - 1)), ch = FW == 64 ? wave : wave * CPW + lane / FW;
    if (plan.trace && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        plan.trace[8 * (size_t)(blockIdx.x + gridDim.x * blockIdx.y)] = ((unsigned long long)xcc << 32) | hwid;
    }
    trace_mark(plan, 1);
    const int len = g.len;
    const long j0 = (long)blockIdx.x * FW;
    const long j = j0 + fl;
    const bool active = j < g.count;
    long base = 0, wbase = 0;
    if (active) {
        long blk, off;
        divmod_nonneg(j, g.inc, blk, off);
        base = blk * g.inc * len + off;
        wbase = blk * g.inc * (len - 1) + off;
    }
    const FarFibre<OP> far{p, base, g.inc, wbase};

    // Window rows are addressed relative to lo = block start - H (negative for the first block: those rows do not
    // exist and are never touched), so that the share of a thread is the same set of slots in every block.
    // Strided sweeps: element u of a thread is row lo + wave + NW*u of its own fibre (each wave instruction = one
    // coalesced 512-byte row); elements H/NW .. H/NW + C - 1 are rows of the block itself -- the ones the same thread
    // streams out at the end, so an operand staged here can wait in registers for the epilogue (Op::KEEP).
    // Dimension-0 sweeps (fibres contiguous): lanes run ALONG the fibre, element u is row lo + 64*(u % RB) + lane of
    // fibre wave + NW*(u / RB), and the tile is transposed on its way into LDS (pitch 65).
    // The loads of a batch of NB elements are all issued before the first is waited for.  NB = NST (the whole window
    // share of the thread) unless that would not fit the register budget: transposed sweeps stage in two batches (three
    // for two-operand inputs: 48 live doubles spill otherwise), two-operand strided sweeps in two.
    constexpr int NB = TRANSPOSED ? (Op<OP>::NIN > 1 ? (NST + 2) / 3 : (NST + 1) / 2) : (Op<OP>::NIN > 1 ? (NST + 1) / 2 : NST);
    // (PTV_KEEP_N: how many of a thread's C own rows keep the operand -- the rest is fetched again.  Round 4: all 16 spilled 28 registers
    // at the 128-VGPR budget (DR row sweep 116.7 -> 115.1 / 114.5 / 144 us at 4 / 8 / 16 rows), so 8 were kept.  Round 5: with the staging
    // addresses in scalar registers all 16 fit but for eight spilled dwords outside the walk: 108.2 -> 105.1 / 104.0 us at 12 / 16, the
    // second read of s' is gone and with it a tenth of the row sweep's traffic (profiles/r05_s4_ab_keep.txt).  The 64-fibre x 8-wave
    // tile (option tile = 0) stays at 8.)
#ifndef PTV_KEEP_N
#define PTV_KEEP_N 16
#endif
    constexpr int KNW = (FW < 64 || WEIGHTED) ? PTV_KEEP_N : (PTV_KEEP_N < 8 ? PTV_KEEP_N : 8);
    constexpr int KN = KEEP ? (KNW < C ? KNW : C) : 0;
    double kept[KEEP ? KN : 1];
    // `inner` blocks (strided tiles): all FW fibres of the tile exist, the whole window lies inside the fibre (but, first block, the
    // zone before sample 0) and the fibre's last sample beyond it -- nearly every block of a large image.  Uniform over the workgroup,
    // so nothing is tested per element there: the window loads run down the fibre from one address (the zone rows before sample 0,
    // which nothing ever reads, take copies of sample 0: a clamped row instead of a mask), the rebuild takes its FULL form, the
    // stream-out its rows as they come.
    const bool tile_whole = !TRANSPOSED && !SHORT && (long)blockIdx.x * FW + FW <= g.count;
    auto inner_block = [&](int q) { return tile_whole && q * NCH * C + NCH * C + TA <= len - 1; };
    auto stage_as = [&](int q, auto inner_tag) {
        constexpr bool inner = decltype(inner_tag)::value;
        const int cs_wg = q * NCH * C;
        const int lo = cs_wg - HA, hi = min(len, cs_wg + NCH * C + TA);
#pragma unroll
        for (int u0 = 0; u0 < NST; u0 += NB) {
            double s0[NB], s1[NB], sw[NB];
#pragma unroll
            for (int v = 0; v < NB; v++) {
                const int u = u0 + v;
                int r;
                long idx, widx;
                bool ok;
                if (!TRANSPOSED) {
                    r = lo + ch + NCH * u;
                    ok = active && r >= 0 && r < hi && r - lo < ROWS;
                    if constexpr (inner) {
                        ok = NCH * u + NCH <= ROWS || r - lo < ROWS;          // (compile time for all but a ragged last share)
                        if (NCH * u < HA) r = max(r, 0);                        // (compile time: the shares that hold zone rows)
                    }
                    idx = base + (long)r * g.inc;
                    widx = wbase + (long)r * g.inc;
                } else {
                    const long jf = j0 + wave + NW * (u / RB);
                    r = lo + (u % RB) * 64 + lane;
                    ok = wave + NW * (u / RB) < 64 && jf < g.count && r >= 0 && r < hi;
                    idx = jf * len + r;
                    widx = jf * (len - 1) + r;
                }
                ok = ok && u < NST;
                s0[v] = s1[v] = 0.0;
                // (strided tiles: the shares that hold rows two workgroups stage -- the zone and look-ahead rows, and the own rows that are
                //  a neighbour's: the block's first TA and last HA -- are loaded without the streaming hint an op may put on a window operand)
                constexpr int kOwn0 = HA / NCH, kOwn1 = HA / NCH + C;
#ifndef PTV_NO_SHARED_HALO   // (A/B switch)
                // (measured, profiles/r06_s13_ab_halo.txt: DR row sweep 105.0 -> 103.9 us; the weighted tile, one or two workgroups per CU, lost
                //  1 % and keeps the hint everywhere)
                const bool shared_rows = !TRANSPOSED && !SHORT && !WEIGHTED && (u < kOwn0 + (TA + NCH - 1) / NCH || u >= kOwn1 - (HA + NCH - 1) / NCH);
#else
                const bool shared_rows = false;
#endif
                if (ok) {
                    if (shared_rows) fetch_in_shared_or_plain<OP>(p, idx, s0[v], s1[v]);
                    else             Op<OP>::fetch_in(p, idx, s0[v], s1[v]);
                }
                if (WEIGHTED) sw[v] = (ok && (inner || r < len - 1)) ? p.w[widx] : 0.0;
            }
#pragma unroll
            for (int v = 0; v < NB; v++) {
                const int u = u0 + v;
                int r, col;
                bool ok;
                if (!TRANSPOSED) {
                    r = lo + ch + NCH * u;
                    col = fl;
                    ok = active && r >= 0 && r < hi && r - lo < ROWS;
                    if constexpr (inner) ok = NCH * u + NCH <= ROWS || r - lo < ROWS;
                } else {
                    col = wave + NW * (u / RB);
                    r = lo + (u % RB) * 64 + lane;
                    ok = col < 64 && j0 + col < g.count && r >= 0 && r < hi;
                }
                if (ok && u < NST) {
                    Yp[(r - lo) * PITCH + col] = Op<OP>::y_of(p, s0[v], s1[v]);
                    if (WEIGHTED) Wp[(r - lo) * PITCH + col] = sw[v];
                }
                if (KEEP && u >= HA / NCH && u < HA / NCH + KN) kept[(KEEP && u >= HA / NCH && u < HA / NCH + KN) ? u - HA / NCH : 0] = s1[v];
            }
        }
    };
    auto stage = [&](int q) {
        if (inner_block(q)) stage_as(q, std::true_type{});
        else                stage_as(q, std::false_type{});
    };

    const int q_first = blockIdx.y * plan.qpw;
    const int nblk = min(plan.qpw, plan.Q - q_first);

    for (int kb = 0; kb < nblk; kb++) {
        const int q = q_first + kb;
        if (plan.ablate & 4) {
            if (kb == 0)
                for (int e = tid; e < ROWS * PITCH; e += 64 * NW) Yp[e] = (double)((e * 2654435761u) >> 20) * 1e-3;
        } else {
            stage(q);
        }
        __syncthreads();
        if (kb == 0) trace_mark(plan, 2);

        const int cs_wg = q * NCH * C;
        const int lo = cs_wg - HA;
        const int hi = min(len, cs_wg + NCH * C + TA);

        // ---- speculative walk of this wave's chunk(s) -----------------------------------------------------------------
        const int cs = cs_wg + ch * C;
        const int ce = min(cs + C, len);
        const bool has_chunk = active && cs < len;
        const int start = max(0, cs - H);
        const LdsWin<WEIGHTED, PITCH> win{(lds_double *)Yp + fl, (lds_double *)Wp + fl, lo};
        ChunkRec rec;
        PiecePrefix head;
        bool certain = false;
        if (has_chunk && !(plan.ablate & 1)) {
            Walker w;
            // A lane that finds a bend known a priori (chunkcore.hpp) among the kLook edges before its chunk starts its
            // walk AT it -- exact by construction, no warm-up zone to walk, no link to prove.  On noisy data with small
            // lambda (the headline: 78 % of all edges qualify) every lane of a wave does; otherwise the lane falls back to
            // the speculative start.
            // (robust instantiation: the whole zone is searched -- a lane that starts at a bend known a priori has no link that could
            // fail, and failed links across workgroups are what the repair kernel is left with at the upper end of rung 1)
#ifndef PTV_TILE_ROBUST_LOOK
#define PTV_TILE_ROBUST_LOOK 14   // (against 8: 4096^2 DR at lambda = 0.4 / 0.5: 10.01 -> 9.89, 11.47 -> 11.33 ms; nothing from 0.6 on)
#endif
            constexpr int kLook = ROUNDS ? PTV_TILE_ROBUST_LOOK : 8;
            static_assert(H >= kLook + 2, "the certain-bend search reads rows of the warm-up zone");
            int cat = -1, ctype = 0;
            if (start > 0 && H <= kWarm && p.lam > 0.0) cat = certain_bend_before<WEIGHTED, kLook>(win, cs, len, p.lam, ctype);
            if (cat >= 0) {
                certain = true;
                walker_restart_with<WEIGHTED>(w, cat, ctype, len, p.lam, win.y(cat), WEIGHTED ? win.r(cat - 1) : 0.0,
                                              (WEIGHTED && cat < len - 1) ? win.r(cat) : 0.0);
                rec.mine = rec.next = rec.last = ((link_t)cat << 1) | (link_t)ctype;
            } else {
                walker_start<WEIGHTED>(w, win, start, p.lam);
            }
            walk_chunk<OP, WEIGHTED, PITCH, ROUNDS, TAB>(w, rec, win, far, hi, cs, ce, len, p.lam, (unsigned)(unsigned long long)rtab);
            // (the rows before the chunk that belong to its first piece, summed while the window holds samples only: an unproven lane's
            //  are not its own to rely on once the rebuild has begun in other waves -- chunkcore.hpp first_piece_prefix)
            if (!GUARD) head = first_piece_prefix(win, rec, cs, start);
        }
        // ---- prove the links between consecutive chunks ------------------------------------------------------------------
        codes[ch * FW + fl] = rec.next;
        if (ROUNDS && tid == 0) anybad[0] = anybad[1] = 0;
        __syncthreads();   // all walks done: link codes visible, window rows no longer read as walk input
        if (kb == 0) trace_mark(plan, 3);
        const int prev_slot = (ch > 0) ? (ch - 1) * FW + fl : (NCH + ((kb + 1) & 1)) * FW + fl;
        bool bad = false;
        // Second chances inside the block (plan.rounds > 0; data whose walks need more than the zone to meet): a lane
        // whose link fails, while its predecessor's holds, walks its chunk again from the predecessor's last bend -- a
        // bend of the true walk if the predecessor is true.  Every round moves the proven frontier of a failing run one
        // chunk on; links are re-examined after every round (a predecessor that walked again may have changed its
        // code), and what is still unproven after the last round goes to the repair kernel as usual.
        for (int round = 0; ; round++) {
            const bool linked = has_chunk && !(start == 0 || certain) && (ch > 0 || kb > 0);   // hangs on its predecessor
            bad = has_chunk && (rec.failed || (linked && (rec.mine == 0 || rec.mine != (codes[prev_slot] & ~kLinkCertain))));
            if (!ROUNDS || round >= plan.rounds) break;
            if (has_chunk) codes[ch * FW + fl] = bad ? rec.next : (rec.next | kLinkCertain);   // same code, plus the flag
            if (bad) anybad[round & 1] = 1;
            __syncthreads();
            if (!anybad[round & 1]) break;               // uniform
            if (tid == 0) anybad[(round + 1) & 1] = 0;   // set again only after the barrier below
            if (bad && (ch > 0 || kb > 0)) {
                const link_t praw = codes[prev_slot];
                const link_t prev = praw & ~kLinkCertain;
                const int at = (int)(prev >> 1);
                if ((praw & kLinkCertain) && prev != 0 && at > max(lo, 0)) {
                    ChunkRec again;
                    Walker w;
                    walker_restart_with<WEIGHTED>(w, at, (int)(prev & 1u), len, p.lam, win.y(at), WEIGHTED ? win.r(at - 1) : 0.0,
                                                  (WEIGHTED && at < len - 1) ? win.r(at) : 0.0);
                    again.mine = again.next = again.last = prev;
                    walk_chunk<OP, WEIGHTED, PITCH, ROUNDS, TAB>(w, again, win, far, hi, cs, ce, len, p.lam, (unsigned)(unsigned long long)rtab);
                    if (!again.failed) {
                        rec = again;
                        certain = false;   // from now on the chunk hangs on its predecessor like any other
                        codes[ch * FW + fl] = rec.next;
                    }
                }
            }
            __syncthreads();
        }
        if (has_chunk) {
            if (rec.failed) {
                rec.mine = kLinkBad;
                rec.next = 0;
            }
            if (bad) flag_chunk(failflags, j, q * NCH + ch, (len + C - 1) / C, plan.dirty, rec.failed);
            // Every chunk publishes its two codes: sweep_repair_kernel proves the links between workgroups with them
            // and, for a fibre with an unproven link, finds where a repair walk may stop.
            const long slot = (long)(q * NCH + ch) * g.count + j;
            code_mine[slot] = (certain && rec.mine != kLinkBad) ? (rec.mine | kLinkCertain) : rec.mine;
            code_next[slot] = rec.next;
            // ... and the workgroup's last chunk, right now, what the next workgroup's first chunk must have begun with
            if (plan.xlink && kb == nblk - 1 && ch == NCH - 1)
                xlink_publish(plan.xlink + (size_t)blockIdx.y * g.count + j, plan.dirty.epoch, rec.next);
        }
        // (the link INTO this workgroup is checked at the very end, when the workgroup before has surely published)
        if (kb == 0 && ch == 0) {
            const link_t began = (has_chunk && !certain) ? rec.mine : kNoCheck;
            if (TRANSPOSED) began_reg = began;
            else stash[fl] = began;
        }
        // carried to the next block's first chunk; two slots in turn, so that no barrier is needed before the write
        if (ch == NCH - 1) codes[(NCH + (kb & 1)) * FW + fl] = (bad || !has_chunk) ? rec.next : (rec.next | kLinkCertain);
        // Non-fused ops: the operand fetches of the epilogue's first batch of rows go out now and fly while the rebuild
        // runs (the walk's registers are free by now); the second batch is fetched while the first is stored.
        const int ce_wg = min(len, cs_wg + NCH * C);
#ifdef PTV_PREFETCH_EPILOGUE   // measured: the 32 VGPRs it holds across the rebuild spill at two workgroups per CU, 14 % slower
        constexpr bool PREFETCH = !TRANSPOSED && !Op<OP>::FUSED && !KEEP;
#else
        constexpr bool PREFETCH = false;
#endif
        constexpr int NPRE = UL;   // (all C rows would not fit the register budget next to the rebuild)
        Ext pre[PREFETCH ? NPRE : 1];
        if (PREFETCH && active && !(plan.ablate & 2)) {
#pragma unroll
            for (int m = 0; m < NPRE; m++) {
                const int k = min(cs_wg + ch + NCH * m, ce_wg - 1);
                pre[PREFETCH ? m : 0] = Op<OP>::fetch(p, base + (long)k * g.inc);
            }
        }
        int wlo = cs_wg;   // first row this lane may write
        if (GUARD) {
            // (second chances may have replaced the record: the sums of first_piece_prefix now, before the barrier every rebuild waits behind)
            if (has_chunk && !(plan.ablate & 1)) head = first_piece_prefix(win, rec, cs, start);
            // one lane mask per wave: FW = 64 -> the wave's chunk ; FW = 32 -> its two chunks, the later one in the high half
            const unsigned long long mask = __ballot(bad);
            if (lane == 0) unproven[wave] = mask;
            __syncthreads();
            for (int k = ch - 1; k >= 0; k--)
                if ((unproven[k / CPW] >> ((k % CPW) * FW + fl)) & 1ull) {
                    wlo = cs_wg + k * C;
                    break;
                }
        }
        const bool inner = inner_block(q);   // (uniform over the workgroup)
        if (inner && !(plan.ablate & 1))
            rebuild_owned<Op<OP>, WEIGHTED, C, PTV_TILE_UNROLL, TAB, lds_double *, (ROUNDS ? TS : 0), 1>(win, rec, cs, ce, len, start, !bad, wlo,
                                                                                                       ch == NCH - 1, p.lam, (lds_double *)rtab, &head);
        else if (has_chunk && !(plan.ablate & 1))
            rebuild_owned<Op<OP>, WEIGHTED, C, PTV_TILE_UNROLL, TAB, lds_double *, (ROUNDS ? TS : 0)>(win, rec, cs, ce, len, start, !bad, wlo,
                                                                                                    ch == NCH - 1 || ce == len, p.lam, (lds_double *)rtab, &head);
        __syncthreads();
        if (kb == 0) trace_mark(plan, 4);

        // ---- stream the block's NW*C rows out: coalesced 512-byte rows, UL operand fetches in flight per lane ------------------
        if (!(plan.ablate & 2)) {
            if (!TRANSPOSED && inner) {
                // every row of the block exists for every fibre of the tile: nothing is tested
#pragma unroll
                for (int m0 = 0; m0 < C; m0 += UL) {
                    Ext ex[UL];
#pragma unroll
                    for (int u = 0; u < UL; u++) {
                        const long idx = base + (long)(cs_wg + ch + NCH * (m0 + u)) * g.inc;
                        if (KEEP && m0 + u < KN) ex[u] = Op<OP>::fetch_rest(p, idx, kept[(KEEP && m0 + u < KN) ? m0 + u : 0]);
                        else if (!Op<OP>::FUSED) ex[u] = Op<OP>::fetch(p, idx);
                    }
#pragma unroll
                    for (int u = 0; u < UL; u++) {
                        const int k = cs_wg + ch + NCH * (m0 + u);
                        const double v = Yp[(k - lo) * PITCH + fl];
                        if (Op<OP>::FUSED) Op<OP>::store_fused(p, base + (long)k * g.inc, v);
                        else               Op<OP>::finish(p, base + (long)k * g.inc, ex[u], v);
                    }
                }
            } else if (!TRANSPOSED) {
                if (active) {
                    // the thread that staged rows cs_wg + ch + NCH*m streams them out (Op::KEEP: with the staged operand)
#pragma unroll
                    for (int m0 = 0; m0 < C; m0 += UL) {
                        Ext ex[UL];
#pragma unroll
                        for (int u = 0; u < UL; u++) {
                            const int k = min(cs_wg + ch + NCH * (m0 + u), ce_wg - 1);
                            if (KEEP && m0 + u < KN) ex[u] = Op<OP>::fetch_rest(p, base + (long)k * g.inc, kept[(KEEP && m0 + u < KN) ? m0 + u : 0]);
                            else if (PREFETCH && m0 == 0) ex[u] = pre[PREFETCH ? u : 0];
                            else if (!Op<OP>::FUSED) ex[u] = Op<OP>::fetch(p, base + (long)k * g.inc);
                        }
#pragma unroll
                        for (int u = 0; u < UL; u++) {
                            const int k = cs_wg + ch + NCH * (m0 + u);
                            if (k < ce_wg) {
                                const double v = Yp[(k - lo) * PITCH + fl];
                                if (Op<OP>::FUSED) Op<OP>::store_fused(p, base + (long)k * g.inc, v);
                                else               Op<OP>::finish(p, base + (long)k * g.inc, ex[u], v);
                            }
                        }
                    }
                }
            } else {
                constexpr int ERB = (NW * C + 63) / 64;
                constexpr int items = FPW * ERB;
#pragma unroll
                for (int t0 = 0; t0 < items; t0 += UL) {
                    Ext ex[UL];
#pragma unroll
                    for (int u = 0; u < UL; u++) {
                        const int t = t0 + u;
                        const long jf = j0 + wave + NW * (t / ERB);
                        const int k = cs_wg + (t % ERB) * 64 + lane;
                        const bool ok = t < items && wave + NW * (t / ERB) < 64 && jf < g.count && k < ce_wg;
                        ex[u] = (ok && !Op<OP>::FUSED) ? Op<OP>::fetch(p, jf * len + k) : Ext{0, 0};
                    }
#pragma unroll
                    for (int u = 0; u < UL; u++) {
                        const int t = t0 + u;
                        const int f = wave + NW * (t / ERB);
                        const int k = cs_wg + (t % ERB) * 64 + lane;
                        if (t < items && f < 64 && j0 + f < g.count && k < ce_wg) {
                            const double v = Yp[(k - lo) * PITCH + f];
                            if (Op<OP>::FUSED) Op<OP>::store_fused(p, (j0 + f) * len + k, v);
                            else               Op<OP>::finish(p, (j0 + f) * len + k, ex[u], v);
                        }
                    }
                }
            }
        }
        if (kb + 1 < nblk) __syncthreads();   // every wave is done reading this block's window
    }
    if (plan.xlink && blockIdx.y > 0 && ch == 0 && active) {
        const link_t began = TRANSPOSED ? began_reg : stash[fl];   // (written by this very thread)
        if (began != kNoCheck) {
            const int why = xlink_check(plan.xlink + (size_t)(blockIdx.y - 1) * g.count + j, plan.dirty.epoch, began);
            if (why) plan.dirty.set(why);
        }
    }
    trace_mark(plan, 5);
}

// ---- kernel 2a: speculative chunks ALONG the fibre (dimension-0 sweeps, unweighted) -------------------------------------
// Fibres of dimension 0 are contiguous in memory, so 64 consecutive chunks of ONE fibre can share a wavefront: lane l
// owns chunk l of a 64-chunk segment.  The segment (+ zone and look-ahead rows) is copied into LDS as it lies in
// memory -- 512-byte coalesced loads, no transposition -- and everything after that stays inside the wave: the walks
// read the shared copy (a lane's zone IS its neighbour's chunk: no row is staged twice but the H + T rows at the two
// ends of a 1088-sample segment, 2 %), a link is proven with one lane shuffle, the rebuild follows the same ownership
// rule, the outputs leave as 512-byte rows.  No workgroup barrier anywhere: the four waves of a workgroup are only
// scheduled together, so the memory phases of one wave overlap the walks of the others on the same CU.
// Chunks are 17 samples long, not 16: lane l walks rows 17 l + t of the linear LDS copy, and 17 is odd, so the 64 lanes
// of a read hit 32 different bank pairs -- the floor for 8-byte accesses -- without any padding (with 16 they would
// hit two).
#ifndef PTV_ALONG_C
#define PTV_ALONG_C 17
#endif
// Weighted sweeps hold two LDS planes per wave (samples, penalties): with 17-sample chunks 17.8 KB, i.e. 8 waves per CU, and the kernel
// idles -- vector pipes 45 % busy, HBM at a third of its rate (profiles/r05_s1_kernel_counters.txt).  Chunks of 9 samples (odd: no bank
// conflicts) halve the segment and with it the LDS: 16 waves per CU.  The walk costs more per sample (a zone per 9 samples instead of
// per 17) and still wins: weighted column sweep 143.3 -> 111.2 us, weighted 4096^2 solve 11.64 -> 10.53 ms; 11 samples (12 waves):
// 117.8 us (profiles/r05_s3_ab_weighted.txt).  (Unweighted: 15 / 13 samples -- 20 waves per CU -- move the column sweep 75.0 ->
// 75.6 / 72.1 us and cost 512-sample fibres their one-segment instantiation: 17 stays.)
#ifndef PTV_ALONG_W_C
#define PTV_ALONG_W_C 9
#endif
constexpr int kAlongC = PTV_ALONG_C;
// Chunk length of the robust instantiation (rungs 1 / 2: pieces of a few samples).  Longer chunks walk the zone less often and --
// what matters more -- shrink the spread between the lanes of a wave, whose walk lasts as long as its slowest lane's: with 31
// samples a wave walks 2.70 -> 2.03 trips per sample at lambda = 0.5 and 4.24 -> 3.03 at 0.7 on the inputs of DR sweeps (host model:
// tools/study/links_study.py).  The price is LDS: 16.9 KB per wave instead of 9.7, eight waves per CU instead of sixteen -- and
// measured (end of round 3) that price is too high: 4096^2 DR at lambda = 0.5 / 0.7 12.7 -> 16.1, 23.7 -> 33.3 ms with 31, 13.7 / 25.0
// with 23.  So: 17, like the plain instantiation.  (An odd number, see above; a chunk's piece ends fit the 32-bit masks of ChunkRec.)
#ifndef PTV_ALONG_ROBUST_C
#define PTV_ALONG_ROBUST_C PTV_ALONG_C
#endif
constexpr int along_chunk(bool robust, bool weighted) { return weighted ? PTV_ALONG_W_C : (robust ? PTV_ALONG_ROBUST_C : kAlongC); }
#ifndef PTV_ALONG_WAVES
#define PTV_ALONG_WAVES 4
#endif
#ifndef PTV_ALONG_UNROLL
#define PTV_ALONG_UNROLL 4
#endif
constexpr int kAlongWaves = PTV_ALONG_WAVES;
// Look-ahead rows after a segment.  Only the segment's LAST lane reads them -- to close the piece that covers its last
// sample -- and a segment is 1088 samples, so they cost next to nothing here: the robust instantiation takes 64 (on DR
// iterates at lambda = 0.5 / 0.7 / 1 on unit noise a walk needs more than 8 rows past its chunk in 2 % / 20 % / 70 % of
// the cases, more than 32 in 0 / 0.01 % / 7 %), where every further sample would be a dependent global read.
constexpr int along_tail_rows(int H, bool robust) { return robust ? 64 : tail_rows(H); }
// Rows kept BEFORE a segment.  The walks start H samples early whatever this is; the robust instantiation keeps 64 so that
// a second chance can start from a bend that lies further back than the zone (the predecessor's last bend sits more than
// 16 samples before the boundary in ~1 % of the cases at lambda = 0.7: a third of the links it still left to the repair kernel).
constexpr int along_zone_rows(int H, bool robust) { return robust && H < 64 ? 64 : H; }

// G lanes share one segment of G chunks: 64 for long fibres; 32 or 16 pack two or four shorter fibres into a wave.
// ROBUST (geometry mode 1: pieces of a few samples, walks that need the whole zone -- or more -- to meet): like the tile
// kernel's robust instantiation, nothing a failed link needs leaves the kernel if it can be helped:
//   * a walk may run past the segment's look-ahead rows (global reads, a few samples, only the lanes that need it);
//   * second chances inside the wave: a lane whose link fails while its predecessor's holds walks its chunk again from the
//     predecessor's last bend (one lane shuffle tells it which); links are re-examined after every round, plan.rounds rounds;
//   * the first lane of a segment, whose predecessor is the last lane of ANOTHER wave, looks that wave's final code up
//     through LDS when both waves sit in the same workgroup (4096-sample fibres: 4 segments = the 4 waves of a workgroup)
//     and gets its second chance from it.
// What is still unproven goes to the repair kernel as before, which also re-checks every link between SEGMENTS (and tile workgroups)
// from the codes the waves finally publish.  Links INSIDE a segment / workgroup that no lane flagged are not re-checked there (the
// repair kernel jumps from flagged chunk to flagged chunk): for those the in-kernel proofs -- code equality, second chances, the
// hand-over across waves -- ARE what exactness rests on.
// ONESEG: fibres of at most one segment (G chunks) -- there is no row before the segment and none after it, so the robust
// instantiation's 64 + 64 rows of look-back / look-ahead are not allocated: a third of its LDS for 512-sample fibres, and with it
// twelve waves per CU become sixteen.
constexpr int kRunsWords = 196;   // RUNS: LDS words per wave (64 runs, 64 + 64 masks, the bend before the segment; padded)
template <int OP, bool WEIGHTED, int H, int G, bool ROBUST, bool ONESEG = false, bool RUNS = false>
__global__ __launch_bounds__(64 * kAlongWaves) void sweep_along_kernel(SweepArgs p, FibreGeom g, ChunkPlan plan, link_t *code_mine,
                                                                        link_t *code_next, int *failflags) {
    constexpr int C = along_chunk(ROBUST, WEIGHTED), SEG = G * C, T = ONESEG ? 0 : along_tail_rows(H, ROBUST), HZ = ONESEG ? 0 : along_zone_rows(H, ROBUST), ROWS = HZ + SEG + T, NG = 64 / G;
    constexpr int NU = (ROWS + G - 1) / G;   // staged elements per lane
    constexpr int UL = (C + 1) / 2;          // epilogue operand fetches in flight per lane (C = 17 rows per lane: 9 + 8)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.gate && *p.gate == 0) return;
    // (the wave number is the same in every lane: said so, everything derived from it -- fibre, segment, base addresses, window
    //  bounds -- lives in scalar registers and the address arithmetic of the memory phases runs on the scalar unit)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gi = lane / G, gl = lane % G;
    double *Yp = reinterpret_cast<double *>(smem) + (size_t)(wave * NG + gi) * (ROWS + 2) * (WEIGHTED ? 2 : 1);
    double *Wp = Yp + (WEIGHTED ? ROWS + 2 : 0);   // per-edge penalties, same rows (weighted sweeps)
    // ROBUST: what a wave's last lane ends up with, for the first lane of the next wave: [kAlongWaves] codes, [kAlongWaves] "ready"
    unsigned *xwave = reinterpret_cast<unsigned *>(reinterpret_cast<double *>(smem) + (size_t)kAlongWaves * NG * (ROWS + 2) * (WEIGHTED ? 2 : 1));
    // plain instantiation: the pull-backs of the walk divide by table (walk_asm.hpp: walk_interior_asm_tab); one table per workgroup
#ifndef PTV_NO_WALK_TABLE   // (the switch stays for A/B builds: the walk then divides with v_rcp_f64 + Newton + residual)
    constexpr bool TAB = H <= kWarm && (ROBUST || H + kAlongC + T < kRecipTable);
#else
    constexpr bool TAB = false;
#endif
    constexpr int TS = ROBUST ? kRecipTableRobust : kRecipTable;
    double *rtab = reinterpret_cast<double *>(xwave + (ROBUST ? 16 : 0));
    if constexpr (TAB) {
        if (threadIdx.x < TS) rtab[threadIdx.x] = threadIdx.x ? 1.0 / (double)threadIdx.x : 0.0;
        if (!ROBUST || G != 64) __syncthreads();   // (before anything else happens: every wave of the workgroup is here; the robust
                                                   //  64-lane instantiation has its own barrier right below)
    }
    if (ROBUST && G == 64) {
        if (lane == 0) xwave[kAlongWaves + wave] = 0u;
        __syncthreads();   // (the only workgroup barrier of the kernel: before anything else happens)
    }
    const int len = g.len;
    const int nseg = (len + SEG - 1) / SEG, NC = (len + C - 1) / C;
    const long wid = (long)blockIdx.x * kAlongWaves + wave;
    const long unit = wid * NG + gi;         // one group of G lanes = one segment of one fibre
    long j, sg_l;
    divmod_nonneg(unit, (long)nseg, j, sg_l);
    const int sg = (int)sg_l;
    const bool live = j < g.count;           // (nothing in this kernel synchronises across waves; a group past the end idles)
    if (plan.trace && lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        plan.trace[8 * (size_t)wid] = ((unsigned long long)xcc << 32) | hwid;
        plan.trace[8 * (size_t)wid + 1] = wall_clock64();
    }
    const long fbase = live ? j * len : 0, wbase = live ? j * (len - 1) : 0;
    const int seg_s = sg * SEG, seg_e = live ? min(len, seg_s + SEG) : seg_s;
    const int lo = seg_s - HZ, hi = min(len, seg_s + SEG + T);

    // ---- stage: the segment as it lies in memory ---------------------------------------------------------------------------
    // `interior`: every row of the window but (first segment) the zone before sample 0 exists -- three segments out of four of a
    // 4096-sample fibre.  Then nothing is tested per element: the loads are scalar base + lane + immediate, the LDS stores lane base +
    // immediate (the first segment's zone rows, which nothing ever reads, take copies of sample 0: a clamped address instead of a mask).
    // Uniform over the wave for whole-wave segments (G = 64: fibre and segment are scalar values there).
    const bool interior = G == 64 && !ONESEG && live && seg_s + SEG + T <= len - 1;
    const unsigned ul = (unsigned)gl;
    auto stage_interior = [&]() {
        constexpr int NB = NU <= 20 ? NU : (NU + 1) / 2;
        const long row0 = fbase + lo, wrow0 = wbase + lo;   // (scalar; the first segment: lo = -HZ, element 0 is clamped below)
#pragma unroll
        for (int b0 = 0; b0 < NU; b0 += NB) {
            double s0[NB], s1[NB], sw[WEIGHTED ? NB : 1];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int rel = G * (b0 + u);
                s0[u] = s1[u] = 0.0;
                if (b0 + u >= NU) continue;
                long idx = (row0 + rel) + (long)ul, widx = (wrow0 + rel) + (long)ul;
                if (rel < HZ) {   // (compile time: the element that holds the zone rows)
                    const int r = max(lo + rel + gl, 0);
                    idx = fbase + r;
                    widx = wbase + r;
                }
                if (rel + G <= ROWS) {
                    Op<OP>::fetch_in(p, idx, s0[u], s1[u]);
                    if (WEIGHTED) sw[WEIGHTED ? u : 0] = p.w[widx];
                } else if (rel + gl < ROWS) {   // (the window's last, partial group of rows)
                    Op<OP>::fetch_in(p, idx, s0[u], s1[u]);
                    if (WEIGHTED) sw[WEIGHTED ? u : 0] = p.w[widx];
                }
            }
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int rel = G * (b0 + u);
                if (b0 + u >= NU) continue;
                if (rel + G <= ROWS || rel + gl < ROWS) {
                    Yp[rel + ul] = Op<OP>::y_of(p, s0[u], s1[u]);
                    if (WEIGHTED) Wp[rel + ul] = sw[WEIGHTED ? u : 0];
                }
            }
        }
    };
    if (interior && !(plan.ablate & 4)) {
        stage_interior();
    } else
    if (live && !(plan.ablate & 4)) {
        // every load of a batch is issued before the first is waited for; NB rows per lane and batch (the 31-sample chunks stage
        // 34 rows per lane: in one batch a two-operand op would hold 136 registers)
        constexpr int NB = NU <= 20 ? NU : (NU + 1) / 2;
#pragma unroll
        for (int b0 = 0; b0 < NU; b0 += NB) {
            double s0[NB], s1[NB], sw[WEIGHTED ? NB : 1];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int r = lo + G * (b0 + u) + gl;
                s0[u] = s1[u] = 0.0;
                if (b0 + u < NU && r >= 0 && r < hi) Op<OP>::fetch_in(p, fbase + r, s0[u], s1[u]);
                if (WEIGHTED) sw[WEIGHTED ? u : 0] = (b0 + u < NU && r >= 0 && r < hi && r < len - 1) ? p.w[wbase + r] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int r = lo + G * (b0 + u) + gl;
                if (b0 + u < NU && r >= 0 && r < hi) {
                    Yp[r - lo] = Op<OP>::y_of(p, s0[u], s1[u]);
                    if (WEIGHTED) Wp[r - lo] = sw[WEIGHTED ? u : 0];
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (plan.trace && lane == 0) plan.trace[8 * (size_t)wid + 2] = wall_clock64();

    // ---- speculative walk of this lane's chunk ---------------------------------------------------------------------------------
    const int cs = seg_s + gl * C;
    const int ce = min(cs + C, len);
    const bool has_chunk = cs < seg_e;
    const int start = max(0, cs - H);
    const LdsWin<WEIGHTED, 1> win{(lds_double *)Yp, (lds_double *)Wp, lo};
    const FarFibre<OP> far{p, fbase, 1, wbase};
    ChunkRec rec;
    bool certain = false;
    // ---- RUNS: the segment cut at the bends known a priori, run by run (chunkcore.hpp "known runs") ------------------------------------
    // Interior segments only (every row of the window exists; the fibre's last sample, with its own tests, is another segment's).  Four
    // phases, all inside the wave: (1) every lane looks at the edges of its chunk -- bends known a priori, and what decides a run of two
    // samples -- and borrows its neighbours' for the runs that cross into and out of its chunk; (2) runs of three and more samples are
    // listed; (3) lane i walks run i from its first bend to the closing one; what it finds goes to the chunks it belongs to; (4) every
    // lane puts its chunk's record together: piece ends and types, the bend before its chunk, the one that closes the segment.
    // Nothing is written to the window before phase 4 is through, and whatever does not fit -- a run longer than a lane takes, more
    // than 64 runs, no known bend within two samples of the segment start or eight of its end -- sends the whole wave to the
    // speculative walk below as if nothing had happened.
    bool solved = false;
    if constexpr (RUNS) {
        if (interior && p.lam > 0.0 && !(plan.ablate & 1)) {
            typedef __attribute__((address_space(3))) unsigned lds_uint;   // (LDS instructions, not flat ones: the atomics below are ds_or / ds_max)
            lds_uint *rl = (lds_uint *)(reinterpret_cast<unsigned *>(rtab + TS) + wave * kRunsWords);   // [0, 64) runs ; [64, 128) ends ; [128, 192) types ; [192] the bend before the segment
            constexpr unsigned CM = (1u << C) - 1u;
            const bool free0 = sg == 0 && lane == 0;   // the fibre starts here: no bend, height 0
            // (1) edges
            const EdgeMasks own = own_edges<C>(win, cs, p.lam);
            EdgeMasks pv, nx;
            pv.K = (unsigned)__shfl_up((int)own.K, 1); pv.P = (unsigned)__shfl_up((int)own.P, 1);
            pv.N = (unsigned)__shfl_up((int)own.N, 1); pv.B = (unsigned)__shfl_up((int)own.B, 1);
            nx.K = (unsigned)__shfl_down((int)own.K, 1); nx.P = (unsigned)__shfl_down((int)own.P, 1);
            nx.N = (unsigned)__shfl_down((int)own.N, 1); nx.B = (unsigned)__shfl_down((int)own.B, 1);
            {   // the edges no chunk of the segment owns: the T = 8 behind it (lanes 0 .. 7), the two before it (lanes 8, 9)
                unsigned k = 0, pp = 0, nn = 0, bb = 0;
                if (lane < 8) one_edge(win.y(seg_e - 1 + lane), win.y(seg_e + lane), p.lam, k, pp, nn, bb);
                else if (lane < 10 && sg > 0) one_edge(win.y(seg_s - 11 + lane), win.y(seg_s - 10 + lane), p.lam, k, pp, nn, bb);
                const unsigned xk = (unsigned)__ballot(k != 0u), xp = (unsigned)__ballot(pp != 0u), xn = (unsigned)__ballot(nn != 0u),
                               xb = (unsigned)__ballot(bb != 0u);
                if (lane == 63) { nx.K = (xk & 0xffu) << kEdgeBias; nx.P = (xp & 0xffu) << kEdgeBias; nx.N = (xn & 0xffu) << kEdgeBias; nx.B = (xb & 0xffu) << kEdgeBias; }
                if (lane == 0)  { pv.K = ((xk >> 8) & 3u) << C; pv.P = ((xp >> 8) & 3u) << C; pv.N = ((xn >> 8) & 3u) << C; pv.B = ((xb >> 8) & 3u) << C; }
            }
            EdgeMasks m;
            m.K = edge_ext<C>(own.K, pv.K, nx.K); m.P = edge_ext<C>(own.P, pv.P, nx.P);
            m.N = edge_ext<C>(own.N, pv.N, nx.N); m.B = edge_ext<C>(own.B, pv.B, nx.B);
            if (free0) m.K = (m.K & ~7u) | 4u;   // (the fibre start delimits the first run like a bend; nothing lies before it)
            unsigned BE, BT, WS;
            settle_short_runs(m, BE, BT, WS);
            if (free0) {   // no rule of thumb across the free end: the first run is walked unless it is one sample long
                BE = (BE & ~8u) | (m.K & 8u);
                BT = (BT & ~8u) | (m.P & m.K & 8u);
                WS = (WS & ~7u) | ((m.K & 8u) ? 0u : 4u);
            }
            // (2) this lane's runs: those whose first sample is its own; lane 0 also the one that comes in from before the segment
            unsigned dom = WS & ((1u << (C + kEdgeBias)) - 1u);
            if (lane > 0) dom &= ~3u;
            bool fail = lane == 0 && !free0 && (m.K & 7u) == 0u;   // (no bend known within two samples of the segment start)
            const int nruns = __popc(dom);
            int pos = nruns;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(pos, o);
                if (lane >= o) pos += t;
            }
            const int total = __shfl(pos, 63);
            pos -= nruns;
            rl[64 + lane] = 0u;
            rl[128 + lane] = 0u;
            if (lane == 0) rl[192] = 0u;
            while (dom) {
                const int b = __ffs((int)dom) - 1;
                dom &= dom - 1u;
                const int e = run_end(m.K, b);
                if (e < 0) fail = true;
                else if (pos < 64) rl[pos] = RunEntry::make(lane, b, e, (int)((m.P >> b) & 1u), free0 && b == kEdgeBias).word;
                pos++;
            }
            bool go = __ballot(fail) == 0ull && total <= 64;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (3) one run per lane
            if (go) {
                bool walked = true;
                if (lane < total) {
                    const RunEntry en{rl[lane]};
                    const int c0 = seg_s + C * en.lane() - kEdgeBias, a = c0 + en.b(), ee = c0 + en.e();
                    ChunkRec rr;
                    Walker w;
                    if (en.free_start()) {
                        walker_start<false>(w, win, 0, p.lam);
                    } else {
                        walker_restart_with<false>(w, a, en.type(), len, p.lam, win.y(a), 0.0, 0.0);
                        rr.mine = rr.next = rr.last = ((link_t)a << 1) | (link_t)en.type();
                    }
                    walk_chunk<OP, false, 1, false, TAB>(w, rr, win, far, hi, a, ee, len, p.lam, (unsigned)(unsigned long long)rtab);
                    walked = rr.done && !rr.failed;
                    if (walked) {
                        const int o = en.b() - kEdgeBias;
                        if (o >= 0) {
                            __hip_atomic_fetch_or(&rl[64 + en.lane()], rr.ends << o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_or(&rl[128 + en.lane()], rr.types << o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {   // (lane 0's run from before the segment: the pieces that end before it only say where the segment hangs)
                            __hip_atomic_fetch_or(&rl[64], rr.ends >> (-o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_or(&rl[128], rr.types >> (-o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const unsigned low = rr.ends & ((1u << (-o)) - 1u);
                            if (low) {
                                const int j = 31 - __clz((int)low);
                                __hip_atomic_fetch_max(&rl[192], ((unsigned)(en.b() + j + 2) << 1) | ((rr.types >> j) & 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
                go = __ballot(!walked) == 0ull;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (4) the chunk's record
            if (go) {
                const unsigned e_own = rl[64 + lane], t_own = rl[128 + lane];
                const unsigned e_prev = lane ? rl[64 + lane - 1] : 0u, t_prev = lane ? rl[128 + lane - 1] : 0u;
                const unsigned ends = ((BE >> (kEdgeBias + 1)) | e_own | (e_prev >> C)) & CM;
                const unsigned types = ((BT >> (kEdgeBias + 1)) | t_own | (t_prev >> C)) & CM & ends;
                const int lastb = ends ? 31 - __clz((int)ends) : -1;
                const link_t lcode = lastb >= 0 ? ((((link_t)(cs + lastb + 1)) << 1) | ((types >> lastb) & 1u)) : 0u;
                // the bend the segment hangs on (lane 0 knows): the last one among the edges before its first sample -- known a priori,
                // settled by rule, or found by the walk that came in from before the segment
                link_t hang = 0u;
                if (lane == 0 && !free0) {
                    unsigned best = rl[192];
                    const unsigned kb = BE & 7u;
                    if (kb) {
                        const int bi = 31 - __clz((int)kb);
                        const unsigned cand = ((unsigned)(bi + 1) << 1) | ((BT >> bi) & 1u);   // (edge + 1: zero means none)
                        best = cand > best ? cand : best;   // (ordered by the edge: the later bend wins; a bend has one type)
                    }
                    if (best) hang = (((link_t)(seg_s + (int)(best >> 1) - 1 - kEdgeBias)) << 1) | (best & 1u);
                }
                hang = (link_t)__shfl((int)hang, 0);
                const unsigned long long has = __ballot(lastb >= 0);
                const unsigned long long lower = has & ((1ull << lane) - 1ull);
                const int src = lower ? 63 - __clzll((long long)lower) : 0;
                const link_t from_lower = (link_t)__shfl((int)lcode, src);
                const link_t mine = lower ? from_lower : hang;
                bool bad_rec = mine == 0u && !(sg == 0 && lane == 0);   // (a first piece longer than a chunk at the fibre start: the walk's)
                link_t tail = 0u;
                if (lane == 63 && !((ends >> (C - 1)) & 1u)) {
                    // the piece that covers the segment's last sample ends behind it: at the first bend among the T edges there
                    const unsigned beyond = (BE >> (kEdgeBias + 1 + C)) | (e_own >> C), tbeyond = (BT >> (kEdgeBias + 1 + C)) | (t_own >> C);
                    if (beyond) {
                        const int j0 = __ffs((int)beyond) - 1;
                        tail = (((link_t)(seg_e + j0 + 1)) << 1) | ((tbeyond >> j0) & 1u);
                    } else {
                        bad_rec = true;
                    }
                }
                if (__ballot(bad_rec) == 0ull) {
                    rec.ends = ends;
                    rec.types = types;
                    rec.mine = mine;
                    rec.next = lastb >= 0 ? lcode : mine;
                    rec.last = lane == 63 && tail ? tail : rec.next;
                    rec.done = true;
                    certain = true;
                    solved = true;
                    if (lane == 0) plan.dirty.note(5);   // (option "why": waves solved run by run)
                } else if (lane == 0) {
                    plan.dirty.note(6);                  // (... that went to the speculative walk after all)
                }
            } else if (lane == 0) {
                plan.dirty.note(6);
            }
        }
    }
    if (has_chunk && !solved && !(plan.ablate & 1)) {
        Walker w;
        // (robust: the whole zone is searched -- a lane that starts at a bend known a priori has no link that could fail)
        constexpr int kLook = ROBUST ? kWarm - 2 : 8;
        int cat = -1, ctype = 0;
        if (start > 0 && H <= kWarm && (WEIGHTED || p.lam > 0.0)) cat = certain_bend_before<WEIGHTED, kLook>(win, cs, len, p.lam, ctype);
        if (cat >= 0) {
            certain = true;
            walker_restart_with<WEIGHTED>(w, cat, ctype, len, p.lam, win.y(cat), WEIGHTED ? win.r(cat - 1) : 0.0,
                                          (WEIGHTED && cat < len - 1) ? win.r(cat) : 0.0);
            rec.mine = rec.next = rec.last = ((link_t)cat << 1) | (link_t)ctype;
        } else {
            walker_start<WEIGHTED>(w, win, start, p.lam);
        }
        walk_chunk<OP, WEIGHTED, 1, ROBUST, TAB>(w, rec, win, far, hi, cs, ce, len, p.lam, (unsigned)(unsigned long long)rtab);
    }
    if (plan.trace && lane == 0) plan.trace[8 * (size_t)wid + 3] = wall_clock64();

    // ---- links: the predecessor is the lane before (a group's first lane: in another group or wave, left to the repair kernel) ----
    bool bad;
    bool head_linked = false, head_bad = false;   // ROBUST: the group's first lane hangs on another wave's last lane / and that link failed
    auto examine = [&]() {
        const link_t prev_next = (link_t)__shfl_up((int)rec.next, 1);
        const bool linked = has_chunk && !(start == 0 || certain) && gl > 0;
        bad = has_chunk && (rec.failed || (linked && (rec.mine == 0 || rec.mine != prev_next)) || (gl == 0 && head_bad));
        return prev_next;
    };
    // one more walk of this lane's chunk from a bend of its predecessor's walk (a bend of the true walk if the predecessor is true)
    auto second_chance = [&](link_t from) {
        const int at = (int)(from >> 1);
        if (from == 0 || at <= max(lo, 0)) return;
        ChunkRec again;
        Walker w;
        walker_restart_with<WEIGHTED>(w, at, (int)(from & 1u), len, p.lam, win.y(at), WEIGHTED ? win.r(at - 1) : 0.0,
                                      (WEIGHTED && at < len - 1) ? win.r(at) : 0.0);
        again.mine = again.next = again.last = from;
        walk_chunk<OP, WEIGHTED, 1, ROBUST, TAB>(w, again, win, far, hi, cs, ce, len, p.lam, (unsigned)(unsigned long long)rtab);
        if (!again.failed) {
            rec = again;
            certain = false;   // from now on the chunk hangs on its predecessor like any other
        }
    };
    auto rounds = [&]() {
        for (int round = 0; round < plan.rounds; round++) {
            const link_t prev_next = examine();
            if (__ballot(bad) == 0ull) break;
            const bool prev_bad = __shfl_up((int)bad, 1) != 0;
            if (bad && gl > 0 && !prev_bad) second_chance(prev_next);
        }
    };
    if constexpr (ROBUST) {
        rounds();
        if constexpr (G == 64) {
            examine();
            // the wave's last lane, as it stands now, for the next wave's first lane
            if (lane == 63) xwave[wave] = (bad || !has_chunk) ? rec.next : (rec.next | kLinkCertain);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 63) __hip_atomic_store(&xwave[kAlongWaves + wave], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            head_linked = has_chunk && lane == 0 && wave > 0 && sg > 0 && !(certain || rec.failed);
            if (head_linked) {
                while (__hip_atomic_load(&xwave[kAlongWaves + wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u)
                    __builtin_amdgcn_s_sleep(1);
                const link_t praw = xwave[wave - 1];
                const link_t prev = praw & ~kLinkCertain;
                head_bad = rec.mine == 0 || rec.mine != prev;
                if (head_bad && (praw & kLinkCertain)) {
                    second_chance(prev);
                    head_bad = rec.failed || rec.mine == 0 || rec.mine != prev;
                }
            }
            if (__ballot(head_linked) != 0ull) rounds();   // (the first lane may have walked again: its successors' links are looked at afresh)
        }
    }
    examine();
    if (gl == 0 && head_bad && !rec.failed) bad = false;   // (not this kernel's to flag: the repair kernel checks the links between segments)
    if (has_chunk) {
        if (rec.failed) {
            rec.mine = kLinkBad;
            rec.next = 0;
        }
        const int chunk = sg * G + gl;
        if (bad) flag_chunk(failflags, j, chunk, NC, plan.dirty, rec.failed);
        code_mine[j * NC + chunk] = (certain && rec.mine != kLinkBad) ? (rec.mine | kLinkCertain) : rec.mine;
        code_next[j * NC + chunk] = rec.next;
        // the segment's last chunk: what the next segment's first chunk must have begun with (checked by that segment at its end)
        if (plan.xlink && gl == G - 1 && sg + 1 < nseg) xlink_publish(plan.xlink + (size_t)j * nseg + sg, plan.dirty.epoch, rec.next);
    }
    // a lane's writes stop at the nearest unproven chunk before it (see GUARD in sweep_chunk_kernel; needed for H > C)
    int wlo = seg_s;
    if (H > C || ROBUST) {
        const unsigned long long all = __ballot(bad || (gl == 0 && head_bad));
        const unsigned long long grp = (G == 64) ? all : ((all >> (gi * G)) & ((1ull << (G & 63)) - 1ull));
        const unsigned long long below = grp & ((1ull << gl) - 1ull);
        if (below) wlo = seg_s + (63 - __clzll((long long)below)) * C;
    }
    if (!WEIGHTED && interior && !(plan.ablate & 1))   // (every lane of the wave holds a whole chunk: the form that keeps it in registers)
        rebuild_owned<Op<OP>, WEIGHTED, C, PTV_ALONG_UNROLL, TAB, lds_double *, (ROBUST ? TS : 0), 2>(win, rec, cs, ce, len, start, !bad, wlo, gl == G - 1, p.lam,
                                                                                      (lds_double *)rtab, nullptr, plan.legacy != 0);
    else if (has_chunk && !(plan.ablate & 1))
        rebuild_owned<Op<OP>, WEIGHTED, C, PTV_ALONG_UNROLL, TAB, lds_double *, (ROBUST ? TS : 0)>(win, rec, cs, ce, len, start, !bad, wlo, gl == G - 1 || ce == len, p.lam,
                                                                                (lds_double *)rtab, nullptr, plan.legacy != 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (plan.trace && lane == 0) plan.trace[8 * (size_t)wid + 4] = wall_clock64();

    // ---- stream the segment out: rows seg_s + G t + gl, t < C ----------------------------------------------------------------------
    if (interior && !(plan.ablate & 2)) {   // (the whole segment exists: scalar base + lane + immediate, nothing tested)
        const long out0 = fbase + seg_s;
#pragma unroll
        for (int t0 = 0; t0 < C; t0 += UL) {
            Ext ex[UL];
#pragma unroll
            for (int u = 0; u < UL; u++)
                ex[u] = (t0 + u < C && !Op<OP>::FUSED) ? Op<OP>::fetch(p, (out0 + G * (t0 + u)) + (long)ul) : Ext{0, 0};
#pragma unroll
            for (int u = 0; u < UL; u++) {
                if (t0 + u >= C) continue;
                const long idx = (out0 + G * (t0 + u)) + (long)ul;
                const double v = Yp[HZ + G * (t0 + u) + ul];
                if (Op<OP>::FUSED) Op<OP>::store_fused(p, idx, v);
                else               Op<OP>::finish(p, idx, ex[u], v);
            }
        }
    } else
    if (live && !(plan.ablate & 2)) {
#pragma unroll
        for (int t0 = 0; t0 < C; t0 += UL) {
            Ext ex[UL];
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const int k = seg_s + G * (t0 + u) + gl;
                ex[u] = (t0 + u < C && k < seg_e && !Op<OP>::FUSED) ? Op<OP>::fetch(p, fbase + k) : Ext{0, 0};
            }
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const int k = seg_s + G * (t0 + u) + gl;
                if (t0 + u < C && k < seg_e) {
                    const double v = Yp[k - lo];
                    if (Op<OP>::FUSED) Op<OP>::store_fused(p, fbase + k, v);
                    else               Op<OP>::finish(p, fbase + k, ex[u], v);
                }
            }
        }
    }
    // the link into this segment, against what the segment before published (it was dispatched earlier and published before its
    // rebuild: almost always there by now -- else the sweep is marked dirty and the repair kernel checks every boundary itself)
    if (plan.xlink && has_chunk && gl == 0 && sg > 0 && !certain) {
        const int why = xlink_check(plan.xlink + (size_t)j * nseg + sg - 1, plan.dirty.epoch, rec.mine);
        if (why) plan.dirty.set(why);
    }
    if (plan.trace && lane == 0) plan.trace[8 * (size_t)wid + 5] = wall_clock64();
}

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

// ---- kernel 3: local repair of unproven stretches -----------------------------------------------------------------------
// One lane per fibre.  Fast path (the common case): every link is proven -> return.  Otherwise the lane scans its
// chunks in order keeping `cur` = the last bend of the TRUE walk (chunk 0 is true by construction; a chunk whose
// `mine` code equals `cur` continues the true walk, so its outputs and its `next` code are true).  At the first
// chunk that does not, a sequential walk restarts from `cur` -- the walker state after a bend is a function of the
// bend alone -- rewrites the outputs from that chunk on, and after every chunk boundary it crosses checks whether the
// chunk recorded there continues ITS walk (same last bend): if so the recorded outputs beyond are exact and the walk
// stops; the scan resumes there.  Cost: the unproven stretches only (plus the overhang of their last piece), not
// the fibre.  Data with pieces much longer than a chunk fail everywhere and degrade to one sequential walk per fibre.
// The scan does not visit the chunks in between two failures (round 4): the chunk kernels PROVED every link they did not flag, so
// only the flagged range and the first chunks of workgroups whose link in failed can be rejected, and the scan jumps from one of
// those to the next (one bit per boundary, found by the check of the links across workgroups anyway).  At the upper end of rung 1
// (lambda = 0.65 - 0.7 on unit noise: ~600 failed links across tile workgroups per row sweep) the lanes of a wave scan in lockstep
// between their walks, and the scan was a third of the kernel: 4096^2 DR 18.3 -> 17.7, 22.1 -> 21.4 ms.
constexpr link_t kFromStart = 1;   // "no bend yet: the true walk is still in its first piece" (real codes are >= 2)

// what a repair walk keeps track of, whatever it reads its samples from
struct RepairBook {
    const link_t *code_mine;   // code of (chunk c, fibre j) at [c * cstride + j * fstride]
    long cstride, fstride, j;
    int C, len;
    int wfrom = 0;             // outputs are (re)written from this sample on
    int boundary = 0;          // next chunk boundary whose chunk may take over
    link_t last = 0;           // last bend of this walk so far
    bool stop = false;
    int resume_chunk = 0;
    link_t resume_code = 0;

    // `cur`: the last bend of the true walk at or before the chunk (0: none, the walk starts at sample 0).  The chunk
    // kernels leave the rows of a piece to the lane in whose chunk it ends, and an unproven lane keeps to its own rows:
    // the rows between that bend and the chunk belong to the repair walk as well.
    __device__ __forceinline__ void begin(int chunk, link_t cur) {
        wfrom = cur ? (int)(cur >> 1) : 0;
        boundary = (chunk + 1) * C;
        last = cur;
        stop = false;
    }
    __device__ __forceinline__ void bend(int at, int type) {
        const link_t code = ((link_t)at << 1) | (link_t)type;
        while (!stop && boundary < len && at >= boundary) {
            const link_t here = (at == boundary) ? code : last;      // this walk's last bend at-or-before `boundary`
            const int c = boundary / C;
            link_t m = code_mine[(long)c * cstride + j * fstride];
            if (m != kLinkBad) m &= ~kLinkCertain;
            if (m != 0 && m == here) {
                stop = true;
                resume_chunk = c;
                resume_code = here;
            } else {
                boundary += C;
            }
        }
        last = code;
    }
    __device__ __forceinline__ bool keep_going(int) const { return !stop; }
    __device__ __forceinline__ int limit() const { return 1 << 30; }
};

// repair walk straight from global memory (the global-memory geometries: long stretches, pipelined walker)
template <int OP, bool WEIGHTED>
struct RepairSource : RepairBook {
    const SweepArgs &p;
    long base, inc, wbase;
    LazyRun<OP> run;
    __device__ __forceinline__ RepairSource(const RepairBook &b, const SweepArgs &p_, long base_, long inc_, long wbase_)
        : RepairBook(b), p(p_), base(base_), inc(inc_), wbase(wbase_) {}
    __device__ __forceinline__ double y(int i) const { return Op<OP>::load_y(p, base + (long)i * inc); }
    __device__ __forceinline__ double r(int i) const { return p.w[wbase + (long)i * inc]; }
    __device__ __forceinline__ void piece(int from, int to, double v) {
        if (to >= wfrom) run.queue(p, base, inc, max(from, wfrom), to, v);
    }
    __device__ __forceinline__ void pump() { run.pump(p, base, inc); }
    __device__ __forceinline__ void flush() { run.flush(p, base, inc); }
};

// Repair walk through a per-lane LDS window (the LDS geometries: short stretches of short pieces, where a dependent
// global access per sample AND per piece is all the cost -- 200 us for a 100-sample repair).  The lane fetches
// kRepairWindow samples of its fibre in batches of 32 independent loads, walks them out of LDS, parks the piece values
// in a second LDS plane and writes the outputs of the whole stretch at the end, 16 operand fetches in flight.
constexpr int kRepairWindow = 64;
constexpr int kRepairBack = 8;   // samples kept before the one that triggered a refill (short rewinds stay inside)

template <int OP, bool WEIGHTED>
struct WindowRepairSource : RepairBook {
    const SweepArgs &p;
    long base, inc, wbase;
    double *Yw, *Xw, *Rw;      // this lane's columns of the LDS planes: window slot s at [s * 64]
    int wlo = 0, whi = 0;      // samples in the window: [wlo, whi)
    int xlo = 0, xhi = 0;      // samples whose outputs wait in Xw: [xlo, xhi)
    __device__ __forceinline__ WindowRepairSource(const RepairBook &b, const SweepArgs &p_, long base_, long inc_,
                                                  long wbase_, double *lds, int lane)
        : RepairBook(b), p(p_), base(base_), inc(inc_), wbase(wbase_), Yw(lds + lane),
          Xw(lds + kRepairWindow * 64 + lane), Rw(lds + 2 * kRepairWindow * 64 + lane) {}

    // (this kernel is a handful of waves, each as slow as its slowest lane's chain of memory round trips: the batches are as
    // large as the registers of a wave that has the SIMD to itself allow)
    static constexpr int kFlushBatch = 16, kFillBatch = WEIGHTED ? 16 : 32;
    __device__ __forceinline__ void flush() {
        for (int k = xlo; k < xhi; k += kFlushBatch) {
            Ext e[kFlushBatch];
#pragma unroll
            for (int u = 0; u < kFlushBatch; u++)
                if (k + u < xhi) e[u] = Op<OP>::fetch(p, base + (long)(k + u) * inc);
#pragma unroll
            for (int u = 0; u < kFlushBatch; u++)
                if (k + u < xhi) Op<OP>::finish(p, base + (long)(k + u) * inc, e[u], Xw[(k + u - wlo) * 64]);
        }
        xlo = xhi = 0;
    }
    __device__ __forceinline__ void refill(int i) {
        flush();   // the parked outputs are addressed relative to the window
        wlo = max(0, i - kRepairBack);
        whi = min(len, wlo + kRepairWindow);
        for (int b = 0; b < kRepairWindow; b += kFillBatch) {
            double t[kFillBatch], rr[WEIGHTED ? kFillBatch : 1];
#pragma unroll
            for (int u = 0; u < kFillBatch; u++) {
                const int k = wlo + b + u;
                t[u] = (k < whi) ? Op<OP>::load_y(p, base + (long)k * inc) : 0.0;
                if (WEIGHTED) rr[WEIGHTED ? u : 0] = (k < whi && k < len - 1) ? p.w[wbase + (long)k * inc] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kFillBatch; u++) {
                Yw[(b + u) * 64] = t[u];
                if (WEIGHTED) Rw[(b + u) * 64] = rr[WEIGHTED ? u : 0];
            }
        }
    }
    __device__ __forceinline__ double y(int i) {
        if (i < wlo || i >= whi) refill(i);
        return Yw[(i - wlo) * 64];
    }
    __device__ __forceinline__ double r(int i) {
        if (i < wlo || i >= whi) refill(i);
        return Rw[(i - wlo) * 64];
    }
    __device__ __forceinline__ void piece(int from, int to, double v) {
        from = max(from, wfrom);
        if (from > to) return;
        if (from < wlo || to >= whi) {   // (partly) outside the window -- a piece longer than the look-back: write it directly
            const int a = (to >= whi) ? to : min(to, wlo - 1);
            if (to >= whi) flush();
            write_run<OP>(p, base, inc, from, a, v);
            from = a + 1;
            if (from > to) return;
        }
        for (int k = from; k <= to; k++) Xw[(k - wlo) * 64] = v;
        if (xlo == xhi) xlo = from;
        xhi = to + 1;
    }
};

// ---- kernel 3a (staged for round 5, option repair_jobs): one lane per FAILURE ---------------------------------------------------------
// At the upper end of rung 1 the repair kernel is a fifth of a solve, and a launch lasts as long as its worst fibre: 3-4 failed
// links across workgroups repaired one after the other by one lane.  Those failures are almost always independent -- a speculative
// walk that missed its link meets the true one within a chunk or two, far from the next failure 8+ chunks on -- so here every
// failing boundary gets its own lane ("job"): it starts from the bend its predecessor chunk recorded, walks through ONE window of
// its fibre and PARKS its values in LDS, writing nothing.  Then the (at most four) jobs of a fibre, four adjacent lanes, are
// looked at in order: job k is valid iff the last valid job before it re-synchronised at a chunk r <= X_k - 1 -- then chunk
// X_k - 1's recorded codes are true, which is all job k assumed; a job the previous valid walk ran through is discarded (that walk
// IS the truth there).  Valid jobs flush, and the fibre is marked handled for the sequential kernel behind.  Anything else -- a walk
// that leaves its window, more than four failures, links flagged inside a workgroup -- touches nothing and leaves the fibre to the
// sequential kernel: exactness never rests on this one.
#ifndef PTV_JOB_WINDOW
#define PTV_JOB_WINDOW 128
#endif
constexpr int kJobWindow = PTV_JOB_WINDOW;   // samples a job may see: from the bend it starts at
constexpr int kJobsPerFibre = 4;
constexpr int kJobAhead = 4;      // codes of the chunks behind the failed link that a job fetches before it walks

// A job is one wave-lane alone with the memory latency (a workgroup per CU, one wave): what it costs is the number of DEPENDENT
// round trips, ~2 us each.  So everything is fetched in as few, as wide batches as the registers allow: the fail flags with all the
// boundary codes (1), the codes around the failed link -- where the walk starts, where it may hand over -- (1), the first 64
// samples of the window (1; the second 64 only for the walk that gets that far), the operands of the outputs 32 at a time (1-2).
template <int OP, bool WEIGHTED>
struct JobSource : RepairBook {
    const SweepArgs &p;
    long base, inc, wbase;
    // The outputs take the place of the samples: a piece covers samples up to its bend, and the walk never looks at or before a bend again.
    double *Yw, *Rw;           // this lane's columns of the LDS planes: window slot s at [s * 64]
    int wlo = 0, whi = 0;      // samples of the window: [wlo, whi), of which the first `got` are in LDS
    int got = 0;
    int xlo = 0, xhi = 0;      // samples whose outputs wait in Yw: [xlo, xhi)
    bool abort = false;        // the walk needed something outside its window
    int ahead0 = 1 << 30;      // ahead[u] = code_mine of chunk ahead0 + u
    link_t ahead[kJobAhead] = {};
    __device__ __forceinline__ JobSource(const RepairBook &b, const SweepArgs &p_, long base_, long inc_, long wbase_, double *lds, int lane)
        : RepairBook(b), p(p_), base(base_), inc(inc_), wbase(wbase_), Yw(lds + lane), Rw(lds + kJobWindow * 64 + lane) {}
    static constexpr int kFillBatch = WEIGHTED ? 32 : 64;
    __device__ __forceinline__ void fill_more() {
        double t[kFillBatch], rr[WEIGHTED ? kFillBatch : 1];
#pragma unroll
        for (int u = 0; u < kFillBatch; u++) {
            const int k = wlo + got + u;
            t[u] = (k < whi) ? Op<OP>::load_y(p, base + (long)k * inc) : 0.0;
            if (WEIGHTED) rr[WEIGHTED ? u : 0] = (k < whi && k < len - 1) ? p.w[wbase + (long)k * inc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kFillBatch; u++) {
            Yw[(got + u) * 64] = t[u];
            if (WEIGHTED) Rw[(got + u) * 64] = rr[WEIGHTED ? u : 0];
        }
        got += kFillBatch;
    }
    __device__ __forceinline__ void fill(int from) {
        wlo = max(0, from - 1);   // (a weighted restart reads the edge before its bend)
        whi = min(len, wlo + kJobWindow);
        got = 0;
        fill_more();
        if (WEIGHTED) fill_more();
    }
    __device__ __forceinline__ double y(int i) {
        if (i < wlo || i >= whi) { abort = true; return 0.0; }
        while (i - wlo >= got) fill_more();
        return Yw[(i - wlo) * 64];
    }
    __device__ __forceinline__ double r(int i) {
        if (i < wlo || i >= whi) { abort = true; return 0.0; }
        while (i - wlo >= got) fill_more();
        return Rw[(i - wlo) * 64];
    }
    // (job_walk: samples known to be in LDS)
    __device__ __forceinline__ double win_y(int i) const { return Yw[(i - wlo) * 64]; }
    __device__ __forceinline__ double win_r(int i) const { return Rw[(i - wlo) * 64]; }
    __device__ __forceinline__ void piece(int from, int to, double v) {
        from = max(from, wfrom);
        if (from > to) return;
        if (from < wlo || to >= whi) { abort = true; return; }
        while (to - wlo >= got) fill_more();   // (the samples behind the piece's end must be in before outputs take their place)
        for (int k = from; k <= to; k++) Yw[(k - wlo) * 64] = v;
        if (xlo == xhi) xlo = from;
        xhi = to + 1;
    }
    // RepairBook::bend with the codes of the first chunks behind the failed link out of registers
    __device__ __forceinline__ void bend(int at, int type) {
        const link_t code = ((link_t)at << 1) | (link_t)type;
        while (!stop && boundary < len && at >= boundary) {
            const link_t here = (at == boundary) ? code : last;
            const int c = boundary / C;
            const int d = c - ahead0;
            link_t m;
            if (d >= 0 && d < kJobAhead) {
                m = ahead[0];
#pragma unroll
                for (int u = 1; u < kJobAhead; u++) m = (d == u) ? ahead[u] : m;
            } else {
                m = code_mine[(long)c * cstride + j * fstride];
            }
            if (m != kLinkBad) m &= ~kLinkCertain;
            if (m != 0 && m == here) {
                stop = true;
                resume_chunk = c;
                resume_code = here;
            } else {
                boundary += C;
            }
        }
        last = code;
    }
    __device__ __forceinline__ bool keep_going(int) const { return !stop && !abort; }
#ifdef PTV_JOB_TABDIV
    // a / span through a table of reciprocals in LDS (span <= kJobWindow) and one correction: the dozen dependent instructions of an IEEE
    // quotient are a tenth of a trip of a walk that has its SIMD to itself
    const double *tab = nullptr;
    __device__ __forceinline__ double over_span(double a, int span) const {
        const double s = (double)span, inv = tab[span];
        const double q = a * inv;
        return __builtin_fma(__builtin_fma(-q, s, a), inv, q);
    }
#endif
    __device__ __forceinline__ void flush() {
        constexpr int kFlushBatch = 32;
        for (int k = xlo; k < xhi; k += kFlushBatch) {
            Ext e[kFlushBatch];
#pragma unroll
            for (int u = 0; u < kFlushBatch; u++)
                if (k + u < xhi) e[u] = Op<OP>::fetch(p, base + (long)(k + u) * inc);
#pragma unroll
            for (int u = 0; u < kFlushBatch; u++)
                if (k + u < xhi) Op<OP>::finish(p, base + (long)(k + u) * inc, e[u], Yw[(k + u - wlo) * 64]);
        }
    }
};

// The walk of a job.  A wave with a SIMD to itself issues an instruction every ~8 cycles whatever it is, and a taken branch costs
// several of them: the trips of walker_run -- ~120 instructions and ~15 branches -- were 45 of the 57 us of a launch, whether the window
// was 64 or 128 samples, whether the quotients were IEEE or a table.  So the interior trips are chunkcore.hpp's walk_interior again:
// straight-line predicated code out of the LDS window, one branch for the books of a bend, SpanDiv quotients; the fibre's last sample
// keeps walker_run.  Same state machine and operation order as walker_run.
template <bool WEIGHTED, class S>
__device__ __forceinline__ void job_walk(Walker &w, S &src, int len, double lam) {
    for (;;) {
        const int lim = min(len - 1, src.wlo + src.got - 1);   // a trip looks one sample ahead
        if (w.i >= lim) {
            if (w.i >= len - 1) break;
            if (src.wlo + src.got >= src.whi) {   // the end of the window
                src.abort = true;
                return;
            }
            src.fill_more();
            continue;
        }
        double yi = src.win_y(w.i);
        while (!src.stop && w.i < lim) {
            const int i = w.i;
            const double ynx = src.win_y(i + 1);   // speculative: most trips advance by one
            const double r = WEIGHTED ? src.win_r(i) : lam;
            const double h1 = w.hlo + (w.lo - yi);
            const double h2 = w.hhi + (w.hi - yi);
            const bool cv = r < h1;
            const bool fv = !cv && (-r > h2);
            const bool bend = cv || fv;
            const int brk = cv ? w.klo : w.khi;
            const int at = brk + 1;                // wfrom < at <= i: an interior sample inside the window
            const double yat = src.win_y(at), yat1 = src.win_y(at + 1);

            // no bend: pull the pieces back inside the tube where they left it
            const SpanDiv over((double)(i - w.k0));
            const double d2 = ptv_min(r - h2, 0.0), d1 = ptv_max(-r - h1, 0.0);
            const double nhi = w.hi + over(d2), nlo = w.lo + over(d1);
            const double nhhi = ptv_min(h2, r), nhlo = ptv_max(h1, -r);
            const int nkhi = (h2 >= r) ? i : w.khi, nklo = (h1 <= -r) ? i : w.klo;

            // bend: closed-form first sample of the new piece (walker_restart_with, at < len - 1)
            double blo, bhi, bhhi, bhlo;
            if (WEIGHTED) {
                const double wp = src.win_r(brk), wc = src.win_r(at);
                const double a = cv ? yat + wp : yat - wp;
                blo = a - wc;
                bhi = a + wc;
                bhhi = wc;
                bhlo = -wc;
            } else {
                blo = cv ? yat : 2 * (-lam) + yat;
                bhi = cv ? 2 * lam + yat : yat;
                bhhi = lam;
                bhlo = -lam;
            }
            if (bend) {   // (every few trips: the finished piece's values into the window, the hand-over test at a chunk boundary)
                src.piece(w.k0 + 1, brk, cv ? w.lo : w.hi);
                src.bend(at, cv ? BEND_CEIL : BEND_FLOOR);
            }
            w.lo = bend ? blo : nlo;
            w.hi = bend ? bhi : nhi;
            w.hlo = bend ? bhlo : nhlo;
            w.hhi = bend ? bhhi : nhhi;
            w.k0 = bend ? brk : w.k0;
            w.klo = bend ? at : nklo;
            w.khi = bend ? at : nkhi;
            w.i = (bend ? at : i) + 1;
            yi = bend ? yat1 : ynx;
        }
        if (src.stop || src.abort) return;
    }
    walker_run<WEIGHTED>(w, src, len, lam);   // the fibre's last sample, with its own tests (and whatever a bend there rewinds to)
}

template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void sweep_repair_jobs_kernel(SweepArgs p, FibreGeom g, int C, int H, int chunks_per_wg,
                                                                const link_t *code_mine, const link_t *code_next, const int *failflags,
                                                                int *failcount, long cstride, long fstride, DirtyMark dirty,
                                                                unsigned *handled) {
    extern __shared__ __attribute__((aligned(16))) double repair_lds[];
    if (__hip_atomic_load(dirty.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != dirty.epoch) return;
    if (p.gate && *p.gate == 0) return;
    const int lane = threadIdx.x, slot = lane & (kJobsPerFibre - 1);
    const long j = (long)blockIdx.x * (64 / kJobsPerFibre) + (lane / kJobsPerFibre);
    const int len = g.len;
    const int NC = (len + C - 1) / C;
    const int nbound = (NC + chunks_per_wg - 1) / chunks_per_wg;
    const bool live = j < g.count && nbound <= 64;
    const long jj = live ? j : 0;
    // the fail flags and the links across workgroups of this fibre in one round trip (every lane of the fibre looks at all of them)
    const int f0 = failflags[2 * jj], f1 = failflags[2 * jj + 1];
    unsigned long long xbad = 0ull;
    {
        constexpr int UB = 32;
        for (int c0 = chunks_per_wg; c0 < NC; c0 += UB * chunks_per_wg) {
            link_t in[UB], out[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int c = min(c0 + u * chunks_per_wg, NC - 1);
                in[u] = code_mine[(long)c * cstride + jj * fstride];
                out[u] = code_next[(long)(c - 1) * cstride + jj * fstride];
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int c = c0 + u * chunks_per_wg;
                const bool certain = (in[u] & kLinkCertain) && in[u] != kLinkBad;
                if (c < NC && c * C - H > 0 && !certain && (in[u] == 0 || in[u] != out[u])) xbad |= 1ull << ((c / chunks_per_wg) & 63);
            }
        }
    }
    const int nb = __popcll(xbad);
    // (links flagged inside a workgroup, more failures than lanes: the sequential kernel's)
    const bool mine_to_do = live && f0 == 0 && f1 == 0 && nb >= 1 && nb <= kJobsPerFibre;
    const bool have_job = mine_to_do && slot < nb;
    // this lane's job: the slot-th failing boundary of its fibre
    int X = NC, r = NC;
    bool abort = false;
    if (have_job) {
        unsigned long long m = xbad;
        for (int k = 0; k < slot; k++) m &= m - 1;
        X = (int)__builtin_ctzll(m) * chunks_per_wg;
    }
    long base, wbase;
    {
        long blk, off;
        divmod_nonneg(jj, g.inc, blk, off);
        base = blk * g.inc * len + off;
        wbase = blk * g.inc * (len - 1) + off;
    }
    const RepairBook book{code_mine, cstride, fstride, jj, C, len};
    JobSource<OP, WEIGHTED> src(book, p, base, g.inc, wbase, repair_lds, lane);
#ifdef PTV_JOB_TABDIV
    {
        double *tab = repair_lds + (size_t)(1 + (WEIGHTED ? 1 : 0)) * kJobWindow * 64;
        for (int k = lane; k <= kJobWindow + 1; k += 64) tab[k] = k ? 1.0 / (double)k : 0.0;
        __syncthreads();
        src.tab = tab;
    }
#endif
    if (have_job) {
        // one round trip: the codes the walk starts from (the last bend before the link) and may hand over at
        constexpr int KB = 4;
        link_t back[KB];
#pragma unroll
        for (int u = 0; u < KB; u++) back[u] = (X - 1 - u >= 0) ? code_next[(long)(X - 1 - u) * cstride + j * fstride] : 0u;
        const link_t mine = code_mine[(long)X * cstride + j * fstride];
#pragma unroll
        for (int u = 0; u < kJobAhead; u++) src.ahead[u] = (X + 1 + u < NC) ? code_mine[(long)(X + 1 + u) * cstride + j * fstride] : 0u;
        src.ahead0 = X + 1;
        link_t cur = 0u;
#pragma unroll
        for (int u = KB - 1; u >= 0; u--) cur = (back[u] != 0) ? back[u] : cur;
        if (cur == 0u) {
            cur = kFromStart;
            for (int b = X - 1 - KB; b >= 0; b--) {
                const link_t nx = code_next[(long)b * cstride + j * fstride];
                if (nx != 0) {
                    cur = nx;
                    break;
                }
            }
        }
        if (mine != 0 && mine != kLinkBad && mine == cur) {
            r = X;   // (its predecessor had no bend of its own: the link holds after all -- what the sequential scan finds too)
        } else {
            const link_t from = (cur == kFromStart) ? 0u : cur;
            src.begin(X, from);
            src.fill(cur == kFromStart ? 0 : (int)(cur >> 1));
            Walker w;
            if (cur == kFromStart) walker_start<WEIGHTED>(w, src, 0, p.lam);
            else walker_restart<WEIGHTED>(w, src, (int)(cur >> 1), (int)(cur & 1u), len, p.lam);
#ifdef PTV_JOB_NOWALK   // (timing diagnostic: everything but the walk and the flush; every fibre goes on to the sequential kernel)
            src.abort = true;
#elif defined(PTV_JOB_PLAIN_WALK)
            walker_run<WEIGHTED>(w, src, len, p.lam);
#else
            job_walk<WEIGHTED>(w, src, len, p.lam);
#endif
            abort = src.abort;
            r = src.stop ? src.resume_chunk : NC;   // (not stopped and not aborted: it walked to the fibre end inside its window)
        }
    }
    // the jobs of a fibre in order (four adjacent lanes): which of them are valid, and is the fibre this kernel's at all
    const int lane0 = lane & ~(kJobsPerFibre - 1);
    bool fibre_ok = mine_to_do, valid = false;
    int lastr = -1;
#pragma unroll
    for (int k = 0; k < kJobsPerFibre; k++) {
        const int Xk = __shfl(X, lane0 + k), rk = __shfl(r, lane0 + k);
        const bool ak = __shfl((int)abort, lane0 + k) != 0;
        const bool jobk = k < nb;
        if (jobk && ak) fibre_ok = false;
        const bool vk = jobk && (lastr < 0 || lastr <= Xk - 1);
        if (vk) lastr = rk;
        if (k == slot) valid = vk;
    }
    if (fibre_ok && have_job && valid) {
        src.flush();
        atomicAdd(failcount + 1, r - X);   // chunks rewritten
    }
    if (fibre_ok && slot == 0) {
        handled[j] = dirty.epoch;
        atomicAdd(failcount, 1);           // fibres that needed a repair
    }
}

template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void sweep_repair_kernel(SweepArgs p, FibreGeom g, int C, int H, int chunks_per_wg,
                                                           const link_t *code_mine, const link_t *code_next,
                                                           int *failflags, int *failcount, long cstride, long fstride,
                                                           DirtyMark dirty, const unsigned *handled = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double repair_lds[];   // (2 + WEIGHTED) planes of kRepairWindow x 64 (LDS geometries only)
    // the common case: the chunk kernel proved every link itself, across its workgroups too, and said so by NOT marking the sweep
    if (dirty.word && __hip_atomic_load(dirty.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != dirty.epoch) return;
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    if (j >= g.count) return;
    if (p.gate && *p.gate == 0) return;
    if (handled && handled[j] == dirty.epoch) return;   // (sweep_repair_jobs_kernel repaired this fibre)
    const int len = g.len;
    const int NC = (len + C - 1) / C;
    // first / last chunk with an unproven link: what the chunk kernel flagged, widened below by the links between workgroups
    int first = NC, lastbad = -1;
    {
        const int f0 = failflags[2 * j], f1 = failflags[2 * j + 1];
        if (f0 > 0) first = NC - f0;
        if (f1 > 0) lastbad = f1 - 1;
    }
    // Which chunks may fail the scan below at all: those inside the range the chunk kernel flagged [ff, fl] (links inside a workgroup,
    // walks that ran off their window) and the first chunk of a workgroup whose link IN failed (xbad: one bit per boundary).  Every
    // other chunk was proven by the chunk kernel to continue its predecessor's walk -- mine == the predecessor's non-zero next code, or
    // a start at a bend known a priori -- so the scan accepts it whenever its predecessor is true: the scan may jump over them.
    const int ff = first, fl = lastbad;
    unsigned long long xbad = 0ull;
    const bool jump = (NC + chunks_per_wg - 1) / chunks_per_wg <= 64;
    // links between workgroups (inside a workgroup they were checked through LDS): 16 boundaries = 32 independent
    // loads in flight per lane -- the cost of the common case is the latency of these reads
    constexpr int UB = 16;
    for (int c0 = chunks_per_wg; c0 < NC; c0 += UB * chunks_per_wg) {
        link_t in[UB], out[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int c = min(c0 + u * chunks_per_wg, NC - 1);
            in[u] = code_mine[(long)c * cstride + j * fstride];
            out[u] = code_next[(long)(c - 1) * cstride + j * fstride];
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int c = c0 + u * chunks_per_wg;
            const bool certain = (in[u] & kLinkCertain) && in[u] != kLinkBad;
            if (c < NC && c * C - H > 0 && !certain && (in[u] == 0 || in[u] != out[u])) {
                first = min(first, c);
                lastbad = max(lastbad, c);
                if (jump) xbad |= 1ull << (c / chunks_per_wg);
            }
        }
    }
    if (lastbad < 0) return;
    failflags[2 * j] = failflags[2 * j + 1] = 0;
    atomicAdd(failcount, 1);       // fibres that needed a repair
    int walks = 0;

    long blk, off;
    divmod_nonneg(j, g.inc, blk, off);
    const long base = blk * g.inc * len + off, wbase = blk * g.inc * (len - 1) + off;
    const RepairBook book{code_mine, cstride, fstride, j, C, len};
    const bool windowed = H <= kWarmLong;
    RepairSource<OP, WEIGHTED> gsrc(book, p, base, g.inc, wbase);
    WindowRepairSource<OP, WEIGHTED> wsrc(book, p, base, g.inc, wbase, repair_lds, (int)threadIdx.x);
    // every chunk before `first` is proven: the true walk's last bend there is the last non-zero `next` code before it
    auto last_bend_before = [&](int chunk) {
        for (int b = chunk - 1; b >= 0; b--) {
            const link_t nx = code_next[(long)b * cstride + j * fstride];
            if (nx != 0) return nx;
        }
        return kFromStart;
    };
    // the first chunk at or after `chunk` that the scan could reject (NC: none)
    auto next_suspect = [&](int chunk) {
        if (!jump || (chunk >= ff && chunk <= fl)) return chunk;
        int best = chunk < ff ? ff : NC;
        const int b0 = (chunk + chunks_per_wg - 1) / chunks_per_wg;
        if (b0 < 64) {
            const unsigned long long m = xbad >> b0;
            if (m) best = min(best, (b0 + (int)__builtin_ctzll(m)) * chunks_per_wg);
        }
        return best;
    };
    link_t cur = last_bend_before(first);
    int c = first;
    // Two-phase loop so that the lanes of a wave repair TOGETHER: first every lane scans ahead to its next unproven
    // chunk, then all lanes that found one walk at the same time (a walk nested inside the scan would serialise the
    // lanes, each reaching its repair at a different trip).
    while (true) {
        // the scan reads the codes of UB chunks at a time (2 UB independent loads), then goes through them in registers:
        // one memory round trip per UB chunks instead of two per chunk
        bool rejected = false;
        while (c < NC && c <= lastbad && !rejected) {   // (everything after the last flagged chunk is proven)
            // jump over the chunks that cannot be rejected (after a repair walk: from the chunk that took its walk over, whose codes
            // and everything after it are true) -- one memory round trip per failure instead of one per UB chunks in between
            const int suspect = next_suspect(c);
            if (suspect > c) {
                if (suspect >= NC || suspect > lastbad) {
                    c = suspect;
                    break;
                }
                // The true walk's last bend before `suspect`: the latest one recorded by the chunks jumped over -- true records, of chunks
                // the chunk kernels proved.  The scan stops at the chunk the jump started from: the chunks BEFORE c may have been rewritten
                // by a repair walk since, and their records are the speculative walks' still.  (A proven chunk's `next` is never zero --
                // DESIGN "exactness" -- so the chunk just before `suspect` ends the scan in practice; the bound makes that argument unnecessary.)
                for (int b = suspect - 1; b >= c; b--) {
                    const link_t nx = code_next[(long)b * cstride + j * fstride];
                    if (nx != 0) {
                        cur = nx;
                        break;
                    }
                }
                c = suspect;
            }
            link_t mm[UB], nn[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int cc = min(c + u, NC - 1);
                mm[u] = code_mine[(long)cc * cstride + j * fstride];
                nn[u] = code_next[(long)cc * cstride + j * fstride];
            }
            const int c0 = c;
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (!rejected && c == c0 + u && c < NC) {
                    const link_t mraw = mm[u];
                    const bool certain = (mraw & kLinkCertain) && mraw != kLinkBad;
                    const link_t m = certain ? (mraw & ~kLinkCertain) : mraw;
                    // a chunk whose walk began at sample 0 (or at a bend known a priori) is the true walk unless it
                    // ran off its window (kLinkBad)
                    const bool accept = (c * C - H <= 0 || certain) ? (m != kLinkBad) : (m != 0 && m == cur);
                    if (accept) {
                        if (nn[u] != 0) cur = nn[u];
                        c++;
                    } else {
                        rejected = true;
                    }
                }
            }
        }
        if (c >= NC || !rejected) break;
        const link_t from = (cur == kFromStart) ? 0u : cur;
        bool stopped;
        int resume_chunk;
        link_t resume_code;
        Walker w;
        if (windowed) {
            wsrc.begin(c, from);
            if (cur == kFromStart) walker_start<WEIGHTED>(w, wsrc, 0, p.lam);
            else walker_restart<WEIGHTED>(w, wsrc, (int)(cur >> 1), (int)(cur & 1u), len, p.lam);
            walker_run<WEIGHTED>(w, wsrc, len, p.lam);
            wsrc.flush();
            stopped = wsrc.stop; resume_chunk = wsrc.resume_chunk; resume_code = wsrc.resume_code;
        } else {   // global-memory geometries: long pieces
            gsrc.begin(c, from);
            if (cur == kFromStart) walker_start<WEIGHTED>(w, gsrc, 0, p.lam);
            else walker_restart<WEIGHTED>(w, gsrc, (int)(cur >> 1), (int)(cur & 1u), len, p.lam);
            walker_run_blocked<WEIGHTED, kGlobalBlock>(w, gsrc, len, p.lam);
            stopped = gsrc.stop; resume_chunk = gsrc.resume_chunk; resume_code = gsrc.resume_code;
        }
        walks += stopped ? (resume_chunk - c) : (NC - c);   // chunks this walk had to rewrite
        if (!stopped) break;                    // walked to the fibre end: everything from chunk c on is rewritten
        c = resume_chunk;                       // that chunk continues this walk: accepted on the next trip
        cur = resume_code;
    }
    atomicAdd(failcount + 1, walks);   // chunks rewritten
}

// ---- kernel 4 (option certify): the optimality conditions of the prox on what a sweep WROTE ------------------------------------------
// x = prox(y) minimises 1/2 |x - y|^2 + sum_k r_k |x_{k+1} - x_k| (the problem every solver of the reference's 1-D path solves:
// src/TVL1opt.cpp:359-564) iff, with u_k = sum_{i <= k} (y_i - x_i):
//     |u_k| <= r_k for every edge k ;  u_k = -r_k where x_{k+1} > x_k ,  u_k = +r_k where x_{k+1} < x_k ;  u_{n-1} = 0
// -- the minimiser is unique, so a fibre that passes IS the prox, whatever kernel wrote it and whatever went wrong on the way.  The
// check reads the sweep's inputs through the op's own input functor and recovers x from the sweep's outputs (Op::recover), so it
// sees exactly what the next sweep will see.  Tolerances: sums of n terms accumulate ~n ulps of the operands' magnitude, a recovered
// x carries a few ulps of it -- a violation counts above kCertifyTol * n * 2^-52 * (largest operand), a step of x above
// kCertifyStep * 2^-52 * that (+ the slack below).  On top of rounding comes the slack the reference itself leaves: its solvers close the fibre's last piece
// with tests against EPSILON = 1e-10 (src/general.h:64-67, src/TVL1opt.cpp:543-557 -- walker.hpp: kEps), so the string may end up to
// EPSILON off the tube centre and every sum along the last piece inherits that; the sequential walks of this library (rung 5, the repair
// kernels) do the same, bit for bit.  kCertifySlack = 4 EPSILON is allowed for it: a wrong sample below ~1e-9 is inside what the
// reference's own solvers disagree by among themselves.  A fibre that fails is flagged; the host re-solves the flagged fibres with the sequential walk
// (sweep_seq_kernel through its fibre gate) and counts them.  Fibres with a negative penalty are not checked (the reference's behaviour
// there is its code, not a minimisation).
constexpr double kCertifyTol = 64.0, kCertifyStep = 256.0, kCertifyUlp = 2.220446049250313e-16, kCertifySlack = 4.0 * kEps;

struct CertifyAcc {
    double u = 0.0, scale = 0.0, viol = 0.0;
    int where = -1, kind = 0;   // sample and test of the largest violation (0 the bound, 1 / 2 a step up / down off its wall, 3 the total)
    bool defined = true;
    __device__ __forceinline__ void note(double v, int k, int what) {
        if (v > viol) {
            viol = v;
            where = k;
            kind = what;
        }
    }
    // one sample: the running sum behind it, its x, the next sample's x (the last sample: anything), the penalty of the edge behind it
    __device__ __forceinline__ void edge(double uk, double x, double xn, double r, bool last, int k) {
        if (last) {
            note(fabs(uk), k, 3);
            return;
        }
        defined = defined && r >= 0.0;
        note(fabs(uk) - r, k, 0);
        // (a step counts as one above rounding AND above what the slack at the last sample does to the last piece's value: a knot whose jump
        //  is zero up to rounding -- late Dykstra / DR iterates are full of them -- next to a last piece that is 1e-10 / n off shows a step
        //  of either sign: seen on the GPU, PD2 at lambda 0.7, sample 517 of 520)
        const double dx = xn - x, step = kCertifyStep * kCertifyUlp * scale + kCertifySlack;
        if (dx > step)       note(fabs(uk + r), k, 1);
        else if (dx < -step) note(fabs(uk - r), k, 2);
    }
    __device__ __forceinline__ double tolerance(int len, double lam) const {
        return kCertifyTol * (double)len * kCertifyUlp * fmax(scale, fabs(lam)) + kCertifySlack;
    }
    __device__ __forceinline__ bool failed(int len, double lam) const { return defined && viol > tolerance(len, lam); }
};
// what the first few failing fibres of a launch looked like (option verbose prints them)
struct CertifyNote {
    long fibre;
    int where, kind;
    double viol, tol;
};
constexpr int kCertifyNotes = 8;
__device__ __forceinline__ void certify_flag(int *flags, unsigned *count, CertifyNote *notes, long j, const CertifyAcc &acc, int len, double lam) {
    flags[j] = 1;
    const unsigned slot = atomicAdd(count, 1u);
    if (notes && slot < (unsigned)kCertifyNotes) notes[slot] = CertifyNote{j, acc.where, acc.kind, acc.viol, acc.tolerance(len, lam)};
}

template <int OP>
__device__ __forceinline__ void certify_sample(const SweepArgs &p, long idx, double &y, double &x, double &scale) {
    double i0, i1;
    Op<OP>::fetch_in(p, idx, i0, i1);
    y = Op<OP>::y_of(p, i0, i1);
    scale = fmax(scale, fmax(fabs(i0), fabs(i1)));
    x = Op<OP>::recover(p, idx, y, scale);
    scale = fmax(scale, fabs(x));
}

// strided fibres: one lane per fibre, 64 adjacent fibres per wave (every access a 512-byte row)
template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void certify_strided_kernel(SweepArgs p, FibreGeom g, int *flags, unsigned *count, CertifyNote *notes) {
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    if (j >= g.count || g.len <= 0) return;
    if (p.gate && *p.gate == 0) return;
    long blk, off;
    divmod_nonneg(j, g.inc, blk, off);
    const long base = blk * g.inc * g.len + off, wbase = blk * g.inc * (g.len - 1) + off;
    CertifyAcc acc;
    double y, x;
    certify_sample<OP>(p, base, y, x, acc.scale);
    for (int k = 0; k < g.len; k++) {
        const bool last = k == g.len - 1;
        double yn = 0.0, xn = x;
        if (!last) certify_sample<OP>(p, base + (long)(k + 1) * g.inc, yn, xn, acc.scale);
        acc.u += y - x;
        const double r = last ? 0.0 : (WEIGHTED ? p.w[wbase + (long)k * g.inc] : p.lam);
        acc.edge(acc.u, x, xn, r, last, k);
        y = yn;
        x = xn;
    }
    if (acc.failed(g.len, WEIGHTED ? 0.0 : p.lam)) certify_flag(flags, count, notes, j, acc, g.len, WEIGHTED ? 0.0 : p.lam);
}

// contiguous fibres: one wave per fibre, 64 consecutive samples per trip, the running sum by a scan across the lanes
template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(256) void certify_along_kernel(SweepArgs p, FibreGeom g, int *flags, unsigned *count, CertifyNote *notes) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= g.count || g.len <= 0) return;
    if (p.gate && *p.gate == 0) return;
    const long base = j * g.len, wbase = j * (g.len - 1);
    CertifyAcc acc;
    double carry = 0.0;
    for (int k0 = 0; k0 < g.len; k0 += 64) {
        const int k = k0 + lane;
        const bool in = k < g.len, last = k == g.len - 1;
        double y = 0.0, x = 0.0, xn = 0.0, yd;
        if (in) certify_sample<OP>(p, base + k, y, x, acc.scale);
        // the next sample's x: the next lane's; the trip's last lane reads it itself
        xn = __shfl_down(x, 1);
        if (lane == 63 && in && !last) certify_sample<OP>(p, base + k + 1, yd, xn, acc.scale);
        // (every lane tests against the largest operand any lane has seen so far)
        for (int o = 32; o > 0; o >>= 1) acc.scale = fmax(acc.scale, __shfl_xor(acc.scale, o));
        double u = in ? y - x : 0.0;
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(u, o);
            if (lane >= o) u += t;
        }
        u += carry;
        carry = __shfl(u, 63);
        if (in) {
            const double r = last ? 0.0 : (WEIGHTED ? p.w[wbase + k] : p.lam);
            acc.edge(u, x, xn, r, last, k);
        }
    }
    bool bad = acc.failed(g.len, WEIGHTED ? 0.0 : p.lam);
    // (a lane that met a negative penalty takes the whole fibre out of the check)
    if (__ballot(!acc.defined) != 0ull) bad = false;
    const unsigned long long who = __ballot(bad);
    if (who != 0ull && lane == __ffsll((long long)who) - 1) certify_flag(flags, count, notes, j, acc, g.len, WEIGHTED ? 0.0 : p.lam);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
template <int OP, bool WEIGHTED>
void launch_seq(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool long_pieces, int *fibre_gate = nullptr) {
    const unsigned blocks = (unsigned)((g.count + 63) / 64);
    if (blocks == 0) return;
    if (long_pieces) hipLaunchKernelGGL((sweep_seq_kernel<OP, WEIGHTED, true>), dim3(blocks), dim3(64), 0, stream, args, g, fibre_gate);
    else             hipLaunchKernelGGL((sweep_seq_kernel<OP, WEIGHTED, false>), dim3(blocks), dim3(64), 0, stream, args, g, fibre_gate);
    count_event(CNT_SWEEP_LAUNCHES);
    PTV_HIP(hipGetLastError());
}

// persistent per-thread state of the chunked path: link codes, fail flags, repair counters and the geometry policy
struct ChunkScratch {
    std::unique_ptr<Scratch> links, flags;
    size_t link_bytes = 0, flag_count = 0;
    link_t *code_mine = nullptr, *code_next = nullptr;   // [chunk][fibre]
    int *failflags = nullptr;
    double sweep_seed_f = -1.0;             // certain fraction of the sweep being launched (-1: not sampled)
    std::unique_ptr<Scratch> handled_buf;   // option "repair_jobs": per fibre, the epoch of the launch whose failures sweep_repair_jobs_kernel repaired
    size_t handled_count = 0;
    unsigned *handled_for(size_t count, hipStream_t s) {
        if (count > handled_count) {
            handled_buf.reset(new Scratch(sizeof(unsigned) * count));
            handled_count = count;
            PTV_HIP(hipMemsetAsync(handled_buf->as<unsigned>(), 0, sizeof(unsigned) * count, s));
        }
        return handled_buf->as<unsigned>();
    }
    std::unique_ptr<Scratch> certify_buf;   // option "certify": [0] fibres that failed the last check, [1 ...] their flags (the sequential kernel's fibre gate)
    std::unique_ptr<Scratch> certify_notes; // ... and what the first few of them looked like (CertifyNote[kCertifyNotes])
    size_t certify_count = 0;
    int *certify_for(size_t count, hipStream_t s) {
        if (count > certify_count) {
            certify_buf.reset(new Scratch(sizeof(int) * (count + 1)));
            certify_count = count;
            PTV_HIP(hipMemsetAsync(certify_buf->as<int>(), 0, sizeof(int) * (count + 1), s));
        }
        return certify_buf->as<int>();
    }
    std::unique_ptr<Scratch> trace;   // option "trace": phase timestamps of the last chunk-kernel launch
    size_t trace_wgs = 0;
    unsigned long long *trace_buffer(size_t wgs) {
        if (!trace || trace->bytes() < wgs * 64) trace.reset(new Scratch(wgs * 64));
        trace_wgs = wgs;
        return trace->as<unsigned long long>();
    }
    int *failcount = nullptr;   // [family][2]: fibres that needed repair, chunks rewritten by repair walks (cumulative per solve)
    // one "dirty" word for all launches (each launch has its own epoch) and the cross-workgroup link words
    unsigned epoch = 0;
    std::unique_ptr<Scratch> dirty_word, xlink;
    size_t xlink_words = 0;
    DirtyMark next_dirty(hipStream_t s) {
        if (!options().xlink) return DirtyMark{nullptr, 0u, nullptr, nullptr};
        ensure_dirty(s);
        if (++epoch == 0u) ++epoch;   // (0 is what freshly allocated words hold)
        return DirtyMark{dirty_word->as<unsigned>(), epoch, options().why ? dirty_word->as<unsigned>() + 1 : nullptr,
                         optimistic ? dirty_word->as<unsigned>() + 9 : nullptr};
    }
    void ensure_dirty(hipStream_t s) {
        if (!dirty_word) {   // [0] the word, [1..8] option "why" counters, [9] the sticky word of an optimistic solve
            dirty_word.reset(new Scratch(sizeof(unsigned) * 10));
            PTV_HIP(hipMemsetAsync(dirty_word->as<unsigned>(), 0, sizeof(unsigned) * 10, s));
        }
    }
    // An optimistic solve: the chunked sweeps launch no repair kernels (an empty one still costs its dependent launch, 2.5-4.5 us behind
    // every sweep: 5 % of the headline solve, half of a 512^2 one); whatever a sweep leaves is recorded in the sticky word, read once at
    // the end of the solve.  Exactness rests on the run with the repairs that follows a solve whose word is set.
    bool optimistic = false;
    int optimistic_backoff = 0;   // eligible solves still to run WITH their repairs after one that had to be run again (sweep.hip)
    void begin_optimistic(hipStream_t s) {
        ensure_dirty(s);
        PTV_HIP(hipMemsetAsync(dirty_word->as<unsigned>() + 9, 0, sizeof(unsigned), s));
        optimistic = true;
    }
    // ends the optimistic stretch; true: every sweep was clean (synchronises the stream)
    bool end_optimistic(hipStream_t s) {
        optimistic = false;
        unsigned mark = 1;
        PTV_HIP(hipMemcpyAsync(&mark, dirty_word->as<unsigned>() + 9, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        PTV_HIP(hipStreamSynchronize(s));
        // (whoever marked the word also flagged chunks -- per fibre, in units of ITS sweep's geometry -- and no repair kernel came to
        //  take the flags back: the run that follows must not find them)
        if (mark != 0 && flags && flag_count > (size_t)kCounters)
            PTV_HIP(hipMemsetAsync(flags->as<int>() + kCounters, 0, sizeof(int) * (flag_count - (size_t)kCounters), s));
        return mark == 0;
    }
    unsigned long long *xlink_for(size_t words, hipStream_t s) {
        if (!options().xlink) return nullptr;
        if (words > xlink_words) {
            xlink.reset(new Scratch(sizeof(unsigned long long) * words));
            xlink_words = words;
            PTV_HIP(hipMemsetAsync(xlink->as<unsigned long long>(), 0, sizeof(unsigned long long) * words, s));
        }
        return xlink->as<unsigned long long>();
    }

    // Geometry policy (policy.hpp), one per sweep family -- fibres along dim 0 / along the other dims see different
    // data: in a DR solve at large lambda the column pieces are several times longer than the row pieces -- plus the
    // plumbing of its measurements: hipEvents around the measured launch, the repair counters read back behind it.
    struct Policy : GeometryPolicy {
        bool meas = false;       // a measurement is in flight
        int meas_mode = 0, meas_slot = -1;
        long meas_sweep = 0;
        hipEvent_t t0 = nullptr, t1 = nullptr;
        long chunks_done = 0;    // chunks processed (host-side count, cumulative per solve)
        long chunks_seen = 0;    // ... at the last evaluation
        int rewritten_seen = 0;
    } pol[FAM_COUNT];

    // Edge statistics of this solve's input, one record per swept dimension (policy_probe): the seed of the policy.
    struct Probe {
        long inc = 0, count = 0;
        int len = 0;
        bool weighted = false;
        unsigned hist[kProbeWords] = {};   // edges | stretches (pointwise.hpp)
    };
    static constexpr int kMaxProbes = 8;
    Probe probes[kMaxProbes];
    int nprobes = 0;
    std::unique_ptr<Scratch> probe_dev;   // kMaxProbes histograms
    const Probe *find_probe(const FibreGeom &g, bool weighted) const {
        for (int k = 0; k < nprobes; k++)
            if (probes[k].inc == g.inc && probes[k].len == g.len && probes[k].count == g.count && probes[k].weighted == weighted)
                return &probes[k];
        return nullptr;
    }
    // fraction of the sampled edges at which the string is known to bend at this penalty (-1: this sweep's input was not sampled)
    double certain_fraction(const FibreGeom &g, double lam, bool weighted) const {
        const Probe *p = find_probe(g, weighted);
        if (!p || p->hist[kProbeBins] == 0) return -1.0;
        if (!weighted && !(lam > 0.0)) return 1.0;
        // (edges in the threshold's own bin do not count: an edge of exactly 4 lambda -- a checkerboard of +-2 lambda -- is not a
        // bend known a priori, and the kernels' test is strict)
        const int b = probe_bin(weighted ? 4.0 : 4.0 * lam);
        unsigned long above = 0;
        for (int k = b + 1; k < kProbeBins; k++) above += p->hist[k];
        return (double)above / (double)p->hist[kProbeBins];
    }
    // fraction of the sampled 16-edge stretches whose total variation is below 2 lambda: stretches the string crosses (all
    // but) flat -- nothing there for a speculative walk to meet the true one at (-1: not sampled)
    double flat_fraction(const FibreGeom &g, double lam, bool weighted) const {
        const Probe *p = find_probe(g, weighted);
        const unsigned *h = p ? p->hist + kProbeBins + 1 : nullptr;
        if (!p || h[kProbeBins] == 0) return -1.0;
        if (!weighted && !(lam > 0.0)) return 0.0;
        const int b = probe_bin(weighted ? 2.0 : 2.0 * lam);
        unsigned long below = 0;
        for (int k = 0; k < b; k++) below += h[k];
        return (double)below / (double)h[kProbeBins];
    }
    // rung the statistics ask for at this penalty (-1: not sampled)
    int seed(const FibreGeom &g, double lam, bool weighted) const {
        const double f = certain_fraction(g, lam, weighted);
        if (f < 0.0) return -1;
        // Spatially uneven data (half an image flat, sparse spikes on a constant background): whatever the average says, the
        // quiet stretches have pieces far longer than any zone and every chunk in them would go to the repair kernel.
        if (flat_fraction(g, lam, weighted) > kSeedFlat) return 3;
        return rung_from_certain_fraction(f);
    }

    static constexpr int kSlots = 8, kCounters = 2 * FAM_COUNT;
    int *h_counts = nullptr;    // pinned: [slot][kCounters]
    hipEvent_t ev[kSlots] = {};
    bool pending[kSlots] = {};
    long pending_chunks[kSlots][FAM_COUNT] = {};
    int next_slot = 0;
    int latest_rewritten[FAM_COUNT] = {0, 0, 0};
    long latest_chunks[FAM_COUNT] = {0, 0, 0};

    void ensure_host() {
        if (h_counts) return;
        PTV_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_counts), sizeof(int) * kSlots * kCounters, hipHostMallocDefault));
        for (auto &e : ev) PTV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &pl : pol) {
            PTV_HIP(hipEventCreate(&pl.t0));
            PTV_HIP(hipEventCreate(&pl.t1));
        }
    }

    void ensure(long count, int NC, hipStream_t s) {
        const size_t need = sizeof(link_t) * (size_t)count * (size_t)NC * 2;
        if (need > link_bytes) {
            links.reset(new Scratch(need));
            link_bytes = need;
        }
        code_mine = links->as<link_t>();
        code_next = code_mine + (size_t)count * (size_t)NC;
        if (2 * (size_t)count + kCounters > flag_count) {
            poll(true);   // read-backs of the old counters must land before the buffer goes away
            flags.reset(new Scratch(sizeof(int) * (2 * (size_t)count + kCounters)));
            flag_count = 2 * (size_t)count + kCounters;
            PTV_HIP(hipMemsetAsync(flags->as<int>(), 0, sizeof(int) * flag_count, s));
            for (int f = 0; f < FAM_COUNT; f++) {
                pol[f].rewritten_seen = 0;
                latest_rewritten[f] = 0;
            }
        }
        failcount = flags->as<int>();
        failflags = failcount + kCounters;
        ensure_host();
    }

    // take in read-backs, oldest first: all of them (wait), or up to and including slot `until` (blocking), or those
    // that have completed
    void poll(bool wait, int until = -1) {
        for (int k = 0; k < kSlots; k++) {
            const int slot = (next_slot + k) % kSlots;
            if (!pending[slot]) continue;
            if (wait || until >= 0) PTV_HIP(hipEventSynchronize(ev[slot]));
            else if (hipEventQuery(ev[slot]) != hipSuccess) break;
            pending[slot] = false;
            for (int f = 0; f < FAM_COUNT; f++) {
                latest_rewritten[f] = h_counts[slot * kCounters + 2 * f + 1];
                latest_chunks[f] = pending_chunks[slot][f];
            }
            if (slot == until) break;
        }
    }

    int enqueue_readback(hipStream_t s) {
        const int slot = next_slot;
        if (pending[slot] || !failcount) return -1;   // ring full: skip this sample
        PTV_HIP(hipMemcpyAsync(h_counts + slot * kCounters, failcount, sizeof(int) * kCounters, hipMemcpyDeviceToHost, s));
        PTV_HIP(hipEventRecord(ev[slot], s));
        pending[slot] = true;
        for (int f = 0; f < FAM_COUNT; f++) pending_chunks[slot][f] = pol[f].chunks_done;
        next_slot = (slot + 1) % kSlots;
        return slot;
    }

    // the measurement in flight: sweep time in ms, and the fraction of the family's chunks that repair walks rewrote
    // since its last evaluation (-1: no counters, e.g. after a sequential sweep)
    void evaluate(int fam, double &t, double &f) {
        Policy &pl = pol[fam];
        PTV_HIP(hipEventSynchronize(pl.t1));
        float ms = 0.f;
        PTV_HIP(hipEventElapsedTime(&ms, pl.t0, pl.t1));
        t = ms;
        f = -1.0;
        if (pl.meas_slot >= 0) {
            poll(false, pl.meas_slot);
            const long d = latest_chunks[fam] - pl.chunks_seen;
            if (d > 0) {
                f = (double)(latest_rewritten[fam] - pl.rewritten_seen) / (double)d;
                pl.chunks_seen = latest_chunks[fam];
                pl.rewritten_seen = latest_rewritten[fam];
            }
        }
        pl.meas = false;
        pl.meas_slot = -1;
    }

    // hand the measurement in flight to the policy
    void settle(int fam, bool adaptive) {
        Policy &pl = pol[fam];
        const int r = pl.meas_mode;
        double t, f;
        evaluate(fam, t, f);
        if (!adaptive) return;
        if (options().verbose)
            fprintf(stderr, "[proxtv_amd] policy: family %d sweep %ld: mode %d %s took %.3f ms, rewrote %.5f of its chunks (incumbent %d: %.3f ms)\n",
                    fam, pl.sweeps, r, !pl.explore ? "(sample)" : pl.trial >= 0 ? "(trial)" : "(incumbent)", t, f, pl.mode, pl.t_mode);
        pl.measured(r, t, f);
    }
};
// per host thread and per device, like the stream and the pool (the one definition: sweep.hip)
ChunkScratch &chunk_state();

// option "repair_jobs": the failures across workgroups first, one lane each (sweep_repair_jobs_kernel); returns the per-fibre marks the
// sequential kernel behind it skips by (nullptr: not run)
template <int OP, bool WEIGHTED>
const unsigned *launch_repair_jobs(const SweepArgs &args, const FibreGeom &g, int C, int H, int chunks_per_wg, int *failcount, long cstride,
                                   long fstride, const DirtyMark &dirty, hipStream_t stream) {
    const int NC = (g.len + C - 1) / C;
    if (!options().repair_jobs || !dirty.word || H > kWarmLong || (NC + chunks_per_wg - 1) / chunks_per_wg > 64) return nullptr;
    // (1: only where links fail in numbers -- a launch that finds the sweep clean still costs its 2 us ; 2: always.  An UNSAMPLED input
    //  -- sweep_seed_f = -1: a pinned rung, a problem too small to sample -- counts as "in numbers": nothing says the sweep is clean,
    //  and the pinned-rung legs of the test suite run the jobs kernel through this door)
    if (options().repair_jobs == 1 && !(chunk_state().sweep_seed_f < kSeedJobs)) return nullptr;
    constexpr size_t lds = sizeof(double) * ((1 + (WEIGHTED ? 1 : 0)) * kJobWindow * 64 + kJobWindow + 2);   // (+ the table of -DPTV_JOB_TABDIV)
    auto kern = sweep_repair_jobs_kernel<OP, WEIGHTED>;
    {
        static thread_local bool attr_done[kMaxDevices] = {};
        bool &attr_set = attr_done[current_device()];
        if (!attr_set) {
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
    }
    unsigned *handled = chunk_state().handled_for((size_t)g.count, stream);
    constexpr int per_block = 64 / kJobsPerFibre;
    hipLaunchKernelGGL(kern, dim3((unsigned)((g.count + per_block - 1) / per_block)), dim3(64), lds, stream, args, g, C, H, chunks_per_wg,
                       chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags, failcount, cstride, fstride, dirty, handled);
    count_event(CNT_REPAIR_JOBS_LAUNCHES);
    return handled;
}

// Chunk geometry: C = 16 samples per chunk, 8 waves (chunks) per block of 128 samples.  LDS per workgroup = one window
// of H + 128 + 8 rows x 512 B: ~77 KiB for H = 16 -> two workgroups = 16 waves per CU; ~101 KiB for H = 64 -> one.
// Weighted sweeps carry a second (penalty) window and exist for H = 16 only.
// (Tried for the robust instantiation and not kept: chunks of 14 samples with the 16 rows that saves spent on look-ahead --
// H 16 / 8 x 14 / T 24, the same 152 rows -- so that the walks of a block's last chunk close their last piece inside the
// window instead of reading on from global memory: on DR iterates at lambda = 0.5 on unit noise 2 % of them need more than
// 8 rows, 0.02 % more than 16.  The row sweep got 13 % slower, 180 -> 203 us: 37 blocks per fibre instead of 32 cost more
// than the reads past the window did.  What makes these sweeps slow is the walk itself -- at lambda = 0.5 the linearized
// taut string re-walks every piece about once: 2.6 x the trips of the headline.)
template <int OP, bool WEIGHTED, bool TRANSPOSED, int H, bool ROBUST = false, int C = 16, int NW = 8,
          int T = tail_rows(H), bool SHORT = false, int FW = 64>
void launch_chunk_h(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam, int rounds_wanted) {
    constexpr int PITCH = TRANSPOSED ? 65 : FW;
    constexpr int NCH = NW * (64 / FW);   // chunks per block
    constexpr int ROWS = SHORT ? NCH * C : H + NCH * C + T;
    ChunkPlan plan{};
    plan.Q = (g.len + NCH * C - 1) / (NCH * C);
    // blocks per workgroup: enough workgroups to fill the chip a few times over
    const long groups = (g.count + FW - 1) / FW;
    int qpw = 8;
    {
        // (weighted strided sweeps run ONE workgroup per CU -- two LDS planes: as many blocks per workgroup as still gives
        // every CU one; 14.5 -> 14.05 ms on the 4096^2 weighted solve.  Keeping the next block's window share in registers
        // while the current one is processed -- the waves own 256 VGPRs there -- was tried and hid the staging phase, but the
        // sweep did not get faster: at 8 waves per CU it is the walk's dependent-instruction latency that bounds it)
        const long want = (WEIGHTED && !TRANSPOSED && !SHORT) ? (FW < 64 ? 512 : 256) : (FW < 64 ? 4096 : 2048);
        while (qpw > 1 && groups * ((plan.Q + qpw - 1) / qpw) < want) qpw >>= 1;
    }
    plan.qpw = qpw < plan.Q ? qpw : plan.Q;
    plan.ablate = options().ablate;
    plan.rounds = rounds_wanted;
    plan.trace = options().trace ? chunk_state().trace_buffer((size_t)groups * (size_t)((plan.Q + plan.qpw - 1) / plan.qpw)) : nullptr;
    const int WQ = (plan.Q + plan.qpw - 1) / plan.qpw;
    plan.dirty = chunk_state().next_dirty(stream);
    plan.xlink = chunk_state().xlink_for((size_t)WQ * (size_t)g.count, stream);
#ifndef PTV_NO_WALK_TABLE   // (the switch stays for A/B builds: the walk then divides with v_rcp_f64 + Newton + residual)
    constexpr size_t tab_bytes = ((WEIGHTED || !TRANSPOSED) && !SHORT && H <= kWarm && NW <= 8) ? sizeof(double) * (ROBUST ? kRecipTableRobust : kRecipTable) : 0;
#else
    constexpr size_t tab_bytes = 0;
#endif
    constexpr size_t lds = sizeof(double) * PITCH * (size_t)ROWS * (WEIGHTED ? 2 : 1) + sizeof(link_t) * ((NCH + 2) * FW + (TRANSPOSED ? 0 : FW) + 4 + 2 * NW) + tab_bytes;
    static_assert(WEIGHTED || H > kWarm || NW > 8 || 2 * lds <= 160 * 1024, "the short-zone geometry is meant to run two workgroups per CU");
    static_assert(FW == 64 || (WEIGHTED ? 2 : 4) * lds <= 160 * 1024, "the 32-fibre tile is meant to run four workgroups per CU (weighted: two)");
    if (SHORT && g.len > NCH * C) {
        set_error("launch_chunk_h: a fibre of %d samples does not fit the single-block geometry (%d)", g.len, NCH * C);
        throw HipFailure{hipErrorInvalidValue};
    }
    static_assert(lds <= 160 * 1024, "chunk geometry does not fit the LDS of a CU");
    const int NC = (g.len + C - 1) / C;
    chunk_state().ensure(g.count, NC, stream);
    // the second-chance rounds are a separate instantiation: their live state costs the plain kernel registers it
    // does not have (it sits at the 128-VGPR budget of two workgroups per CU)
    static_assert(!ROBUST || H <= kWarm, "second chances exist for the short-zone geometry");
    if (!ROBUST) plan.rounds = 0;
    auto kern = sweep_chunk_kernel<OP, WEIGHTED, TRANSPOSED, C, NW, H, ROBUST, T, SHORT, FW>;
    static thread_local bool attr_done[kMaxDevices] = {};   // function attributes are per device
    bool &attr_set = attr_done[current_device()];
    if (!attr_set) {
        PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr_set = true;
        if (options().verbose) {
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 64 * NW, lds);
            fprintf(stderr, "[proxtv_amd] tile kernel op %d%s%s: %d fibres x %d chunks of %d in %d waves, %zu B of LDS -> %d workgroups per CU\n", OP,
                    WEIGHTED ? " weighted" : "", ROBUST ? " robust" : "", FW, NCH, C, NW, lds, per_cu);
        }
    }
    const dim3 grid((unsigned)groups, (unsigned)WQ);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, args, g, plan, chunk_state().code_mine,
                       chunk_state().code_next, chunk_state().failflags);
    count_event(CNT_SWEEP_LAUNCHES);
    if (!plan.ablate && !plan.dirty.sticky) {   // (an optimistic solve: the sweep has marked the sticky word if it left anything)
        constexpr size_t rlds = sizeof(double) * (2 + (WEIGHTED ? 1 : 0)) * kRepairWindow * 64;
        auto rkern = sweep_repair_kernel<OP, WEIGHTED>;
        static thread_local bool rattr_done[kMaxDevices] = {};
        bool &rattr_set = rattr_done[current_device()];
        if (!rattr_set) {
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rkern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)rlds));
            rattr_set = true;
        }
        const unsigned *handled = launch_repair_jobs<OP, WEIGHTED>(args, g, C, H, plan.qpw * NCH, chunk_state().failcount + 2 * fam, (long)g.count, 1L, plan.dirty, stream);
        hipLaunchKernelGGL(rkern, dim3((unsigned)((g.count + 63) / 64)), dim3(64), rlds, stream, args, g, C, H, plan.qpw * NCH,
                           chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags, chunk_state().failcount + 2 * fam, (long)g.count, 1L,
                           plan.dirty, handled);
        count_event(CNT_REPAIR_LAUNCHES);
    }
    PTV_HIP(hipGetLastError());
    chunk_state().pol[fam].chunks_done += (long)NC * g.count;
}

// Chunks along the fibre (kernel 2a): dimension-0 sweeps, unweighted.  Codes are laid out [fibre][chunk] (a wave writes
// the codes of 64 consecutive chunks of one fibre).
template <int OP, bool WEIGHTED, int H, int G, bool ROBUST, bool ONESEG = false, bool RUNS = false>
void launch_along_g(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam, int rounds_wanted) {
    static_assert(!RUNS || (!WEIGHTED && !ROBUST && !ONESEG && G == 64 && H == kWarm), "known runs: the plain unweighted 64-lane instantiation");
    constexpr int C = along_chunk(ROBUST, WEIGHTED), SEG = G * C, ROWS = ONESEG ? SEG : along_zone_rows(H, ROBUST) + SEG + along_tail_rows(H, ROBUST), NG = 64 / G;
    const int nseg = (g.len + SEG - 1) / SEG;
    if (ONESEG && nseg != 1) {
        set_error("launch_along_g: a fibre of %d samples is more than one segment (%d)", g.len, SEG);
        throw HipFailure{hipErrorInvalidValue};
    }
    const int NC = (g.len + C - 1) / C;
    const long units = g.count * nseg;
    const long waves = (units + NG - 1) / NG;
    ChunkPlan plan{};
    plan.ablate = options().ablate;
    plan.rounds = ROBUST ? rounds_wanted : 0;
    chunk_state().ensure(g.count, NC, stream);
    plan.trace = options().trace ? chunk_state().trace_buffer((size_t)waves) : nullptr;
    plan.dirty = chunk_state().next_dirty(stream);
    plan.xlink = chunk_state().xlink_for((size_t)g.count * (size_t)nseg, stream);
    plan.legacy = options().debug_legacy_rebuild;
    constexpr size_t lds = sizeof(double) * (size_t)(ROWS + 2) * NG * kAlongWaves * (WEIGHTED ? 2 : 1) + (ROBUST ? 64 + sizeof(double) * kRecipTableRobust : sizeof(double) * kRecipTable) +
                           (RUNS ? sizeof(unsigned) * kRunsWords * kAlongWaves : 0);
    static_assert(lds <= 160 * 1024, "along-fibre geometry does not fit the LDS of a CU");
    static_assert(!RUNS || (16 / kAlongWaves) * lds <= 160 * 1024, "known runs: still sixteen waves per CU");
    auto kern = sweep_along_kernel<OP, WEIGHTED, H, G, ROBUST, ONESEG, RUNS>;
    if (lds > 64 * 1024) {   // above the default dynamic-LDS limit
        static thread_local bool attr_done[kMaxDevices] = {};
        bool &attr_set = attr_done[current_device()];
        if (!attr_set) {
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
    }
    // (One workgroup per kAlongWaves segments, dispatched as slots free up.  Tried in round 5 and dropped, profiles/NOTES_r05.md: as many
    // workgroups as the device holds, each taking its segments in static turns -- 76 -> 84 us, nothing rebalances the slow workgroups --
    // or drawing them from atomic counters -- the wave slots stay 98 % full instead of 70 % and the sweep takes as long: the vector
    // pipes, not the dispatcher, are what the waves wait for.)
    hipLaunchKernelGGL(kern, dim3((unsigned)((waves + kAlongWaves - 1) / kAlongWaves)), dim3(64 * kAlongWaves), lds, stream, args, g,
                       plan, chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags);
    count_event(CNT_SWEEP_LAUNCHES);
    if (!plan.ablate && !plan.dirty.sticky) {   // (an optimistic solve: the sweep has marked the sticky word if it left anything)
        constexpr size_t rlds = sizeof(double) * (2 + (WEIGHTED ? 1 : 0)) * kRepairWindow * 64;
        auto rkern = sweep_repair_kernel<OP, WEIGHTED>;
        static thread_local bool rattr_done[kMaxDevices] = {};
        bool &rattr_set = rattr_done[current_device()];
        if (!rattr_set) {
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rkern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
            rattr_set = true;
        }
        const unsigned *handled = launch_repair_jobs<OP, WEIGHTED>(args, g, C, H, G, chunk_state().failcount + 2 * fam, 1L, (long)NC, plan.dirty, stream);
        hipLaunchKernelGGL(rkern, dim3((unsigned)((g.count + 63) / 64)), dim3(64), rlds, stream, args, g, C, H, G, chunk_state().code_mine,
                           chunk_state().code_next, chunk_state().failflags, chunk_state().failcount + 2 * fam, 1L, (long)NC, plan.dirty, handled);
        count_event(CNT_REPAIR_LAUNCHES);
    }
    PTV_HIP(hipGetLastError());
    chunk_state().pol[fam].chunks_done += (long)NC * g.count;
}

// Chunks along the fibre (kernel 2a): dimension-0 sweeps, unweighted.  Codes are laid out [fibre][chunk] (a group writes
// the codes of consecutive chunks of one fibre).  Lanes per segment: a whole wave for long fibres; half or a quarter of
// one when the fibre fits 32 or 16 chunks.
template <int OP, bool WEIGHTED, int H, bool ROBUST, bool RUNS = false>
void launch_along(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam, int rounds) {
    constexpr int C = along_chunk(ROBUST, WEIGHTED);
    if constexpr (RUNS) {   // (fibres of more than a segment: the others have no interior segment to solve run by run)
        if (g.len > 64 * C) {
            launch_along_g<OP, WEIGHTED, H, 64, ROBUST, false, true>(args, g, stream, fam, rounds);
            return;
        }
    }
    // (fibres of one segment: the robust instantiation without its look-back / look-ahead rows)
    if (g.len <= 16 * C)      launch_along_g<OP, WEIGHTED, H, 16, ROBUST, true>(args, g, stream, fam, rounds);
    else if (g.len <= 32 * C) launch_along_g<OP, WEIGHTED, H, 32, ROBUST, true>(args, g, stream, fam, rounds);
    else if (ROBUST && H <= kWarm && g.len <= 64 * C) launch_along_g<OP, WEIGHTED, H, 64, ROBUST, ROBUST && H <= kWarm>(args, g, stream, fam, rounds);
    else                            launch_along_g<OP, WEIGHTED, H, 64, ROBUST>(args, g, stream, fam, rounds);
}

// Global-memory chunks (kernel 2b): chunk C and zone H are run-time values; every link is checked by the repair kernel.
template <int OP, bool WEIGHTED>
void launch_gchunk(const SweepArgs &args, const FibreGeom &g, int C, int H, hipStream_t stream, int fam) {
    const long groups = (g.count + 63) / 64;
    const int NC = (g.len + C - 1) / C;
    chunk_state().ensure(g.count, NC, stream);
    hipLaunchKernelGGL((sweep_gchunk_kernel<OP, WEIGHTED>), dim3((unsigned)groups, (unsigned)NC), dim3(64), 0, stream, args,
                       g, C, H, chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags);
    hipLaunchKernelGGL((sweep_repair_kernel<OP, WEIGHTED>), dim3((unsigned)groups), dim3(64), 0, stream, args, g, C, H, 1,
                       chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags, chunk_state().failcount + 2 * fam, (long)g.count, 1L,
                       DirtyMark{nullptr, 0u, nullptr, nullptr});
    count_event(CNT_SWEEP_LAUNCHES);
    count_event(CNT_REPAIR_LAUNCHES);
    PTV_HIP(hipGetLastError());
    chunk_state().pol[fam].chunks_done += (long)NC * g.count;
}

// ---- strided sweeps through the along-fibre kernel: transpose, sweep, transpose back ---------------------------------------
// When the walks of a strided sweep need long zones (pieces of ~10 samples and more: the row family's modes 1 and 2) the
// 64-fibre tile pays for them in LDS -- one workgroup per CU, every zone staged again -- while the along-fibre kernel
// gets them for free: a lane's zone is its neighbours' chunks.  So the operands are transposed (fibres become
// contiguous; a tiled copy at HBM speed), the sweep runs as a dimension-0 sweep, and the outputs are transposed back.
// Fibre numbering is unchanged: fibre j = slab * inc + off sits at j * len after the transposition of every
// (inc x len) slab.

template <int OP, int H>
void launch_row_along(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam, int rounds) {
    TransposedOperands tr(args, Op<OP>::IN_MASK, Op<OP>::OUT_MASK, g, stream);
    launch_along<OP, false, H, true>(tr.args(), tr.geom(), stream, fam, 2 * rounds);
    tr.finish();
}

template <int OP, bool WEIGHTED, bool TRANSPOSED>
void launch_chunk(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam) {
    // The aliasing contract of sweep.hpp, enforced: a chunked sweep stages windows of its input operand(s) while other workgroups
    // write outputs, and an op with KEEP finishes rows with the operand value it captured at staging time -- an output array that
    // IS a staged operand would be read half-written / finished with stale values.  (Epilogue-only operands may alias an output
    // element for element: the thread that writes the element is the one that read it.)  The solvers ping-pong; this is the check.
    {
        const void *staged[2] = {args.a, Op<OP>::NIN > 1 ? args.b : nullptr};
        for (const void *in : staged)
            if (in && (in == args.o0 || in == args.o1)) {
                set_error("launch_sweep: op %d writes an array its chunked sweep stages as fibre samples (outputs must not alias window operands)", OP);
                throw HipFailure{hipErrorInvalidValue};
            }
    }
    ChunkScratch &st = chunk_state();
    ChunkScratch::Policy &pl = st.pol[fam];
    const bool pinned = options().chunk_mode >= 0;
    const bool pin_ok = options().pin && pin_supports((OpId)OP, WEIGHTED, g, args.lam);
    const int seed = st.seed(g, args.lam, WEIGHTED);
    const double seed_f = st.certain_fraction(g, args.lam, WEIGHTED);
    st.sweep_seed_f = seed_f;
    if (pl.workload(g.len, g.count, WEIGHTED, pin_ok, seed) && pl.meas) {   // a new workload: the measurement in flight is of the old one
        double t, f;
        st.evaluate(fam, t, f);
    }
    int mode;
    bool measure = false;
    if (pinned) {
        mode = pl.mode = options().chunk_mode < kModeSeq ? options().chunk_mode : kModeSeq;
        if (!pl.available(mode, true)) mode = pl.up(mode);
    } else if (options().deterministic) {
        // the rung is a function of the sweep's (sampled) input and penalty alone; unsampled inputs (tiny problems) take the
        // rung whose cost and exactness do not depend on the data
        mode = seed >= 0 ? seed : (pin_ok ? 3 : 0);
        if (mode == 1 && WEIGHTED && !pl.available(1, true)) mode = pl.up(mode);
        if (options().verbose && (pl.sweeps == 0 || mode != pl.mode))
            fprintf(stderr, "[proxtv_amd] policy: family %d (len %d x %ld fibres, lambda %g): seed %d -> mode %d\n", fam, g.len, g.count,
                    args.lam, seed, mode);
        pl.mode = mode;
    } else {
        st.ensure_host();
        if (pl.meas && (pl.explore || pl.sweeps - pl.meas_sweep >= kMonitorLag)) st.settle(fam, true);
        mode = pl.choose();
        measure = pl.wants_measurement(pl.meas);
        if (measure) PTV_HIP(hipEventRecord(pl.t0, stream));
    }
    const int rounds = (mode == 1 || mode == 2) ? kRounds : 0;
    // Geometry ladder.  Dimension 0 (chunks along the fibre once a fibre fills most of a lane group): 0 = 16-sample zones,
    // 1 / 2 = 64-sample zones, 3 = the pinning solver (pin.hip; where it does not apply: chunks from global memory, zone
    // 256), 4 = chunks from global memory (zone 1024), 5 = one sequential walk per fibre.
    // Strided sweeps: 0 / 1 = the 64-fibre tile (1: robust instantiation), 2 = transposed copies + the along-fibre kernel
    // with 64-sample zones (or the tile with 64-sample zones), 3 / 4 / 5 as above.
    const bool along_ok = options().along && g.len >= kAlongMinLen;
    bool pinned_done = false;
    if (mode == 3 && pin_ok) {
        int *pieces = nullptr;
        if (measure) {   // the policy's hint from this rung: pieces per sample (numerator and denominator of evaluate())
            st.ensure(g.count, 1, stream);
            pieces = st.failcount + 2 * fam + 1;
        }
        // false: the grid-wide variant wrote nothing (its instantiation does not fit this device at once after all, or it
        // hit the level cap on periodic data) -- the global-memory chunks below take the sweep
        // (knots known a priori: none to be had where the sampled input shows no edge above 4 lambda -- lambda = 3 on unit noise: the
        // search costs 3-6 % of such a sweep; unsampled inputs search)
        const bool seeds = options().pin_seed && (seed_f < 0.0 || seed_f >= kSeedPins);
        pinned_done = launch_pin((OpId)OP, WEIGHTED, args, g, stream, pieces, seeds);
        count_event(pinned_done ? CNT_PIN_SWEEPS : CNT_PIN_CAP_NEXT_RUNG);
        if (pinned_done) count_event(CNT_SWEEP_LAUNCHES);
        if (pinned_done && measure) pl.chunks_done += (long)g.len * g.count;
    }
    if (pinned_done) {}
    else if (mode >= kModeSeq)  launch_seq<OP, WEIGHTED>(args, g, stream, true);
    else if (mode == 3)    launch_gchunk<OP, WEIGHTED>(args, g, 64, 256, stream, fam);
    else if (mode == 4)    launch_gchunk<OP, WEIGHTED>(args, g, 256, 1024, stream, fam);
    else if (TRANSPOSED && along_ok) {
        // chunks along the fibre: 0 = 16-sample zones ; 1 = the same, robust (second chances inside the wave and across the
        // waves of a workgroup, walks past the look-ahead rows) ; 2 = 64-sample zones, robust
        // (second chances cost the along-fibre kernel a wave's re-walk, and only in waves that need one: twice the rounds of the tile)
        // (rung 0 on data most of whose edges are bends known a priori: interior segments are cut there and solved run by run)
        if (mode == 0 && !WEIGHTED && options().runs && seed_f >= kSeedRuns) {
            if constexpr (!WEIGHTED) launch_along<OP, false, kWarm, false, true>(args, g, stream, fam, 0);
        }
        else if (mode == 0) launch_along<OP, WEIGHTED, kWarm, false>(args, g, stream, fam, 0);
        else if (mode == 1) launch_along<OP, WEIGHTED, kWarm, true>(args, g, stream, fam, 2 * rounds);
        else                launch_along<OP, WEIGHTED, kWarmLong, true>(args, g, stream, fam, 2 * rounds);
    }
    else if (!TRANSPOSED && !WEIGHTED && along_ok && mode == 2) {
        if constexpr (!WEIGHTED) launch_row_along<OP, kWarmLong>(args, g, stream, fam, rounds);
    }
    else if constexpr (!WEIGHTED) {
        if (mode == 2)      launch_chunk_h<OP, false, TRANSPOSED, kWarmLong>(args, g, stream, fam, 0);
        else if (mode == 1 && !TRANSPOSED && options().tile == 1) launch_chunk_h<OP, false, false, kWarm, true, 16, 4, kTail, false, 32>(args, g, stream, fam, rounds);
        else if (mode == 1) launch_chunk_h<OP, false, TRANSPOSED, kWarm, true>(args, g, stream, fam, rounds);
        else if (!TRANSPOSED && options().tile == 1) launch_chunk_h<OP, false, false, kWarm, false, 16, 4, kTail, false, 32>(args, g, stream, fam, 0);
        else                launch_chunk_h<OP, false, TRANSPOSED, kWarm>(args, g, stream, fam, 0);
    } else {
        if (mode == 1 && !TRANSPOSED && options().tile == 1) launch_chunk_h<OP, true, false, kWarm, true, 16, 4, kTail, false, 32>(args, g, stream, fam, rounds);
        else if (mode == 1) launch_chunk_h<OP, true, TRANSPOSED, kWarm, true>(args, g, stream, fam, rounds);
        else if (!TRANSPOSED && options().tile == 1) launch_chunk_h<OP, true, false, kWarm, false, 16, 4, kTail, false, 32>(args, g, stream, fam, 0);
        else           launch_chunk_h<OP, true, TRANSPOSED, kWarm>(args, g, stream, fam, 0);
    }
    pl.sweeps++;
    if (measure) {
        PTV_HIP(hipEventRecord(pl.t1, stream));
        pl.meas = true;
        pl.meas_mode = mode;
        pl.meas_sweep = pl.sweeps;
        pl.meas_slot = (mode < kModeSeq) ? st.enqueue_readback(stream) : -1;
    }
}

template <int OP, bool WEIGHTED>
void launch_op_w(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool allow_chunked, int fam) {
    // chunking pays once a fibre spans several blocks; short fibres stay sequential
    const bool chunked = allow_chunked && g.len >= options().chunk_min_len;
    if (!chunked && !WEIGHTED && options().whole && g.len >= 16 && g.len <= kWholeMax && args.lam >= 0.0) {   // (negative
        // penalties -- tvgen lets them through -- keep the sequential kernel, whose reads past the fibre mirror the reference's;
        // so do fibres of a handful of samples: that kernel divides like the CPU, bit for bit, and loops that end at a
        // bitwise fixed point -- Kolmogorov2_TV -- count their iterations on the last bit)
        // Three kernels for short fibres.  Measured on 512 x 512 x L volumes, unit noise (tools/short_probe.py): up to 32
        // samples the sequential walk wins (0.084 ms per sweep at L = 32 against 0.093 / 0.120); beyond, with pieces of a few
        // samples (the policy's seed says rung 0), ONE block of the chunk kernel -- 4 or 6 chunks of 16 samples walk in
        // parallel -- beats the whole-fibre kernel, which walks 64-96 samples in sequence at five waves per CU (L = 64:
        // 0.136 against 0.206 ms); with longer pieces the whole-fibre kernel, which has no links to lose (0.300 against 0.356).
        if (options().whole == 1 && g.len <= 32) {
            launch_seq<OP, WEIGHTED>(args, g, stream, false);
            return;
        }
        if (options().whole == 1 && chunk_state().seed(g, args.lam, false) == 0) {
            chunk_state().sweep_seed_f = chunk_state().certain_fraction(g, args.lam, false);   // (this sweep's, not the last chunked sweep's: the jobs gate reads it)
            if (g.len <= 64) {
                if (g.inc == 1) launch_chunk_h<OP, false, true, kWarm, false, 16, 4, kTail, true>(args, g, stream, fam, 0);
                else            launch_chunk_h<OP, false, false, kWarm, false, 16, 4, kTail, true>(args, g, stream, fam, 0);
            } else {
                if (g.inc == 1) launch_chunk_h<OP, false, true, kWarm, false, 16, 6, kTail, true>(args, g, stream, fam, 0);
                else            launch_chunk_h<OP, false, false, kWarm, false, 16, 6, kTail, true>(args, g, stream, fam, 0);
            }
            return;
        }
        // short fibres whole in LDS, one lane per fibre (kernel 1b)
        const unsigned blocks = (unsigned)((g.count + 63) / 64);
        if (g.inc == 1) {
            hipLaunchKernelGGL((sweep_whole_kernel<OP, true>), dim3(blocks), dim3(64), sizeof(double) * 65 * (size_t)g.len, stream, args, g);
        } else {
            hipLaunchKernelGGL((sweep_whole_kernel<OP, false>), dim3(blocks), dim3(64), sizeof(double) * 64 * (size_t)g.len, stream, args, g);
        }
        count_event(CNT_SWEEP_LAUNCHES);
        PTV_HIP(hipGetLastError());
    }
    else if (!chunked) launch_seq<OP, WEIGHTED>(args, g, stream, false);
    else if (g.inc == 1) launch_chunk<OP, WEIGHTED, true>(args, g, stream, fam);
    else launch_chunk<OP, WEIGHTED, false>(args, g, stream, fam);
}

// option certify: check what the sweep just wrote (kernel 4), re-solve the fibres that fail, count them.  One small read-back per
// sweep: a validation mode, not a fast path.
// (returns the number of fibres that failed -- their flags are set in `*flags_out` --, or -1 when the sweep cannot be checked)
template <int OP, bool WEIGHTED>
long certify_count(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int **flags_out) {
    if (g.count <= 0 || g.len <= 0) return 0;
    // what cannot be checked: no minimisation behind the sweep (lambda <= 0 -- the identity, or the reference's code as it stands),
    // or an output that IS an operand (in-place sweeps of the sequential kernels: the inputs are gone)
    const void *ins[3] = {(Op<OP>::IN_MASK & 1u) ? args.a : nullptr, (Op<OP>::IN_MASK & 2u) ? args.b : nullptr,
                          (Op<OP>::IN_MASK & 4u) ? args.c : nullptr};
    bool aliased = false;
    for (const void *in : ins) aliased = aliased || (in && (in == args.o0 || in == args.o1));
    if (aliased || (!WEIGHTED && !(args.lam > 0.0))) {
        count_event(CNT_CERTIFY_SKIPPED);
        return -1;
    }
    int *buf = chunk_state().certify_for((size_t)g.count, stream);
    unsigned *count = reinterpret_cast<unsigned *>(buf);
    int *flags = buf + 1;
    if (!chunk_state().certify_notes) chunk_state().certify_notes.reset(new Scratch(sizeof(CertifyNote) * kCertifyNotes));
    CertifyNote *notes = chunk_state().certify_notes->as<CertifyNote>();
    if (g.inc == 1 && g.len >= 64)
        hipLaunchKernelGGL((certify_along_kernel<OP, WEIGHTED>), dim3((unsigned)((g.count + 3) / 4)), dim3(256), 0, stream, args, g, flags, count, notes);
    else
        hipLaunchKernelGGL((certify_strided_kernel<OP, WEIGHTED>), dim3((unsigned)((g.count + 63) / 64)), dim3(64), 0, stream, args, g, flags, count, notes);
    PTV_HIP(hipGetLastError());
    unsigned failed = 0;
    PTV_HIP(hipMemcpyAsync(&failed, count, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    PTV_HIP(hipStreamSynchronize(stream));
    count_event(CNT_CERTIFY_SWEEPS);
    if (failed) PTV_HIP(hipMemsetAsync(count, 0, sizeof(unsigned), stream));
    if (failed && options().verbose) {
        CertifyNote h[kCertifyNotes];
        PTV_HIP(hipMemcpyAsync(h, notes, sizeof(h), hipMemcpyDeviceToHost, stream));
        PTV_HIP(hipStreamSynchronize(stream));
        static const char *const kinds[4] = {"|u| above the penalty", "a step up off the floor wall", "a step down off the ceiling wall", "the total"};
        for (unsigned k = 0; k < failed && k < (unsigned)kCertifyNotes; k++)
            fprintf(stderr, "[proxtv_amd] certify: op %d%s, fibre %ld of %ld (%d samples, stride %ld, lambda %g): %s at sample %d: %.3e against a tolerance of %.3e\n",
                    OP, WEIGHTED ? " weighted" : "", h[k].fibre, g.count, g.len, g.inc, args.lam, kinds[h[k].kind & 3], h[k].where, h[k].viol, h[k].tol);
    }
    if (flags_out) *flags_out = flags;
    return (long)failed;
}

template <int OP, bool WEIGHTED>
void launch_certify(const SweepArgs &args, const FibreGeom &g, hipStream_t stream) {
    int *flags = nullptr;
    const long failed = certify_count<OP, WEIGHTED>(args, g, stream, &flags);
    if (failed > 0) {
        count_event(CNT_CERTIFY_FAILURES, failed);
        if (options().verbose)
            fprintf(stderr, "[proxtv_amd] certify: op %d, %ld fibres of %d samples (stride %ld): %ld failed the optimality conditions -- re-solved sequentially\n",
                    OP, g.count, g.len, g.inc, failed);
        launch_seq<OP, WEIGHTED>(args, g, stream, true, flags);   // (walks the flagged fibres only, and clears their flags)
    }
}

}  // namespace swp

// One translation unit per (op, weighted) pair defines these two (sweep_unit.hip); sweep.hip dispatches to them.
namespace swp {
template <int OP, bool WEIGHTED> void unit_launch(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool allow_chunked, int fam);
template <int OP, bool WEIGHTED> void unit_gated(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *flags);
template <int OP, bool WEIGHTED> void unit_warm();
template <int OP, bool WEIGHTED> long unit_certify(const SweepArgs &args, const FibreGeom &g, hipStream_t stream);
// (op, weighted) pairs that exist: X(op, weighted)
#define PTV_SWEEP_UNITS(X)                                                                                                   \
    X(OP_PROX, false) X(OP_PROX, true) X(OP_DR_COL, false) X(OP_DR_COL, true) X(OP_DR_COL_FINAL, false) X(OP_DR_COL_FINAL, true) \
    X(OP_DR_ROW, false) X(OP_DR_ROW, true) X(OP_DR_ROW_FINAL, false) X(OP_DRW_ROW_FINAL, true) X(OP_PD2_A, false)            \
    X(OP_PD2_B, false) X(OP_YANG, false) X(OP_DR_COL_V, false) X(OP_DR_COL_V, true) X(OP_DR_ROW_V, false) X(OP_DR_ROW_V, true)
#define PTV_DECLARE_UNIT(ID, W)                                                                                                         \
    template <> void unit_launch<ID, W>(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool allow_chunked, int fam); \
    template <> void unit_gated<ID, W>(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *flags);           \
    template <> void unit_warm<ID, W>();                                                                                     \
    template <> long unit_certify<ID, W>(const SweepArgs &args, const FibreGeom &g, hipStream_t stream);
PTV_SWEEP_UNITS(PTV_DECLARE_UNIT)
#undef PTV_DECLARE_UNIT
}  // namespace swp

}  // namespace ptv
