// sweep_links.hpp -- what the chunk kernels and the repair kernels tell each other: link codes, the dirty word, the words across workgroups.
// (One of the pieces of sweep_kernels.hpp, which includes them in order; not meant to be included on its own.)
#pragma once

namespace ptv {
namespace swp {


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

}  // namespace swp
}  // namespace ptv
