// chunkcore.hpp -- what ONE lane of the speculative-chunk kernels does with its chunk: find a start, walk, rebuild.
//
// Host-testable like walker.hpp (tests/host_harness.cpp compiles it with g++ and emulates the lanes of a workgroup one
// after the other; tests/test_chunk_host.py checks the result against the oracle).  Everything here works on a
// *window* of the lane's fibre through a small accessor (concept `Win`):
//     double y(int i)            sample i            (rows [lo, hi) of the fibre are present; reading row hi is allowed
//     double r(int i)            edge penalty i       and yields an unspecified value)
//     void   put(int i, double)  replace row i (rebuild only)
//
// The walk (walk_interior) is the state machine of walker.hpp -- same arithmetic, same operation order -- written as
// ONE branch-free trip per sample: the no-bend update and the post-bend state are both computed and selected, the two
// "pull back inside the tube" updates are a min / max and an unconditional add (a zero correction is an exact no-op),
// and a bend that rewinds costs two extra window reads instead of a divergent region.  The walk records no piece
// values: a piece end costs two bits (end, bend type); values are rebuilt from the window afterwards in closed form.
//
// Ownership of outputs (rebuild_owned).  A piece is rewritten by the lane in whose chunk it ENDS -- that lane reads
// and replaces all rows of the piece, including rows that lie in earlier chunks.  The lane of an earlier chunk leaves
// the rows after its last piece end alone, so every row has exactly one writer as long as consecutive walks agree
// (proven links).  Exceptions: rows outside the block are never written (the zone before the first chunk and the
// look-ahead after the last are read-only copies), so the block's last lane also writes the part inside the block of
// the piece that covers the block's last sample, and the first lane writes its first piece from the block start only;
// a lane whose link is unproven keeps to its own rows (its output is rewritten by the repair kernel anyway, from the
// last proven bend on).
#pragma once

#include "walker.hpp"

#ifdef PTV_HOST_TEST
#include <cmath>
#endif

namespace ptv {

#ifdef PTV_HOST_TEST
inline double ptv_min(double a, double b) { return a < b ? a : b; }
inline double ptv_max(double a, double b) { return a > b ? a : b; }
#else
// plain compare-and-select: no NaN canonicalisation around v_min_f64 / v_max_f64 (inputs are finite)
__device__ __forceinline__ double ptv_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double ptv_max(double a, double b) { return __builtin_fmax(a, b); }
#endif

constexpr unsigned kCodeCertain = 0x80000000u;   // flag on a published `mine` code: the walk began AT a bend known a priori
constexpr unsigned kCodeBad = 0xfffffffeu;       // the walk ran off its window: trust nothing it recorded

// What a chunk's walk leaves behind.
struct ChunkRec {
    unsigned ends = 0;     // bit u: a piece ends at sample cs + u (0 <= u < ce - cs)
    unsigned types = 0;    // bit u: that piece was ended by a FLOOR bend (0: CEIL bend, or the fibre end)
    unsigned mine = 0;     // (restart << 1 | type) of the last bend at or before cs; 0 = none
    unsigned next = 0;     // ... at or before ce (read by the lane of the following chunk)
    unsigned last = 0;     // the walk's most recent bend (once done: the one that closed the piece covering ce - 1)
    double vclose = 0.0;   // value of the piece covering ce - 1, when the slow tail of the walk closed it (have_vclose)
    bool have_vclose = false;
    bool done = false;     // the piece covering ce - 1 is closed
    bool failed = false;   // the walk ran off its window
};

// A bend known without walking: x_k - x_{k-1} = (y_k - y_{k-1}) + (u_k - u_{k-1}) - (u_{k-1} - u_{k-2}) with every dual
// |u_j| <= r_j, so a jump |y_k - y_{k-1}| > r_k + 2 r_{k-1} + r_{k-2} (4 lambda) keeps its sign in x: the string bends
// there (up-jump: off the floor), and the state after a bend depends on the bend alone.  Looks at the LOOK edges
// (k - 1, k), k = cs - LOOK + 1 .. cs; the nearest one to cs wins.  Returns the restart sample (-1: none) and the type.
template <bool WEIGHTED, int LOOK, class Win>
__device__ __forceinline__ int certain_bend_before(const Win &win, int cs, int len, double lam, int &type) {
    double yv[LOOK + 1], rv[LOOK + 2];
#pragma unroll
    for (int u = 0; u <= LOOK; u++) yv[u] = win.y(cs - u);
    if (WEIGHTED) {
#pragma unroll
        for (int u = 0; u <= LOOK + 1; u++) rv[u] = (cs - u < len - 1) ? win.r(cs - u) : 0.0;
    }
    int cat = -1;
#pragma unroll
    for (int u = LOOK - 1; u >= 0; u--) {   // edge (k - 1, k), k = cs - u
        const double d = yv[u] - yv[u + 1];
        double thr = 4.0000001 * lam;
        bool ok = true;
        if (WEIGHTED) {
            thr = 1.0000001 * (rv[u] + 2.0 * rv[u + 1] + rv[u + 2]);
            ok = (rv[u] >= 0.0) & (rv[u + 1] > 0.0) & (rv[u + 2] >= 0.0);
        }
        cat = (ok & (fabs(d) > thr)) ? cs - u : cat;
    }
    // (the bend's type from the chosen edge alone -- two more window reads -- instead of a compare and a select per edge looked at)
    type = 0;
    if (cat >= 0) type = (win.y(cat) - win.y(cat - 1) > 0) ? BEND_FLOOR : BEND_CEIL;
    return cat;
}

// The interior steps of a chunk's walk: samples i < lim, where lim <= len - 1 (the fibre's last sample has its own
// tests: walker_run) and lim <= the window end.  Leaves the walker at the first sample it does not handle.
template <bool WEIGHTED, class Win>
__device__ __forceinline__ void walk_interior(Walker &w, ChunkRec &rec, const Win &win, int lim, int cs, int ce, double lam) {
    if (w.i >= lim || rec.done) return;
    const unsigned span_own = (unsigned)(ce - cs);
    double yi = win.y(w.i);
    while (!rec.done && w.i < lim) {
        const int i = w.i;
        const double ynx = win.y(i + 1);   // speculative: most trips advance by one
        const double r = WEIGHTED ? win.r(i) : lam;
        const double h1 = w.hlo + (w.lo - yi);
        const double h2 = w.hhi + (w.hi - yi);
        const bool cv = r < h1;
        const bool fv = !cv && (-r > h2);
        const bool bend = cv || fv;
        const int brk = cv ? w.klo : w.khi;
        const int at = brk + 1;            // at <= i < len - 1: the restart is an interior sample
        const double yat = win.y(at), yat1 = win.y(at + 1);

        // no bend: pull the pieces back inside the tube where they left it
        const SpanDiv over((double)(i - w.k0));
        const double d2 = ptv_min(r - h2, 0.0), d1 = ptv_max(-r - h1, 0.0);
        const double nhi = w.hi + over(d2), nlo = w.lo + over(d1);
        const double nhhi = ptv_min(h2, r), nhlo = ptv_max(h1, -r);
        const int nkhi = (h2 >= r) ? i : w.khi, nklo = (h1 <= -r) ? i : w.klo;

        // bend: closed-form first sample of the new piece (walker_restart_with, at < len - 1)
        double blo, bhi, bhhi, bhlo;
        if (WEIGHTED) {
            const double wp = win.r(brk), wc = win.r(at);
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

        w.lo = bend ? blo : nlo;
        w.hi = bend ? bhi : nhi;
        w.hlo = bend ? bhlo : nhlo;
        w.hhi = bend ? bhhi : nhhi;
        w.k0 = bend ? brk : w.k0;
        w.klo = bend ? at : nklo;
        w.khi = bend ? at : nkhi;
        w.i = (bend ? at : i) + 1;
        yi = bend ? yat1 : ynx;

        // what the bend leaves behind
        const unsigned code = ((unsigned)at << 1) | (unsigned)fv;
        const int sh = brk - cs;
        const bool own = bend && (unsigned)sh < span_own;
        const unsigned bit = 1u << (sh & 31);
        rec.ends |= own ? bit : 0u;
        rec.types |= (own && fv) ? bit : 0u;
        rec.mine = (bend && at <= cs) ? code : rec.mine;
        rec.next = (bend && at <= ce) ? code : rec.next;
        rec.last = bend ? code : rec.last;
        rec.done = bend && brk >= ce - 1;
    }
}

// Source for walker_run (walker.hpp) over the same window and record: the slow tail of a chunk's walk -- the fibre's
// last sample, or (PAST) samples beyond the window, fetched one at a time through `far` (double far_y(int), far_r(int)).
template <bool WEIGHTED, bool PAST, int OVERFLOW, class Win, class Far>
struct TailSource {
    const Win &win;
    const Far &far;
    ChunkRec &rec;
    int cs, ce, hi, len;
    __device__ __forceinline__ double y(int i) const {
        if (!PAST || i < hi) return win.y(i < hi ? i : hi - 1);
        return far.far_y(i < len ? i : len - 1);
    }
    __device__ __forceinline__ double r(int i) const {
        if (!PAST || i < hi) return win.r(i < hi ? i : hi - 1);
        return (i < len - 1) ? far.far_r(i) : 0.0;
    }
    __device__ __forceinline__ void piece(int, int to, double v) {
        if (to >= cs && to < ce) rec.ends |= 1u << (to - cs);
        if (to >= ce - 1) {
            rec.vclose = v;
            rec.have_vclose = true;
            rec.done = true;
        }
    }
    __device__ __forceinline__ void bend(int at, int type) {
        const unsigned code = ((unsigned)at << 1) | (unsigned)type;
        rec.mine = (at <= cs) ? code : rec.mine;
        rec.next = (at <= ce) ? code : rec.next;
        rec.last = code;
        const int e = at - 1 - cs;   // the piece that this bend ended, relative to the chunk
        if (e >= 0 && e < ce - cs) rec.types |= (unsigned)type << e;
    }
    __device__ __forceinline__ bool keep_going(int i) {
        if (rec.done) return false;
        if (i >= hi + (PAST ? OVERFLOW : 0)) {   // (a walk that reaches the fibre end never gets here: i < len <= hi + OVERFLOW then)
            rec.failed = true;
            return false;
        }
        return true;
    }
};

struct NoFar {
    __device__ __forceinline__ double far_y(int) const { return 0.0; }
    __device__ __forceinline__ double far_r(int) const { return 0.0; }
};

// Piece values from piece ends.  Between two knots of the taut string the prox is constant, and the string's height
// above the tube centre at a knot is -r after a CEIL bend (the knot sits on the tube floor), +r after a FLOOR bend,
// 0 at the fibre ends (r = the tube half-width there: lambda, or the edge's own penalty).  Summing x - y over a
// piece [a, b] therefore gives   v = ( sum_{a..b} y + h_b - h_{a-1} ) / (b - a + 1).
// For one-sample pieces this is bit-for-bit the walker's closed-form restart value (y, y +- 2 lambda); for longer
// pieces it agrees with the walker's running slope to a few ulps (checked on the host: < 1e-15 relative).
//
// One lane, its chunk [cs, ce), C = the compile-time chunk length (ce - cs <= C).  `wlo`: first row of the block (rows
// before it are read-only); `block_last`: this is the last chunk of its block (or of the fibre); `link_ok`: the lane's
// walk is known to continue its predecessor's (or began at the fibre start / at a bend known a priori).
// Rows are replaced by F::fuse(y, v): the prox value itself, or directly the sweep's output when it depends on (y, x) only.
// UNROLL: how many rows of the two passes are in flight together (1 where registers are scarce: the 64-fibre tile at two
// workgroups per CU; more where the LDS latency of a row would otherwise be paid row by row).
// TAB / rt: piece lengths are bounded by the window (plain geometries: zone + chunk + look-ahead rows), so the division by the
// length is one product with rt[length] = the correctly rounded 1.0 / length (see walk_asm.hpp: walk_interior_asm_tab).
// TSZ > 0: the table has TSZ entries and longer pieces are possible (robust instantiations): those divide.
// FULL >= 1: the caller vouches that every lane of the wave holds a whole chunk (ce - cs == C) and that the fibre's last sample lies
// beyond the chunk -- the interior of a fibre, i.e. nearly everything: no row is tested for being in range, no knot for being the
// fibre's end.  FULL == 2 (unweighted; the along-fibre kernel, which has the registers): the chunk's C samples also stay in registers
// between the two passes, so the forward pass costs a row one LDS read, the backward pass none, and ops whose output needs the row's
// own sample (DR_COL) take the same array-free passes as the others instead of a dynamic back-fill loop per piece.  Same arithmetic
// in the same order as the general form: bit-identical (tests/host_harness.cpp runs all three).
// The rows BEFORE a lane's chunk that belong to the first piece ending in it: their sum and count.  rebuild_owned reads them itself where
// nothing can have replaced them yet (lanes of one wave, in lockstep); where the lanes of a fibre sit in different waves (the strided
// tiles) an UNPROVEN lane's rows before its chunk may be somebody else's to write -- the two walks disagree, that is what unproven
// means -- so the caller takes the sums while the window still holds samples only, before the barrier behind the walks.
struct PiecePrefix {
    double s = 0.0, cnt = 0.0;
};
template <class Win>
__device__ __forceinline__ PiecePrefix first_piece_prefix(const Win &win, const ChunkRec &rec, int cs, int start) {
    PiecePrefix pre;
    int a0 = cs;
    if (rec.mine != 0 && rec.mine != kCodeBad) a0 = (int)(rec.mine >> 1);
    else if (start == 0) a0 = 0;
    for (int k = a0; k < cs; k++) {
        pre.s += win.y(k);
        pre.cnt += 1.0;
    }
    return pre;
}

template <class F, bool WEIGHTED, int C, int UNROLL = 1, bool TAB = false, class RT = const double *, int TSZ = 0, int FULL = 0, class Win>
__device__ __forceinline__ void rebuild_owned(Win &win, const ChunkRec &rec, int cs, int ce, int len, int start, bool link_ok,
                                              int wlo, bool block_last, double lam, RT rt = RT(), const PiecePrefix *pre = nullptr,
                                              bool legacy = false) {
    auto quotient = [&](double num, double count) {
        if constexpr (TAB) {
            if (TSZ > 0 && count >= (double)TSZ) {
                const SpanDiv over(count);
                return over(num);
            }
            return num * rt[(int)count];
        } else {
            const SpanDiv over(count);
            return over(num);
        }
    };
    // An unproven lane keeps to its own rows (w0 = cs) -- but what it writes there must be what ITS walk found, the first piece
    // included: a repair walk whose last bend is this lane's `mine` hands over to this chunk and trusts every row of it behind the
    // piece the repair walk itself was in.  So the first piece is summed from its true first row (a0 = the bend `mine`) whether the
    // link is proven or not.  (Until round 5 an unproven lane summed from cs: harmless as long as the repair walk and the lane's
    // walk end that piece in the same place -- the repair walk rewrites exactly those rows -- and wrong when a knot with a jump of
    // exactly zero, as the late iterations of a Dykstra / DR loop produce them, is a bend to one walk and none to the other: the two
    // walks round differently.  tools/case_diag.py, profiles/NOTES_r05.md "session 17".)
    int a0 = cs, w0 = cs;
    double hprev = 0.0;
    if (rec.mine != 0 && rec.mine != kCodeBad) {
        const int at = (int)(rec.mine >> 1);
        const double r = WEIGHTED ? win.r(at - 1) : lam;
        hprev = (rec.mine & 1u) ? r : -r;
        a0 = (legacy && !link_ok) ? cs : at;   // (legacy: option debug_legacy_rebuild -- the hole of rounds 1-4, planted so that a test can watch
                                               //  the certifier catch it; never set otherwise)
        w0 = link_ok ? at : cs;
    } else if (link_ok && start == 0) {
        a0 = w0 = 0;   // no bend yet and the walk began at the fibre start: the first piece starts at sample 0, height 0
    }
    double s = 0.0, cnt = 0.0;
    if (pre && a0 < cs) {             // (summed by the caller while the window held samples only: first_piece_prefix, same order)
        s = pre->s;
        cnt = pre->cnt;
    } else {
        for (int k = a0; k < cs; k++) {   // rows of earlier chunks that belong to the piece ending here (usually none or a few)
            s += win.y(k);
            cnt += 1.0;
        }
#ifndef PTV_HOST_TEST
        // Those rows lie in the predecessor's chunk, and the predecessor -- another lane of this wave in the along-fibre kernel -- may end a
        // piece there and replace them in its own rebuild.  The lanes run in lockstep and the LDS serves a wave's accesses in order, so the
        // sums above are taken before any lane's first put below AS LONG AS the compiler keeps the order; it reasons about one lane's
        // addresses (rows < cs read, rows >= cs written: disjoint) and may not.  A scheduling barrier, no instruction: (measured: the sums
        // in a pass of their own behind a wavefront fence cost the column sweep 1 us of 75, profiles/r06_s2_ab_fence.txt)
        __builtin_amdgcn_wave_barrier();
#endif
    }
    // the piece that covers ce - 1 and ends beyond it: the block's last lane writes its rows inside the block
    auto tail_value = [&](double s_, double cnt_, double hprev_, double &cur_) {
        if (!(block_last && rec.done && !((rec.ends >> (ce - 1 - cs)) & 1u))) return false;
        if (rec.have_vclose) {
            cur_ = rec.vclose;
        } else {
            const int brk = (int)(rec.last >> 1) - 1;   // inside the window: the bend was found by the interior walk
            for (int k = ce; k <= brk; k++) {
                s_ += win.y(k);
                cnt_ += 1.0;
            }
            const double r = WEIGHTED ? win.r(brk) : lam;
            const double hk = (rec.last & 1u) ? r : -r;
            cur_ = quotient(s_ + (hk - hprev_), cnt_);
        }
        return true;
    };
    // height of the string above the tube centre at the knot after own row u, where a piece ends: 0 at the fibre end, else the
    // tube's half-width with the sign of the bend type (unweighted, device: lambda with its sign bit flipped -- a shift and a xor)
    const int uend = len - 1 - cs;   // own row of the fibre's last sample, if it lies in this chunk
    auto knot_height = [&](int u) {
        if (FULL == 0 && u == uend) return 0.0;
        if constexpr (WEIGHTED) {
            const double r = win.r(cs + u);
            return ((rec.types >> u) & 1u) ? r : -r;
        } else {
#ifdef PTV_HOST_TEST
            return ((rec.types >> u) & 1u) ? lam : -lam;
#else
            return __hiloint2double(__double2hiint(lam) ^ (int)((~(rec.types >> u) & 1u) << 31), __double2loint(lam));
#endif
        }
    };
    double cur = 0.0;
    bool have = false;
    if constexpr (FULL >= 2 && !WEIGHTED) {
        double slot[C];   // row u: its sample, or -- once a piece has ended there -- that piece's value
        // (all C reads in flight at once: one LDS latency per chunk instead of one per row)
#pragma unroll
        for (int u = 0; u < C; u++) slot[u] = win.y(cs + u);
        // (unweighted, no fibre end in the chunk: the knot's height is lambda with the sign of the bend type)
        auto height = [&](int u) {
#ifdef PTV_HOST_TEST
            return ((rec.types >> u) & 1u) ? lam : -lam;
#else
            return __hiloint2double(__double2hiint(lam) ^ (int)((~(rec.types >> u) & 1u) << 31), __double2loint(lam));
#endif
        };
        // the length of the piece in hand as an integer: it indexes the reciprocal table as it is (a double count costs a quarter-rate
        // conversion per piece end)
        int n = (int)cnt;
        auto over_n = [&](double num, int count) {
            if constexpr (TAB) {
                if (TSZ > 0 && count >= TSZ) {
                    const SpanDiv over((double)count);
                    return over(num);
                }
                return num * rt[count];
            } else {
                const SpanDiv over((double)count);
                return over(num);
            }
        };
#pragma unroll
        for (int u = 0; u < C; u++) {
            const double yu = slot[u];
            s += yu;
            n += 1;
            if ((rec.ends >> u) & 1u) {
                const double hk = height(u);
                const double v = over_n(s + (hk - hprev), n);
                win.put(cs + u, F::fuse(yu, v));
                slot[u] = v;
                s = 0.0;
                n = 0;
                hprev = hk;
            }
        }
        have = tail_value(s, (double)n, hprev, cur);
#pragma unroll
        for (int u = C - 1; u >= 0; u--) {
            const bool e = (rec.ends >> u) & 1u;
            if (have && !e) win.put(cs + u, F::fuse(slot[u], cur));
            cur = e ? slot[u] : cur;
            have = have || e;
        }
    } else
    if (!F::USES_Y) {
        // The output does not depend on the row's own sample: the value of a piece is parked in the row where the piece
        // ends (forward pass), then every other row takes the value of the next piece end after it (backward pass).
        // (A row that ends no piece costs an add and a count: height, quotient, store and resets sit under the piece-end branch.)
#pragma unroll UNROLL
        for (int u = 0; u < C; u++) {
            const bool in = FULL != 0 || cs + u < ce;
            s += in ? win.y(cs + u) : 0.0;
            cnt += 1.0;
            if (in && ((rec.ends >> u) & 1u)) {
                const double hk = knot_height(u);
                win.put(cs + u, F::fuse(0.0, quotient(s + (hk - hprev), cnt)));
                s = 0.0;
                cnt = 0.0;
                hprev = hk;
            }
        }
        have = tail_value(s, cnt, hprev, cur);
        if (have) cur = F::fuse(0.0, cur);
#pragma unroll UNROLL
        for (int u = C - 1; u >= 0; u--) {
            const bool e = (rec.ends >> u) & 1u;
            const bool in = FULL != 0 || cs + u < ce;
            const double here = in ? win.y(cs + u) : 0.0;
            cur = e ? here : cur;
            if (have && !e && in) win.put(cs + u, cur);
            have = have || e;
        }
    } else {
        // The output needs the row's own sample too: when a piece ends its rows are replaced there and then, last row
        // first (it is in a register), the earlier ones -- none for a one-sample piece -- in a short loop.  No second pass,
        // nothing waits in registers (sixteen parked values spill at the 128-VGPR budget of two workgroups per CU).
        // Same recurrence, same branch form: everything but the running sum and the count happens where a piece ends.
        int first = w0 > wlo ? w0 : wlo;
#pragma unroll UNROLL
        for (int u = 0; u < C; u++) {
            const bool in = FULL != 0 || cs + u < ce;
            const double yu = in ? win.y(cs + u) : 0.0;
            s += yu;
            cnt += 1.0;
            if (in && ((rec.ends >> u) & 1u)) {
                const double hk = knot_height(u);
                const double v = quotient(s + (hk - hprev), cnt);
                win.put(cs + u, F::fuse(yu, v));
                for (int k = cs + u - 1; k >= first; k--) win.put(k, F::fuse(win.y(k), v));
                first = cs + u + 1;
                s = 0.0;
                cnt = 0.0;
                hprev = hk;
            }
        }
        if (tail_value(s, cnt, hprev, cur))
            for (int k = ce - 1; k >= first; k--) win.put(k, F::fuse(win.y(k), cur));
        return;
    }
    if (have) {
        const int from = w0 > wlo ? w0 : wlo;
        for (int k = cs - 1; k >= from; k--) win.put(k, F::fuse(win.y(k), cur));
    }
}

// ---- known runs: the fibre cut at the bends known a priori (round 6; sweep_along_kernel, RUNS) ---------------------------------------
// An edge (k - 1, k) with |y_k - y_{k-1}| > 4 lambda is a bend whatever the rest of the fibre looks like (certain_bend_before), and the
// state right after a bend depends on the bend alone: the stretch between two such edges -- a RUN -- is a problem of its own, with the
// string's height given at both ends.  On the headline's data 78 % of the edges qualify, and the runs are short:
//   * one sample (61 % of all samples): a piece of its own, nothing to decide;
//   * two samples (27 %): the one edge inside bends or does not, and a compare says which -- with d the edge's jump (|d| <= 4 lambda: it
//     is not a bend known a priori) and s0, s2 the bend types at the run's ends: both CEIL: it bends (CEIL) iff d < 0 ; both FLOOR:
//     (FLOOR) iff d > 0 ; otherwise iff |d| > 2 lambda, with the jump's sign.  (The walk from the first bend makes exactly these decisions in its two trips: a CEIL start has
//     h1 = -lambda - d, h2 = 3 lambda - d at the second sample -- it bends there iff d < -2 lambda, and touches the floor, klo = the
//     second sample, iff d >= 0, which is where the closing bend then breaks; symmetrically for a FLOOR start.  Ties on the last bit
//     may fall differently: a knot with a jump of zero up to rounding, both cuts of which are the prox to rounding.)
//   * three and more (12 %): walked -- but a run needs no zone, no link and no second chance: ONE lane walks ONE run from its first
//     bend to the closing one (walk_chunk with the run as its "chunk"), so a wave of 64 chunks walks its ~44 runs in one pass of
//     ~8 trips where the chunk-per-lane scheme takes 19.5 (tools/study in the notes: profiles/NOTES_r06.md).
// What comes out is the ChunkRec the rebuild wants -- piece ends and bend types per chunk, the codes of the bends around it -- with no
// speculation behind it: a segment solved this way is exact by construction and publishes its codes as certain.
// Bit b of an edge mask of the lane with chunk start cs: the edge before sample cs + b - kEdgeBias (two edges of the chunk before,
// the chunk's own C, the first 30 - C of the chunk behind).
constexpr int kEdgeBias = 2;
constexpr int kRunMax = 14;    // longest run a lane takes (its piece ends must fit a 32-bit mask behind any offset inside a chunk)
struct EdgeMasks {
    unsigned K = 0, P = 0, N = 0, B = 0;   // |d| > 4 lambda (a bend known a priori) ; d > 0 ; d < 0 ; |d| > 2 lambda
};
// the chunk's own edges (bits kEdgeBias .. kEdgeBias + C - 1): rows cs - 1 .. cs + C - 1 are read
template <int C, class Win>
__device__ __forceinline__ EdgeMasks own_edges(const Win &win, int cs, double lam) {
    double yv[C + 1];
#pragma unroll
    for (int u = 0; u <= C; u++) yv[u] = win.y(cs - 1 + u);
    EdgeMasks m;
    // (4 lambda EXACTLY, not certain_bend_before's 4.0000001: the rule for two-sample runs below rests on "an edge that is not known
    //  a priori jumps by 4 lambda at most" -- with a margin on the threshold an inner edge of -4.00000006 lambda between two FLOOR bends,
    //  which bends CEIL, was left unbent: 3e-9 off in two rows of one fibre of a 4096^2 PD2 solve, found by the certifier
    //  (tests/golden/sliver_edge_fibre.npz).  |d| > 4 lambda is a bend in exact arithmetic; within rounding of 4 lambda its jump is zero
    //  within rounding, and either cut is the prox to rounding.)
    const double t4 = 4 * lam, t2 = 2 * lam;
#pragma unroll
    for (int u = C - 1; u >= 0; u--) {
        const double d = yv[u + 1] - yv[u];
#ifdef PTV_HOST_TEST
        const double a = fabs(d);
        m.K = (m.K << 1) | (unsigned)(a > t4);
        m.B = (m.B << 1) | (unsigned)(a > t2);
        m.P = (m.P << 1) | (unsigned)(d > 0);
        m.N = (m.N << 1) | (unsigned)(d < 0);
#else
        // mask = 2 mask + (the compare): one compare into the carry, one add with carry -- two instructions a bit where the compiler
        // takes three (compare, select 0 / 1, shift-or); 17 edges x 4 masks a lane
        asm("v_cmp_gt_f64 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m.K) : "v"(d), "s"(t4) : "vcc");
        asm("v_cmp_gt_f64 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m.B) : "v"(d), "s"(t2) : "vcc");
        asm("v_cmp_lt_f64 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m.P) : "v"(d) : "vcc");
        asm("v_cmp_gt_f64 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m.N) : "v"(d) : "vcc");
#endif
    }
    m.K <<= kEdgeBias; m.B <<= kEdgeBias; m.P <<= kEdgeBias; m.N <<= kEdgeBias;
    return m;
}
// one edge on its own (the few around a segment that no chunk owns)
__device__ __forceinline__ void one_edge(double ya, double yb, double lam, unsigned &k, unsigned &pp, unsigned &nn, unsigned &bb) {
    const double d = yb - ya, a = fabs(d);
    k = (unsigned)(a > 4 * lam);
    bb = (unsigned)(a > 2 * lam);
    pp = (unsigned)(d > 0);
    nn = (unsigned)(d < 0);
}
// ... with the last two edges of the chunk before (its own bits C, C + 1) and the first edges of the chunk behind (its own bits 2 ...)
template <int C>
__device__ __forceinline__ unsigned edge_ext(unsigned own, unsigned prev_own, unsigned next_own) {
    return own | ((prev_own >> C) & 3u) | ((next_own >> kEdgeBias) << (C + kEdgeBias));
}
// What is settled without a walk: BE = bend edges (known a priori, or the inner edge of a two-sample run that bends), BT their types,
// WS = first edges of the runs of three and more samples.  (Bits whose neighbours lie outside the mask are the caller's to ignore.)
__device__ __forceinline__ void settle_short_runs(const EdgeMasks &m, unsigned &BE, unsigned &BT, unsigned &WS) {
    const unsigned K = m.K, Km = K << 1, Kp = K >> 1;
    const unsigned mid = ~K & Km & Kp;                       // the inner edge of a two-sample run
    const unsigned Pm = m.P << 1, Pp = m.P >> 1;             // types of the bends before / behind it
    const unsigned bend = mid & ((~Pm & ~Pp & m.N) | (Pm & Pp & m.P) | ((Pm ^ Pp) & m.B));
    BE = K | bend;
    BT = m.P & BE;
    WS = K & ~Kp & ~(K >> 2);
}
// a run to walk, as the lane that owns its first sample describes it to the lane that will walk it
struct RunEntry {
    unsigned word;
    __device__ __forceinline__ static RunEntry make(int lane, int b, int e, int type, bool free_start) {
        return RunEntry{(unsigned)lane | ((unsigned)b << 6) | ((unsigned)e << 11) | ((unsigned)type << 16) | ((unsigned)free_start << 17)};
    }
    __device__ __forceinline__ int lane() const { return (int)(word & 63u); }
    __device__ __forceinline__ int b() const { return (int)((word >> 6) & 31u); }     // bit of the run's first edge in its owner's mask
    __device__ __forceinline__ int e() const { return (int)((word >> 11) & 31u); }    // ... of the bend that closes it
    __device__ __forceinline__ int type() const { return (int)((word >> 16) & 1u); }
    __device__ __forceinline__ bool free_start() const { return (word >> 17) & 1u; }
};
// the end of the run that starts at edge b: the next bend known a priori (-1: none inside the mask, or further than a lane takes)
__device__ __forceinline__ int run_end(unsigned K, int b) {
    const unsigned up = (b >= 31) ? 0u : (K >> (b + 1));
    if (up == 0u) return -1;
#ifdef PTV_HOST_TEST
    const int e = b + 1 + __builtin_ctz(up);
#else
    const int e = b + 1 + (__ffs((int)up) - 1);
#endif
    return (e - b <= kRunMax) ? e : -1;
}

}  // namespace ptv
