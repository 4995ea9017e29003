// pincore.hpp -- the taut string of ONE fibre found by all lanes of a group at once: "pin the worst violator".
//
// Why.  The speculative-chunk kernels (chunkcore.hpp) are fast while a walk that starts a few samples early meets the
// true walk before its chunk begins, i.e. while pieces are short.  When pieces are tens to thousands of samples long
// (lambda >~ the noise level, block images, the later iterates of a DR solve) they degrade to one sequential lane per
// fibre, and the linearized walk itself re-walks long pieces again and again (the reference answers that with the
// convex-hull taut string behind its hybrid switch: src/TVL1opt_tautstring.cpp:256-340,
// src/TVL1opt_hybridtautstring.cpp:31-35,73).  This is the device counterpart: exact for any input, its cost is
// (number of levels) x n with ~12-16 levels for anything from white noise to a single flat piece, and every level is
// data-parallel over the samples of the fibre.
//
// The problem as a string.  With S_j = sum_{i<j} y_i (knots j = 0..n), the prox of y is the slope sequence of the
// shortest path s through the tube S_j - r_j <= s_j <= S_j + r_j (r = lambda at interior knots, 0 at both ends):
// x_i = s_{i+1} - s_i (the classic taut-string reading of src/TVL1opt_tautstring.cpp:262-280).
//
// The rule.  Let A = (ja, ha), B = (jb, hb) be two points known to lie on the string and c the chord between them.
// If the chord stays inside the tube, it IS the string between A and B.  Otherwise let ku maximise c_j - (S_j + r_j)
// (how far the chord pokes through the upper wall) and kl maximise (S_j - r_j) - c_j: when positive, BOTH walls' worst
// knots lie on the string, at their wall's height.  Proof for the upper wall: let P be the string between A and B and
// g = c - P; g(ja) = g(jb) = 0.  If P passed strictly below the wall at ku then max g > c_ku - (S_ku + r_ku) = the
// largest upper violation; at a maximum of g the string is convex (its slope grows), and a taut string grows its slope
// only where it touches the UPPER wall -- so that maximiser k is an upper contact, P_k = S_k + r_k, and its violation
// c_k - P_k = g(k) exceeds the maximum.  Contradiction; the lower wall is symmetric.  So every level pins, in every
// unfinished segment, the worst knot of each wall, and segments whose chord fits are final; the fibre ends are pinned
// from the start.  Levels are independent of the order in which segments are looked at: the result is a function of
// the input only (the maxima are reduced with atomic max on the bit pattern and ties go to the smallest knot).
//
// Work split (concept).  A group of G lanes owns one fibre; lane t owns the interior knots [1 + tP, 1 + (t+1)P) and
// keeps their pins in two bit masks in registers, plus the nearest pin on either side of its range (la, rb) with the
// string's height there.  A segment that lies inside one lane's range is resolved by that lane alone.  A segment that
// spans lanes is reduced through a slot in shared memory keyed by the lane that holds its left pin -- unique, because
// only one segment can leave a lane's range to the right -- in three steps separated by group barriers:
//     scan    every lane evaluates its knots against their chords, resolves inner segments, posts the maxima of its
//             two possible spanning runs (the one entering from the left, the one leaving to the right)
//     claim   lanes whose run maximum equals the slot's maximum post their knot (atomic min: smallest knot wins)
//     update  every lane reads the slots of its spanning segments and moves la / rb / its own masks
// A lane none of whose segments changed in a level is final (a segment with a violation anywhere always gains a pin,
// which changes something for every lane it touches); the group stops when no lane gained a pin.
//
// Numerics.  Decisions are taken on S, a running sum (centred by the fibre mean by the caller), so values are exact to
// ~1e-16 |S| / piece length instead of the walker's 1e-16 |y|: 1e-14 relative on unit noise of 4096 samples.  Knots
// whose violation is rounding noise may or may not be pinned; either way the string moves by that noise only.
//
// Host-testable: tests/host_harness.cpp runs the three steps for all lanes of a group in turn (a barrier is the end of
// a loop), tests/test_pin_host.py compares with the oracle.
#pragma once

namespace ptv {

#ifndef PTV_HOST_TEST
#define PTV_PIN_FN __device__ __forceinline__
PTV_PIN_FN unsigned long long pin_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
PTV_PIN_FN double pin_double(unsigned long long b) { return __longlong_as_double((long long)b); }
PTV_PIN_FN double pin_max(double a, double b) {   // one v_max_f64: the builtin would first quiet both operands (it cannot
    double r;                                      // know that a tagged violation is never a signalling NaN)
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
PTV_PIN_FN double pin_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a / b for a positive integer-valued b: reciprocal + one Newton step, then one residual correction of the quotient (within one
// ulp of the IEEE quotient at a third of the instructions of the full division sequence; the values of this solver carry the
// rounding of their running sums anyway -- see "Numerics" above)
PTV_PIN_FN double pin_div(double a, double b) {
    const double x = __builtin_amdgcn_rcp(b);
    const double inv = __builtin_fma(__builtin_fma(-b, x, 1.0), x, x);
    const double q = a * inv;
    return __builtin_fma(__builtin_fma(-q, b, a), inv, q);
}
#else
#define PTV_PIN_FN inline
inline double pin_div(double a, double b) { return a / b; }
inline unsigned long long pin_bits(double v) { unsigned long long b; std::memcpy(&b, &v, 8); return b; }
inline double pin_double(unsigned long long b) { double v; std::memcpy(&v, &b, 8); return v; }
inline double pin_max(double a, double b) { return a > b ? a : b; }
inline double pin_min(double a, double b) { return a < b ? a : b; }
#endif

// A violation carries the lane-local index of its knot in the lowest bits of its mantissa (log2 P bits: a relative
// perturbation of 2^-46 at most, far below the rounding of the running sums it is computed from), so that "largest
// violation and where" is ONE floating-point max per knot and wall instead of a compare and three selects.
template <int P>
struct PinTag {
    static constexpr unsigned long long kMask = (P <= 16 ? 15ull : P <= 32 ? 31ull : 63ull);
    PTV_PIN_FN static double tag(double v, int k) { return pin_double((pin_bits(v) & ~kMask) | (unsigned long long)k); }
    PTV_PIN_FN static int index(double v) { return (int)(pin_bits(v) & kMask); }
};

// window seeds: lanes of a wave (of 64: 1024 knots) that must have found a knot with 64-knot windows for the 16-knot stage to run, and with
// 16-knot windows for the 4-knot stage (operands of DR sweeps on unit noise, lanes with a knot per stage 64 / 16 / 4: lambda 0.8: 27 / 43 / 21;
// 1: 24 / 29 / 6; 1.5: 11 / 8 / 0; 2: 6 / 3 / 0; 3: 2 / 0 / 0 -- a stage costs a third of a level, a handful of knots does not save one)
constexpr int kSeedStage16 = 8, kSeedStage4 = 24;

// Shared-memory side of a group (concept `Sh`):
//     static constexpr bool kWeighted          per-knot half-widths (else r(j) is one constant)
//     double S(int j)                          running sum at knot j (0 <= j <= n)
//     double r(int j)                          tube half-width at interior knot j
//     double own(int t, int k)                 = S(1 + tP + k): lane t's k-th knot; k is a compile-time constant wherever this
//                                                is called (unrolled loops), so an implementation may keep the values in registers
//     double own_at(int t, int k)              the same for a run-time k -- EXCEPT at a knot the lane has pinned and settled
//                                                (PinLane::settle): there the slot holds the string's height S -+ r, which is
//                                                all anybody ever asks of a pinned knot again
//     void   set_own(int t, int k, double v)   overwrite that slot (settle only)
//     double rown(int t, int k)                = r(1 + tP + k)
//     void   post(int wall, int slot, double v)     slot maximum <- max(., v)          (v > 0)
//     double best(int wall, int slot)               slot maximum (0: nothing posted)
//     void   claim(int wall, int slot, Key key)     slot key <- min(., key)      (PinLane::claim_key: the knot in the low half)
//     int    knot(int wall, int slot)               the knot of the smallest key (negative: nobody claimed)
//     void   clear_best(int slot), clear_knot(int slot)     both walls of a slot back to "nothing"
// One buffer of slots serves all levels: a lane clears the knots of the slots it owns (slot t + 1; lane 0 also slot 0)
// in its scan step -- every reader of the previous level is past the barrier that ended it, the next claims come after
// the next barrier -- and their maxima in its update step, which reads knots only.
// wall 0 = upper, 1 = lower.

template <bool SMALL> struct PinMask { using type = unsigned long long; };
template <> struct PinMask<true> { using type = unsigned; };

// Key: the type of a claim key -- 32 bits (knot and distance in 16 bits each: fibres held in one workgroup's LDS) or 64 bits
// (32 + 32: fibres spread over a grid of workgroups, pinlong.hip).
template <int P, class Key = unsigned>
struct PinLane {
    static_assert(P >= 1 && P <= 64, "a lane's pins live in one 64-bit mask per wall");
    // (a 32-bit claim key keeps the knot in 16 bits: fibres of at most 65535 samples -- the LDS kernel's planes end at 16384)
    // a lane's pins: one bit per own knot and wall -- 32-bit words where they suffice (half the instructions per shift / scan)
    using Mask = typename PinMask<(P <= 32)>::type;
    PTV_PIN_FN static int ctz(Mask m) { return sizeof(Mask) == 4 ? __builtin_ctz((unsigned)m) : __builtin_ctzll((unsigned long long)m); }
    int n = 0, t = 0;
    int j0 = 0, j1 = 0;                        // own candidate knots [j0, j1) (interior knots are 1 .. n-1)
    Mask pinU = 0, pinL = 0;                   // bit k: knot j0 + k is pinned to the upper / lower wall
    Mask pendU = 0, pendL = 0;                 // ... gained in the last update: their slots still hold the running sum (settle)
    int la = 0, rb = 0;                        // nearest pinned knot before j0 / at or after j1
    double hl = 0.0, hr = 0.0;                 // the string's height there
    bool final_ = false;                       // no segment of this lane can change any more
    // what the scan step found for the two runs that may span lanes
    double eU = 0.0, eL = 0.0, xU = 0.0, xL = 0.0;   // entering run / leaving run: largest (scaled) violation per wall (0: none)
    int eUk = 0, eLk = 0, xUk = 0, xLk = 0;          // ... and the knot
    int eEnds = 0, xEnds = 0;                         // ja + jb of the two runs' segments (twice their midpoint)
    bool leaving = false;                             // the lane has pins of its own, so a second run leaves to the right
    Mask newU = 0, newL = 0;                          // pins found this level inside the lane's own range

    PTV_PIN_FN static int slot_of(int ja) { return ja == 0 ? 0 : (ja - 1) / P + 1; }

    template <class Sh>
    PTV_PIN_FN void init(int n_, int t_, const Sh &sh) {
        n = n_;
        t = t_;
        j0 = 1 + t * P < n ? 1 + t * P : n;
        j1 = j0 + P < n ? j0 + P : n;
        pinU = pinL = 0;
        pendU = pendL = 0;
        la = 0;
        hl = sh.S(0);
        rb = n;
        hr = sh.S(n);
        final_ = (t * P >= n);   // a lane without samples never acts (it still passes the barriers); one with a sample
                                 // but no knot (the fibre's last) follows its neighbours' pins
    }

    // Knots known before the first level.  Where |y_j - y_{j-1}| > r_{j+1} + 2 r_j + r_{j-1} (4 lambda) the prox keeps the sign of
    // that jump whatever the rest of the fibre looks like (chunkcore.hpp: bends known a priori), i.e. the string bends at knot j:
    // upwards -- a convex bend, on the UPPER wall -- for an up-jump, on the lower wall for a down-jump.  Pinning them at once
    // spares the levels that would find them: on the inputs of DR sweeps at lambda = 0.8 x noise 13 levels become 6-10
    // (tools/study/pin_study.py).  up / lo: the lane's own such knots; (la_, hl_) / (rb_, hr_): the nearest one on either side of
    // the lane's range (or the fibre end) and the string's height there.  Call after init(), before the first scan().
    PTV_PIN_FN void seed(Mask up, Mask lo, int la_, double hl_, int rb_, double hr_) {
        pinU = pendU = up;
        pinL = pendL = lo;
        la = la_;
        hl = hl_;
        rb = rb_;
        hr = hr_;
    }

    template <class Sh>
    PTV_PIN_FN double height(const Sh &sh, int j, bool lower) const {
        return lower ? sh.S(j) - sh.r(j) : sh.S(j) + sh.r(j);
    }

    // ---- knots known before the first level, by windows (round 6) ------------------------------------------------------------
    // The 4 lambda rule above is the smallest case of a rule about ANY window of knots a < b, pinned or not:
    //     let l be the line through the upper wall at a and b, and k the interior knot where the wall lies deepest below it;
    //     if that depth exceeds the tube's width at both ends, V = l_k - (S_k + r_k) > max(2 r_a, 2 r_b), the string touches the
    //     upper wall at k (and symmetrically the lower wall where it rises highest above its own line).
    // Proof.  Let P be the string and g = l - P on [a, b].  At the ends P is inside the tube, so g(a) <= l_a - (S_a - r_a) = 2 r_a
    // < V, likewise g(b) < V.  If P passed strictly below the wall at k, g(k) > V: the maximum of g over [a, b] is then larger
    // than V and attained at an interior knot k*, where P is convex (its slope grows) -- and a taut string grows its slope only
    // where it touches the upper wall, so g(k*) = l_k* - (S_k* + r_k*) <= V by the choice of k.  Contradiction.  (Nothing is
    // assumed about the string at a and b: the window needs no pins.  Adjacent knots a = k - 1, b = k + 1 with one penalty:
    // V = |y_k - y_{k-1}| / 2 > 2 lambda is the 4 lambda rule.)
    // With one penalty for all interior knots both walls are copies of the sums, so ONE number per knot serves both walls:
    // W (chord_S(j) - S_j) = (S_a - S_j) W + (S_b - S_a)(j - a), above 2 lambda W: upper contact; below -2 lambda W: lower.
    // Windows used (unit noise at lambda = 1: pieces of 4-10 samples; tools measured on DR iterates: 10-12 levels become 4-6):
    // 4, 16 and 64 knots, each on its own grid and on the grid shifted by half a window.  P = 16; windows that touch the
    // fibre ends (r = 0 there) are left out.  With per-knot penalties the two walls are evaluated separately.
    // A lane evaluates its own knots; windows that span lanes are joined by the caller.  Values carry the knot's distance from
    // the window start in the low six bits of their mantissa (as PinTag does), so that "deepest and where" is one max and one
    // min per knot.
    struct Win {
        double mx, mn;
    };
    PTV_PIN_FN static double wtag(double v, int d) { return pin_double((pin_bits(v) & ~63ull) | (unsigned long long)d); }
    PTV_PIN_FN static int wdist(double v) { return (int)(pin_bits(v) & 63ull); }
    PTV_PIN_FN static double wmin(double a, double b) { return pin_min(a, b); }
    PTV_PIN_FN static Win wjoin(Win a, Win b) { return Win{pin_max(a.mx, b.mx), wmin(a.mn, b.mn)}; }
    // window of W knots starting at own index e = A (own knot e, 1 <= e <= P, is knot t P + e; e = 0 and e > P are the
    // neighbours'): the part over own knots e = E0 .. E1.  (Sa, ra), (Sb, rb): the sums and half-widths at the window's ends.
    // One penalty: one number per knot (see above).  Per-knot penalties: the upper wall's depth below its line in mx, the lower
    // wall's height above its own -- negated -- in mn.
    template <int W, int A, int E0, int E1, class Sh>
    PTV_PIN_FN Win win_part(const Sh &sh, double Sa, double ra, double Sb, double rb) const {
        Win w{0.0, 0.0};
        if constexpr (!Sh::kWeighted) {
            const double dl = Sb - Sa;
            double lin = dl * (double)(E0 - A);
#pragma unroll
            for (int e = E0; e <= E1; e++) {
                const double v = wtag((Sa - sh.own(t, e - 1)) * (double)W + lin, e - A);
                w.mx = pin_max(w.mx, v);
                w.mn = wmin(w.mn, v);
                lin += dl;
            }
        } else {
            const double Ua = Sa + ra, La = Sa - ra, du = (Sb + rb) - Ua, dw = (Sb - rb) - La;
            double linu = du * (double)(E0 - A), linl = dw * (double)(E0 - A);
#pragma unroll
            for (int e = E0; e <= E1; e++) {
                const double sj = sh.own(t, e - 1), rj = sh.rown(t, e - 1);
                w.mx = pin_max(w.mx, wtag((Ua - (sj + rj)) * (double)W + linu, e - A));
                w.mn = wmin(w.mn, wtag((La - (sj - rj)) * (double)W + linl, e - A));
                linu += du;
                linl += dw;
            }
        }
        return w;
    }
    // the deepest knots of a finished window, if deep enough and this lane's own (own index e = A + distance in 1 .. P)
    template <int W, int A>
    PTV_PIN_FN void win_take(Win w, bool valid, double thr, Mask &up, Mask &lo) const {
        const int eu = A + wdist(w.mx), el = A + wdist(w.mn);
        if (valid && w.mx > thr * (double)W && eu >= 1 && eu <= P) up |= (Mask)1 << (eu - 1);
        if (valid && -w.mn > thr * (double)W && el >= 1 && el <= P) lo |= (Mask)1 << (el - 1);
    }
    // Three stages, coarse to fine: 64-knot windows, then 16, then 4.  A stage runs only where the stage before it found knots in
    // enough lanes of the lane's wave (kSeedStage16 / kSeedStage4; the caller's ballot): features deep enough for a small window
    // show in the large ones around them, and data whose pieces are far longer than any window (lambda many times the noise) pay
    // for one stage, not three.
    // Each stage: the lane's parts -> the caller joins them with the neighbours' (lane shuffles on the device, arrays in the
    // host harness; what would cross a wave is left to the levels) -> the lane takes what is its own.
    // A window's threshold: the tube's width at its wider end (and a hair: rounding of the sums must not pin a knot ON the threshold).
    PTV_PIN_FN static double seed_threshold(double ra, double rb) { return 2.0000002 * (ra > rb ? ra : rb); }
    // -- 64 knots: four lanes a window on the plain grid (lanes 4 m ..) and on the shifted grid (lanes 4 m + 2 ..); lane q of the
    // four holds the distances 16 q + 1 .. 16 q + 16 from the window's start (the last one is the window's end: left out).
    template <class Sh>
    PTV_PIN_FN Win win64_part(const Sh &sh, double Sa, double ra, double Sb, double rb, int q) const {
        static_assert(P == 16, "window seeds: sixteen knots per lane");
        Win w{0.0, 0.0};
        if constexpr (!Sh::kWeighted) {
            const double dl = Sb - Sa;
            double lin = dl * (double)(16 * q + 1);
#pragma unroll
            for (int e = 1; e <= P; e++) {
                double v = wtag((Sa - sh.own(t, e - 1)) * 64.0 + lin, (16 * q + e) & 63);
                if (e == P) v = (q == 3) ? 0.0 : v;   // (distance 64: the window's own end)
                w.mx = pin_max(w.mx, v);
                w.mn = wmin(w.mn, v);
                lin += dl;
            }
        } else {
            const double Ua = Sa + ra, La = Sa - ra, du = (Sb + rb) - Ua, dw = (Sb - rb) - La;
            double linu = du * (double)(16 * q + 1), linl = dw * (double)(16 * q + 1);
#pragma unroll
            for (int e = 1; e <= P; e++) {
                const double sj = sh.own(t, e - 1), rj = sh.rown(t, e - 1);
                double vu = wtag((Ua - (sj + rj)) * 64.0 + linu, (16 * q + e) & 63), vl = wtag((La - (sj - rj)) * 64.0 + linl, (16 * q + e) & 63);
                if (e == P) {
                    vu = (q == 3) ? 0.0 : vu;
                    vl = (q == 3) ? 0.0 : vl;
                }
                w.mx = pin_max(w.mx, vu);
                w.mn = wmin(w.mn, vl);
                linu += du;
                linl += dw;
            }
        }
        return w;
    }
    // all: the joined parts of the window's four lanes; thr: the window's threshold; q: the lane's place among the four; first: the first
    // lane index such a window may start at (4 on the plain grid, 2 on the shifted one: the windows before touch the fibre's first knot)
    PTV_PIN_FN void win64_take(double thr, Win all, bool in_reach, int q, int first, Mask &up, Mask &lo) const {
        const int a = -16 * q;   // own index of the window's start
        const bool valid = in_reach && t - q >= first && a + 64 <= n - 1 - t * P;
        const int eu = a + wdist(all.mx), el = a + wdist(all.mn);
        if (valid && all.mx > thr * 64.0 && eu >= 1 && eu <= P) up |= (Mask)1 << (eu - 1);
        if (valid && -all.mn > thr * 64.0 && el >= 1 && el <= P) lo |= (Mask)1 << (el - 1);
    }
    // -- 16 knots: the lane's own window 0-16 ((Sl, rl): knot t P, the one before the lane's first); of the shifted grid the first
    // half of 8-24 ((Sfar, rfar): knot t P + 24) and the second half of (-8)-8 ((Sback, rback): knot t P - 8).  thr_tail / thr_head:
    // the thresholds of the two shared windows, for win16_take.
#define PTV_S_AT(e) ((e) == 0 ? Sl : sh.own(t, ((e) > 0 ? (e) : 1) - 1))   // (macros: the index stays a compile-time constant)
#define PTV_R_AT(e) ((e) == 0 ? rl : sh.rown(t, ((e) > 0 ? (e) : 1) - 1))
    template <class Sh>
    PTV_PIN_FN void win16_parts(const Sh &sh, double Sl, double rl, double Sfar, double rfar, double Sback, double rback, Mask &up, Mask &lo,
                                Win &tail, Win &head, double &thr_tail, double &thr_head) const {
        const Win w = win_part<16, 0, 1, 15>(sh, Sl, rl, PTV_S_AT(16), PTV_R_AT(16));
        win_take<16, 0>(w, t > 0 && 16 <= n - 1 - t * P, seed_threshold(rl, PTV_R_AT(16)), up, lo);   // (the knot before lane 0's first is the fibre end)
        tail = win_part<16, 8, 9, 16>(sh, PTV_S_AT(8), PTV_R_AT(8), Sfar, rfar);
        head = win_part<16, -8, 1, 7>(sh, Sback, rback, PTV_S_AT(8), PTV_R_AT(8));
        thr_tail = seed_threshold(PTV_R_AT(8), rfar);
        thr_head = seed_threshold(rback, PTV_R_AT(8));
    }
    PTV_PIN_FN void win16_take(Win tail, double thr_tail, Win next_head, bool has_next, Win prev_tail, Win head, double thr_head, bool has_prev,
                               Mask &up, Mask &lo) const {
        const int room = n - 1 - t * P;
        win_take<16, 8>(wjoin(tail, next_head), has_next && 24 <= room, thr_tail, up, lo);
        win_take<16, -8>(wjoin(prev_tail, head), has_prev && t * P >= 9 && 8 <= room, thr_head, up, lo);
    }
    // -- 4 knots: plain grid 0-4, 4-8, 8-12, 12-16; shifted 2-6, 6-10, 10-14, and 14-18, which the lane evaluates alone with the next
    // lane's first two knots ((Sr1, rr1), (Sr2, rr2): knots t P + 17, t P + 18); give: 1 / 2 = the NEXT lane's first knot touches the
    // upper / lower wall
    template <class Sh>
    PTV_PIN_FN void win4_all(const Sh &sh, double Sl, double rl, double Sr1, double rr1, double Sr2, double rr2, Mask &up, Mask &lo, int &give) const {
        const int room = n - 1 - t * P;
#define PTV_WIN4(A)                                                                                                        \
        {                                                                                                                  \
            const Win w = win_part<4, A, A + 1, A + 3>(sh, PTV_S_AT(A), PTV_R_AT(A), PTV_S_AT(A + 4), PTV_R_AT(A + 4));    \
            win_take<4, A>(w, (A > 0 || t > 0) && A + 4 <= room, seed_threshold(PTV_R_AT(A), PTV_R_AT(A + 4)), up, lo);     \
        }
        PTV_WIN4(0) PTV_WIN4(4) PTV_WIN4(8) PTV_WIN4(12) PTV_WIN4(2) PTV_WIN4(6) PTV_WIN4(10)
#undef PTV_WIN4
        const double Sa = PTV_S_AT(14), ra = PTV_R_AT(14);
        Win w = win_part<4, 14, 15, 16>(sh, Sa, ra, Sr2, rr2);
        if constexpr (!Sh::kWeighted) {
            const double v = wtag((Sa - Sr1) * 4.0 + (Sr2 - Sa) * 3.0, 3);
            w.mx = pin_max(w.mx, v);
            w.mn = wmin(w.mn, v);
        } else {
            const double Ua = Sa + ra, La = Sa - ra;
            w.mx = pin_max(w.mx, wtag((Ua - (Sr1 + rr1)) * 4.0 + ((Sr2 + rr2) - Ua) * 3.0, 3));
            w.mn = wmin(w.mn, wtag((La - (Sr1 - rr1)) * 4.0 + ((Sr2 - rr2) - La) * 3.0, 3));
        }
        const double thr = seed_threshold(ra, rr2);
        const bool valid = 18 <= room;
        win_take<4, 14>(w, valid, thr, up, lo);
        give = 0;
        if (valid && w.mx > thr * 4.0 && wdist(w.mx) == 3) give = 1;
        if (valid && -w.mn > thr * 4.0 && wdist(w.mn) == 3) give |= 2;
    }
#undef PTV_S_AT
#undef PTV_R_AT
    PTV_PIN_FN void win4_take(int prev_give, bool has_prev, Mask &up, Mask &lo) const {
        if (has_prev && (prev_give & 1)) up |= (Mask)1;
        if (has_prev && (prev_give & 2)) lo |= (Mask)1;
    }

    // ---- scan -------------------------------------------------------------------------------------------------------------
    // Written for full unrolling: k is a compile-time constant in every copy of the body, so a pin test is one bit test,
    // a knot's sum is read at a constant offset from the lane's part of the plane (sh.own), and the distances to the two
    // ends of the run are doubles counted up and down by one (exact) instead of converted integers.
    // The slots of the knots this lane pinned in the last update take the string's height there (S + r at the upper wall,
    // S - r at the lower): from now on a run that ends at such a knot reads its height with one load, no sign to work out.
    // Every lane runs this at the start of a level (final lanes included) and once more after the last one: nobody reads a
    // knot's sum after the level in which it was claimed (a pinned knot never violates, and place() looks at this level's
    // claims only).
    template <class Sh>
    PTV_PIN_FN void settle(Sh &sh) {
        Mask m = pendU | pendL;
        while (m) {
            const int k = ctz(m);
            m &= m - 1;
            const double w = Sh::kWeighted ? sh.rown(t, k) : sh.r(j0 + k);
            sh.set_own(t, k, ((pendL >> k) & 1) ? sh.own_at(t, k) - w : sh.own_at(t, k) + w);
        }
        pendU = pendL = 0;
    }

    template <class Sh>
    PTV_PIN_FN void scan(Sh &sh) {
        settle(sh);
        newU = newL = 0;
        eU = eL = xU = xL = 0.0;
        leaving = false;
        if (final_) return;
        sh.clear_knot(t + 1);
        if (t == 0) sh.clear_knot(0);
        const Mask pinned = pinU | pinL;
        int ca = la;
        double cha = hl;
        // end of the run that enters from the left: the lane's first pin, or the pin beyond its range
        int cb;
        double chb;
        if (pinned) {
            const int b = ctz(pinned);
            cb = j0 + b;
            chb = own_height(sh, b);
        } else {
            cb = rb;
            chb = hr;
        }
        // Violations are compared scaled by the segment's length D = cb - ca: D (c_j - S_j) = (cha - S_j)(cb - j) + (chb - S_j)(j - ca),
        // so a level divides nothing, and all lanes of a segment -- same ends, same heights -- compare like with like.
        double D = (double)(cb - ca);
        double da = (double)(cb - j0), db = (double)(j0 - ca);   // distances of the knot in hand to the run's two ends
        double rd = Sh::kWeighted ? 0.0 : sh.r(j0) * D;
        double bu = 0.0, bl = 0.0;   // largest violation of the run so far, per wall, tagged with its knot (PinTag)
        bool entering = true;
        const int cnt = j1 - j0;
#pragma unroll
        for (int k = 0; k < P; k++) {
            if (k < cnt) {   // (fewer than P knots: the fibre's last lane only)
                const int j = j0 + k;
                const bool is_pin = (pinned >> k) & 1;
                if (is_pin) {
                    // a run closes at this pin
                    if (entering) {
                        eU = bu; eL = bl;
                        entering = false;
                    } else {
                        if (bu > 0.0) newU |= (Mask)1 << PinTag<P>::index(bu);
                        if (bl > 0.0) newL |= (Mask)1 << PinTag<P>::index(bl);
                    }
                    ca = j;
                    cha = chb;   // (the run ended exactly here)
                    const Mask rest = (k + 1 < (int)(8 * sizeof(Mask))) ? (Mask)(pinned >> (k + 1)) : (Mask)0;
                    if (rest) {
                        const int b = k + 1 + ctz(rest);
                        cb = j0 + b;
                        chb = own_height(sh, b);
                    } else {
                        cb = rb;
                        chb = hr;
                    }
                    D = (double)(cb - ca);
                    da = D;
                    db = 0.0;
                    if (!Sh::kWeighted) rd = sh.r(j) * D;
                    bu = bl = 0.0;
                }
                // every knot is evaluated, pins included (straight-line code for the common case); a pin's own numbers
                // are replaced by "no violation"
                const double s = sh.own(t, k);
                const double q = (cha - s) * da + (chb - s) * db;
                const double wd = Sh::kWeighted ? sh.rown(t, k) * D : rd;
                const double vu = is_pin ? -1.0 : q - wd, vl = is_pin ? -1.0 : -q - wd;
                bu = pin_max(bu, PinTag<P>::tag(vu, k));
                bl = pin_max(bl, PinTag<P>::tag(vl, k));
                da -= 1.0;
                db += 1.0;
            }
        }
        // the last run ends at rb, beyond the lane's range
        if (entering) {
            eU = bu; eL = bl;
        } else {
            leaving = true;
            xU = bu; xL = bl;
        }
        eUk = j0 + PinTag<P>::index(eU); eLk = j0 + PinTag<P>::index(eL);
        xUk = j0 + PinTag<P>::index(xU); xLk = j0 + PinTag<P>::index(xL);
        eEnds = la + (pinned ? j0 + ctz(pinned) : rb);
        xEnds = ca + rb;   // (ca: the lane's last pin when a run leaves)
        const int se = slot_of(la);
        if (eU > 0.0) sh.post(0, se, eU);
        if (eL > 0.0) sh.post(1, se, eL);
        if (leaving) {
            if (xU > 0.0) sh.post(0, t + 1, xU);
            if (xL > 0.0) sh.post(1, t + 1, xL);
        }
    }

    // height of the string at the lane's own (settled) pin k
    template <class Sh>
    PTV_PIN_FN double own_height(const Sh &sh, int k) const { return sh.own_at(t, k); }

    // ---- claim ------------------------------------------------------------------------------------------------------------
    // Among the lanes that hold a segment's largest violation the one whose knot lies closest to the middle of the
    // segment wins (then the smaller knot): on data with exact ties -- stripes, checkerboards, staircases -- a fixed
    // preference for one end would peel one knot off a segment per level (n / 2 levels for +-a alternating samples),
    // the middle halves it.  Key: distance to the midpoint (doubled) in the high half, the knot in the low half.
    static constexpr int kKeyHalf = 4 * (int)sizeof(Key);   // bits of the knot (low) and of the distance (high)
    PTV_PIN_FN static Key claim_key(int k, int ends) {
        const int d = 2 * k - ends;
        return ((Key)(unsigned)(d < 0 ? -d : d) << kKeyHalf) | (Key)(unsigned)k;
    }
    PTV_PIN_FN static int claimed_knot(Key key) { return (int)(key & (((Key)1 << kKeyHalf) - 1)); }

    template <class Sh>
    PTV_PIN_FN void claim(Sh &sh) {
        if (final_) return;
        const int se = slot_of(la);
        if (eU > 0.0 && eU == sh.best(0, se)) sh.claim(0, se, claim_key(eUk, eEnds));
        if (eL > 0.0 && eL == sh.best(1, se)) sh.claim(1, se, claim_key(eLk, eEnds));
        if (leaving) {
            if (xU > 0.0 && xU == sh.best(0, t + 1)) sh.claim(0, t + 1, claim_key(xUk, xEnds));
            if (xL > 0.0 && xL == sh.best(1, t + 1)) sh.claim(1, t + 1, claim_key(xLk, xEnds));
        }
    }

    // ---- update: returns true when the lane gained a pin of its own ------------------------------------------------------------
    template <class Sh>
    PTV_PIN_FN bool update(Sh &sh) {
        if (final_) return false;
        sh.clear_best(t + 1);
        if (t == 0) sh.clear_best(0);
        bool moved = false;
        const int se = slot_of(la);   // (before la moves)
        for (int wall = 0; wall < 2; wall++) {
            const int ke = sh.knot(wall, se);
            if (ke >= 0) moved |= place(sh, ke, wall);
            if (leaving) {
                const int kx = sh.knot(wall, t + 1);
                if (kx >= 0) moved |= place(sh, kx, wall);
            }
        }
        const bool gained = (newU | newL) != 0;
        pinU |= newU;
        pinL |= newL;
        pendU |= newU;
        pendL |= newL;
        if (!moved && !gained) final_ = true;
        return gained;
    }

    template <class Sh>
    PTV_PIN_FN bool place(const Sh &sh, int k, int wall) {
        if (k < j0) {
            if (k > la) {
                la = k;
                hl = height(sh, k, wall != 0);
                return true;
            }
            return false;
        }
        if (k >= j1) {
            if (k < rb) {
                rb = k;
                hr = height(sh, k, wall != 0);
                return true;
            }
            return false;
        }
        if (wall) newL |= (Mask)1 << (k - j0);
        else      newU |= (Mask)1 << (k - j0);
        return true;
    }

    // ---- values: slope of the string over the lane's own samples i = tP + k (between knots i and i + 1), k < P ---------------
    // `mean` is what the caller subtracted from the samples before summing them.
    template <class Sh, class Put>
    PTV_PIN_FN void values(const Sh &sh, double mean, Put &&put) const {   // (after a last settle())
        const Mask pinned = pinU | pinL;
        const int i0 = t * P;
        if (i0 >= n) return;
        int ca = la;
        double cha = hl;
        int cb;
        double chb;
        if (pinned) {
            const int b = ctz(pinned);
            cb = j0 + b;
            chb = own_height(sh, b);
        } else {
            cb = rb;
            chb = hr;
        }
        double v = pin_div(chb - cha, (double)(cb - ca)) + mean;
        const int cnt = (i0 + P <= n ? P : n - i0);
#pragma unroll
        for (int k = 0; k < P; k++) {
            if (k < cnt) {
                const int i = i0 + k;   // knot i = j0 + k - 1: the lane's own knot k - 1
                if (k >= 1 && ((pinned >> (k - 1)) & 1)) {
                    ca = i;
                    cha = chb;
                    const Mask rest = (k < (int)(8 * sizeof(Mask))) ? (Mask)(pinned >> k) : (Mask)0;
                    if (rest) {
                        const int b = k + ctz(rest);
                        cb = j0 + b;
                        chb = own_height(sh, b);
                    } else {
                        cb = rb;
                        chb = hr;
                    }
                    v = pin_div(chb - cha, (double)(cb - ca)) + mean;
                }
                put(i, k, v);
            }
        }
    }
};

}  // namespace ptv
