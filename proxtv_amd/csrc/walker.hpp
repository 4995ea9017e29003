// walker.hpp -- the exact 1-D TV-L1 prox state machine as run by ONE wavefront lane.
//
// The prox  argmin_x 1/2||x-y||^2 + sum_i r_i |x_i - x_{i+1}|  is the derivative of the shortest path through the
// tube (cumsum(y) +- r).  The walker advances sample by sample keeping two candidate straight pieces that leave
// the last knot k0: `lo` hugging the tube floor and `hi` hugging the ceiling; hlo/hhi are their heights relative
// to the tube centre at the current sample, klo/khi the last samples where each touched its own wall.  When the
// low piece pokes through the ceiling (or the high one through the floor) the string must bend at klo (khi): the
// piece up to there is final, and the walk restarts at the next sample in closed form.
//
// Behaviourally this is the reference's linearized taut string (src/TVL1opt.cpp:359-564, identical arithmetic and
// operation order; weighted form src/TVL1Wopt.cpp:364-567), restructured as ONE loop with one sample per trip so
// that the 64 lanes of a wavefront, each owning a different fibre, execute the same instruction stream with
// predication instead of diverging into separate inner loops.
//
// The key structural fact used by the chunked kernels (sweep.hip): right after a bend the walker state depends
// only on (restart index, bend type) -- everything before is forgotten.
#pragma once

#ifndef PTV_HOST_TEST   // tests/host_harness.cpp compiles this header with g++ to check the logic without a GPU
#include <hip/hip_runtime.h>
#endif
#include <type_traits>
#include <utility>

namespace ptv {

constexpr double kEps = 1e-10;  // absolute tolerance of the last-sample tests (reference: src/general.h:64-67)

enum BendType : int { BEND_CEIL = 0, BEND_FLOOR = 1 };

// Per-lane walker registers.
struct Walker {
    double lo, hi, hlo, hhi;
    int i, k0, klo, khi;
};

// Source concept (duck-typed template parameter `S`):
//   double y(int i)            sample i of this lane's fibre
//   double r(int i)            per-edge half-width (weighted only; edge i joins samples i and i+1)
//   void   piece(int from, int to, double v)   samples from..to (inclusive) of the prox equal v
//   void   bend(int restart, int type)         a bend happened; the walk restarts at sample `restart`
//   bool   keep_going(int i)   polled once per trip (chunked kernels stop early; sequential returns true)

// A source may bring its own quotient by the span of a piece (a small positive integer): `double over_span(double a, int span)`.
template <class S, class = void>
struct SourceDivides : std::false_type {};
template <class S>
struct SourceDivides<S, std::void_t<decltype(std::declval<S &>().over_span(0.0, 0))>> : std::true_type {};
template <class S>
__device__ __forceinline__ double over_span(S &src, double a, int span) {
    if constexpr (SourceDivides<S>::value) return src.over_span(a, span);
    else return a / span;
}

// Start a walk at sample `at` as if the fibre began there (free left end: string height 0).
template <bool WEIGHTED, class S>
__device__ __forceinline__ void walker_start(Walker &w, S &src, int at, double lam) {
    const double y0 = src.y(at);
    const double r0 = WEIGHTED ? src.r(at) : lam;
    w.hlo = w.hhi = 0.0;
    w.lo = -r0 + y0;
    w.hi = r0 + y0;
    w.k0 = at - 1;
    w.klo = w.khi = at;
    w.i = at;
}

// Put the walker in the state it has right after a bend of `type` whose new piece starts at sample `at` (0 < at < n).
// This is the whole point of the chunked kernels: that state depends on nothing but (at, type).  For at < n-1 it is
// the closed-form first sample used by walker_run's interior branch (a bend met at the last sample, which restarts
// without stepping, reaches the very same state one trip later); for at == n-1 the no-step form applies.
// Operands: yn = y(at); weighted: wp = r(at-1), wc = r(at) (wc unused at the last sample).
template <bool WEIGHTED>
__device__ __forceinline__ void walker_restart_with(Walker &w, int at, int type, int n, double lam, double yn, double wp,
                                                    double wc) {
    const int last = n - 1;
    const bool cv = (type == BEND_CEIL);
    w.k0 = at - 1;
    w.klo = w.khi = at;
    if (at < last) {
        if (WEIGHTED) {
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
        w.i = at + 1;
    } else {
        if (WEIGHTED) {
            if (cv) { w.lo = yn + wp; w.hi = yn + wp; w.hhi = w.hlo = -wp; }
            else    { w.hi = yn - wp; w.lo = yn - wp; w.hhi = w.hlo = wp; }
        } else {
            if (cv) { w.lo = yn; w.hi = 2 * lam + yn; w.hhi = w.hlo = -lam; }
            else    { w.hi = yn; w.lo = 2 * (-lam) + yn; w.hhi = w.hlo = lam; }
        }
        w.i = at;
    }
}

template <bool WEIGHTED, class S>
__device__ __forceinline__ void walker_restart(Walker &w, S &src, int at, int type, int n, double lam) {
    const double yn = src.y(at);
    double wp = 0.0, wc = 0.0;
    if (WEIGHTED) {
        wp = src.r(at - 1);
        if (at < n - 1) wc = src.r(at);
    }
    walker_restart_with<WEIGHTED>(w, at, type, n, lam, yn, wp, wc);
}

// Run until the fibre end (sample n-1 closed) or until src.keep_going() says stop (SINGLE: at most one trip).
// Returns true when the fibre end was reached and the closing piece was emitted.
template <bool WEIGHTED, class S, bool SINGLE = false>
__device__ __forceinline__ bool walker_run(Walker &w, S &src, int n, double lam) {
    const int last = n - 1;
    while (w.i < n) {
        if (!src.keep_going(w.i)) return false;
        const int i = w.i;
        const double yi = src.y(i);
        const bool interior = i < last;
        const double r = interior ? (WEIGHTED ? src.r(i) : lam) : 0.0;

        // low piece vs ceiling, then (only if that held) high piece vs floor -- reference order
        w.hlo += w.lo - yi;
        const bool cv = interior ? (r < w.hlo) : (w.hlo > kEps);
        bool fv = false;
        if (!cv) {
            w.hhi += w.hi - yi;
            fv = interior ? (-r > w.hhi) : (w.hhi < -kEps);
        }

        if (cv || fv) {
            const int brk = cv ? w.klo : w.khi;
            src.piece(w.k0 + 1, brk, cv ? w.lo : w.hi);
            const int at = brk + 1;
            src.bend(at, cv ? BEND_CEIL : BEND_FLOOR);
            w.k0 = brk;
            w.klo = w.khi = at;
            w.i = at;
            if (at < n) {  // at == n only for negative lambda (reference reads y[n] there and exits)
                const double yn = (at == i) ? yi : src.y(at);
                if (interior) {
                    // closed-form first sample of the new piece, then step past it
                    if (WEIGHTED) {
                        const double wp = src.r(at - 1), wc = src.r(at);
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
                    w.i = at + 1;
                } else {
                    // bend detected at the last sample: restart WITHOUT stepping (may itself be the last sample)
                    if (WEIGHTED) {
                        const double wp = src.r(at - 1);
                        const double wc = (at == last) ? 0.0 : src.r(at);
                        if (cv) { w.lo = yn + wp - wc; w.hi = yn + wp + wc; w.hhi = w.hlo = -wp; }
                        else    { w.hi = yn - wp + wc; w.lo = yn - wp - wc; w.hhi = w.hlo = wp; }
                    } else {
                        if (cv) { w.lo = yn; w.hi = 2 * lam + yn; w.hhi = w.hlo = -lam; }
                        else    { w.hi = yn; w.lo = 2 * (-lam) + yn; w.hhi = w.hlo = lam; }
                    }
                }
            }
        } else {
            if (interior) {
                // pull the pieces back inside the tube where they left it
                const int span = i - w.k0;
                if (w.hhi >= r) {
                    w.hi += over_span(src, r - w.hhi, span);
                    w.hhi = r;
                    w.khi = i;
                }
                if (w.hlo <= -r) {
                    w.lo += over_span(src, -r - w.hlo, span);
                    w.hlo = -r;
                    w.klo = i;
                }
            } else {
                if (w.hlo <= 0) w.lo += over_span(src, -w.hlo, i - w.k0);
            }
            w.i = i + 1;
        }
        if (SINGLE) break;
    }
    if (SINGLE && w.i < n) return false;
    src.piece(w.k0 + 1, last, w.lo);
    return true;
}

// a / s for a small positive integer s held as a double.  Device: reciprocal (v_rcp_f64) + one Newton step, then one
// residual correction of the quotient -- agrees with IEEE division to the last bit in all but rare ties (at most one
// ulp off) at a third of the instructions of the full v_div_* sequence; both tube pieces share the reciprocal.
// The host harness divides.
struct SpanDiv {
    double s, inv;
#ifdef PTV_HOST_TEST
#ifdef PTV_TABLE_RECIP   // design study: the quotient as ONE product with the correctly rounded reciprocal (a table entry on the device)
    explicit SpanDiv(double s_) : s(s_), inv(1.0 / s_) {}
    double operator()(double a) const { return a * inv; }
#else
    explicit SpanDiv(double s_) : s(s_), inv(0.0) {}
    double operator()(double a) const { return a / s; }
#endif
#else
    __device__ __forceinline__ explicit SpanDiv(double s_) : s(s_) {
        const double x = __builtin_amdgcn_rcp(s_);
        inv = __builtin_fma(__builtin_fma(-s_, x, 1.0), x, x);
    }
    __device__ __forceinline__ double operator()(double a) const {
        const double q = a * inv;
        return __builtin_fma(__builtin_fma(-q, s, a), inv, q);
    }
#endif
};

// Same walk for sources that live in global memory, where every dependent load is a memory round trip and a wave
// has few neighbours to hide it behind (one lane per fibre: 64 waves for 4096 fibres).  The loop is software-
// pipelined: each trip of the outer loop first ISSUES the loads the next trip will use -- the K samples that follow
// the block in hand (the walk is predicted to run straight on) and the operands of the next K queued outputs
// (src.pump) -- and then works out of registers on the block loaded one trip earlier: the restart after the previous
// bend, up to K interior trips as straight-line predicated code.  A lane whose prediction failed (it bent and
// rewound, or just started) issues the right block and sits one trip out instead of stalling its wave for a round
// trip.  A bend only books the finished piece (src.piece queues it; it drains K samples per trip while the walk
// goes on) and the restart, whose sample arrives with the lane's next block.  The last sample of the fibre, with its
// own tests, goes through one trip of walker_run.  Same state machine and operation order as walker_run; the tube
// updates divide through SpanDiv.
// Extra source members:
//   int  limit()   the walk never processes a sample >= limit() (keep_going must refuse those)
//   void pump()    store the outputs whose operands were fetched one trip ago; fetch the operands of the next K
//   void flush()   write everything still queued (called before returning)
template <bool WEIGHTED, int K, class S>
__device__ __forceinline__ bool walker_run_blocked(Walker &w, S &src, int n, double lam) {
    const int last = n - 1;
    int rs = -1;                         // >= 0: a bend of this type happened and the walker has to restart at sample w.i
    double yn[K], rn[K], rnprev = 0.0;   // block being loaded, for samples nb .. nb + K - 1
    int nb = w.i;
#pragma unroll
    for (int u = 0; u < K; u++) {
        const int at = nb + u < last ? nb + u : last;
        yn[u] = src.y(at);
        if (WEIGHTED) rn[u] = src.r(at < last ? at : last - 1);
    }
    while (w.i < n) {
        if (!src.keep_going(w.i)) {
            src.flush();
            return false;
        }
        if (rs < 0 && w.i >= last) {
            if (walker_run<WEIGHTED, S, true>(w, src, n, lam)) {
                src.flush();
                return true;
            }
            continue;
        }
        // the block in hand, and the loads for the next one
        double yb[K], rb[K];
        const double rprev = rnprev;
        const int base = nb;
#pragma unroll
        for (int u = 0; u < K; u++) {
            yb[u] = yn[u];
            if (WEIGHTED) rb[u] = rn[u];
        }
        const bool useful = (base == w.i);
        nb = useful ? base + K : w.i;
        src.pump();
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int at = nb + u < last ? nb + u : last;
            yn[u] = src.y(at);
            if (WEIGHTED) rn[u] = src.r(at < last ? at : last - 1);
        }
        if (WEIGHTED) rnprev = src.r(nb > 0 ? (nb - 1 < last ? nb - 1 : last - 1) : 0);   // (nb may lie past the fibre end)
        if (!useful) continue;

        const int lim = last < src.limit() ? last : src.limit();   // interior trips handle i < lim
        if (rs >= 0) {   // base < last here: a bend found at an interior sample restarts at or before it
            walker_restart_with<WEIGHTED>(w, base, rs, n, lam, yb[0], rprev, rb[0]);
            rs = -1;
        }
        int pend = -1;
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int i = base + u;
            const bool act = (pend < 0) & (w.i == i) & (i < lim);
            const double yi = yb[u];
            const double r = WEIGHTED ? rb[u] : lam;
            const double h1 = w.hlo + (w.lo - yi);
            const bool cv = r < h1;
            const double h2 = w.hhi + (w.hi - yi);
            const bool fv = !cv & (-r > h2);
            const bool bent = act & (cv | fv);
            const bool adv = act & !(cv | fv);
            pend = bent ? (cv ? BEND_CEIL : BEND_FLOOR) : pend;
            const SpanDiv over((double)(i - w.k0));
            const bool th = adv & (h2 >= r), tl = adv & (h1 <= -r);
            const double nhi = w.hi + over(r - h2);
            const double nlo = w.lo + over(-r - h1);
            w.hi = th ? nhi : w.hi;
            w.hhi = th ? r : (adv ? h2 : w.hhi);
            w.khi = th ? i : w.khi;
            w.lo = tl ? nlo : w.lo;
            w.hlo = tl ? -r : (adv ? h1 : w.hlo);
            w.klo = tl ? i : w.klo;
            w.i = adv ? i + 1 : w.i;
        }
        if (pend >= 0) {
            const bool cv = (pend == BEND_CEIL);
            const int brk = cv ? w.klo : w.khi;
            src.piece(w.k0 + 1, brk, cv ? w.lo : w.hi);
            src.bend(brk + 1, pend);
            w.i = brk + 1;
            rs = pend;
        }
    }
    src.piece(w.k0 + 1, last, w.lo);
    src.flush();
    return true;
}

}  // namespace ptv
