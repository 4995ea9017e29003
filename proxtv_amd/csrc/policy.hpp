// policy.hpp -- which geometry a sweep family runs: pure host logic, no HIP (tests/host_harness.cpp drives it on the CPU).
//
// Geometry ladder.  The zone must be a few pieces long for a speculative walk to meet the true one, and piece length
// grows like (lambda / noise)^2:
//   0  LDS window, 16-sample zones            pieces of a few samples (the headline regime)
//   1  the same, robust instantiation         pieces of ~5 samples: walks may run past the window (global reads), failed
//                                             links are walked again inside the block (second-chance rounds)
//   2  LDS window, 64-sample zones            pieces of ~10 samples
//   3  the pinning solver (pin.hip) where it applies: exact and data-parallel inside the fibre, same cost whatever the
//      pieces; else global memory, zone 256 / chunk 64 (pieces of ~50 samples)
//   4  global memory, zone 1024 / chunk 256   pieces of a few hundred samples
//   5  one sequential walk per fibre          pieces comparable to the fibre: speculation cannot pay
//
// The policy is a hill climb on MEASURED sweep time with the repair kernel's counters as the hint for where to look:
//   * a solve starts by measuring one sweep at the incumbent mode (the family's next launch waits for the
//     measurement -- the host enqueues a whole solve long before the device finishes its first sweep, so a decision
//     needs a stall: one per family and solve in the steady state);
//   * many rewritten chunks -> trial of the next rung up (straight to the sequential walk if most chunks failed, then
//     back down); none at all -> trial of the next rung down; the fastest of the modes tried wins, and the climb
//     goes on in that direction while the counters say there is something to find;
//   * afterwards a sample (time + counters) is taken every few sweeps and looked at kMonitorLag sweeps later, when
//     the device has reached it but still has work queued (no bubble): the data of a solve drift -- DR iterates at
//     large lambda grow longer pieces sweep after sweep -- and a drift re-opens the exploration.
// The mode persists across solves; one-sweep solves (batched 1-D prox calls) explore across calls.
#pragma once

namespace ptv {

constexpr int kModeSeq = 5;
constexpr int kRounds = 4;          // second-chance rounds of mode 1
constexpr int kAlongMinLen = 160;   // dimension-0 sweeps take the along-fibre kernel from this fibre length on (16, 32 or 64 lanes share a segment of 17-sample chunks)
// rewritten-chunk fraction above which the next rung is worth a trial, per rung (from mode 0 the next rung is the same
// geometry made robust: a handful of repaired fibres per sweep already costs more than that), and below which the rung
// below is
constexpr double kTryUpAt[kModeSeq + 1] = {4e-6, 2e-4, 5e-4, 5e-4, 5e-4, 1.0};
constexpr double kCleanAt[kModeSeq + 1] = {1e-6, 1e-6, 5e-5, 5e-5, 5e-5, 1.0};
constexpr double kJump = 0.5;       // ... above which the trial goes straight to the sequential walk
constexpr double kBetter = 0.9;     // a trial wins if its sweep took less than this times the incumbent's
constexpr double kDrift = 1.5;      // steady state: re-explore when the sweep time moved by this factor
// The pinning rung has no repair counters; it reports the pieces per sample of its result instead.  Short pieces (at least
// kPinShort per sample) read as a clean sweep: the chunk kernels below are worth a trial.  Anything else reads as kPinHold,
// neither clean nor dirty: stay -- a trial of 64-sample zones on long pieces costs a hundred sweeps' time in repairs.
constexpr double kPinShort = 0.2;
constexpr double kPinHold = 1e-4;   // (between kCleanAt[3] and kTryUpAt[3])
constexpr double kPinMargin = 1.1;  // a chunk geometry measured slower than this times the pinning rung's sweep gives way to it
constexpr int kMonitorLag = 2;      // family sweeps between a steady-state sample and its evaluation
constexpr int kHoldSolves = 2;      // solves during which a rejected direction is not tried again
constexpr int kQuietSolves = 16;    // one-sweep solves between explorations

// Seeding.  Before a solve the input's edges |y_k - y_{k-1}| are sampled along every dimension it sweeps (pointwise.hip's
// edge_histogram); the fraction f of them above 4 lambda -- the edges at which the string is KNOWN to bend
// (chunkcore.hpp) -- says which rung the data want: on unit noise f = 0.78 / 0.40 / 0.16 / 0.05 / 0.005 at lambda = 0.1 /
// 0.3 / 0.5 / 0.7 / 1, where the pieces of a DR solve's iterates average 1.1 / 1.5 / 2.4 / 4 / 8 samples and a speculative
// walk meets the true one within 6 / 10 / 15 / 30 / 70 samples.  A second statistic guards against spatially uneven data:
// the total variation of sampled stretches of 16 edges -- where it is below 2 lambda the string runs (all but) flat, a
// speculative walk has nothing to meet the true walk at, and an average over the lively rest of the array would not tell.
// With option "deterministic" (the default) the rung of a
// sweep is this function of (input, lambda) and nothing else -- two runs on the same input take the same kernels and agree
// to the last bit; without it the hill climb below starts from the seed instead of exploring from scratch.
constexpr double kSeedNoisy = 0.45;   // f at or above: rung 0
constexpr double kSeedMid = 0.03;     // f at or above: rung 1 ; below: rung 3  (as sampled -- edges in the threshold's own eighth-octave bin do not count -- lambda = 0.7 on unit noise gives 0.034, 0.75-0.8 gives 0.022; at 0.8: 35.0 ms on rung 1, 34.0 on rung 3)
constexpr double kSeedPins = 0.001;    // rung 3: the pinning solver searches for knots known a priori when f is at least this (lambda = 1 on unit noise: 0.005,
                                       // 29.4 against 30.0 ms per 4096^2 DR solve; 0.8: 25.3 against 31.8; 3: none to find, 18.3 against 19.1 with the search)
constexpr double kSeedLongZones = 0.1;  // rung 3 on fibres of more than 16384 samples (the grid-wide pinning solver): 64-sample zones instead when at least this fraction
                                       // of the edges is above ONE penalty (one fibre of 4 M samples, unit noise, rung 2 / rung 3: lambda 1: 0.13 / 27 ms, 2: 0.18 / 26, 3: 36 / 26)
constexpr double kSeedJobs = 0.055;    // rung 1, option "repair_jobs" = 1: f below (as sampled: lambda >= 0.65 on unit noise -- 0.6 gives 0.065, 0.65 gives 0.048) the failed
                                       // links across workgroups go one lane each (4096^2 DR: lambda 0.6 14.27 -> 14.42 ms, 0.65 17.67 -> 17.33, 0.7 21.41 -> 19.97)
constexpr double kSeedRuns = 0.70;     // rung 0, dimension-0 sweeps, f at or above: interior segments are cut at the bends known a priori and solved run by run
                                       // (sweep_along_kernel RUNS): one pass of a wave takes 64 runs of three and more samples -- 44 per segment at
                                       // f = 0.78 (lambda = 0.1 on unit noise), 80 at 0.67 (0.15), where the speculative walk is the faster one again
constexpr double kSeedFlat = 0.02;     // more than this fraction of the sampled 16-edge stretches (all but) flat at lambda: rung 3
// DR2L1W: the second form of the iteration (ops.hpp, OP_DR_COL_V) pays below this certain fraction only -- the weighted column
// sweep is the heavier one to begin with (4096^2, weights U(0.5, 1.5) lambda: lambda = 0.4: 17.4 -> 18.1 ms, 0.6: 25.3 -> 24.0)
constexpr double kSeedDrFormWeighted = 0.2;
// ... of the operands of Dykstra sweeps, sampled mid-solve (sweep.hip: policy_reprobe): x + p walks like noisier data than its certain
// fraction says (4096^2 PD2 on unit noise, rung 1 / rung 3: lambda 0.5, f = 0.022: 20.4 / 28.5 ms; 0.6, f = 0.009: 23.8 / 29.3; 0.7, f = 0.001:
// 39.0 / 29.9 -- where a DR solve at f = 0.022 takes 31.2 / 20.0).  The operands of Yang's ADMM sweeps, X - U / rho, keep the general threshold:
// their certain fraction falls all through the solve (lambda = 2: 0.35 at iteration 3, 0.005 at 9, 0 at 33), and rung 1 at 0.005 on its way
// to 0 cost 73 ms a solve against 30.
constexpr double kSeedMidDykstra = 0.004;
// Small sweeps (up to kSmallSweep samples: 1024^2): a repair launch costs what it costs whatever the image, a sweep of the pinning solver a
// sixteenth of what it costs at 4096^2 -- the chunk kernels give way earlier (1024^2 unit noise, rung 1 / rung 3: DR at lambda = 0.7, f = 0.034:
// 6.0 / 3.3 ms; PD2 at lambda = 0.5, f = 0.022: 6.4 / 5.4 ms; DR at lambda = 0.5, f = 0.16: 2.4 / 3.1)
constexpr long kSmallSweep = 1L << 21;
constexpr double kSeedMidSmall = 0.06;
// Weighted sweeps: the robust tile of a weighted row sweep runs at half the unweighted one's occupancy (two LDS planes), and the pinning solver
// starts from knots known by windows there too: the chunk kernels give way earlier (4096^2 weighted DR, penalties U(0.5, 1.5) x s, rung 1 /
// rung 3: s = 0.6, f = 0.095: 21.5 / 29.8 ms; 0.7, f = 0.060: 29.4 / 30.9; 0.8, f = 0.038: 45.6 / 32.0)
constexpr double kSeedMidWeighted = 0.055;
inline int rung_from_certain_fraction(double f, bool dykstra = false, bool small = false, bool weighted = false) {
    const double mid = small ? kSeedMidSmall : (weighted ? kSeedMidWeighted : (dykstra ? kSeedMidDykstra : kSeedMid));
    return f >= kSeedNoisy ? 0 : (f >= mid ? 1 : 3);
}

// Mid-solve samples of the operands of Dykstra / ADMM sweeps (sweep.hip: policy_reprobe) are taken before the sweeps of these iterations
// (1-based) of a loop: 2, 3, 5, 9, 17, 33, ... -- Dykstra's operands settle within a few iterations; `steady`: and every fourth one from 9 on --
// ADMM's keep drifting
inline bool reprobe_at(int it, bool steady = false) { return it >= 2 && (((it - 1) & (it - 2)) == 0 || (steady && it > 9 && (it - 9) % 4 == 0)); }

struct GeometryPolicy {
    int mode = 0;            // incumbent geometry
    double t_mode = 0.0;     // ms of its last measured sweep
    int trial = -1;          // >= 0: geometry under trial
    int dir = 0;
    int best = 0;            // fastest geometry of the exploration under way, and its sweep time
    double best_t = 0.0;
    bool explore = true;
    int hold_up = 0, hold_down = 0, quiet = 0;
    long sweeps = 0;         // sweeps of this family since the solve started
    bool single = false;     // the previous solve was ONE sweep (batched 1-D prox calls): every call is timed, so that a call whose
                             // data got harder than the pinning rung's yardstick is the last slow one
    // the workload of the last sweep (a different shape = a different workload)
    bool weighted = false;   // no mode 2 for weighted sweeps (two LDS windows)
    int len = 0;             // no mode 4 below 1024 samples
    long count = 0;
    bool pin = false;        // rung 3 is the pinning solver: nothing above it is worth a trial
    double t_pin = 0.0;      // ... and what a sweep of this workload takes there (0: not measured yet): a yardstick whose
                             // cost barely depends on the data, so one cheap measurement tells when the chunk kernels lose
    int changes = 0;         // workload changes seen in this solve

    // (with a pinning rung the adaptive ladder is 0, 1, 3: 64-sample zones -- rung 2 -- only ever won where the pinning
    // solver now does as well, and a trial of it on long pieces is expensive; an explicitly chosen mode still gets it)
    bool available(int m, bool forced = false) const {
        return !(m == 2 && (weighted || (pin && !forced))) && !(m == 4 && (len < 1024 || (pin && !forced)));
    }
    int top() const { return pin ? 3 : kModeSeq; }   // where a trial jumps to when most chunks failed
    int up(int m) const {
        do m++; while (m < kModeSeq && !available(m));
        return m;
    }
    int down(int m) const {
        do m--; while (m > 0 && !available(m));
        return m;
    }
    void conclude() {
        trial = -1;
        explore = false;
    }

    // A sweep of this shape is about to be launched.  Returns true when the workload differs from the last one's
    // (explore afresh -- unless shapes keep alternating inside one solve: 4-D+ tensors share a family).
    bool workload(int len_, long count_, bool weighted_, bool pin_ = false, int seed = -1) {
        const bool fresh = (len != len_ || count != count_ || weighted != weighted_ || pin != pin_) && changes++ < 4;
        if (fresh) {
            explore = true;
            trial = -1;
            hold_up = hold_down = quiet = 0;
            t_pin = 0.0;
            // A new workload opens on the rung its input's edge statistics ask for (`seed`).  Unknown data (no statistics:
            // tiny problems): on the pinning rung where there is one -- its sweep costs the same whatever the pieces and it
            // reports how long they are, so the way down is taken only when it is worth it, whereas one sweep of 16-sample
            // zones on long pieces costs a hundred sweeps' time in repair walks.
            if (changes <= 1) {
                if (seed >= 0) mode = seed;
                else if (pin_) mode = 3;
            }
        }
        len = len_;
        count = count_;
        weighted = weighted_;
        pin = pin_;
        return fresh;
    }

    // geometry of the next sweep
    int choose() {
        if (!available(mode)) mode = up(mode);
        return (explore && trial >= 0) ? trial : mode;
    }
    // should the next sweep be measured?  (`pending`: a measurement is still in flight)
    bool wants_measurement(bool pending) const {
        return !pending && (explore || single || (sweeps > 0 && sweeps % (mode == 0 ? 8 : 4) == 0));
    }

    // one step of an exploration: the sweep just measured ran geometry r in t ms and had the fraction f of its chunks
    // rewritten by repair walks (f < 0: no counters, e.g. the sequential kernel)
    void step(int r, double t, double f) {
        const int rr = r < kModeSeq ? r : kModeSeq;
        const bool dirty = f > kTryUpAt[rr], clean = f < 0 || f <= kCleanAt[rr];
        int next = -1;
        if (trial < 0) {   // the incumbent: where to look, if anywhere
            t_mode = best_t = t;
            best = r;
            if (r < kModeSeq && dirty && hold_up == 0) {
                next = (f > kJump && top() > r) ? top() : up(r);
                dir = (next > up(r)) ? -1 : +1;   // skipped rungs on the way up: look at them from above
            } else if (r > 0 && clean && hold_down == 0) {
                next = down(r);
                dir = -1;
            }
        } else {           // a trial: remember the fastest, walk on while the counters say there is something to find
            // going up a trial must win clearly; going down a clean one only has to be no slower (noise on small
            // problems must not leave the policy on a heavier geometry than the data need)
            if (t < (dir < 0 && clean ? 1.05 : kBetter) * best_t) {
                best = r;
                best_t = t;
            }
            if (t < 3.0 * best_t) {
                if (dir > 0 && r < kModeSeq && dirty) next = up(r);
                if (dir < 0 && r > 0 && clean) next = down(r);
            }
        }
        if (next >= 0) {
            trial = next;
            return;
        }
        if (trial >= 0 && best == mode) {   // looked and found nothing: leave that direction alone for a while
            if (dir > 0) hold_up = kHoldSolves;
            else hold_down = single ? 4 * kHoldSolves : kHoldSolves;   // (one-sweep solves: a rejected look below costs a whole call)
        }
        mode = best;
        t_mode = best_t;
        conclude();
    }

    // steady-state sample of the incumbent
    void monitor(int r, double t, double f) {
        const int rr = r < kModeSeq ? r : kModeSeq;
        const bool dirty = f > kTryUpAt[rr], clean = f < 0 || f <= kCleanAt[rr];
        const bool slower = t_mode > 0 && t > kDrift * t_mode;
        const bool harder = r < kModeSeq && dirty && (hold_up == 0 || slower);
        const bool easier = r > 0 && clean && t_mode > 0 && t * kDrift < t_mode;
        if (harder) hold_up = 0;
        if (easier) hold_down = 0;
        if (harder || easier) {
            explore = true;
            trial = -1;
            step(r, t, f);
        }
    }

    // a measurement arrived: exploration step or steady-state sample
    void measured(int r, double t, double f) {
        if (pin) {
            if (r == 3) {
                f = (f >= kPinShort) ? 0.0 : kPinHold;
                t_pin = t;
            } else if (t_pin > 0.0 && t > kPinMargin * t_pin) {   // no need to explore: the yardstick is known and beats this
                mode = 3;
                t_mode = t_pin;
                hold_down = kHoldSolves;
                conclude();
                return;
            } else if (t_pin <= 0.0 && explore && trial < 0) {    // first look at this workload: take the yardstick next
                t_mode = best_t = t;
                best = r;
                trial = 3;
                dir = +1;
                return;
            }
        }
        // one-sweep solves on the pinning rung whose result has short pieces: look below now, not at the next periodic exploration
        if (pin && r == 3 && single && !explore && f == 0.0 && hold_down == 0) {
            explore = true;
            trial = -1;
        }
        if (explore) step(r, t, f);
        else monitor(r, t, f);
    }

    // a new solve starts (after the previous solve's last measurement, if any, went through measured())
    void begin_solve() {
        if (hold_up > 0) hold_up--;
        if (hold_down > 0) hold_down--;
        single = pin && sweeps == 1;
        if (sweeps > 1) {            // a real solve: every solve opens with a measured sweep of the incumbent
            explore = true;
            trial = -1;
        } else if (!explore) {       // one-sweep solves: an exploration every kQuietSolves calls
            if (quiet > 0) quiet--;
            else {
                explore = true;
                trial = -1;
                quiet = kQuietSolves;
            }
        }
        sweeps = 0;
        changes = 0;
    }
};

}  // namespace ptv
