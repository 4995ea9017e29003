// host_harness.cpp -- TEST-ONLY: compiles the device walker (proxtv_amd/csrc/walker.hpp) with g++ so that the
// state machine the HIP kernels run can be checked against the oracle on a machine without a GPU.
// Never linked into libproxtv_amd.so; built on demand by tests/test_walker_host.py into a temp directory.
#define PTV_HOST_TEST 1
#define __device__
#define __forceinline__ inline
#include <cstring>

#include "../proxtv_amd/csrc/policy.hpp"
#include "../proxtv_amd/csrc/walker.hpp"

using namespace ptv;

namespace {
struct HostSource {
    const double *yy;
    const double *ww;
    double *x;
    int stop_after;   // chunk emulation: stop once a piece reached this sample (<0: never)
    int bends = 0;
    bool done = false;
    double y(int i) const { return yy[i]; }
    double r(int i) const { return ww[i]; }
    void piece(int a, int b, double v) {
        for (int j = a; j <= b; j++) x[j] = v;
        if (stop_after >= 0 && b >= stop_after) done = true;
    }
    void bend(int, int) { bends++; }
    bool keep_going(int) const { return !done; }
    int limit() const { return 1 << 30; }
    void pump() const {}
    void flush() const {}
};
}  // namespace

extern "C" {

// full sequential walk, what sweep_seq_kernel does per lane
int host_walk(const double *y, const double *w, double lam, double *x, int n) {
    if (n <= 0) return 0;
    if (w && n == 1) { x[0] = y[0]; return 0; }
    HostSource s{y, w, x, -1};
    Walker wk;
    if (w) { walker_start<true>(wk, s, 0, lam); walker_run<true>(wk, s, n, lam); }
    else   { walker_start<false>(wk, s, 0, lam); walker_run<false>(wk, s, n, lam); }
    return s.bends;
}

// the same walk in blocks of K samples, what the global-memory kernels do per lane (walker_run_blocked)
int host_walk_blocked(const double *y, const double *w, double lam, double *x, int n, int from, int until) {
    if (n <= 0) return 0;
    if (w && n == 1) { x[0] = y[0]; return 0; }
    HostSource s{y, w, x, until};
    Walker wk;
    if (w) { walker_start<true>(wk, s, from, lam); walker_run_blocked<true, 8>(wk, s, n, lam); }
    else   { walker_start<false>(wk, s, from, lam); walker_run_blocked<false, 8>(wk, s, n, lam); }
    return s.bends;
}

// speculative walk: start at `from` as if the fibre began there, stop once sample `until` is covered.
// Writes only what it covers; returns the first sample index whose value is final-and-equal to the true walk
// if that can be told (the restart index of the first bend), else -1.
int host_walk_from(const double *y, const double *w, double lam, double *x, int n, int from, int until) {
    HostSource s{y, w, x, until};
    Walker wk;
    if (w) { walker_start<true>(wk, s, from, lam); walker_run<true>(wk, s, n, lam); }
    else   { walker_start<false>(wk, s, from, lam); walker_run<false>(wk, s, n, lam); }
    return s.bends;
}

// The geometry policy (policy.hpp) under a cost model: `cost` / `frac` hold, for two phases of a solve (before / from
// sweep `switch_at` on), the sweep time in ms and the rewritten-chunk fraction of each of the 6 modes (frac < 0: no
// counters).  The loop mirrors launch_chunk / chunk_stats_reset in sweep.hip: measurements are looked at before the
// family's next launch while exploring, kMonitorLag sweeps later in the steady state, and at the next solve's start.
// Returns the incumbent mode at the end; total_ms = time of the LAST solve; trace (if not null) = mode of every sweep
// of the last solve.
int policy_sim(const double *cost, const double *frac, int switch_at, int solves, int sweeps, int len, int weighted,
               int start_mode, double *total_ms, int *trace) {
    GeometryPolicy pl;
    pl.mode = start_mode;
    bool meas = false;
    int meas_mode = 0, meas_phase = 0;
    long meas_sweep = 0;
    double total = 0;
    for (int s = 0; s < solves; s++) {
        if (meas) {
            pl.measured(meas_mode, cost[6 * meas_phase + meas_mode], frac[6 * meas_phase + meas_mode]);
            meas = false;
        }
        pl.begin_solve();
        total = 0;
        for (int k = 0; k < sweeps; k++) {
            const int phase = k >= switch_at ? 1 : 0;
            if (pl.workload(len, 4096, weighted != 0)) meas = false;
            if (meas && (pl.explore || pl.sweeps - meas_sweep >= kMonitorLag)) {
                pl.measured(meas_mode, cost[6 * meas_phase + meas_mode], frac[6 * meas_phase + meas_mode]);
                meas = false;
            }
            const int mode = pl.choose();
            const bool measure = pl.wants_measurement(meas);
            total += cost[6 * phase + mode];
            if (trace) trace[k] = mode;
            pl.sweeps++;
            if (measure) {
                meas = true;
                meas_mode = mode;
                meas_phase = phase;
                meas_sweep = pl.sweeps;
            }
        }
    }
    if (total_ms) *total_ms = total;
    return pl.mode;
}
}
