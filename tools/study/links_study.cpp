// links_study.cpp -- HOST-ONLY design study (never part of the library): how a chunk's link could be proven without the
// walk past the chunk end.  Compiles the device walker (walker.hpp, chunkcore.hpp) with g++ like tests/host_harness.cpp.
//
// Today a lane walks until the piece that covers its chunk's last sample closes -- it must, to know its last bend at or
// before the chunk end, the code its successor's link is checked against -- and a wave pays the slowest of its 64 lanes.
// The alternative measured here: a lane stops the first moment its walk position reaches its chunk end, and the link of the
// NEXT lane is proven by comparing complete walker states at that moment (two walks in the same state at the same position
// are identical from there on).  Per chunk the study records the trips of both schemes, whether each scheme's link holds,
// and the trips the true walk spends inside the chunk (the cost of a second chance).
#define PTV_HOST_TEST 1
#define __device__
#define __forceinline__ inline
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../proxtv_amd/csrc/walker.hpp"
#include "../../proxtv_amd/csrc/chunkcore.hpp"

using namespace ptv;

namespace {
struct CountWin {   // the whole fibre as one window; every trip of walk_interior reads three samples (+ one per call)
    const double *p;
    mutable long reads = 0;
    double y(int i) const { reads++; return p[i]; }
    double r(int) const { return 0.0; }
};
struct BendLog {
    const double *yy;
    std::vector<unsigned> codes;
    double y(int i) const { return yy[i]; }
    double r(int) const { return 0.0; }
    void piece(int, int, double) {}
    void bend(int at, int type) { codes.push_back(((unsigned)at << 1) | (unsigned)type); }
    bool keep_going(int) const { return true; }
};
long trips_of(const CountWin &w, long reads_before, int calls) { return (w.reads - reads_before - calls) / 3; }
bool same(const Walker &a, const Walker &b) {
    return a.i == b.i && a.k0 == b.k0 && a.klo == b.klo && a.khi == b.khi && !memcmp(&a.lo, &b.lo, sizeof(double)) &&
           !memcmp(&a.hi, &b.hi, sizeof(double)) && !memcmp(&a.hlo, &b.hlo, sizeof(double)) && !memcmp(&a.hhi, &b.hhi, sizeof(double));
}
}  // namespace

extern "C" {
// out: 8 ints per chunk c (0 <= c < nchunks = len / C; chunks too close to either end of the fibre have out[8c] = -1):
//   [0] valid  [1] trips today (start .. piece over ce-1 closed)  [2] trips to the first moment the position reaches ce
//   [3] link proven today (codes)  [4] link proven by state at cs  [5] started at a bend known a priori
//   [6] trips of the TRUE walk between the moments it reaches cs and ce  [7] samples walked past ce today
int study_fibre(const double *y, int len, double lam, int C, int H, int look, int *out) {
    const int nchunks = len / C;
    for (int c = 0; c < nchunks; c++) out[8 * c] = -1;
    // the true walk: its bends, and its state / trip count at the first moment its position reaches each chunk boundary
    BendLog log{y, {}};
    {
        Walker w;
        walker_start<false>(w, log, 0, lam);
        walker_run<false>(w, log, len, lam);
    }
    std::vector<Walker> at_boundary((size_t)nchunks + 1);
    std::vector<long> trips_at((size_t)nchunks + 1, 0);
    {
        CountWin win{y};
        Walker w;
        walker_start<false>(w, win, 0, lam);
        ChunkRec rec;
        long total = 0;
        for (int c = 1; c <= nchunks && c * C < len - 1; c++) {
            const long before = win.reads;
            const bool ran = w.i < c * C;
            walk_interior<false>(w, rec, win, c * C, 0, len + 64, lam);
            if (ran) total += trips_of(win, before, 1);
            at_boundary[(size_t)c] = w;
            trips_at[(size_t)c] = total;
        }
    }
    auto true_code_at_or_before = [&](int pos) {
        unsigned best = 0;
        for (unsigned code : log.codes) {   // (emitted in increasing position)
            if ((int)(code >> 1) <= pos) best = code; else break;
        }
        return best;
    };
    int done = 0;
    for (int c = 1; c < nchunks; c++) {
        const int cs = c * C, ce = cs + C;
        if (cs - 64 < 1 || ce + 200 >= len - 1) continue;
        const int start = cs - H;
        CountWin win{y};
        int cat = -1, ctype = 0;
        if (look == 8) cat = certain_bend_before<false, 8>(win, cs, len, lam, ctype);
        else if (look == 14) cat = certain_bend_before<false, 14>(win, cs, len, lam, ctype);
        auto begin = [&](Walker &w, ChunkRec &rec) {
            if (cat >= 0) {
                walker_restart_with<false>(w, cat, ctype, len, lam, y[cat], 0.0, 0.0);
                rec.mine = rec.next = rec.last = ((unsigned)cat << 1) | (unsigned)ctype;
            } else {
                walker_start<false>(w, win, start, lam);
            }
        };
        // today's scheme
        Walker w;
        ChunkRec rec;
        begin(w, rec);
        long before = win.reads;
        walk_interior<false>(w, rec, win, len - 1, cs, ce, lam);
        const long trips_old = trips_of(win, before, 1);
        const unsigned truth = true_code_at_or_before(cs);
        const bool link_old = cat >= 0 || (rec.mine != 0 && rec.mine == truth);
        // the alternative: state at the first moment the position reaches cs, then on to ce
        Walker v;
        ChunkRec dummy;
        begin(v, dummy);
        before = win.reads;
        int calls = 0;
        if (v.i < cs) { walk_interior<false>(v, dummy, win, cs, 0, len + 64, lam); calls++; }
        const bool link_state = cat >= 0 || same(v, at_boundary[(size_t)c]);
        if (v.i < ce) { walk_interior<false>(v, dummy, win, ce, 0, len + 64, lam); calls++; }
        const long trips_new = trips_of(win, before, calls);
        int *o = out + 8 * c;
        o[0] = 1;
        o[1] = (int)trips_old;
        o[2] = (int)trips_new;
        o[3] = link_old;
        o[4] = link_state;
        o[5] = cat >= 0;
        o[6] = (int)(trips_at[(size_t)c + 1] - trips_at[(size_t)c]);
        o[7] = w.i - ce;
        done++;
    }
    return done;
}

// Temporal seeding (round 4): inside a splitting loop the input of a sweep changes little from one iteration to the next, so a
// lane that finds no bend known a priori may start its walk AT the last bend the PREVIOUS iteration's solution had at or before
// its chunk start (restart state, like a bend known a priori -- but a guess: the link must still be proven) instead of a free
// end H samples back.  y_prev / y_cur: the same fibre at two consecutive iterations.  back: how far before the chunk start a seed
// may lie (rows the window keeps before the chunk).
// out: 8 ints per chunk:  [0] valid  [1] trips today  [2] trips seeded  [3] link proven today  [4] link proven seeded
//   [5] started at a bend known a priori (both schemes alike)  [6] trips of the true walk inside the chunk  [7] a seed was available
int study_seeded(const double *y_prev, const double *y, int len, double lam, int C, int H, int look, int back, int *out) {
    const int nchunks = len / C;
    for (int c = 0; c < nchunks; c++) out[8 * c] = -1;
    BendLog prev{y_prev, {}}, log{y, {}};
    {
        Walker w;
        walker_start<false>(w, prev, 0, lam);
        walker_run<false>(w, prev, len, lam);
        walker_start<false>(w, log, 0, lam);
        walker_run<false>(w, log, len, lam);
    }
    std::vector<long> trips_at((size_t)nchunks + 1, 0);
    {
        CountWin win{y};
        Walker w;
        walker_start<false>(w, win, 0, lam);
        ChunkRec rec;
        long total = 0;
        for (int c = 1; c <= nchunks && c * C < len - 1; c++) {
            const long before = win.reads;
            const bool ran = w.i < c * C;
            walk_interior<false>(w, rec, win, c * C, 0, len + 64, lam);
            if (ran) total += trips_of(win, before, 1);
            trips_at[(size_t)c] = total;
        }
    }
    auto code_at_or_before = [](const std::vector<unsigned> &codes, int pos) {
        unsigned best = 0;
        for (unsigned code : codes) {
            if ((int)(code >> 1) <= pos) best = code; else break;
        }
        return best;
    };
    int done = 0;
    for (int c = 1; c < nchunks; c++) {
        const int cs = c * C, ce = cs + C;
        if (cs - 80 < 1 || ce + 200 >= len - 1) continue;
        CountWin win{y};
        int cat = -1, ctype = 0;
        if (look == 8) cat = certain_bend_before<false, 8>(win, cs, len, lam, ctype);
        else if (look == 14) cat = certain_bend_before<false, 14>(win, cs, len, lam, ctype);
        const unsigned truth = code_at_or_before(log.codes, cs);
        const unsigned seed = code_at_or_before(prev.codes, cs);
        const bool have_seed = seed != 0 && cs - (int)(seed >> 1) <= back;
        long trips[2];
        bool link[2];
        for (int scheme = 0; scheme < 2; scheme++) {
            Walker w;
            ChunkRec rec;
            if (cat >= 0) {
                walker_restart_with<false>(w, cat, ctype, len, lam, y[cat], 0.0, 0.0);
                rec.mine = rec.next = rec.last = ((unsigned)cat << 1) | (unsigned)ctype;
            } else if (scheme == 1 && have_seed) {
                const int at = (int)(seed >> 1);
                walker_restart_with<false>(w, at, (int)(seed & 1u), len, lam, y[at], 0.0, 0.0);
                rec.mine = rec.next = rec.last = seed;
            } else {
                walker_start<false>(w, win, cs - H, lam);
            }
            const long before = win.reads;
            walk_interior<false>(w, rec, win, len - 1, cs, ce, lam);
            trips[scheme] = trips_of(win, before, 1);
            link[scheme] = cat >= 0 || (rec.mine != 0 && rec.mine == truth);
        }
        int *o = out + 8 * c;
        o[0] = 1;
        o[1] = (int)trips[0];
        o[2] = (int)trips[1];
        o[3] = link[0];
        o[4] = link[1];
        o[5] = cat >= 0;
        o[6] = (int)(trips_at[(size_t)c + 1] - trips_at[(size_t)c]);
        o[7] = have_seed;
        done++;
    }
    return done;
}

// The true walk's trip counter at every bend: out_at[k] = restart sample, out_trip[k] = trips walked when that bend was made
// (the walk restarts there in closed form).  Returns the number of bends (at most cap are stored).
int study_bends(const double *y, int len, double lam, int *out_at, long *out_trip, int cap) {
    struct Log {
        const double *yy;
        int *at; long *trip; int cap, n = 0;
        mutable long trips = 0;
        double y(int i) const { return yy[i]; }
        double r(int) const { return 0.0; }
        void piece(int, int, double) {}
        void bend(int a, int) { if (n < cap) { at[n] = a; trip[n] = trips; } n++; }
        bool keep_going(int) const { trips++; return true; }
    } log{y, out_at, out_trip, cap};
    Walker w;
    walker_start<false>(w, log, 0, lam);
    walker_run<false>(w, log, len, lam);
    return log.n;
}

// sequential walk of one fibre (to generate DR iterates without the oracle)
void study_prox(const double *y, int n, double lam, double *x) {
    struct Src {
        const double *yy; double *x;
        double y(int i) const { return yy[i]; }
        double r(int) const { return 0.0; }
        void piece(int a, int b, double v) { for (int j = a; j <= b; j++) x[j] = v; }
        void bend(int, int) {}
        bool keep_going(int) const { return true; }
    } s{y, x};
    if (n == 1) { x[0] = y[0]; return; }
    Walker w;
    walker_start<false>(w, s, 0, lam);
    walker_run<false>(w, s, n, lam);
}
}
