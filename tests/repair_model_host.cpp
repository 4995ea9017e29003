// repair_model.cpp -- HOST-ONLY model (never part of the library) of the stage behind the chunk kernels: speculative chunk walks leave
// link codes and outputs; a repair pass finds the links that do not hold and re-walks from the last true bend.  Compiles the device
// walker (walker.hpp) with g++ like tests/host_harness.cpp.  Every chunk is its own "workgroup" here (all links are links across
// workgroups: the case the repair kernel's jump and the jobs repair are about).
//
// Four repairs of the same speculative state, each compared with the true prox of the fibre:
//   SEQ_OLD   the sequential scan that JUMPS to the next link in doubt and looks the starting bend up with an unbounded scan back
//             through the recorded codes (round 4 until its last commit)
//   SEQ_NEW   the same with the scan bounded by the chunk the jump started from (the bend in hand stays if nothing in between bent)
//   JOBS      one walk per failing link, all from the RECORDED codes, validity decided afterwards in order (sweep_repair_jobs_kernel);
//             what it declines goes to SEQ_NEW
//   JOBS_G    JOBS with the guard: a valid job must have started from a record at or behind the chunk that took the previous valid
//             walk over
#define PTV_HOST_TEST 1
#define __device__
#define __forceinline__ inline
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../proxtv_amd/csrc/walker.hpp"
#include "../proxtv_amd/csrc/chunkcore.hpp"

using namespace ptv;

namespace {
typedef unsigned link_t;
constexpr link_t kBad = 0xfffffffeu, kCertain = 0x80000000u, kFromStart = 1u;

struct Fibre {
    const double *y;
    const double *w;   // per-edge penalties (len - 1), or nullptr: lam on every edge
    int len, C, H, NC;
    double lam;
    std::vector<link_t> mine, next;   // as the chunk kernels publish them
    std::vector<double> spec;         // the speculative outputs
    std::vector<char> doubt;          // link INTO chunk c in doubt
};

// one chunk's speculative walk: from a bend known a priori, else from a free end H samples before the chunk; it owns the pieces that
// END inside the chunk -- their rows before the chunk too if its link is proven, else only its own rows (the rows between the last
// true bend and an unproven chunk belong to the repair walk) -- and stops when the piece over its last sample is closed
// g_legacy (unweighted): an UNPROVEN chunk values the first piece that ends in it the way the device's rebuild did until round 5 -- the
// closed form over its OWN rows only, (sum_{cs..to} y + h_to - h_mine) / (to - cs + 1) -- instead of the piece's value.  Harmless while
// the repair walk and the chunk's walk end that piece in the same place; test_repair_model_host.py shows that it is not with g_mirror.
int g_legacy = 0;
int g_table = 0;
struct SpecSource {
    const Fibre &f;
    double *x;       // nullptr: codes only
    int cs, ce;
    bool proven;
    link_t mine = 0, next = 0;
    bool done = false;
    int pend_to = -1;          // (g_legacy: the piece waits for the bend that ends it -- its type is part of the closed form)
    link_t began = 0;          // the bend the piece in hand began at (0: the walk's free end)
    double y(int i) const { return f.y[i]; }
    double r(int i) const { return f.w[i]; }
    // g_table: the quotient by a piece's span the way the device's chunk walks take it (walk_asm.hpp, table loops: ONE product with the
    // correctly rounded reciprocal) while the repair walks divide (walker.hpp over_span): the device's two roundings, on the host
    double over_span(double a, int span) const { return g_table ? a * (1.0 / (double)span) : a / span; }
    void legacy_flush(int end_type) {   // end_type < 0: the fibre's end
        if (pend_to < 0) return;
        double sum = 0.0;
        for (int k = cs; k <= pend_to; k++) sum += f.y[k];
        const double hprev = began ? ((began & 1u) ? f.lam : -f.lam) : 0.0;
        const double hk = end_type < 0 ? 0.0 : (end_type ? f.lam : -f.lam);
        const double v = (sum + (hk - hprev)) / (double)(pend_to - cs + 1);
        for (int k = cs; k <= pend_to; k++) x[k] = v;
        pend_to = -1;
    }
    void piece(int from, int to, double v) {
        if (x && to >= cs && to < ce) {
            if (g_legacy && !f.w && !proven && from < cs) pend_to = to;
            else for (int k = proven ? from : std::max(from, cs); k <= to; k++) x[k] = v;
        }
        if (to >= ce - 1) done = true;
    }
    void bend(int at, int type) {
        if (x) legacy_flush(type);
        const link_t code = ((link_t)at << 1) | (link_t)type;
        began = code;
        if (at <= cs) mine = code;
        if (at <= ce) next = code;
    }
    bool keep_going(int) const { return !done; }
};

// The same walk on the mirrored fibre scaled by three (-3 y, penalties x 3, bend types swapped): the same problem, so a valid walk of it
// -- but every product and quotient rounds differently (the walker by itself is mirror-symmetric: the scaling is what does it).  With
// g_mirror set the speculative chunk walks run like that while the repair walks do not: two walks that cut a fibre differently wherever
// the string touches the tube to the last bit, as the device's two walks do (walk_interior's table reciprocals against walker_run's
// quotients).  What the repairs join must agree to rounding all the same.
int g_mirror = 0;
constexpr double kMirrorScale = 3.0;
template <class S>
struct Mirrored {
    S &s;
    double y(int i) const { return -kMirrorScale * s.y(i); }
    double r(int i) const { return kMirrorScale * s.r(i); }
    void piece(int from, int to, double v) { s.piece(from, to, -v / kMirrorScale); }
    void bend(int at, int type) { s.bend(at, type ^ 1); }
    bool keep_going(int i) const { return s.keep_going(i); }
};

struct CertainWin {
    const double *p, *w;
    double y(int i) const { return p[i]; }
    double r(int i) const { return w[i]; }
};

template <bool W>
void speculate(Fibre &f) {
    f.mine.assign(f.NC, 0);
    f.next.assign(f.NC, 0);
    f.doubt.assign(f.NC, 0);
    f.spec.assign(f.len, 0.0);
    for (int pass = 0; pass < 2; pass++) {
    for (int c = 0; c < f.NC; c++) {
        const int cs = c * f.C, ce = std::min(cs + f.C, f.len);
        SpecSource s{f, pass ? f.spec.data() : nullptr, cs, ce, pass ? !f.doubt[c] : false};
        Walker w;
        bool certain = false;
        Mirrored<SpecSource> ms{s};
        // (EVERY chunk walk, the ones from the fibre's start too: on the device all lanes of a chunk kernel run the same instructions, and
        //  two of them that have bent at the same place are in the same state to the last bit -- walker_restart_with is walk_interior's
        //  post-bend state -- so they cut the fibre alike from there on.  A model that mixed roundings AMONG the chunk walks leaves rows
        //  unwritten between a chunk that starts at a bend known a priori and its predecessor; the device cannot.)
        const bool mirror = g_mirror != 0;
        if (cs - f.H <= 0) {
            if (mirror) walker_start<W>(w, ms, 0, kMirrorScale * f.lam);
            else walker_start<W>(w, s, 0, f.lam);
        } else {
            int type = 0;
            CertainWin win{f.y, f.w};
            const int cat = certain_bend_before<W, 14>(win, cs, f.len, f.lam, type);
            if (cat >= 0) {
                if (mirror) walker_restart<W>(w, ms, cat, type ^ 1, f.len, kMirrorScale * f.lam);
                else walker_restart<W>(w, s, cat, type, f.len, f.lam);
                s.mine = s.next = ((link_t)cat << 1) | (link_t)type;
                certain = true;
            } else if (mirror) {
                walker_start<W>(w, ms, cs - f.H, kMirrorScale * f.lam);
            } else {
                walker_start<W>(w, s, cs - f.H, f.lam);
            }
        }
        s.began = s.mine;   // (a start at a bend known a priori: the first piece began there)
        if (mirror) walker_run<W>(w, ms, f.len, kMirrorScale * f.lam);
        else walker_run<W>(w, s, f.len, f.lam);
        if (pass) s.legacy_flush(-1);
        if (pass) continue;
        f.mine[c] = certain ? (s.mine | kCertain) : s.mine;
        f.next[c] = s.next;
    }
    if (pass) break;
    for (int c = 1; c < f.NC; c++) {
        const link_t in = f.mine[c], out = f.next[c - 1];
        const bool certain = (in & kCertain) && in != kBad;
        f.doubt[c] = (c * f.C - f.H > 0 && !certain && (in == 0 || in != out));
    }
    }
}

// the walk of a repair: from `cur`, writing from that bend on, until a chunk's recorded start agrees with it at a boundary
struct RepairSrc {
    const Fibre &f;
    double *x;
    int wfrom = 0, boundary = 0;
    link_t last = 0;
    bool stop = false;
    int resume_chunk = 0;
    link_t resume_code = 0;
    int lo = 0, hi = 1 << 30;    // (jobs: the window; a walk that needs more aborts)
    bool abort = false;
    std::vector<std::pair<int, double>> parked;   // (jobs: outputs wait for the verdict)
    bool park = false;
    double y(int i) { if (i < lo || i >= hi) { abort = true; return 0.0; } return f.y[i]; }
    double r(int i) { if (i < lo || i >= hi) { abort = true; return 0.0; } return f.w[i]; }
    void begin(int chunk, link_t cur) {
        wfrom = cur ? (int)(cur >> 1) : 0;
        boundary = (chunk + 1) * f.C;
        last = cur;
        stop = false;
    }
    void piece(int from, int to, double v) {
        from = std::max(from, wfrom);
        if (from > to) return;
        if (to >= hi) { abort = true; return; }
        for (int k = from; k <= to; k++) {
            if (park) parked.emplace_back(k, v); else x[k] = v;
        }
    }
    void bend(int at, int type) {
        const link_t code = ((link_t)at << 1) | (link_t)type;
        while (!stop && boundary < f.len && at >= boundary) {
            const link_t here = (at == boundary) ? code : last;
            const int c = boundary / f.C;
            link_t m = f.mine[c];
            if (m != kBad) m &= ~kCertain;
            if (m != 0 && m == here) {
                stop = true;
                resume_chunk = c;
                resume_code = here;
            } else {
                boundary += f.C;
            }
        }
        last = code;
    }
    bool keep_going(int) const { return !stop && !abort; }
};

link_t last_bend_before(const Fibre &f, int chunk, int floor_chunk, int *from_chunk = nullptr) {
    for (int b = chunk - 1; b >= floor_chunk; b--)
        if (f.next[b] != 0) {
            if (from_chunk) *from_chunk = b;
            return f.next[b];
        }
    if (from_chunk) *from_chunk = -1;
    return 0;   // (none in range)
}

template <bool W>
void run_walk(const Fibre &f, RepairSrc &s, int chunk, link_t cur) {
    const link_t from = (cur == kFromStart) ? 0u : cur;
    s.begin(chunk, from);
    Walker w;
    if (cur == kFromStart) walker_start<W>(w, s, 0, f.lam);
    else walker_restart<W>(w, s, (int)(cur >> 1), (int)(cur & 1u), f.len, f.lam);
    walker_run<W>(w, s, f.len, f.lam);
}

// the sequential repair with the jump; `bounded`: the scan behind a jump stops at the chunk the jump started from.
// `skip`: fibres' chunks a jobs pass has dealt with are not in doubt any more (pass nullptr otherwise)
template <bool W>
int repair_seq(const Fibre &f, double *x, bool bounded, long *stale_reads) {
    int first = f.NC, lastbad = -1;
    for (int c = 1; c < f.NC; c++)
        if (f.doubt[c]) { first = std::min(first, c); lastbad = c; }
    if (lastbad < 0) return 0;
    auto next_suspect = [&](int c) {
        for (int b = c; b < f.NC; b++) if (f.doubt[b]) return b;
        return f.NC;
    };
    link_t cur = last_bend_before(f, first, 0);
    if (cur == 0) cur = kFromStart;
    int c = first, walks = 0;
    int rewritten_from = f.NC, rewritten_to = -1;   // chunks whose records a repair walk has made stale: [from, to)
    while (true) {
        bool rejected = false;
        while (c < f.NC && c <= lastbad && !rejected) {
            const int suspect = next_suspect(c);
            if (suspect > c) {
                if (suspect >= f.NC || suspect > lastbad) { c = suspect; break; }
                if (bounded) {
                    const link_t found = last_bend_before(f, suspect, c);
                    if (found) cur = found;
                } else {
                    int from_chunk = -1;
                    link_t found = last_bend_before(f, suspect, 0, &from_chunk);
                    if (found == 0) found = kFromStart;
                    if (from_chunk >= rewritten_from && from_chunk < rewritten_to && found != cur && stale_reads) ++*stale_reads;
                    cur = found;
                }
                c = suspect;
            }
            const link_t mraw = f.mine[c];
            const bool certain = (mraw & kCertain) && mraw != kBad;
            const link_t m = certain ? (mraw & ~kCertain) : mraw;
            const bool accept = (c * f.C - f.H <= 0 || certain) ? (m != kBad) : (m != 0 && m == cur);
            if (accept) {
                if (f.next[c] != 0) cur = f.next[c];
                c++;
            } else {
                rejected = true;
            }
        }
        if (c >= f.NC || !rejected) break;
        RepairSrc s{f, x};
        run_walk<W>(f, s, c, cur);
        walks++;
        if (!s.stop) break;
        rewritten_from = std::min(rewritten_from, c);
        rewritten_to = std::max(rewritten_to, s.resume_chunk);
        c = s.resume_chunk;
        cur = s.resume_code;
    }
    return walks;
}

// the jobs repair: returns false if it declines the fibre (nothing written then)
template <bool W>
bool repair_jobs(const Fibre &f, double *x, bool guard, int window, int max_jobs) {
    std::vector<int> X;
    for (int c = 1; c < f.NC; c++) if (f.doubt[c]) X.push_back(c);
    if (X.empty()) return true;
    if ((int)X.size() > max_jobs) return false;
    struct Job { int X, r, from_chunk; bool abort; std::vector<std::pair<int, double>> out; };
    std::vector<Job> jobs;
    for (int Xk : X) {
        Job j{Xk, f.NC, -1, false, {}};
        link_t cur = last_bend_before(f, Xk, 0, &j.from_chunk);
        if (cur == 0) cur = kFromStart;
        const link_t mine = f.mine[Xk];
        if (mine != 0 && mine != kBad && mine == cur) {
            j.r = Xk;
        } else {
            RepairSrc s{f, nullptr};
            s.park = true;
            const int at = (cur == kFromStart) ? 0 : (int)(cur >> 1);
            s.lo = std::max(0, at - 1);
            s.hi = std::min(f.len, s.lo + window);
            run_walk<W>(f, s, Xk, cur);
            j.abort = s.abort;
            j.r = s.stop ? s.resume_chunk : f.NC;
            j.out.swap(s.parked);
        }
        jobs.push_back(std::move(j));
    }
    int lastr = -1;
    std::vector<char> valid(jobs.size(), 0);
    for (size_t k = 0; k < jobs.size(); k++) {
        if (jobs[k].abort) return false;
        const bool vk = lastr < 0 || lastr <= jobs[k].X - 1;
        if (vk && guard && lastr >= 0 && jobs[k].from_chunk < lastr) return false;
        if (vk) lastr = jobs[k].r;
        valid[k] = vk;
    }
    for (size_t k = 0; k < jobs.size(); k++)
        if (valid[k]) for (auto &kv : jobs[k].out) x[kv.first] = kv.second;
    return true;
}

double worst_diff(const std::vector<double> &a, const std::vector<double> &b) {
    double w = 0.0;
    for (size_t k = 0; k < a.size(); k++) w = std::max(w, std::fabs(a[k] - b[k]));
    return w;
}

template <bool W>
int model_run(const double *Y, const double *Wt, int count, int len, double lam, int C, int H, int window, int max_jobs, long *out, double *worst) {
    int first_bad = -1;
    for (int j = 0; j < count; j++) {
        Fibre f{Y + (size_t)j * len, W ? Wt + (size_t)j * len : nullptr, len, C, H, (len + C - 1) / C, lam, {}, {}, {}, {}};
        speculate<W>(f);
        std::vector<double> truth(len);
        {
            struct Src {
                const double *yy, *ww; double *x;
                std::vector<link_t> codes;
                double y(int i) const { return yy[i]; }
                double r(int i) const { return ww[i]; }
                void piece(int a, int b, double v) { for (int k = a; k <= b; k++) x[k] = v; }
                void bend(int at, int type) { codes.push_back(((link_t)at << 1) | (link_t)type); }
                bool keep_going(int) const { return true; }
            } s{f.y, f.w, truth.data(), {}};
            Walker w;
            walker_start<W>(w, s, 0, lam);
            walker_run<W>(w, s, len, lam);
            // A first guess at why the repairs are sound -- "a chunk whose recorded start IS the true walk's last bend at or before it is never
            // in doubt" -- counted: it is FALSE (a predecessor's walk can still be off at its end where the successor's warm-up has already
            // met the true walk).  The reason that holds is about the codes: a proven chunk's `next` is never zero.
            size_t k = 0;
            link_t true_before = 0;
            for (int c = 1; c < f.NC; c++) {
                while (k < s.codes.size() && (int)(s.codes[k] >> 1) <= c * f.C) true_before = s.codes[k++];
                const link_t m = f.mine[c] == kBad ? kBad : (f.mine[c] & ~kCertain);
                if (f.doubt[c] && m != 0 && m != kBad && m == true_before) out[11]++;
            }
        }
        int doubts = 0;
        for (int c = 1; c < f.NC; c++) doubts += f.doubt[c];
        if (doubts == 0) {
            out[10] += worst_diff(f.spec, truth) <= 1e-9;
            continue;
        }
        out[0]++;
        out[1] += doubts;
        const double tol = 1e-9;
        {
            std::vector<double> x = f.spec;
            repair_seq<W>(f, x.data(), false, &out[9]);
            const double e = worst_diff(x, truth);
            worst[0] = std::max(worst[0], e);
            if (e > tol) { out[3]++; if (first_bad < 0) first_bad = j; }
        }
        {
            std::vector<double> x = f.spec;
            out[2] += repair_seq<W>(f, x.data(), true, nullptr);
            const double e = worst_diff(x, truth);
            worst[1] = std::max(worst[1], e);
            if (e > tol) out[4]++;
        }
        for (int g = 0; g < 2; g++) {
            std::vector<double> x = f.spec;
            if (!repair_jobs<W>(f, x.data(), g == 1, window, max_jobs)) {
                out[7 + g]++;
                repair_seq<W>(f, x.data(), true, nullptr);
            }
            const double e = worst_diff(x, truth);
            worst[2 + g] = std::max(worst[2 + g], e);
            if (e > tol) out[5 + g]++;
        }
    }
    return first_bad;
}
template <bool W>
int repair_state(Fibre &f, double *x, int which) {
    if (which == 0) return repair_seq<W>(f, x, true, nullptr);
    if (!repair_jobs<W>(f, x, true, 128, 4)) {
        repair_seq<W>(f, x, true, nullptr);
        return 0;
    }
    return 1;
}
}  // namespace

extern "C" {
// Y: count fibres of len samples.  Wt: per-edge penalties, count x len (the last of a fibre unused), or nullptr: lam on every edge.
// out[0..11]: fibres with a link in doubt ; links in doubt ; walks of SEQ_NEW ; fibres where SEQ_OLD / SEQ_NEW / JOBS / JOBS_G end wrong (4) ;
//             fibres JOBS / JOBS_G declined (2) ; stale records read by SEQ_OLD ; fibres where the speculation alone is already exact ;
//             chunks in doubt although their recorded start is the true bend (it happens)
// worst[0..3]: largest absolute error of the four repairs.  Returns the index of the first fibre SEQ_OLD gets wrong (-1: none).
// The repairs on a state that was NOT speculated here: outputs, codes and flags as tests/host_harness.cpp's chunk_fibre left them (the
// device's lane code: walk_interior, the links, rebuild_owned).  which: 0 = bounded sequential repair, 1 = jobs (what it declines: sequential).
// Returns the number of walks (sequential) or 1 / 0 (jobs took the fibre / declined).
// (wt: per-edge penalties, len - 1 of them, or nullptr: lam on every edge)
int model_repair_state(const double *y, const double *wt, int len, double lam, int C, int H, double *x, const unsigned *mine, const unsigned *next,
                       const char *bad, int which) {
    Fibre f{y, wt, len, C, H, (len + C - 1) / C, lam, {}, {}, {}, {}};
    f.mine.assign(mine, mine + f.NC);
    f.next.assign(next, next + f.NC);
    f.doubt.assign(bad, bad + f.NC);
    f.doubt[0] = 0;
    f.spec.assign(x, x + len);
    return wt ? repair_state<true>(f, x, which) : repair_state<false>(f, x, which);
}
void model_set_mirror(int on) { g_mirror = on; }
void model_set_legacy(int on) { g_legacy = on; }
void model_set_table(int on) { g_table = on; }
// one unweighted fibre laid open: the speculative outputs, the codes, the links in doubt, the bounded sequential repair's result
void model_one(const double *y, int len, double lam, int C, int H, double *spec, double *repaired, unsigned *mine, unsigned *next, char *doubt) {
    Fibre f{y, nullptr, len, C, H, (len + C - 1) / C, lam, {}, {}, {}, {}};
    speculate<false>(f);
    std::vector<double> x = f.spec;
    repair_seq<false>(f, x.data(), true, nullptr);
    for (int k = 0; k < len; k++) { spec[k] = f.spec[k]; repaired[k] = x[k]; }
    for (int c = 0; c < f.NC; c++) { mine[c] = f.mine[c]; next[c] = f.next[c]; doubt[c] = f.doubt[c]; }
}
int model_fibres(const double *Y, const double *Wt, int count, int len, double lam, int C, int H, int window, int max_jobs, long *out, double *worst) {
    return Wt ? model_run<true>(Y, Wt, count, len, lam, C, H, window, max_jobs, out, worst)
              : model_run<false>(Y, nullptr, count, len, lam, C, H, window, max_jobs, out, worst);
}
}
