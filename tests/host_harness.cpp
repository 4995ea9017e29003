// host_harness.cpp -- TEST-ONLY: compiles the device walker (proxtv_amd/csrc/walker.hpp) with g++ so that the
// state machine the HIP kernels run can be checked against the oracle on a machine without a GPU.
// Never linked into libproxtv_amd.so; built on demand by tests/test_walker_host.py into a temp directory.
#define PTV_HOST_TEST 1
#define __device__
#define __forceinline__ inline
#include <cstring>
#include <atomic>
#include <thread>
#include <pthread.h>

#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

#include "../proxtv_amd/csrc/policy.hpp"
#include "../proxtv_amd/csrc/walker.hpp"
#include "../proxtv_amd/csrc/chunkcore.hpp"
#include "../proxtv_amd/csrc/pincore.hpp"

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

namespace {
// the LDS window of one workgroup, one fibre's column of it: rows [lo, hi) of the fibre (+ one slack row)
struct HostWin {
    std::vector<double> yy, ww;
    std::vector<int> writes;
    int lo = 0, hi = 0;
    double y(int i) const { return yy[(size_t)(i - lo)]; }
    double r(int i) const { return ww[(size_t)(i - lo)]; }
    void put(int i, double v) {
        yy[(size_t)(i - lo)] = v;
        writes[(size_t)(i - lo)]++;
    }
};
struct HostFar {
    const double *y, *w;
    double far_y(int i) const { return y[i]; }
    double far_r(int i) const { return w[i]; }
};
// an op whose output depends on the input sample as well (like DR_COL): a row that is read after another lane replaced
// it, or replaced twice, shows
struct Reflect {
    static constexpr bool USES_Y = true;
    static double fuse(double y, double x) { return y - 2.0 * x; }
};
struct Identity {
    static constexpr bool USES_Y = false;
    static double fuse(double, double x) { return x; }
};

// What the chunk kernels leave for the repair kernel, per chunk (global index: block * NW + lane): the published codes and the flag.
// Filled by chunk_fibre when set (host_set_state_buffers): tests/test_sweep_end_to_end_host.py hands them, with the outputs, to the repair
// model (tests/repair_model_host.cpp model_repair_state) -- one sweep end to end on the host.
int g_rounds = 0;   // second-chance rounds inside a block (host_set_rounds; 0 = the plain instantiations)
unsigned *g_state_mine = nullptr, *g_state_next = nullptr;
char *g_state_bad = nullptr;
int g_state_cap = 0, g_state_C = 0;

template <bool WEIGHTED, bool PAST, class F, int C = 16>
int chunk_fibre(const double *y, const double *w, double lam, int len, int H, int T, int NW, int seed, double *x,
                int *first_bad, int *write_errors) {
    constexpr int LOOK = 8;
    std::mt19937 rng((unsigned)seed);
    const int Q = (len + NW * C - 1) / (NW * C);
    *first_bad = -1;
    *write_errors = 0;
    unsigned carried = 0;       // `next` code of the previous block's last lane
    bool carried_proven = false;   // ... and whether that lane's link was proven (second chances walk from proven predecessors only)
    int cur = 0;                // restart of the true walk's last bend so far (what the repair kernel would restart from)
    int bends = 0;
    for (int q = 0; q < Q; q++) {
        const int cs_wg = q * NW * C;
        const int ce_wg = std::min(len, cs_wg + NW * C);
        HostWin win;
        win.lo = std::max(0, cs_wg - H);
        win.hi = std::min(len, cs_wg + NW * C + T);
        win.yy.assign(y + win.lo, y + win.hi);
        win.yy.push_back(1e300);   // the slack row: never used for anything that matters
        win.yy.push_back(1e300);
        if (WEIGHTED) {
            win.ww.assign((size_t)(win.hi - win.lo) + 2, 0.0);
            for (int k = win.lo; k < win.hi && k < len - 1; k++) win.ww[(size_t)(k - win.lo)] = w[k];
        }
        win.writes.assign(win.yy.size(), 0);
        HostFar far{y, w};
        std::vector<ChunkRec> recs((size_t)NW);
        std::vector<int> starts((size_t)NW, 0);
        std::vector<char> has((size_t)NW, 0), certain((size_t)NW, 0), bad((size_t)NW, 0);
        for (int wave = 0; wave < NW; wave++) {
            const int cs = cs_wg + wave * C, ce = std::min(cs + C, len);
            if (cs >= len) continue;
            has[(size_t)wave] = 1;
            const int start = std::max(0, cs - H);
            starts[(size_t)wave] = start;
            ChunkRec &rec = recs[(size_t)wave];
            Walker wk;
            int cat = -1, ctype = 0;
            if (start > 0 && H <= 16 && lam > 0.0) cat = certain_bend_before<WEIGHTED, LOOK>(win, cs, len, lam, ctype);
            if (cat >= 0) {
                certain[(size_t)wave] = 1;
                walker_restart_with<WEIGHTED>(wk, cat, ctype, len, lam, win.y(cat), WEIGHTED ? win.r(cat - 1) : 0.0,
                                              (WEIGHTED && cat < len - 1) ? win.r(cat) : 0.0);
                rec.mine = rec.next = rec.last = ((unsigned)cat << 1) | (unsigned)ctype;
            } else {
                walker_start<WEIGHTED>(wk, win, start, lam);
            }
            walk_interior<WEIGHTED>(wk, rec, win, std::min(len - 1, win.hi), cs, ce, lam);
            TailSource<WEIGHTED, PAST, 48, HostWin, HostFar> tail{win, far, rec, cs, ce, win.hi, len};
            walker_run<WEIGHTED>(wk, tail, len, lam);
            if (rec.failed) rec.next = 0;
            bends += __builtin_popcount(rec.ends);
        }
        // links (kernel: through LDS inside the block, by the repair kernel between blocks)
        auto examine = [&]() {
            for (int wave = 0; wave < NW; wave++) {
                if (!has[(size_t)wave]) continue;
                const ChunkRec &rec = recs[(size_t)wave];
                const bool linked = !(starts[(size_t)wave] == 0 || certain[(size_t)wave]) && (wave > 0 || q > 0);
                const unsigned prev = wave > 0 ? recs[(size_t)wave - 1].next : carried;
                bad[(size_t)wave] = rec.failed || (linked && (rec.mine == 0 || rec.mine != prev));
            }
        };
        examine();
        // Second chances (the robust instantiations, plan.rounds > 0): a lane whose link fails while its predecessor's holds walks its chunk
        // again from the predecessor's last bend; links are looked at afresh after every round (sweep_chunk_kernel / sweep_along_kernel).
        for (int round = 0; round < g_rounds; round++) {
            bool any = false;
            for (int wave = 0; wave < NW; wave++) any = any || (has[(size_t)wave] && bad[(size_t)wave]);
            if (!any) break;
            std::vector<char> proven((size_t)NW, 0);
            for (int wave = 0; wave < NW; wave++) proven[(size_t)wave] = has[(size_t)wave] && !bad[(size_t)wave];
            std::vector<ChunkRec> again((size_t)NW);
            std::vector<char> walked((size_t)NW, 0);
            for (int wave = 0; wave < NW; wave++) {
                if (!has[(size_t)wave] || !bad[(size_t)wave] || !(wave > 0 || q > 0)) continue;
                const unsigned prev = wave > 0 ? recs[(size_t)wave - 1].next : carried;
                const bool prev_proven = wave > 0 ? proven[(size_t)wave - 1] != 0 : carried_proven;
                const int at = (int)(prev >> 1);
                if (!prev_proven || prev == 0 || at <= std::max(win.lo, 0)) continue;
                const int cs = cs_wg + wave * C, ce = std::min(cs + C, len);
                ChunkRec &rec = again[(size_t)wave];
                Walker wk;
                walker_restart_with<WEIGHTED>(wk, at, (int)(prev & 1u), len, lam, win.y(at), WEIGHTED ? win.r(at - 1) : 0.0,
                                              (WEIGHTED && at < len - 1) ? win.r(at) : 0.0);
                rec.mine = rec.next = rec.last = prev;
                walk_interior<WEIGHTED>(wk, rec, win, std::min(len - 1, win.hi), cs, ce, lam);
                TailSource<WEIGHTED, PAST, 48, HostWin, HostFar> tail{win, far, rec, cs, ce, win.hi, len};
                walker_run<WEIGHTED>(wk, tail, len, lam);
                walked[(size_t)wave] = !rec.failed;
            }
            for (int wave = 0; wave < NW; wave++)
                if (walked[(size_t)wave]) {
                    recs[(size_t)wave] = again[(size_t)wave];
                    certain[(size_t)wave] = 0;   // from now on the chunk hangs on its predecessor like any other
                }
            examine();
        }
        for (int wave = 0; wave < NW; wave++) {
            if (!has[(size_t)wave]) continue;
            if (bad[(size_t)wave] && *first_bad < 0) *first_bad = cur;
            if (*first_bad < 0 && recs[(size_t)wave].next != 0) cur = (int)(recs[(size_t)wave].next >> 1);
        }
        if (g_state_mine) {   // as sweep_chunk_kernel / sweep_along_kernel publish them
            g_state_C = C;
            for (int wave = 0; wave < NW; wave++) {
                const int c = q * NW + wave;
                if (!has[(size_t)wave] || c >= g_state_cap) continue;
                const ChunkRec &rec = recs[(size_t)wave];
                g_state_mine[c] = rec.failed ? kCodeBad : ((certain[(size_t)wave] && rec.mine != kCodeBad) ? (rec.mine | kCodeCertain) : rec.mine);
                g_state_next[c] = rec.failed ? 0u : rec.next;
                g_state_bad[c] = bad[(size_t)wave];
            }
        }
        // the rows before each chunk that belong to its first piece, summed before anything is replaced (chunkcore.hpp first_piece_prefix:
        // the tile kernels take them before the barrier behind the walks, the lanes of an along-fibre wave read them in lockstep)
        std::vector<PiecePrefix> pres((size_t)NW);
        for (int wave = 0; wave < NW; wave++)
            if (has[(size_t)wave] && !recs[(size_t)wave].failed)
                pres[(size_t)wave] = first_piece_prefix(win, recs[(size_t)wave], cs_wg + wave * C, starts[(size_t)wave]);
        // An UNPROVEN lane's own rows, up to the last piece that ends in its chunk, must be what a walk restarted at the lane's `mine` bend
        // produces: a repair walk whose last bend is that one hands over to the chunk and trusts them (sweep_kernels.hpp RepairBook::bend).
        // Checked on a copy of the window -- in the block itself lanes that are right overwrite rows of lanes that are wrong.
        for (int wave = 0; wave < NW; wave++) {
            const ChunkRec &rec = recs[(size_t)wave];
            if (!has[(size_t)wave] || !bad[(size_t)wave] || rec.failed || rec.mine == 0 || rec.ends == 0) continue;
            const int cs = cs_wg + wave * C, ce = std::min(cs + C, len);
            HostWin copy = win;
            rebuild_owned<Identity, WEIGHTED, C>(copy, rec, cs, ce, len, starts[(size_t)wave], false, cs_wg, wave == NW - 1 || ce == len, lam,
                                                 (const double *)nullptr, &pres[(size_t)wave]);
            const int e_last = cs + 31 - __builtin_clz(rec.ends);
            std::vector<double> xr((size_t)len, 0.0);
            HostSource src{y, w, xr.data(), e_last};
            Walker wk;
            walker_restart<WEIGHTED>(wk, src, (int)(rec.mine >> 1), (int)(rec.mine & 1u), len, lam);
            walker_run<WEIGHTED>(wk, src, len, lam);
            double scale = 1.0;
            for (int k = cs; k <= e_last; k++) scale = std::max(scale, std::fabs(y[k]));
            for (int k = cs; k <= e_last; k++)
                if (!(std::fabs(copy.y(k) - xr[(size_t)k]) <= 1e-10 * scale)) (*write_errors)++;   // (the fibre's last piece: closed form against the walker's kEps tests, ~1e-11)
            for (int k = win.lo; k < cs; k++)
                if (copy.writes[(size_t)(k - win.lo)] != 0) (*write_errors)++;   // ... and it keeps to its own rows
        }
        // rebuild, lanes in random order (they run concurrently on the device)
        std::vector<int> order;
        for (int wave = 0; wave < NW; wave++)
            if (has[(size_t)wave]) order.push_back(wave);
        std::shuffle(order.begin(), order.end(), rng);
        for (int wave : order) {
            const int cs = cs_wg + wave * C, ce = std::min(cs + C, len);
            // (a whole chunk strictly inside the fibre: the forms that skip the per-row range tests (FULL = 1) and that keep the chunk in
            //  registers (FULL = 2), as the kernels' interior blocks / segments run them -- in turn with the general form, so that every
            //  form meets every kind of neighbour)
            const int form = (ce - cs == C && ce <= len - 1) ? (q + wave) % 3 : 0;
            const PiecePrefix *pre = &pres[(size_t)wave];
            // zones longer than a chunk: a lane's writes stop at the nearest unproven chunk before it (GUARD in sweep_chunk_kernel, `wlo` in
            // sweep_along_kernel) -- there may be PROVEN chunks before that one, rows the repair kernel will not touch
            int wlo = cs_wg;
            if (H > C || g_rounds > 0)   // (second-chance walks reach back anywhere in the block)
                for (int k = wave - 1; k >= 0; k--)
                    if (bad[(size_t)k]) { wlo = cs_wg + k * C; break; }
            if (form == 2 && !WEIGHTED)
                rebuild_owned<F, WEIGHTED, C, 1, false, const double *, 0, 2>(win, recs[(size_t)wave], cs, ce, len, starts[(size_t)wave],
                                                                              !bad[(size_t)wave], wlo, wave == NW - 1 || ce == len, lam,
                                                                              (const double *)nullptr, pre);
            else if (form >= 1)
                rebuild_owned<F, WEIGHTED, C, 1, false, const double *, 0, 1>(win, recs[(size_t)wave], cs, ce, len, starts[(size_t)wave],
                                                                              !bad[(size_t)wave], wlo, wave == NW - 1 || ce == len, lam,
                                                                              (const double *)nullptr, pre);
            else
                rebuild_owned<F, WEIGHTED, C>(win, recs[(size_t)wave], cs, ce, len, starts[(size_t)wave], !bad[(size_t)wave], wlo,
                                              wave == NW - 1 || ce == len, lam, (const double *)nullptr, pre);
        }
        bool clean = true;
        for (int wave = 0; wave < NW; wave++) clean = clean && !bad[(size_t)wave];
        for (int k = cs_wg; k < ce_wg; k++) {
            // undo the reflection (exact only if the row was fused once, from its own sample)
            x[k] = F::USES_Y ? 0.5 * (y[k] - win.y(k)) : win.y(k);
            if (clean && *first_bad < 0 && win.writes[(size_t)(k - win.lo)] != 1) (*write_errors)++;
        }
        for (int k = win.lo; k < cs_wg; k++)
            if (win.writes[(size_t)(k - win.lo)] != 0) (*write_errors)++;
        for (int k = ce_wg; k < win.hi; k++)
            if (win.writes[(size_t)(k - win.lo)] != 0) (*write_errors)++;
        int lastw = NW - 1;
        while (lastw > 0 && !has[(size_t)lastw]) lastw--;
        carried = recs[(size_t)lastw].next;
        carried_proven = !bad[(size_t)lastw];
    }
    return bends;
}
}  // namespace

// The pinning solver (pincore.hpp) for one fibre, its group of lanes emulated one after the other: a group barrier is the
// end of a loop over lanes, an atomic is a plain max / min.  P = knots per lane (16, 32 or 64).  w (n - 1 edge penalties)
// may be null.  Returns the number of levels.
template <int P, bool W, class Key = unsigned>
struct PinHostShared {
    std::vector<double> s, rr;
    std::vector<double> mx[2];
    std::vector<Key> arg[2];
    static constexpr bool kWeighted = W;
    double S(int j) const { return s[(size_t)j]; }
    double r(int j) const { return rr[(size_t)j]; }
    double own(int t, int k) const { return s[(size_t)(1 + t * P + k)]; }
    double own_at(int t, int k) const { return own(t, k); }
    void set_own(int t, int k, double v) { s[(size_t)(1 + t * P + k)] = v; }
    double rown(int t, int k) const { return rr[(size_t)(1 + t * P + k)]; }
    void post(int wall, int slot, double v) { if (v > mx[wall][(size_t)slot]) mx[wall][(size_t)slot] = v; }
    double best(int wall, int slot) const { return mx[wall][(size_t)slot]; }
    void claim(int wall, int slot, Key key) { if (key < arg[wall][(size_t)slot]) arg[wall][(size_t)slot] = key; }
    int knot(int wall, int slot) const { return arg[wall][(size_t)slot] == (Key)~(Key)0 ? -1 : PinLane<P, Key>::claimed_knot(arg[wall][(size_t)slot]); }
    void clear_best(int slot) { mx[0][(size_t)slot] = mx[1][(size_t)slot] = 0.0; }
    void clear_knot(int slot) { arg[0][(size_t)slot] = arg[1][(size_t)slot] = (Key)~(Key)0; }
};

template <int P, bool W, class Key = unsigned>
static int pin_fibre(const double *y, const double *w, double lam, double *x, int n, int seeded = 0, int *nseeds = nullptr) {
    PinHostShared<P, W, Key> sh;
    const int lanes = (n + P - 1) / P;
    double mean = 0;
    for (int i = 0; i < n; i++) mean += y[i];
    mean /= n;
    sh.s.assign((size_t)(lanes + 64) * P + 2, 0.0);    // (a lane may read the slots of knots it does not have; the window seeds run whole waves)
    sh.rr.assign((size_t)(lanes + 64) * P + 2, 0.0);
    for (int i = 0; i < n; i++) sh.s[(size_t)i + 1] = sh.s[(size_t)i] + (y[i] - mean);
    for (int j = 1; j < n; j++) sh.rr[(size_t)j] = w ? w[j - 1] : lam;
    std::vector<PinLane<P, Key>> lane((size_t)lanes);
    (void)W;
    for (int t = 0; t < lanes; t++) lane[(size_t)t].init(n, t, sh);
    if (seeded) {   // knots known a priori (PinLane::seed; the device kernel's criterion: pin.hip)
        std::vector<int> wall((size_t)n + 1, -1);
        for (int j = 1; j < n; j++) {
            const double d = y[j] - y[j - 1];
            double thr = 4.0000001 * lam;
            bool ok = lam > 0.0;
            if (w) {
                const double rm = j > 1 ? w[j - 2] : 0.0, r0 = w[j - 1], rp = j + 1 < n ? w[j] : 0.0;
                thr = 1.0000001 * (rp + 2.0 * r0 + rm);
                ok = rm >= 0.0 && r0 > 0.0 && rp >= 0.0;
            }
            if (ok && std::fabs(d) > thr) wall[(size_t)j] = d > 0 ? 0 : 1;
        }
        // knots known by windows (pincore.hpp: win64 / win16 / win4): the lanes of a wave exchange their parts through arrays where the
        // kernel shuffles, a stage runs in the waves where the one before it found a knot (the kernel's ballots); what spans waves is
        // left out there, and here
        if constexpr (P == 16) {
            if (seeded >= 2 && (W || lam > 0.0)) {
                using Lane = PinLane<P, Key>;
                using Win = typename Lane::Win;
                using Mask = typename Lane::Mask;
                auto Sk = [&](int j) { return sh.S(j < 0 ? 0 : (j > n ? n : j)); };
                auto Rk = [&](int j) { return W ? sh.r(j < 1 ? 1 : (j > n - 1 ? n - 1 : j)) : lam; };
                const int padded = (lanes + 63) / 64 * 64;   // (lanes beyond the fibre run the same code on the device; their parts are never used)
                std::vector<Mask> ups((size_t)padded, 0), los((size_t)padded, 0), up16((size_t)padded, 0), lo16((size_t)padded, 0);
                std::vector<Lane> ghost((size_t)padded);
                std::vector<Win> pa((size_t)padded), pb((size_t)padded), tail((size_t)padded), head((size_t)padded);
                std::vector<double> thr_tail((size_t)padded, 0.0), thr_head((size_t)padded, 0.0);
                std::vector<int> give((size_t)padded, 0);
                for (int t = 0; t < padded; t++) {
                    ghost[(size_t)t].init(n, t, sh);
                    const int qa = t & 3, qb = (t + 2) & 3, a0 = (t - qa) * P, b0 = (t - qb) * P;
                    pa[(size_t)t] = ghost[(size_t)t].win64_part(sh, Sk(a0), Rk(a0), Sk(a0 + 64), Rk(a0 + 64), qa);
                    pb[(size_t)t] = ghost[(size_t)t].win64_part(sh, Sk(b0), Rk(b0), Sk(b0 + 64), Rk(b0 + 64), qb);
                }
                auto wave_has = [&](int base, const std::vector<Mask> &a, const std::vector<Mask> &b, int lanes_needed) {
                    int found = 0;
                    for (int l = 0; l < 64; l++) found += (a[(size_t)(base + l)] | b[(size_t)(base + l)]) != 0;
                    return found >= lanes_needed;
                };
                for (int t = 0; t < padded; t++) {   // 64 knots, the plain grid
                    const int l = t & 63, base = t - l, qa = t & 3, a0 = (t - qa) * P;
                    auto in_wave = [&](int lane_in_wave) { return (size_t)(base + (lane_in_wave & 63)); };
                    const int m = l ^ 2;   // (second step of the butterfly: the partner's joined pair)
                    const Win all_a = Lane::wjoin(Lane::wjoin(pa[in_wave(l)], pa[in_wave(l ^ 1)]), Lane::wjoin(pa[in_wave(m)], pa[in_wave(m ^ 1)]));
                    ghost[(size_t)t].win64_take(Lane::seed_threshold(Rk(a0), Rk(a0 + 64)), all_a, true, qa, 4, ups[(size_t)t], los[(size_t)t]);
                }
                for (int base = 0; base < padded; base += 64) {   // ... the shifted grid, in the waves where the plain one found a knot
                    if (!wave_has(base, ups, los, 1)) continue;
                    for (int t = base; t < base + 64; t++) {
                        const int l = t - base, qb = (t + 2) & 3, b0 = (t - qb) * P;
                        auto in_wave = [&](int lane_in_wave) { return (size_t)(base + (lane_in_wave & 63)); };
                        const int partner = qb < 2 ? l + 2 : l - 2, mb = partner & 63;
                        const Win all_b = Lane::wjoin(Lane::wjoin(pb[in_wave(l)], pb[in_wave(l ^ 1)]), Lane::wjoin(pb[in_wave(mb)], pb[in_wave(mb ^ 1)]));
                        ghost[(size_t)t].win64_take(Lane::seed_threshold(Rk(b0), Rk(b0 + 64)), all_b, partner >= 0 && partner < 64, qb, 2, ups[(size_t)t], los[(size_t)t]);
                    }
                }
                for (int base = 0; base < padded; base += 64) {
                    if (!wave_has(base, ups, los, kSeedStage16)) continue;
                    for (int t = base; t < base + 64; t++)
                        ghost[(size_t)t].win16_parts(sh, Sk(t * P), Rk(t * P), Sk(t * P + 24), Rk(t * P + 24), Sk(t * P - 8), Rk(t * P - 8), up16[(size_t)t], lo16[(size_t)t],
                                                     tail[(size_t)t], head[(size_t)t], thr_tail[(size_t)t], thr_head[(size_t)t]);
                    for (int t = base; t < base + 64; t++) {
                        const int l = t - base;
                        ghost[(size_t)t].win16_take(tail[(size_t)t], thr_tail[(size_t)t], head[(size_t)(base + ((l + 1) & 63))], l < 63, tail[(size_t)(base + ((l + 63) & 63))],
                                                    head[(size_t)t], thr_head[(size_t)t], l > 0, up16[(size_t)t], lo16[(size_t)t]);
                    }
                    const bool finer = wave_has(base, up16, lo16, kSeedStage4);
                    for (int t = base; t < base + 64; t++) {
                        ups[(size_t)t] |= up16[(size_t)t];
                        los[(size_t)t] |= lo16[(size_t)t];
                    }
                    if (!finer) continue;
                    for (int t = base; t < base + 64; t++)
                        ghost[(size_t)t].win4_all(sh, Sk(t * P), Rk(t * P), Sk(t * P + P + 1), Rk(t * P + P + 1), Sk(t * P + P + 2), Rk(t * P + P + 2), ups[(size_t)t], los[(size_t)t],
                                                  give[(size_t)t]);
                    for (int t = base; t < base + 64; t++) {
                        const int l = t - base;
                        ghost[(size_t)t].win4_take(give[(size_t)(base + ((l + 63) & 63))], l > 0, ups[(size_t)t], los[(size_t)t]);
                    }
                }
                for (int t = 0; t < lanes; t++) {
                    if (t * P >= n) continue;
                    for (int k = 0; k < P; k++) {
                        const int j = 1 + t * P + k;
                        if (j >= n || wall[(size_t)j] >= 0) continue;
                        if ((ups[(size_t)t] >> k) & 1) wall[(size_t)j] = 0;
                        else if ((los[(size_t)t] >> k) & 1) wall[(size_t)j] = 1;
                    }
                }
            }
        }
        if (nseeds) {
            *nseeds = 0;
            for (int j = 1; j < n; j++) *nseeds += wall[(size_t)j] >= 0;
        }
        for (int t = 0; t < lanes; t++) {
            PinLane<P, Key> &L = lane[(size_t)t];
            if (t * P >= n) continue;
            typename PinLane<P, Key>::Mask up = 0, lo = 0;
            for (int j = L.j0; j < L.j1; j++) {
                if (wall[(size_t)j] == 0) up |= (typename PinLane<P, Key>::Mask)1 << (j - L.j0);
                if (wall[(size_t)j] == 1) lo |= (typename PinLane<P, Key>::Mask)1 << (j - L.j0);
            }
            int la = 0, rb = n;
            double hl = sh.S(0), hr = sh.S(n);
            for (int j = L.j0 - 1; j >= 1; j--)
                if (wall[(size_t)j] >= 0) { la = j; hl = L.height(sh, j, wall[(size_t)j] != 0); break; }
            for (int j = L.j1; j < n; j++)
                if (wall[(size_t)j] >= 0) { rb = j; hr = L.height(sh, j, wall[(size_t)j] != 0); break; }
            L.seed(up, lo, la, hl, rb, hr);
        }
    }
    int levels = 0;
    for (int wall = 0; wall < 2; wall++) {   // one buffer of slots for all levels: the lanes clear what they own (pincore.hpp)
        sh.mx[wall].assign((size_t)lanes + 1, 0.0);
        sh.arg[wall].assign((size_t)lanes + 1, (Key)~(Key)0);
    }
    for (;;) {
        levels++;
        for (int t = 0; t < lanes; t++) lane[(size_t)t].scan(sh);
        for (int t = 0; t < lanes; t++) lane[(size_t)t].claim(sh);
        bool any = false;
        for (int t = 0; t < lanes; t++) any |= lane[(size_t)t].update(sh);
        if (!any) break;
    }
    for (int t = 0; t < lanes; t++) lane[(size_t)t].settle(sh);   // (all lanes, then a barrier, on the device)
    for (int t = 0; t < lanes; t++) lane[(size_t)t].values(sh, mean, [&](int i, int, double v) { x[i] = v; });
    return levels;
}


// The same protocol under REAL concurrency: the lanes are dealt to host threads, the three steps of a level are separated
// by thread barriers, and the slots are relaxed atomics (max by compare-and-swap on the bit pattern, min on the key) -- what
// the kernels do with ds_max_u64 / ds_min_u32 and global atomics.  The sequential emulation above cannot see an ordering
// mistake in the one-buffer slot protocol (who clears what, between which barriers); this one can.
template <int P, bool W>
struct PinThreadShared {
    using Key = unsigned long long;
    std::vector<double> s, rr;
    std::vector<std::atomic<unsigned long long>> mx[2];
    std::vector<std::atomic<Key>> arg[2];
    static constexpr bool kWeighted = W;
    double S(int j) const { return s[(size_t)j]; }
    double r(int j) const { return rr[(size_t)j]; }
    double own(int t, int k) const { return s[(size_t)(1 + t * P + k)]; }
    double own_at(int t, int k) const { return own(t, k); }
    void set_own(int t, int k, double v) { s[(size_t)(1 + t * P + k)] = v; }
    double rown(int t, int k) const { return rr[(size_t)(1 + t * P + k)]; }
    void post(int wall, int slot, double v) {
        const unsigned long long b = pin_bits(v);
        unsigned long long cur = mx[wall][(size_t)slot].load(std::memory_order_relaxed);
        while (cur < b && !mx[wall][(size_t)slot].compare_exchange_weak(cur, b, std::memory_order_relaxed)) {}
    }
    double best(int wall, int slot) const { return pin_double(mx[wall][(size_t)slot].load(std::memory_order_relaxed)); }
    void claim(int wall, int slot, Key key) {
        Key cur = arg[wall][(size_t)slot].load(std::memory_order_relaxed);
        while (key < cur && !arg[wall][(size_t)slot].compare_exchange_weak(cur, key, std::memory_order_relaxed)) {}
    }
    int knot(int wall, int slot) const {
        const Key key = arg[wall][(size_t)slot].load(std::memory_order_relaxed);
        return key == ~0ull ? -1 : PinLane<P, Key>::claimed_knot(key);
    }
    void clear_best(int slot) { mx[0][(size_t)slot].store(0, std::memory_order_relaxed); mx[1][(size_t)slot].store(0, std::memory_order_relaxed); }
    void clear_knot(int slot) { arg[0][(size_t)slot].store(~0ull, std::memory_order_relaxed); arg[1][(size_t)slot].store(~0ull, std::memory_order_relaxed); }
};

template <bool W>
static int pin_fibre_threads(const double *y, const double *w, double lam, double *x, int n, int nthreads) {
    constexpr int P = 16;
    PinThreadShared<P, W> sh;
    const int lanes = (n + P - 1) / P;
    double mean = 0;
    for (int i = 0; i < n; i++) mean += y[i];
    mean /= n;
    sh.s.assign((size_t)n + 1 + P, 0.0);
    sh.rr.assign((size_t)n + 1 + P, 0.0);
    for (int i = 0; i < n; i++) sh.s[(size_t)i + 1] = sh.s[(size_t)i] + (y[i] - mean);
    for (int j = 1; j < n; j++) sh.rr[(size_t)j] = w ? w[j - 1] : lam;
    for (int wall = 0; wall < 2; wall++) {
        sh.mx[wall] = std::vector<std::atomic<unsigned long long>>((size_t)lanes + 1);
        sh.arg[wall] = std::vector<std::atomic<unsigned long long>>((size_t)lanes + 1);
        for (int k = 0; k <= lanes; k++) {
            sh.mx[wall][(size_t)k].store(0);
            sh.arg[wall][(size_t)k].store(~0ull);
        }
    }
    std::vector<PinLane<P, unsigned long long>> lane((size_t)lanes);
    for (int t = 0; t < lanes; t++) lane[(size_t)t].init(n, t, sh);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, (unsigned)nthreads);
    std::atomic<int> gained[3];
    for (auto &g : gained) g.store(0);
    std::atomic<int> levels{0};
    auto body = [&](int tid) {
        // lanes are dealt round-robin, so that neighbouring lanes (which share segments and slots) run on different threads
        for (int level = 0;; level++) {
            if (tid == 0) gained[(level + 1) % 3].store(0, std::memory_order_relaxed);
            for (int t = tid; t < lanes; t += nthreads) lane[(size_t)t].scan(sh);
            pthread_barrier_wait(&bar);
            for (int t = tid; t < lanes; t += nthreads) lane[(size_t)t].claim(sh);
            pthread_barrier_wait(&bar);
            bool any = false;
            for (int t = tid; t < lanes; t += nthreads) any |= lane[(size_t)t].update(sh);
            if (any) gained[level % 3].store(1, std::memory_order_relaxed);
            pthread_barrier_wait(&bar);
            if (tid == 0) levels.store(level + 1);
            if (gained[level % 3].load(std::memory_order_relaxed) == 0) break;
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < nthreads; k++) th.emplace_back(body, k);
    for (auto &t : th) t.join();
    pthread_barrier_destroy(&bar);
    for (int t = 0; t < lanes; t++) lane[(size_t)t].settle(sh);   // (all lanes, then a barrier, on the device)
    for (int t = 0; t < lanes; t++) lane[(size_t)t].values(sh, mean, [&](int i, int, double v) { x[i] = v; });
    return levels.load();
}

extern "C" {

// One fibre through the speculative-chunk scheme exactly as a workgroup column does it (chunkcore.hpp): blocks of NW
// chunks of 16 samples, window [block - H, block + T), certain-bend starts, branch-free interior walk + walker_run
// tail, links, ownership rebuild with the lanes in random order.  first_bad = -1 when every link is proven, else the
// sample from which the repair kernel would rewrite (the restart of the last proven bend before the first unproven
// chunk): outputs before it are final.  write_errors counts rows of clean blocks that were not
// written exactly once, and rows outside a block that were written at all.
// (nullptr: off)  Returns nothing; host_state_chunk() tells the chunk length of the last run.
void host_set_state_buffers(unsigned *mine, unsigned *next, char *bad, int cap) {
    g_state_mine = mine; g_state_next = next; g_state_bad = bad; g_state_cap = cap;
}
int host_state_chunk() { return g_state_C; }
void host_set_rounds(int rounds) { g_rounds = rounds; }

int host_chunk_fibre(const double *y, const double *w, double lam, int len, int H, int T, int NW, int past, int seed,
                     double *x, int *first_bad, int *write_errors) {
    const bool refl = seed & 1;   // both rebuild flavours: outputs that depend on the row's own sample, and that do not
    if (!w && (NW == 64 || NW == 32 || NW == 16)) {   // the along-fibre kernel's geometries: 64 / 32 / 16 chunks per segment,
        // of 17 samples, or of 31 in the robust instantiation (64 look-ahead rows)
        if (T >= 64)
            return refl ? chunk_fibre<false, false, Reflect, 31>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors)
                        : chunk_fibre<false, false, Identity, 31>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors);
        return refl ? chunk_fibre<false, false, Reflect, 17>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors)
                    : chunk_fibre<false, false, Identity, 17>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors);
    }
    if (w && (NW == 64 || NW == 32 || NW == 16))   // the along-fibre kernel's weighted geometry: chunks of 9 samples (two LDS planes), zones of 16
        return refl ? chunk_fibre<true, false, Reflect, 9>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors)
                    : chunk_fibre<true, false, Identity, 9>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors);
    if (w) {
        if (refl) return past ? chunk_fibre<true, true, Reflect>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors)
                              : chunk_fibre<true, false, Reflect>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors);
        return past ? chunk_fibre<true, true, Identity>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors)
                    : chunk_fibre<true, false, Identity>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors);
    }
    if (refl) return past ? chunk_fibre<false, true, Reflect>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors)
                          : chunk_fibre<false, false, Reflect>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors);
    return past ? chunk_fibre<false, true, Identity>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors)
                : chunk_fibre<false, false, Identity>(y, w, lam, len, H, T, NW, seed, x, first_bad, write_errors);
}

// One short fibre the way sweep_whole_kernel does it (whole fibre in the window, 32 samples of outputs at a time, each
// 32 restarting at the last bend at or before its first sample).
int host_whole_fibre(const double *y, double lam, int len, int reflect, double *x) {
    constexpr int C = 32;
    HostWin win;
    win.lo = 0;
    win.hi = len;
    win.yy.assign(y, y + len);
    win.yy.push_back(1e300);
    win.yy.push_back(1e300);
    win.writes.assign(win.yy.size(), 0);
    HostFar far{y, nullptr};
    unsigned carry = 0;
    for (int cs = 0; cs < len; cs += C) {
        const int ce = std::min(cs + C, len);
        ChunkRec rec;
        Walker wk;
        int start = 0;
        if (carry != 0) {
            start = (int)(carry >> 1);
            walker_restart_with<false>(wk, start, (int)(carry & 1u), len, lam, win.y(start), 0.0, 0.0);
            rec.mine = rec.next = rec.last = carry;
        } else {
            walker_start<false>(wk, win, 0, lam);
        }
        walk_interior<false>(wk, rec, win, std::min(len - 1, win.hi), cs, ce, lam);
        TailSource<false, false, 48, HostWin, HostFar> tail{win, far, rec, cs, ce, win.hi, len};
        walker_run<false>(wk, tail, len, lam);
        if (reflect) rebuild_owned<Reflect, false, C>(win, rec, cs, ce, len, start, true, 0, ce == len, lam);
        else         rebuild_owned<Identity, false, C>(win, rec, cs, ce, len, start, true, 0, ce == len, lam);
        carry = rec.next;
    }
    int bad = 0;
    for (int k = 0; k < len; k++) {
        x[k] = reflect ? 0.5 * (y[k] - win.y(k)) : win.y(k);
        if (win.writes[(size_t)k] != 1) bad++;
    }
    return bad;
}

// One call of the interior walk of a chunk (chunkcore.hpp: walk_interior -- the specification of the assembly loops in
// walk_asm.hpp) on a plain array standing for the LDS window: state in `wk` (lo, hi, hlo, hhi) / `wi` (i, k0, klo, khi) and
// `rc` (ends, types, mine, next, last, done), in and out.  tests/test_walk_asm_emulated.py interprets the assembly text lane
// by lane and compares.  w (may be null): per-edge penalties.
void host_walk_interior(const double *y, const double *w, int n, int lo, int lim, int cs, int ce, double lam, double *wk, int *wi,
                        unsigned *rc) {
    HostWin win;
    win.lo = lo;
    win.hi = n;
    win.yy.assign(y + lo, y + n);
    win.yy.push_back(1e300);
    win.yy.push_back(1e300);
    if (w) {
        win.ww.assign((size_t)(n - lo) + 2, 0.0);
        for (int k = lo; k < n - 1; k++) win.ww[(size_t)(k - lo)] = w[k];
    }
    Walker wa;
    wa.lo = wk[0]; wa.hi = wk[1]; wa.hlo = wk[2]; wa.hhi = wk[3];
    wa.i = wi[0]; wa.k0 = wi[1]; wa.klo = wi[2]; wa.khi = wi[3];
    ChunkRec rec;
    rec.ends = rc[0]; rec.types = rc[1]; rec.mine = rc[2]; rec.next = rc[3]; rec.last = rc[4]; rec.done = rc[5] != 0;
    if (w) walk_interior<true>(wa, rec, win, lim, cs, ce, lam);
    else   walk_interior<false>(wa, rec, win, lim, cs, ce, lam);
    wk[0] = wa.lo; wk[1] = wa.hi; wk[2] = wa.hlo; wk[3] = wa.hhi;
    wi[0] = wa.i; wi[1] = wa.k0; wi[2] = wa.klo; wi[3] = wa.khi;
    rc[0] = rec.ends; rc[1] = rec.types; rc[2] = rec.mine; rc[3] = rec.next; rc[4] = rec.last; rc[5] = rec.done ? 1u : 0u;
}

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
// of the last solve.  pin: rung 3 is the pinning solver (its `frac` entry = pieces per sample of its result).
int policy_sim_pin(const double *cost, const double *frac, int switch_at, int solves, int sweeps, int len, int weighted,
                   int start_mode, int pin, double *total_ms, int *trace) {
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
            const int phase = (sweeps == 1 ? s : k) >= switch_at ? 1 : 0;   // (one-sweep solves: the data change at solve `switch_at`)
            if (pl.workload(len, 4096, weighted != 0, pin != 0)) meas = false;
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

// concurrent emulation (pthread barriers, atomic slots): same result as the sequential one, bit for bit
int host_pin_fibre_threads(const double *y, const double *w, double lam, double *x, int n, int nthreads) {
    return w ? pin_fibre_threads<true>(y, w, lam, x, n, nthreads) : pin_fibre_threads<false>(y, w, lam, x, n, nthreads);
}

// the same with 64-bit claim keys: fibres longer than one workgroup's LDS (the lanes of pinlong.hip's grid are emulated
// exactly like the lanes of a workgroup: the protocol is the same, its barriers span the grid)
int host_pin_fibre_long(const double *y, const double *w, double lam, double *x, int n) {
    return w ? pin_fibre<16, true, unsigned long long>(y, w, lam, x, n) : pin_fibre<16, false, unsigned long long>(y, w, lam, x, n);
}

// the same, starting from the knots known a priori
// seeded with the knots known by windows as well (sixteen knots a lane, one penalty: the kernel's option pin_seed = 2); *nseeds: how many knots
// the levels started from
int host_pin_fibre_windows(const double *y, double lam, double *x, int n, int *nseeds) {
    return pin_fibre<16, false>(y, nullptr, lam, x, n, 2, nseeds);
}
// ... with per-edge penalties w (n - 1 of them)
int host_pin_fibre_windows_weighted(const double *y, const double *w, double *x, int n, int *nseeds) {
    return pin_fibre<16, true>(y, w, 0.0, x, n, 2, nseeds);
}

int host_pin_fibre_seeded(const double *y, const double *w, double lam, double *x, int n, int P) {
    if (P == 16) return w ? pin_fibre<16, true>(y, w, lam, x, n, 1) : pin_fibre<16, false>(y, w, lam, x, n, 1);
    if (P == 32) return w ? pin_fibre<32, true>(y, w, lam, x, n, 1) : pin_fibre<32, false>(y, w, lam, x, n, 1);
    if (P == 64) return w ? pin_fibre<64, true>(y, w, lam, x, n, 1) : pin_fibre<64, false>(y, w, lam, x, n, 1);
    if (P == 4) return w ? pin_fibre<4, true>(y, w, lam, x, n, 1) : pin_fibre<4, false>(y, w, lam, x, n, 1);
    return -1;
}

int host_pin_fibre(const double *y, const double *w, double lam, double *x, int n, int P) {
    if (P == 16) return w ? pin_fibre<16, true>(y, w, lam, x, n) : pin_fibre<16, false>(y, w, lam, x, n);
    if (P == 32) return w ? pin_fibre<32, true>(y, w, lam, x, n) : pin_fibre<32, false>(y, w, lam, x, n);
    if (P == 64) return w ? pin_fibre<64, true>(y, w, lam, x, n) : pin_fibre<64, false>(y, w, lam, x, n);
    if (P == 4) return w ? pin_fibre<4, true>(y, w, lam, x, n) : pin_fibre<4, false>(y, w, lam, x, n);
    return -1;
}

// the seeded policy's pure functions (policy.hpp): rung of a sweep from the certain fraction of its (sampled) input; iterations before
// whose sweeps a Dykstra / ADMM loop samples its operands again
int policy_rung(double f, int dykstra, int small, int weighted) { return rung_from_certain_fraction(f, dykstra != 0, small != 0, weighted != 0); }
int policy_reprobe_at(int it, int steady) { return reprobe_at(it, steady != 0) ? 1 : 0; }

// (the same without a pinning rung: rung 3 is the global-memory chunk kernel)
int policy_sim(const double *cost, const double *frac, int switch_at, int solves, int sweeps, int len, int weighted,
               int start_mode, double *total_ms, int *trace) {
    return policy_sim_pin(cost, frac, switch_at, solves, sweeps, len, weighted, start_mode, 0, total_ms, trace);
}

// ---- known runs (chunkcore.hpp; sweep_along_kernel RUNS): one fibre, segment by segment, the way the kernel's wave does it -------------
// Interior segments take the four phases of the kernel -- edges, the list of runs, one run per lane, the chunks' records -- with the lanes
// emulated one after the other; rows of segments that are not interior, or that fall back (stats[1]), come from the exact walk of the
// whole fibre, as the speculative path would leave them.  stats: [0] segments solved run by run, [1] fallen back, [2] runs walked,
// [3] most runs in one segment.
int host_runs_fibre(const double *y, double lam, int len, double *x, int *stats) {
    constexpr int C = 17, G = 64, H = 16, T = 8, SEG = C * G;
    constexpr unsigned CM = (1u << C) - 1u;
    for (int k = 0; k < 10; k++) stats[k] = 0;   // ([4..9]: why segments fell back: no bend at the start, a run without end, > 64 runs, a walk that did not close, no bend before a chunk, no bend behind the segment)
    {
        HostSource src{y, nullptr, x, -1};
        Walker w;
        walker_start<false>(w, src, 0, lam);
        walker_run<false>(w, src, len, lam);
    }
    const int nseg = (len + SEG - 1) / SEG;
    for (int sg = 0; sg < nseg; sg++) {
        const int seg_s = sg * SEG, seg_e = std::min(len, seg_s + SEG);
        if (!(seg_s + SEG + T <= len - 1) || !(lam > 0.0)) continue;   // (`interior`)
        HostWin win;
        win.lo = seg_s - H;
        win.hi = seg_s + SEG + T;
        for (int i = win.lo; i < win.hi; i++) win.yy.push_back(y[std::max(i, 0)]);   // (the first segment's zone: copies of sample 0)
        win.yy.push_back(1e300);
        win.yy.push_back(1e300);
        win.writes.assign(win.yy.size(), 0);
        HostFar far{y, nullptr};
        // (1) edges
        EdgeMasks own[G], m[G];
        for (int l = 0; l < G; l++) own[l] = own_edges<C>(win, seg_s + l * C, lam);
        unsigned xk = 0, xp = 0, xn = 0, xb = 0;
        for (int l = 0; l < 10; l++) {
            unsigned k = 0, pp = 0, nn = 0, bb = 0;
            if (l < 8) one_edge(win.y(seg_e - 1 + l), win.y(seg_e + l), lam, k, pp, nn, bb);
            else if (sg > 0) one_edge(win.y(seg_s - 11 + l), win.y(seg_s - 10 + l), lam, k, pp, nn, bb);
            xk |= k << l; xp |= pp << l; xn |= nn << l; xb |= bb << l;
        }
        unsigned BE[G], BT[G], WS[G];
        for (int l = 0; l < G; l++) {
            EdgeMasks pv = l ? own[l - 1] : EdgeMasks{}, nx = l < G - 1 ? own[l + 1] : EdgeMasks{};
            if (l == G - 1) { nx.K = (xk & 0xffu) << kEdgeBias; nx.P = (xp & 0xffu) << kEdgeBias; nx.N = (xn & 0xffu) << kEdgeBias; nx.B = (xb & 0xffu) << kEdgeBias; }
            if (l == 0)     { pv.K = ((xk >> 8) & 3u) << C; pv.P = ((xp >> 8) & 3u) << C; pv.N = ((xn >> 8) & 3u) << C; pv.B = ((xb >> 8) & 3u) << C; }
            m[l].K = edge_ext<C>(own[l].K, pv.K, nx.K); m[l].P = edge_ext<C>(own[l].P, pv.P, nx.P);
            m[l].N = edge_ext<C>(own[l].N, pv.N, nx.N); m[l].B = edge_ext<C>(own[l].B, pv.B, nx.B);
            const bool free0 = sg == 0 && l == 0;
            if (free0) m[l].K = (m[l].K & ~7u) | 4u;
            settle_short_runs(m[l], BE[l], BT[l], WS[l]);
            if (free0) {
                BE[l] = (BE[l] & ~8u) | (m[l].K & 8u);
                BT[l] = (BT[l] & ~8u) | (m[l].P & m[l].K & 8u);
                WS[l] = (WS[l] & ~7u) | ((m[l].K & 8u) ? 0u : 4u);
            }
        }
        // (2) the list
        std::vector<unsigned> list;
        bool fail = sg > 0 && (m[0].K & 7u) == 0u;
        if (fail) stats[4]++;
        for (int l = 0; l < G; l++) {
            unsigned dom = WS[l] & ((1u << (C + kEdgeBias)) - 1u);
            if (l > 0) dom &= ~3u;
            while (dom) {
                const int b = __builtin_ctz(dom);
                dom &= dom - 1u;
                const int e = run_end(m[l].K, b);
                if (e < 0) { fail = true; stats[5]++; }
                else list.push_back(RunEntry::make(l, b, e, (int)((m[l].P >> b) & 1u), sg == 0 && l == 0 && b == kEdgeBias).word);
            }
        }
        stats[3] = std::max(stats[3], (int)list.size());
        bool go = !fail && list.size() <= 64;
        if (list.size() > 64) stats[6]++;
        // (3) one run per lane
        unsigned E[G] = {}, Tt[G] = {}, hang_word = 0;
        if (go) {
            for (unsigned word : list) {
                const RunEntry en{word};
                const int c0 = seg_s + C * en.lane() - kEdgeBias, a = c0 + en.b(), ee = c0 + en.e();
                ChunkRec rr;
                Walker w;
                if (en.free_start()) {
                    walker_start<false>(w, win, 0, lam);
                } else {
                    walker_restart_with<false>(w, a, en.type(), len, lam, win.y(a), 0.0, 0.0);
                    rr.mine = rr.next = rr.last = ((unsigned)a << 1) | (unsigned)en.type();
                }
                walk_interior<false>(w, rr, win, std::min(len - 1, win.hi), a, ee, lam);
                TailSource<false, false, 48, HostWin, HostFar> tail{win, far, rr, a, ee, win.hi, len};
                walker_run<false>(w, tail, len, lam);
                if (!(rr.done && !rr.failed)) { go = false; stats[7]++; break; }
                stats[2]++;
                const int o = en.b() - kEdgeBias;
                if (o >= 0) {
                    E[en.lane()] |= rr.ends << o;
                    Tt[en.lane()] |= rr.types << o;
                } else {
                    E[0] |= rr.ends >> (-o);
                    Tt[0] |= rr.types >> (-o);
                    const unsigned low = rr.ends & ((1u << (-o)) - 1u);
                    if (low) {
                        const int j = 31 - __builtin_clz(low);
                        hang_word = std::max(hang_word, ((unsigned)(en.b() + j + 2) << 1) | ((rr.types >> j) & 1u));
                    }
                }
            }
        }
        // (4) the chunks' records
        ChunkRec recs[G];
        if (go) {
            unsigned ends[G], types[G], lcode[G];
            int lastb[G];
            for (int l = 0; l < G; l++) {
                const unsigned e_prev = l ? E[l - 1] : 0u, t_prev = l ? Tt[l - 1] : 0u;
                ends[l] = ((BE[l] >> (kEdgeBias + 1)) | E[l] | (e_prev >> C)) & CM;
                types[l] = ((BT[l] >> (kEdgeBias + 1)) | Tt[l] | (t_prev >> C)) & CM & ends[l];
                lastb[l] = ends[l] ? 31 - __builtin_clz(ends[l]) : -1;
                lcode[l] = lastb[l] >= 0 ? ((((unsigned)(seg_s + l * C + lastb[l] + 1)) << 1) | ((types[l] >> lastb[l]) & 1u)) : 0u;
            }
            unsigned hang = 0;
            if (sg > 0) {
                unsigned best = hang_word;
                const unsigned kb = BE[0] & 7u;
                if (kb) {
                    const int bi = 31 - __builtin_clz(kb);
                    best = std::max(best, ((unsigned)(bi + 1) << 1) | ((BT[0] >> bi) & 1u));
                }
                if (best) hang = (((unsigned)(seg_s + (int)(best >> 1) - 1 - kEdgeBias)) << 1) | (best & 1u);
            }
            for (int l = 0; l < G && go; l++) {
                unsigned mine = hang;
                for (int k = l - 1; k >= 0; k--)
                    if (lastb[k] >= 0) { mine = lcode[k]; break; }
                if (mine == 0u && !(sg == 0 && l == 0)) { go = false; stats[8]++; }
                unsigned tailc = 0;
                if (l == G - 1 && !((ends[l] >> (C - 1)) & 1u)) {
                    const unsigned beyond = (BE[l] >> (kEdgeBias + 1 + C)) | (E[l] >> C), tbeyond = (BT[l] >> (kEdgeBias + 1 + C)) | (Tt[l] >> C);
                    if (beyond) {
                        const int j0 = __builtin_ctz(beyond);
                        tailc = (((unsigned)(seg_e + j0 + 1)) << 1) | ((tbeyond >> j0) & 1u);
                    } else {
                        go = false;
                        stats[9]++;
                    }
                }
                ChunkRec &rec = recs[l];
                rec.ends = ends[l];
                rec.types = types[l];
                rec.mine = mine;
                rec.next = lastb[l] >= 0 ? lcode[l] : mine;
                rec.last = (l == G - 1 && tailc) ? tailc : rec.next;
                rec.done = true;
            }
        }
        if (!go) {
            stats[1]++;
            continue;
        }
        stats[0]++;
        std::vector<PiecePrefix> pres((size_t)G);
        for (int l = 0; l < G; l++) pres[(size_t)l] = first_piece_prefix(win, recs[l], seg_s + l * C, std::max(0, seg_s + l * C - H));
        for (int l = 0; l < G; l++) {
            const int cs = seg_s + l * C;
            rebuild_owned<Identity, false, C, 1, false, const double *, 0, 2>(win, recs[l], cs, cs + C, len, std::max(0, cs - H), true, seg_s, l == G - 1, lam,
                                                                          (const double *)nullptr, &pres[(size_t)l]);
        }
        int errs = 0;
        for (int k = seg_s; k < seg_e; k++) {
            if (win.writes[(size_t)(k - win.lo)] != 1) errs++;
            x[k] = win.y(k);
        }
        for (int k = win.lo; k < win.hi; k++)
            if ((k < seg_s || k >= seg_e) && win.writes[(size_t)(k - win.lo)] != 0) errs++;
        if (errs) return -errs;   // a row of the segment not written exactly once, or a row outside it written at all
    }
    return 0;
}
}
