// pin_study.cpp -- HOST-ONLY design study (never part of the library): how many levels the pinning solver (pincore.hpp) saves when
// the knots known a priori -- edges with |y_k - y_{k-1}| > 4 lambda, where the string is known to bend (chunkcore.hpp) -- are pinned
// before the first level instead of being found by it.  The lanes of a group are emulated one after the other like in
// tests/host_harness.cpp.
#define PTV_HOST_TEST 1
#define __device__
#define __forceinline__ inline
#include <cmath>
#include <cstring>
#include <vector>

#include "../../proxtv_amd/csrc/pincore.hpp"

using namespace ptv;

namespace {
template <int P>
struct Shared {
    std::vector<double> s, rr;
    std::vector<double> mx[2];
    std::vector<unsigned> arg[2];
    static constexpr bool kWeighted = false;
    double S(int j) const { return s[(size_t)j]; }
    double r(int j) const { return rr[(size_t)j]; }
    double own(int t, int k) const { return s[(size_t)(1 + t * P + k)]; }
    double own_at(int t, int k) const { return own(t, k); }
    void set_own(int t, int k, double v) { s[(size_t)(1 + t * P + k)] = v; }
    double rown(int t, int k) const { return rr[(size_t)(1 + t * P + k)]; }
    void post(int wall, int slot, double v) { if (v > mx[wall][(size_t)slot]) mx[wall][(size_t)slot] = v; }
    double best(int wall, int slot) const { return mx[wall][(size_t)slot]; }
    void claim(int wall, int slot, unsigned key) { if (key < arg[wall][(size_t)slot]) arg[wall][(size_t)slot] = key; }
    int knot(int wall, int slot) const { return arg[wall][(size_t)slot] == ~0u ? -1 : PinLane<P>::claimed_knot(arg[wall][(size_t)slot]); }
    void clear_best(int slot) { mx[0][(size_t)slot] = mx[1][(size_t)slot] = 0.0; }
    void clear_knot(int slot) { arg[0][(size_t)slot] = arg[1][(size_t)slot] = ~0u; }
};
}  // namespace

extern "C" {
// seed: 0 = the fibre ends only (today) ; 1 = also every knot known a priori.  Returns the number of levels; *npins = knots pinned
// before the first level.
int pin_levels(const double *y, int n, double lam, double *x, int seed, int *npins) {
    constexpr int P = 16;
    Shared<P> sh;
    const int lanes = (n + P - 1) / P;
    double mean = 0;
    for (int i = 0; i < n; i++) mean += y[i];
    mean /= n;
    sh.s.assign((size_t)n + 1 + P, 0.0);
    sh.rr.assign((size_t)n + 1 + P, 0.0);
    for (int i = 0; i < n; i++) sh.s[(size_t)i + 1] = sh.s[(size_t)i] + (y[i] - mean);
    for (int j = 1; j < n; j++) sh.rr[(size_t)j] = lam;
    std::vector<PinLane<P>> lane((size_t)lanes);
    for (int t = 0; t < lanes; t++) lane[(size_t)t].init(n, t, sh);
    *npins = 0;
    if (seed) {
        // knot j (between samples j - 1 and j) is known a priori when |y_j - y_{j-1}| > 4 lambda: an up-jump is a convex bend, the
        // string touches the UPPER wall there
        std::vector<int> wall((size_t)n + 1, -1);
        for (int j = 1; j < n; j++) {
            const double d = y[j] - y[j - 1];
            if (std::fabs(d) > 4.0000001 * lam) wall[(size_t)j] = d > 0 ? 0 : 1;
        }
        for (int t = 0; t < lanes; t++) {
            PinLane<P> &L = lane[(size_t)t];
            for (int j = L.j0; j < L.j1; j++) {
                if (wall[(size_t)j] == 0) { L.pinU |= 1u << (j - L.j0); L.pendU |= 1u << (j - L.j0); (*npins)++; }
                if (wall[(size_t)j] == 1) { L.pinL |= 1u << (j - L.j0); L.pendL |= 1u << (j - L.j0); (*npins)++; }
            }
            for (int j = L.j0 - 1; j >= 1; j--)
                if (wall[(size_t)j] >= 0) { L.la = j; L.hl = L.height(sh, j, wall[(size_t)j] != 0); break; }
            for (int j = L.j1; j < n; j++)
                if (wall[(size_t)j] >= 0) { L.rb = j; L.hr = L.height(sh, j, wall[(size_t)j] != 0); break; }
        }
    }
    for (int w = 0; w < 2; w++) {
        sh.mx[w].assign((size_t)lanes + 1, 0.0);
        sh.arg[w].assign((size_t)lanes + 1, ~0u);
    }
    int levels = 0;
    for (;;) {
        levels++;
        for (int t = 0; t < lanes; t++) lane[(size_t)t].scan(sh);
        for (int t = 0; t < lanes; t++) lane[(size_t)t].claim(sh);
        bool any = false;
        for (int t = 0; t < lanes; t++) any |= lane[(size_t)t].update(sh);
        if (!any) break;
    }
    for (int t = 0; t < lanes; t++) lane[(size_t)t].settle(sh);
    for (int t = 0; t < lanes; t++) lane[(size_t)t].values(sh, mean, [&](int i, int, double v) { x[i] = v; });
    return levels;
}
}
