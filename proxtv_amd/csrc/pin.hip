// pin.hip -- the pinning solver (pincore.hpp) as a sweep kernel: the exact taut string of a fibre found by all lanes of a
// group at once; cost ~ (12-16 levels) x n whatever the length of the pieces.  The device counterpart of the reference's
// worst-case-linear solver behind its hybrid switch (src/TVL1opt_tautstring.cpp:256-340,
// src/TVL1opt_hybridtautstring.cpp:31-35,73) and the top rung of the geometry ladder of sweep.hip: where speculative
// chunks need zones of hundreds of samples (lambda >~ noise, block images, late DR iterates) this kernel takes the same
// time as on white noise.
//
// One group of G lanes per fibre (G = 256: a workgroup; G = 64: a wave, four fibres per workgroup), P knots per lane.
// LDS per group: ONE plane of doubles -- the samples as they arrive, then their running sums in place, then the prox
// values in place -- padded so that the lanes of an access (P doubles apart in the fibre) fall on different banks without
// any transposition (PinGeom); a second plane for the penalties of a weighted sweep; and one small buffer of reduction
// slots (pincore.hpp).  4096-sample fibres: 39 KB per workgroup, four workgroups = 16 waves per CU.
//
//   stage     coalesced loads through the op's input functor (every load of a batch issued before the first is waited for)
//   sums      mean of the fibre (group reduction), centred running sums (lane-local pass + group scan), in place
//   levels    scan / claim / update of pincore.hpp, three group barriers per level, until no lane gained a pin
//   values    every lane turns its part of the string into slopes, in place
//   stream    coalesced, through the op's output functor
//
// Fibres must be contiguous (dimension 0); a strided sweep runs on transposed copies of its operands (pointwise.hip's
// slab_transpose), like launch_row_along in sweep.hip.  Exact for every input: no links, no repair kernel, no counters.
#include "pin.hpp"

#include <memory>

#include "pin_device.hpp"
#include "transposed.hpp"

namespace ptv {

namespace {

using namespace pin;

template <int OP, bool WEIGHTED, int P, int G>
__global__ __launch_bounds__(kPinThreads) void sweep_pin_kernel(SweepArgs p, FibreGeom g, int *pieces, int *gaveup, int seeded) {
    using Geo = PinGeom<P, G, WEIGHTED>;
    using Sh = PinShared<P, G, WEIGHTED>;
    constexpr int SLOTS = Geo::SLOTS;
    constexpr int UB = 8;   // global loads in flight per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.gate && *p.gate == 0) return;
    const int tid = threadIdx.x, gi = tid / G, t = tid % G;
    char *base = smem + Geo::group_bytes * (size_t)gi;
    double *Sp = reinterpret_cast<double *>(base);
    double *Wp = Sp + (WEIGHTED ? Geo::ROWS : 0);
    unsigned long long *mx = reinterpret_cast<unsigned long long *>(base + Geo::plane_bytes * (WEIGHTED ? 2 : 1));   // [wall][slot]
    unsigned *arg = reinterpret_cast<unsigned *>(mx + 2 * SLOTS);
    const long fibre = (long)blockIdx.x * Geo::NG + gi;
    if (G < kPinThreads && fibre >= g.count) return;   // (a whole wave; such groups share no barrier with the others)
    const int n = g.len;
    const long fbase = fibre * n, wbase = fibre * (long)(n - 1);

    // ---- stage ------------------------------------------------------------------------------------------------------------------
#pragma unroll 1
    for (int u0 = 0; u0 < P; u0 += UB) {
        double s0[UB], s1[UB], sw[WEIGHTED ? UB : 1];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int i = (u0 + u) * G + t;
            s0[u] = s1[u] = 0.0;
            if (i < n) Op<OP>::fetch_in(p, fbase + i, s0[u], s1[u]);
            if (WEIGHTED) sw[WEIGHTED ? u : 0] = (i >= 1 && i < n) ? p.w[wbase + i - 1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int i = (u0 + u) * G + t;
            if (i < n) {
                Sp[Geo::sa(i + 1)] = Op<OP>::y_of(p, s0[u], s1[u]);
                if (WEIGHTED && i >= 1) Wp[Geo::sa(i)] = sw[WEIGHTED ? u : 0];   // penalty of the edge (i - 1, i) = half-width at knot i
            }
        }
    }
    if (t == 0) Sp[0] = 0.0;
    group_sync<G>();

    // ---- centred running sums, in place ---------------------------------------------------------------------------------------------
    double *own = Sp + Geo::lane_base(t);   // own[k]: sample t P + k, then the sum at knot t P + k + 1
    const int cnt = n - t * P < 0 ? 0 : (n - t * P < P ? n - t * P : P);
    double *red = reinterpret_cast<double *>(mx);
    // ---- knots known a priori (PinLane::seed), while the plane still holds the samples: own knot k = 1 + t P + k lies between the
    // samples own[k] and the one after it (the lane's last knot looks at its neighbour's first sample: the barriers of the sums
    // below come before anybody overwrites a sample)
    typename PinLane<P>::Mask seedU = 0, seedL = 0;
    if (seeded & 1) {
        double prev = cnt > 0 ? own[0] : 0.0;
#pragma unroll
        for (int k = 0; k < P; k++) {
            const int j = 1 + t * P + k;
            if (j < n) {
                const double cur = Sp[Geo::sa(j + 1)];
                const double d = cur - prev;
                double thr = 4.0000001 * p.lam;
                bool ok = p.lam > 0.0;
                if (WEIGHTED) {   // r_{j+1} + 2 r_j + r_{j-1} ; the fibre ends have no wall (half-width 0)
                    const double rm = j > 1 ? Wp[Geo::sa(j - 1)] : 0.0, r0 = Wp[Geo::sa(j)], rp = j + 1 < n ? Wp[Geo::sa(j + 1)] : 0.0;
                    thr = 1.0000001 * (rp + 2.0 * r0 + rm);
                    ok = (rm >= 0.0) & (r0 > 0.0) & (rp >= 0.0);
                }
                const bool hit = ok & (fabs(d) > thr);
                if (hit && d > 0) seedU |= (typename PinLane<P>::Mask)1 << k;
                if (hit && !(d > 0)) seedL |= (typename PinLane<P>::Mask)1 << k;
                prev = cur;
            }
        }
    }
    double mean;
    {
        double ls = 0.0;
        for (int k = 0; k < cnt; k++) ls += own[k];
        double total;
        group_scan<G>(ls, t, red, total);
        mean = total / (double)n;
        double lc = 0.0;
        for (int k = 0; k < cnt; k++) lc += own[k] - mean;
        double tot2;
        double acc = group_scan<G>(lc, t, red, tot2) - lc;
        for (int k = 0; k < cnt; k++) {
            acc += own[k] - mean;
            own[k] = acc;
        }
    }
    group_sync<G>();   // the sums are in place; `red` is free again
    PinLane<P> ln;
    Sh sh{Sp, Wp, own, Wp + Geo::lane_base(t), {}, p.lam, mx, arg};
    if (Sh::kCached) {
#pragma unroll
        for (int k = 0; k < (Sh::kCached ? P : 1); k++) sh.cached[k] = own[k];
    }
    ln.init(n, t, sh);
    // knots known by windows (pincore.hpp: sixteen knots a lane, one penalty), coarse to fine, a stage only where the one before found
    // a knot somewhere in the wave; windows that span lanes are joined by lane shuffles -- the few that also span waves are left to
    // the levels
    if constexpr (P == 16) {
        if ((seeded & 2) && (WEIGHTED || p.lam > 0.0)) {
            using Lane = PinLane<P>;
            using Win = typename Lane::Win;
            using Mask = typename Lane::Mask;
            const int l = t & 63;
            auto Sk = [&](int j) { return Sp[Geo::sa(j < 0 ? 0 : (j > n ? n : j))]; };
            auto Rk = [&](int j) { return WEIGHTED ? Wp[Geo::sa(j < 1 ? 1 : (j > n - 1 ? n - 1 : j))] : p.lam; };   // (windows that reach a fibre end are not taken)
            auto shfl_win = [](Win w, int src) { return Win{__shfl(w.mx, src), __shfl(w.mn, src)}; };
            Mask up = 0, lo = 0;
            {   // 64 knots: the plain grid first; the shifted one where that found a knot somewhere in the wave
                const int qa = t & 3, qb = (t + 2) & 3;
                const int a0 = (t - qa) * P, b0 = (t - qb) * P;
                const Win pa = ln.win64_part(sh, Sk(a0), Rk(a0), Sk(a0 + 64), Rk(a0 + 64), qa);
                Win all_a = Lane::wjoin(pa, shfl_win(pa, l ^ 1));
                all_a = Lane::wjoin(all_a, shfl_win(all_a, l ^ 2));
                ln.win64_take(Lane::seed_threshold(Rk(a0), Rk(a0 + 64)), all_a, true, qa, 4, up, lo);
                if (__ballot((up | lo) != 0) != 0ull) {
                    const Win pb = ln.win64_part(sh, Sk(b0), Rk(b0), Sk(b0 + 64), Rk(b0 + 64), qb);
                    Win all_b = Lane::wjoin(pb, shfl_win(pb, l ^ 1));
                    const int partner = qb < 2 ? l + 2 : l - 2;
                    all_b = Lane::wjoin(all_b, shfl_win(all_b, partner & 63));
                    ln.win64_take(Lane::seed_threshold(Rk(b0), Rk(b0 + 64)), all_b, partner >= 0 && partner < 64, qb, 2, up, lo);
                }
            }
            if (__popcll(__ballot((up | lo) != 0)) >= kSeedStage16) {
                const double Sl = Sk(t * P), rl = Rk(t * P);
                Mask up16 = 0, lo16 = 0;
                Win tail, head;
                double thr_tail, thr_head;
                ln.win16_parts(sh, Sl, rl, Sk(t * P + 24), Rk(t * P + 24), Sk(t * P - 8), Rk(t * P - 8), up16, lo16, tail, head, thr_tail, thr_head);
                ln.win16_take(tail, thr_tail, shfl_win(head, (l + 1) & 63), l < 63, shfl_win(tail, (l + 63) & 63), head, thr_head, l > 0, up16, lo16);
                up |= up16;
                lo |= lo16;
                if (__popcll(__ballot((up16 | lo16) != 0)) >= kSeedStage4) {
                    int give;
                    ln.win4_all(sh, Sl, rl, Sk(t * P + P + 1), Rk(t * P + P + 1), Sk(t * P + P + 2), Rk(t * P + P + 2), up, lo, give);
                    ln.win4_take(__shfl(give, (l + 63) & 63), l > 0, up, lo);
                }
            }
            if (t * P < n) {
                seedU |= up;
                seedL |= lo & ~seedU;
            }
        }
    }
    // the nearest seeded knot on either side of the lane's range, and the string's height there (read before any lane settles:
    // settle() turns the sum of a pinned knot into that height in place)
    int s_la = 0, s_rb = n;
    double s_hl = 0.0, s_hr = 0.0;
    using SeedMask = typename PinLane<P>::Mask;
    if (seeded) seeded = group_any<G>((seedU | seedL) != 0) ? seeded : 0;   // (nothing found anywhere in the fibre: the levels start from its ends)
    if (seeded) {
        using Mask = SeedMask;
        const Mask both = seedU | seedL;
        const int j0 = 1 + t * P;
        int last = 0, first = 0x7fffffff;
        if (both) {
            const int kl = (int)(8 * sizeof(Mask)) - 1 - (sizeof(Mask) == 4 ? __builtin_clz((unsigned)both) : __builtin_clzll((unsigned long long)both));
            const int kf = PinLane<P>::ctz(both);
            last = ((j0 + kl) << 1) | (int)((seedL >> kl) & 1);
            first = ((j0 + kf) << 1) | (int)((seedL >> kf) & 1);
        }
        int before, after;
        group_neighbour_seeds<G>(last, first, t, reinterpret_cast<int *>(red), before, after);
        if (before) s_la = before >> 1;
        if (after != 0x7fffffff) s_rb = after >> 1;
        const double wl = (s_la > 0) ? (WEIGHTED ? Wp[Geo::sa(s_la)] : p.lam) : 0.0, wr = (s_rb < n) ? (WEIGHTED ? Wp[Geo::sa(s_rb)] : p.lam) : 0.0;
        s_hl = Sp[Geo::sa(s_la)] + ((before & 1) ? -wl : wl);
        s_hr = Sp[Geo::sa(s_rb)] + ((after != 0x7fffffff && (after & 1)) ? -wr : wr);
        group_sync<G>();
    }
    // reduction slots: empty (from here on the lanes clear what they own as the levels go: pincore.hpp)
    for (int wall = 0; wall < 2; wall++) {
        mx[wall * SLOTS + t + 1] = 0ull;
        arg[wall * SLOTS + t + 1] = ~0u;
        if (t == 0) {
            mx[wall * SLOTS] = 0ull;
            arg[wall * SLOTS] = ~0u;
        }
    }
    group_sync<G>();

    // ---- levels ----------------------------------------------------------------------------------------------------------------------
    if (seeded && t * P < n) ln.seed(seedU, seedL, s_la, s_hl, s_rb, s_hr);
    bool capped = false;
#pragma unroll 1
    for (int level = 0;; level++) {
        ln.scan(sh);
        group_sync<G>();
        ln.claim(sh);
        group_sync<G>();
        const bool gained = ln.update(sh);
        if (!group_any<G>(gained)) break;
        if (level + 1 >= kPinMaxLevels) {   // (uniform over the group) periodic ties: this fibre goes to the walker
            capped = true;
            break;
        }
    }
    if (capped) {
        if (t == 0) gaveup[fibre] = 1;
        return;   // nothing of this fibre was written; groups share no barrier after this point
    }

    // ---- values, in place (a lane reads nothing but its own part of the plane and what it cached of its neighbours') -----------------
    ln.settle(sh);
    ln.values(sh, mean, [&](int, int k, double v) { own[k] = v; });
    if (pieces) {   // a measured launch: pieces of this sweep, for the geometry policy (one atomic per wave)
        int c = __popcll(ln.pinU | ln.pinL) + (t == 0 ? 1 : 0);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
        if ((t & 63) == 0 && c > 0) atomicAdd(pieces, c);
    }
    group_sync<G>();

    // ---- stream out -------------------------------------------------------------------------------------------------------------------
#pragma unroll 1
    for (int u0 = 0; u0 < P; u0 += UB) {
        Ext ex[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int i = (u0 + u) * G + t;
            ex[u] = (i < n) ? Op<OP>::fetch(p, fbase + i) : Ext{0, 0};
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int i = (u0 + u) * G + t;
            if (i < n) Op<OP>::finish(p, fbase + i, ex[u], Sp[Geo::sa(i + 1)]);
        }
    }
}

// per-fibre "gave up at the level cap" flags: zero between launches (launch_seq_gated clears what it consumes)
struct GaveUp {
    std::unique_ptr<Scratch> buf;
    size_t count = 0;
    int *get(long fibres, hipStream_t s) {
        if ((size_t)fibres > count) {
            buf.reset(new Scratch(sizeof(int) * (size_t)fibres));
            count = (size_t)fibres;
            PTV_HIP(hipMemsetAsync(buf->as<int>(), 0, sizeof(int) * count, s));
        }
        return buf->as<int>();
    }
};
static thread_local GaveUp g_gaveup[kMaxDevices];
static thread_local int g_seeded = 3;   // this sweep starts from the knots known a priori: bit 0 = jumps above 4 lambda, bit 1 = the deepest knots of windows (launch_pin's argument, for the launchers below)

template <int OP, bool WEIGHTED, int P, int G>
void launch_geom(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *pieces) {
    using Geo = PinGeom<P, G, WEIGHTED>;
    auto kern = sweep_pin_kernel<OP, WEIGHTED, P, G>;
    if (Geo::lds > 64 * 1024) {   // above the default dynamic-LDS limit
        static thread_local bool attr_done[kMaxDevices] = {};
        bool &attr_set = attr_done[current_device()];
        if (!attr_set) {
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo::lds));
            attr_set = true;
        }
    }
    const long wgs = (g.count + Geo::NG - 1) / Geo::NG;
    int *gaveup = g_gaveup[current_device()].get(g.count, stream);
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(kPinThreads), Geo::lds, stream, args, g, pieces, gaveup, g_seeded);
    PTV_HIP(hipGetLastError());
    // fibres that hit the level cap (none on anything but periodic data: the kernel returns at once)
    launch_seq_gated((OpId)OP, WEIGHTED, args, g, stream, gaveup);
}

// contiguous fibres: pick the group geometry from the fibre length
template <int OP, bool WEIGHTED>
void launch_contig(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *pieces) {
    if (g.len <= 64 * 16)        launch_geom<OP, WEIGHTED, 16, 64>(args, g, stream, pieces);
    else if (g.len <= 256 * 16)  launch_geom<OP, WEIGHTED, 16, 256>(args, g, stream, pieces);
    else if (g.len <= 256 * 32)  launch_geom<OP, WEIGHTED, 32, 256>(args, g, stream, pieces);
    else if constexpr (!WEIGHTED) launch_geom<OP, false, 64, 256>(args, g, stream, pieces);
}

template <int OP, bool WEIGHTED>
void launch_op(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *pieces) {
    if (g.inc == 1) {
        launch_contig<OP, WEIGHTED>(args, g, stream, pieces);
        return;
    }
    // strided: the same sweep along dimension 0 of transposed copies (transposed.hpp)
    // (Tried in round 5 and removed: ranges of fibres alternating between two streams so that the copies of one range move under the
    //  levels of another -- four quarter-size launches per sweep lose more than the hidden copies win: 4096^2 DR at lambda 0.8 / 1 / 3
    //  25.1 -> 27.1, 29.3 -> 30.9, 18.2 -> 21.0 ms.)
    TransposedOperands tr(args, Op<OP>::IN_MASK, Op<OP>::OUT_MASK, g, stream);
    launch_contig<OP, WEIGHTED>(tr.args(), tr.geom(), stream, pieces);
    tr.finish();
}

}  // namespace

bool launch_pin(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *pieces, int seeds) {
    g_seeded = seeds;
    if (!pin_supports(op, weighted, g, args.lam)) {
        set_error("launch_pin: sweep not supported (len %d, inc %ld, weighted %d)", g.len, g.inc, (int)weighted);
        throw HipFailure{hipErrorInvalidValue};
    }
    if (pin_is_long(weighted, g)) return launch_pin_long(op, weighted, args, g, stream, pieces);
#define PTV_PIN_CASE(ID)                                         \
    case ID:                                                     \
        if (weighted) launch_op<ID, true>(args, g, stream, pieces);      \
        else          launch_op<ID, false>(args, g, stream, pieces);     \
        break;
#define PTV_PIN_CASE_U(ID) case ID: launch_op<ID, false>(args, g, stream, pieces); break;
#define PTV_PIN_CASE_W(ID) case ID: launch_op<ID, true>(args, g, stream, pieces); break;
    switch (op) {
        PTV_PIN_CASE(OP_PROX)
        PTV_PIN_CASE(OP_DR_COL)
        PTV_PIN_CASE(OP_DR_COL_FINAL)
        PTV_PIN_CASE(OP_DR_ROW)
        PTV_PIN_CASE_U(OP_DR_ROW_FINAL)
        PTV_PIN_CASE_W(OP_DRW_ROW_FINAL)
        PTV_PIN_CASE_U(OP_PD2_A)
        PTV_PIN_CASE_U(OP_PD2_B)
        PTV_PIN_CASE_U(OP_YANG)
        PTV_PIN_CASE(OP_DR_COL_V)
        PTV_PIN_CASE(OP_DR_ROW_V)
        default:
            set_error("launch_pin: unknown op %d", (int)op);
            throw HipFailure{hipErrorInvalidValue};
    }
#undef PTV_PIN_CASE
#undef PTV_PIN_CASE_U
#undef PTV_PIN_CASE_W
    return true;
}


void warm_pin() {
    hipFuncAttributes attr;
    PTV_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>((sweep_pin_kernel<OP_PROX, false, 16, 256>))));
}

}  // namespace ptv
