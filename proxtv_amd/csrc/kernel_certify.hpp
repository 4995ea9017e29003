// kernel_certify.hpp -- kernel 4 (option certify): the optimality conditions of the prox on what a sweep wrote.
// (One of the pieces of sweep_kernels.hpp, which includes them in order; not meant to be included on its own.)
#pragma once

namespace ptv {
namespace swp {

// ---- kernel 4 (option certify): the optimality conditions of the prox on what a sweep WROTE ------------------------------------------
// x = prox(y) minimises 1/2 |x - y|^2 + sum_k r_k |x_{k+1} - x_k| (the problem every solver of the reference's 1-D path solves:
// src/TVL1opt.cpp:359-564) iff, with u_k = sum_{i <= k} (y_i - x_i):
//     |u_k| <= r_k for every edge k ;  u_k = -r_k where x_{k+1} > x_k ,  u_k = +r_k where x_{k+1} < x_k ;  u_{n-1} = 0
// -- the minimiser is unique, so a fibre that passes IS the prox, whatever kernel wrote it and whatever went wrong on the way.  The
// check reads the sweep's inputs through the op's own input functor and recovers x from the sweep's outputs (Op::recover), so it
// sees exactly what the next sweep will see.  Tolerances: sums of n terms accumulate ~n ulps of the operands' magnitude, a recovered
// x carries a few ulps of it -- a violation counts above kCertifyTol * n * 2^-52 * (largest operand), a step of x above
// kCertifyStep * 2^-52 * that (+ the slack below).  On top of rounding comes the slack the reference itself leaves: its solvers close the fibre's last piece
// with tests against EPSILON = 1e-10 (src/general.h:64-67, src/TVL1opt.cpp:543-557 -- walker.hpp: kEps), so the string may end up to
// EPSILON off the tube centre and every sum along the last piece inherits that; the sequential walks of this library (rung 5, the repair
// kernels) do the same, bit for bit.  kCertifySlack = 4 EPSILON is allowed for it: a wrong sample below ~1e-9 is inside what the
// reference's own solvers disagree by among themselves.  A fibre that fails is flagged; the host re-solves the flagged fibres with the sequential walk
// (sweep_seq_kernel through its fibre gate) and counts them.  Fibres with a negative penalty are not checked (the reference's behaviour
// there is its code, not a minimisation).
constexpr double kCertifyTol = 64.0, kCertifyStep = 256.0, kCertifyUlp = 2.220446049250313e-16, kCertifySlack = 4.0 * kEps;

struct CertifyAcc {
    double u = 0.0, scale = 0.0, viol = 0.0;
    int where = -1, kind = 0;   // sample and test of the largest violation (0 the bound, 1 / 2 a step up / down off its wall, 3 the total)
    bool defined = true;
    __device__ __forceinline__ void note(double v, int k, int what) {
        if (v > viol) {
            viol = v;
            where = k;
            kind = what;
        }
    }
    // one sample: the running sum behind it, its x, the next sample's x (the last sample: anything), the penalty of the edge behind it
    __device__ __forceinline__ void edge(double uk, double x, double xn, double r, bool last, int k) {
        if (last) {
            note(fabs(uk), k, 3);
            return;
        }
        defined = defined && r >= 0.0;
        note(fabs(uk) - r, k, 0);
        // (a step counts as one above rounding AND above what the slack at the last sample does to the last piece's value: a knot whose jump
        //  is zero up to rounding -- late Dykstra / DR iterates are full of them -- next to a last piece that is 1e-10 / n off shows a step
        //  of either sign: seen on the GPU, PD2 at lambda 0.7, sample 517 of 520)
        const double dx = xn - x, step = kCertifyStep * kCertifyUlp * scale + kCertifySlack;
        if (dx > step)       note(fabs(uk + r), k, 1);
        else if (dx < -step) note(fabs(uk - r), k, 2);
    }
    __device__ __forceinline__ double tolerance(int len, double lam) const {
        return kCertifyTol * (double)len * kCertifyUlp * fmax(scale, fabs(lam)) + kCertifySlack;
    }
    __device__ __forceinline__ bool failed(int len, double lam) const { return defined && viol > tolerance(len, lam); }
};
// what the first few failing fibres of a launch looked like (option verbose prints them)
struct CertifyNote {
    long fibre;
    int where, kind;
    double viol, tol;
};
constexpr int kCertifyNotes = 8;
__device__ __forceinline__ void certify_flag(int *flags, unsigned *count, CertifyNote *notes, long j, const CertifyAcc &acc, int len, double lam) {
    flags[j] = 1;
    const unsigned slot = atomicAdd(count, 1u);
    if (notes && slot < (unsigned)kCertifyNotes) notes[slot] = CertifyNote{j, acc.where, acc.kind, acc.viol, acc.tolerance(len, lam)};
}

template <int OP>
__device__ __forceinline__ void certify_sample(const SweepArgs &p, long idx, double &y, double &x, double &scale) {
    double i0, i1;
    Op<OP>::fetch_in(p, idx, i0, i1);
    y = Op<OP>::y_of(p, i0, i1);
    scale = fmax(scale, fmax(fabs(i0), fabs(i1)));
    x = Op<OP>::recover(p, idx, y, scale);
    scale = fmax(scale, fabs(x));
}

// strided fibres: one lane per fibre, 64 adjacent fibres per wave (every access a 512-byte row)
template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void certify_strided_kernel(SweepArgs p, FibreGeom g, int *flags, unsigned *count, CertifyNote *notes) {
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    if (j >= g.count || g.len <= 0) return;
    if (p.gate && *p.gate == 0) return;
    long blk, off;
    divmod_nonneg(j, g.inc, blk, off);
    const long base = blk * g.inc * g.len + off, wbase = blk * g.inc * (g.len - 1) + off;
    CertifyAcc acc;
    double y, x;
    certify_sample<OP>(p, base, y, x, acc.scale);
    for (int k = 0; k < g.len; k++) {
        const bool last = k == g.len - 1;
        double yn = 0.0, xn = x;
        if (!last) certify_sample<OP>(p, base + (long)(k + 1) * g.inc, yn, xn, acc.scale);
        acc.u += y - x;
        const double r = last ? 0.0 : (WEIGHTED ? p.w[wbase + (long)k * g.inc] : p.lam);
        acc.edge(acc.u, x, xn, r, last, k);
        y = yn;
        x = xn;
    }
    if (acc.failed(g.len, WEIGHTED ? 0.0 : p.lam)) certify_flag(flags, count, notes, j, acc, g.len, WEIGHTED ? 0.0 : p.lam);
}

// contiguous fibres: one wave per fibre, 64 consecutive samples per trip, the running sum by a scan across the lanes
template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(256) void certify_along_kernel(SweepArgs p, FibreGeom g, int *flags, unsigned *count, CertifyNote *notes) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= g.count || g.len <= 0) return;
    if (p.gate && *p.gate == 0) return;
    const long base = j * g.len, wbase = j * (g.len - 1);
    CertifyAcc acc;
    double carry = 0.0;
    for (int k0 = 0; k0 < g.len; k0 += 64) {
        const int k = k0 + lane;
        const bool in = k < g.len, last = k == g.len - 1;
        double y = 0.0, x = 0.0, xn = 0.0, yd;
        if (in) certify_sample<OP>(p, base + k, y, x, acc.scale);
        // the next sample's x: the next lane's; the trip's last lane reads it itself
        xn = __shfl_down(x, 1);
        if (lane == 63 && in && !last) certify_sample<OP>(p, base + k + 1, yd, xn, acc.scale);
        // (every lane tests against the largest operand any lane has seen so far)
        for (int o = 32; o > 0; o >>= 1) acc.scale = fmax(acc.scale, __shfl_xor(acc.scale, o));
        double u = in ? y - x : 0.0;
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(u, o);
            if (lane >= o) u += t;
        }
        u += carry;
        carry = __shfl(u, 63);
        if (in) {
            const double r = last ? 0.0 : (WEIGHTED ? p.w[wbase + k] : p.lam);
            acc.edge(u, x, xn, r, last, k);
        }
    }
    bool bad = acc.failed(g.len, WEIGHTED ? 0.0 : p.lam);
    // (a lane that met a negative penalty takes the whole fibre out of the check)
    if (__ballot(!acc.defined) != 0ull) bad = false;
    const unsigned long long who = __ballot(bad);
    if (who != 0ull && lane == __ffsll((long long)who) - 1) certify_flag(flags, count, notes, j, acc, g.len, WEIGHTED ? 0.0 : p.lam);
}

}  // namespace swp
}  // namespace ptv
