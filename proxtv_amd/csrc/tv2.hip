// tv2.hip -- batched exact 1-D TV-L2 prox over the fibres of an N-D array (the p = 2 arm of TV(),
// reference: morePG_TV2, src/TVL2opt.cpp:190-445; dispatch src/TVgenopt.cpp:43-45).
//
//     prox(y) = argmin_x 1/2 ||x - y||^2 + lambda ||Dx||_2
//
// Dual: min_u 1/2 ||D'u - y||^2 over the ball ||u||_2 <= lambda, x = y + D'u (DUAL2PRIMAL, src/TVmacros.h:10-14) -- a
// trust-region problem on T = DD' = tridiag(-1, 2, -1).  Either u = T^-1 Dy is inside the ball (x is then the mean of y),
// or u = (T + mu I)^-1 Dy with ||u|| = lambda; mu is the root of the secular equation 1/||u(mu)|| = 1/lambda, found by
// More-Sorensen's Newton iteration (the reference's method, :340-385): factor T + mu I, solve for u, solve again for
// v = (T + mu I)^-1 u, mu += (||u||^2 / u'v) (||u|| - lambda) / lambda.  From mu = 0 the iterates increase monotonically
// to the root, quadratically at the end (3-9 iterations in practice).
//
// Unlike the reference this solver has no projected-gradient prelude, does not stop at a duality gap of 1e-5 and does not
// warm-start a fibre from the previous fibre of the same OpenMP thread: every fibre starts at mu = 0 and iterates until
// | ||u|| - lambda | <= 1e-14 lambda, so the result depends on (y, lambda) only -- the same decision as the oracle's
// orc_TV2_exact, which the parity tests compare with to 1e-9; the reference itself is matched within its own guarantee
// (||dx||_2 <= sqrt(2 * 1e-5)).
//
// Mapping: one lane per fibre, 64 adjacent fibres per wave (every access of the wave is a coalesced 512-byte row, as in
// sweep_seq_kernel); dimension-0 sweeps go through a tiled transpose so that they, too, are strided.  The tridiagonal
// solves are LDL' sweeps (what dpttrf_/dpttrs_ do): three scratch arrays of the data's size (pivots d, forward solution
// z, dual u).  Not a hot path of the headline; exact, batched, device-resident.
#include "tv2.hpp"

#include "pointwise.hpp"

namespace ptv {

namespace {

constexpr int kBlock = 8;   // samples handled per step of a sweep: their loads are independent of the recurrence

struct Tv2Args {
    const double *y;
    double *x;
    double *d, *z, *u;   // scratch, laid out like the data
    double lam;
};

// (tridiag(-1, a, -1)) out = rhs over one fibre.  REFACTOR: compute and store the pivots d (else reuse them).
// RHS_DIFF: rhs_i = y_{i+1} - y_i (the first solve of an iteration), else rhs = u (read before it is overwritten? no:
// the second solve writes nothing but z, its solution is only dotted with u).
// Returns sum out_i^2 (RHS_DIFF) or sum u_i out_i (second solve).
template <bool RHS_DIFF>
__device__ __forceinline__ double tri_solve(const Tv2Args &p, long base, long inc, int nn, double a) {
    // forward: d_i = a - 1/d_{i-1} ; z_i = rhs_i + z_{i-1}/d_{i-1}
    double dprev = 0.0, zprev = 0.0;
    for (int i0 = 0; i0 < nn; i0 += kBlock) {
        double r[kBlock], yn[kBlock + 1];
        if (RHS_DIFF) {
#pragma unroll
            for (int k = 0; k <= kBlock; k++) yn[k] = (i0 + k <= nn) ? p.y[base + (long)(i0 + k) * inc] : 0.0;
#pragma unroll
            for (int k = 0; k < kBlock; k++) r[k] = yn[k + 1] - yn[k];
        } else {
#pragma unroll
            for (int k = 0; k < kBlock; k++) r[k] = (i0 + k < nn) ? p.u[base + (long)(i0 + k) * inc] : 0.0;
        }
        double dv[kBlock];
        if (!RHS_DIFF) {
#pragma unroll
            for (int k = 0; k < kBlock; k++) dv[k] = (i0 + k < nn) ? p.d[base + (long)(i0 + k) * inc] : 1.0;
        }
#pragma unroll
        for (int k = 0; k < kBlock; k++) {
            const int i = i0 + k;
            if (i < nn) {
                double di, zi;
                if (i == 0) {
                    di = a;
                    zi = r[k];
                } else {
                    di = RHS_DIFF ? a - 1.0 / dprev : dv[k];
                    zi = r[k] + zprev / dprev;
                }
                if (RHS_DIFF) p.d[base + (long)i * inc] = di;
                p.z[base + (long)i * inc] = zi;
                dprev = di;
                zprev = zi;
            }
        }
    }
    // backward: out_i = (z_i + out_{i+1}) / d_i
    double acc = 0.0, onext = 0.0;
    for (int i1 = nn - 1; i1 >= 0; i1 -= kBlock) {
        double zv[kBlock], dv[kBlock], uv[kBlock];
#pragma unroll
        for (int k = 0; k < kBlock; k++) {
            const int i = i1 - k;
            zv[k] = (i >= 0) ? p.z[base + (long)i * inc] : 0.0;
            dv[k] = (i >= 0) ? p.d[base + (long)i * inc] : 1.0;
            if (!RHS_DIFF) uv[k] = (i >= 0) ? p.u[base + (long)i * inc] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kBlock; k++) {
            const int i = i1 - k;
            if (i >= 0) {
                const double o = (i == nn - 1) ? zv[k] / dv[k] : (zv[k] + onext) / dv[k];
                if (RHS_DIFF) {
                    p.u[base + (long)i * inc] = o;
                    acc += o * o;
                } else {
                    acc += uv[k] * o;
                }
                onext = o;
            }
        }
    }
    return acc;
}

__global__ __launch_bounds__(64) void tv2_fibres_kernel(Tv2Args p, FibreGeom g) {
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    if (j >= g.count || g.len <= 0) return;
    long blk, off;
    divmod_nonneg(j, g.inc, blk, off);
    const long base = blk * g.inc * g.len + off, inc = g.inc;
    const int n = g.len, nn = n - 1;
    if (nn == 0 || !(p.lam > 0.0)) {
        for (int i = 0; i < n; i++) p.x[base + (long)i * inc] = p.y[base + (long)i * inc];
        return;
    }
    double mu = 0.0;
    double nu2 = tri_solve<true>(p, base, inc, nn, 2.0 + mu);
    double nu = sqrt(nu2);
    if (nu > p.lam) {
        for (int it = 0; it < 200; it++) {
            const double q2 = tri_solve<false>(p, base, inc, nn, 2.0 + mu);
            const double next = mu + (nu2 / q2) * (nu - p.lam) / p.lam;
            if (!(next > mu)) break;
            mu = next;
            nu2 = tri_solve<true>(p, base, inc, nn, 2.0 + mu);
            nu = sqrt(nu2);
            if (fabs(nu - p.lam) <= 1e-14 * p.lam) break;
        }
    }
    // x = y + D'u
    double uprev = 0.0;
    for (int i0 = 0; i0 < n; i0 += kBlock) {
        double yv[kBlock], uv[kBlock];
#pragma unroll
        for (int k = 0; k < kBlock; k++) {
            const int i = i0 + k;
            yv[k] = (i < n) ? p.y[base + (long)i * inc] : 0.0;
            uv[k] = (i < nn) ? p.u[base + (long)i * inc] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kBlock; k++) {
            const int i = i0 + k;
            if (i < n) {
                p.x[base + (long)i * inc] = yv[k] - uprev + uv[k];   // (u_{-1} = u_{nn} = 0)
                uprev = uv[k];
            }
        }
    }
}

// ---- one LONG fibre: the same Newton iteration, every tridiagonal solve parallel inside the fibre ------------------------------------
// With one lane per fibre a single signal of 10^6 samples is 10^6 dependent steps per sweep, a dozen sweeps: ~1 s.  For a handful
// of long contiguous fibres the solves are done by all lanes of the chip instead:
//   * the pivots of the LDL' factorisation of tridiag(-1, a, -1), a = 2 cosh(theta), have a closed form -- d_i = sinh((i + 2) theta) /
//     sinh((i + 1) theta), at theta = 0: (i + 2) / (i + 1) -- so nothing is factored;
//   * what is left of a solve are two first-order linear recurrences (forward z_i = r_i + z_{i-1} / d_{i-1}, backward o_i =
//     (z_i + o_{i+1}) / d_i): compositions of affine maps t -> c t + r, i.e. an (associative) scan.  Three launches per recurrence:
//     every block reduces its 2048 elements to one map, one block scans the block maps, every block applies.
// The dot products of the Newton step (||u||^2, u'v) are summed per block in a fixed order (deterministic).
constexpr int kLongThreads = 256, kLongItems = 8, kLongBlock = kLongThreads * kLongItems;
constexpr int kLongMaxBlocks = 4096;          // block maps scanned by one workgroup: fibres up to 8.4 M samples
constexpr int kLongMinLen = 16384;            // shorter fibres: the lane-per-fibre kernel (its 200 us are not worth 100 launches)

struct LongArgs {
    const double *y;    // the fibre (n samples)
    double *z, *u;      // nn = n - 1 each
    double *amap;       // [blocks][2]: block maps (c, r), then in place: the value entering each block
    double *partial;    // [blocks]: dot-product partials
    int nn;
    double theta, etheta;   // a = 2 cosh(theta) ; exp(-theta)
};

// g(k) = sinh(k theta) / sinh((k + 1) theta) = 1 / d_{k-1}  (g(0) = 0)
__device__ __forceinline__ double long_g(const LongArgs &p, long k) {
    if (k <= 0) return 0.0;
    if (p.theta == 0.0) return (double)k / (double)(k + 1);
    const double a = -2.0 * p.theta * (double)k, b = a - 2.0 * p.theta;
    if (a < -80.0) return p.etheta;
    return p.etheta * (expm1(a) / expm1(b));
}

// PASS 0: forward, rhs = Dy -> z ; 1: forward, rhs = u -> z ; 2: backward -> u, partial = sum o^2 ; 3: backward, partial = sum u o
// logical index k runs the way the recurrence does: forward i = k, backward i = nn - 1 - k.
template <int PASS>
__device__ __forceinline__ void long_coeffs(const LongArgs &p, long k, double &c, double &r) {
    if (PASS <= 1) {
        c = long_g(p, k);
        r = PASS == 0 ? p.y[k + 1] - p.y[k] : p.u[k];
    } else {
        const long i = (long)p.nn - 1 - k;
        c = long_g(p, i + 1);
        r = p.z[i] * c;
    }
}

// ordered composition of the maps held by the threads of a block: after the call thread 0 holds the block's map; with SCAN every
// thread gets, in (pc, pr), the composition of the maps of the threads BEFORE it (identity for thread 0)
template <bool SCAN>
__device__ __forceinline__ void long_block_compose(double &c, double &r, double &pc, double &pr, double (*sc)[kLongThreads]) {
    const int t = threadIdx.x;
    sc[0][t] = c;
    sc[1][t] = r;
    __syncthreads();
    // Hillis-Steele inclusive scan of maps: m_t <- m_t o m_{t-d}
    for (int d = 1; d < kLongThreads; d <<= 1) {
        double c2 = 1.0, r2 = 0.0;
        const bool on = t >= d;
        if (on) { c2 = sc[0][t - d]; r2 = sc[1][t - d]; }
        __syncthreads();
        if (on) {
            r = r + c * r2;
            c = c * c2;
            sc[0][t] = c;
            sc[1][t] = r;
        }
        __syncthreads();
    }
    if (SCAN) {
        pc = t > 0 ? sc[0][t - 1] : 1.0;
        pr = t > 0 ? sc[1][t - 1] : 0.0;
    }
}

template <int PASS>
__global__ __launch_bounds__(kLongThreads) void tv2_long_reduce_kernel(LongArgs p) {
    __shared__ double sc[2][kLongThreads];
    const long k0 = (long)blockIdx.x * kLongBlock + (long)threadIdx.x * kLongItems;
    double c = 1.0, r = 0.0;
#pragma unroll
    for (int j = 0; j < kLongItems; j++) {
        if (k0 + j < p.nn) {
            double cj, rj;
            long_coeffs<PASS>(p, k0 + j, cj, rj);
            r = rj + cj * r;
            c = cj * c;
        }
    }
    double pc, pr;
    long_block_compose<false>(c, r, pc, pr, sc);
    if (threadIdx.x == kLongThreads - 1) {
        p.amap[2 * blockIdx.x] = c;
        p.amap[2 * blockIdx.x + 1] = r;
    }
}

// one workgroup: block maps -> the value entering every block (the recurrences start from 0: z_{-1} = o_{nn} = 0)
__global__ __launch_bounds__(kLongThreads) void tv2_long_scan_kernel(LongArgs p, int blocks) {
    __shared__ double sc[2][kLongThreads];
    constexpr int PER = kLongMaxBlocks / kLongThreads;
    const int b0 = threadIdx.x * PER;
    double c = 1.0, r = 0.0;
    for (int j = 0; j < PER; j++)
        if (b0 + j < blocks) {
            const double cj = p.amap[2 * (b0 + j)], rj = p.amap[2 * (b0 + j) + 1];
            r = rj + cj * r;
            c = cj * c;
        }
    double pc, pr;
    long_block_compose<true>(c, r, pc, pr, sc);
    double v = pr;   // (the map of everything before this thread's blocks, applied to 0)
    for (int j = 0; j < PER; j++)
        if (b0 + j < blocks) {
            const double cj = p.amap[2 * (b0 + j)], rj = p.amap[2 * (b0 + j) + 1];
            p.amap[2 * (b0 + j)] = v;     // entering block b0 + j
            v = rj + cj * v;
        }
}

template <int PASS>
__global__ __launch_bounds__(kLongThreads) void tv2_long_apply_kernel(LongArgs p) {
    __shared__ double sc[2][kLongThreads];
    const long k0 = (long)blockIdx.x * kLongBlock + (long)threadIdx.x * kLongItems;
    double cs[kLongItems], rs[kLongItems];
    double c = 1.0, r = 0.0;
#pragma unroll
    for (int j = 0; j < kLongItems; j++) {
        cs[j] = 1.0;
        rs[j] = 0.0;
        if (k0 + j < p.nn) {
            long_coeffs<PASS>(p, k0 + j, cs[j], rs[j]);
            r = rs[j] + cs[j] * r;
            c = cs[j] * c;
        }
    }
    double pc, pr;
    long_block_compose<true>(c, r, pc, pr, sc);
    double v = pr + pc * p.amap[2 * blockIdx.x];   // entering this thread's first element
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < kLongItems; j++) {
        if (k0 + j < p.nn) {
            v = rs[j] + cs[j] * v;
            if (PASS <= 1) {
                p.z[k0 + j] = v;
            } else {
                const long i = (long)p.nn - 1 - (k0 + j);
                if (PASS == 2) {
                    p.u[i] = v;
                    acc += v * v;
                } else {
                    acc += p.u[i] * v;
                }
            }
        }
    }
    if (PASS >= 2) {   // block sum in a fixed order
        __syncthreads();
        sc[0][threadIdx.x] = acc;
        __syncthreads();
        for (int d = kLongThreads / 2; d > 0; d >>= 1) {
            if ((int)threadIdx.x < d) sc[0][threadIdx.x] += sc[0][threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) p.partial[blockIdx.x] = sc[0][0];
    }
}

__global__ __launch_bounds__(kLongThreads) void tv2_long_sum_kernel(const double *partial, int blocks, double *out) {
    __shared__ double sc[kLongThreads];
    double a = 0.0;
    for (int b = threadIdx.x; b < blocks; b += kLongThreads) a += partial[b];
    sc[threadIdx.x] = a;
    __syncthreads();
    for (int d = kLongThreads / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) sc[threadIdx.x] += sc[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sc[0];
}

// x = y + D'u  (u_{-1} = u_{nn} = 0)
__global__ __launch_bounds__(kLongThreads) void tv2_long_primal_kernel(const double *y, const double *u, double *x, int n, int with_u) {
    for (long i = (long)blockIdx.x * kLongThreads + threadIdx.x; i < n; i += (long)gridDim.x * kLongThreads) {
        const double up = (with_u && i > 0) ? u[i - 1] : 0.0, uc = (with_u && i < n - 1) ? u[i] : 0.0;
        x[i] = y[i] - up + uc;
    }
}

// sum of the result of one solve: returns sum o^2 (rhs = Dy; u <- o) or sum u o (rhs = u)
double tv2_long_solve(LongArgs &p, bool rhs_diff, double mu, int blocks, double *dsum, hipStream_t s) {
    p.theta = mu > 0.0 ? log1p(0.5 * mu + sqrt(mu + 0.25 * mu * mu)) : 0.0;   // acosh(1 + mu / 2)
    p.etheta = exp(-p.theta);
    const dim3 grid((unsigned)blocks), block(kLongThreads);
    if (rhs_diff) {
        hipLaunchKernelGGL(tv2_long_reduce_kernel<0>, grid, block, 0, s, p);
        hipLaunchKernelGGL(tv2_long_scan_kernel, dim3(1), block, 0, s, p, blocks);
        hipLaunchKernelGGL(tv2_long_apply_kernel<0>, grid, block, 0, s, p);
        hipLaunchKernelGGL(tv2_long_reduce_kernel<2>, grid, block, 0, s, p);
        hipLaunchKernelGGL(tv2_long_scan_kernel, dim3(1), block, 0, s, p, blocks);
        hipLaunchKernelGGL(tv2_long_apply_kernel<2>, grid, block, 0, s, p);
    } else {
        hipLaunchKernelGGL(tv2_long_reduce_kernel<1>, grid, block, 0, s, p);
        hipLaunchKernelGGL(tv2_long_scan_kernel, dim3(1), block, 0, s, p, blocks);
        hipLaunchKernelGGL(tv2_long_apply_kernel<1>, grid, block, 0, s, p);
        hipLaunchKernelGGL(tv2_long_reduce_kernel<3>, grid, block, 0, s, p);
        hipLaunchKernelGGL(tv2_long_scan_kernel, dim3(1), block, 0, s, p, blocks);
        hipLaunchKernelGGL(tv2_long_apply_kernel<3>, grid, block, 0, s, p);
    }
    hipLaunchKernelGGL(tv2_long_sum_kernel, dim3(1), block, 0, s, p.partial, blocks, dsum);
    PTV_HIP(hipGetLastError());
    double h = 0.0;
    PTV_HIP(hipMemcpyAsync(&h, dsum, sizeof(double), hipMemcpyDeviceToHost, s));
    PTV_HIP(hipStreamSynchronize(s));
    return h;
}

// one contiguous fibre of n samples; same iteration and stopping rule as tv2_fibres_kernel
void tv2_long_fibre(const double *y, double *x, int n, double lam, hipStream_t s) {
    const int nn = n - 1;
    const int blocks = (nn + kLongBlock - 1) / kLongBlock;
    Scratch z(sizeof(double) * (size_t)nn), u(sizeof(double) * (size_t)nn), amap(sizeof(double) * 2 * (size_t)blocks),
        partial(sizeof(double) * (size_t)blocks), dsum(sizeof(double));
    LongArgs p{y, z.d(), u.d(), amap.d(), partial.d(), nn, 0.0, 1.0};
    double mu = 0.0;
    double nu2 = tv2_long_solve(p, true, mu, blocks, dsum.d(), s);
    double nu = sqrt(nu2);
    const bool inside = !(nu > lam);   // the unconstrained dual lies in the ball: x = y + D'u with T u = Dy, the mean of y
    if (!inside) {
        for (int it = 0; it < 200; it++) {
            const double q2 = tv2_long_solve(p, false, mu, blocks, dsum.d(), s);
            const double next = mu + (nu2 / q2) * (nu - lam) / lam;
            if (!(next > mu)) break;
            mu = next;
            nu2 = tv2_long_solve(p, true, mu, blocks, dsum.d(), s);
            nu = sqrt(nu2);
            if (fabs(nu - lam) <= 1e-14 * lam) break;
        }
    }
    const int pb = (n + kLongThreads - 1) / kLongThreads;
    hipLaunchKernelGGL(tv2_long_primal_kernel, dim3((unsigned)(pb < 4096 ? pb : 4096)), dim3(kLongThreads), 0, s, y, u.d(), x, n, 1);
    PTV_HIP(hipGetLastError());
}

}  // namespace

void tv2_fibres(const double *in, double *out, const int *ns, int nds, int dim, double lam, hipStream_t s) {
    long n = 1;
    for (int i = 0; i < nds; i++) n *= ns[i];
    if (n <= 0) return;
    const size_t bytes = sizeof(double) * (size_t)n;
    FibreGeom g = fibres_along(ns, nds, dim);
    // A handful of long contiguous fibres (a single signal through tv2_1d / TV(p = 2)): the solves run parallel inside the fibre,
    // one fibre after the other -- ~14 launches and two host round trips per Newton iteration, about a millisecond per fibre --
    // so only where that beats one lane per fibre, whose time grows with the fibre LENGTH (~1 us per sample: 0.2 s at 2 x 10^5)
    // whatever the count: count <= len / 2048 (8 fibres of 16 384 samples, 14 of 30 000; 48 of 16 384 stay lane-per-fibre), at most 32.
    const long long_max = g.len / 2048 < 32 ? g.len / 2048 : 32;
    if (g.inc == 1 && g.count <= long_max && g.len >= kLongMinLen && (g.len - 1 + kLongBlock - 1) / kLongBlock <= kLongMaxBlocks && lam > 0.0) {
        for (long j = 0; j < g.count; j++) tv2_long_fibre(in + j * g.len, out + j * g.len, g.len, lam, s);
        count_event(CNT_TV2_LONG_FIBRES, g.count);
        return;
    }
    Scratch d(bytes), z(bytes), u(bytes);   // (the long path above brings its own, fibre-sized)
    if (g.inc == 1 && g.count > 1) {
        // dimension 0: fibres are contiguous, so lanes would stride by the fibre length -- transpose (len x count ->
        // count x len), solve along dimension 1 of the transposed array, transpose back
        Scratch tin(bytes), tout(bytes);
        slab_transpose(in, tin.d(), g.len, g.count, 1, s);
        const FibreGeom gt{g.count, g.len, g.count};
        const Tv2Args a{tin.d(), tout.d(), d.d(), z.d(), u.d(), lam};
        hipLaunchKernelGGL(tv2_fibres_kernel, dim3((unsigned)((gt.count + 63) / 64)), dim3(64), 0, s, a, gt);
        PTV_HIP(hipGetLastError());
        slab_transpose(tout.d(), out, g.count, g.len, 1, s);
        return;
    }
    const Tv2Args a{in, out, d.d(), z.d(), u.d(), lam};
    hipLaunchKernelGGL(tv2_fibres_kernel, dim3((unsigned)((g.count + 63) / 64)), dim3(64), 0, s, a, g);
    PTV_HIP(hipGetLastError());
}


void warm_tv2() {
    hipFuncAttributes attr;
    PTV_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(tv2_fibres_kernel)));
}

}  // namespace ptv
