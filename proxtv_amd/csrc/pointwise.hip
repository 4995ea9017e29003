// pointwise.hip -- streaming pointwise updates and reductions of the splitting loops, gfx950.
// All HBM-bound: grid-stride loops, 256-thread blocks, fixed partial-sum layout so results are deterministic.
#include "pointwise.hpp"

#include "transposed.hpp"

namespace ptv {

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sum of one value per thread; result valid in thread 0
__device__ __forceinline__ double block_sum(double v) {
    __shared__ double part[kThreads / 64];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) part[wv] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0)
        for (int k = 0; k < kThreads / 64; k++) r += part[k];
    return r;
}

// one segment (image) per blockIdx.y; the partial layout per segment does not depend on the segment count, so a
// batched solve sums each image exactly like a single-image solve
__global__ __launch_bounds__(kThreads) void sum_kernel(const double *a, long n, double *partials) {
    const double *seg = a + (long)blockIdx.y * n;
    double acc = 0;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) acc += seg[i];
    acc = block_sum(acc);
    if (threadIdx.x == 0) partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}

__global__ __launch_bounds__(kThreads) void absdiff_kernel(const double *a, const double *b, long n, double *partials) {
    double acc = 0;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
        acc += fabs(a[i] - b[i]);
    acc = block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ __launch_bounds__(kThreads) void finish_kernel(const double *partials, int count, double *out) {
    const double *seg = partials + (long)blockIdx.x * count;
    double acc = 0;
    for (int i = threadIdx.x; i < count; i += kThreads) acc += seg[i];
    acc = block_sum(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(kThreads) void dr_fill_kernel(double *t, long n, const double *sums, double sign) {
    const double v = sign * (2 * sums[blockIdx.y] / n);
    double *seg = t + (long)blockIdx.y * n;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) seg[i] = v;
}

// Plain 8-bytes-per-lane stream copy: the access width of every sweep kernel.  Used to calibrate the rocprofv3
// FETCH_SIZE / WRITE_SIZE counters against a known byte count (tools/pmc_traffic.sh), nothing else.
__global__ __launch_bounds__(kThreads) void calib_copy_kernel(const double *src, double *dst, long n) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) dst[i] = src[i];
}

__global__ __launch_bounds__(kThreads) void scale_kernel(const double *y, double *x, double divisor, long n) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) x[i] = y[i] / divisor;
}

// Two adjacent elements of a stream as one 16-byte access where the array allows it.  Caller arrays (the *_dev entry points) are
// only guaranteed 8-byte alignment -- a tensor view that starts at an odd element -- so the host side checks every operand and
// picks WIDE = false (two 8-byte accesses, same element -> lane map, same results) when any of them is not 16-byte aligned.
template <bool WIDE>
__device__ __forceinline__ double2 load_pair(const double *a, long k) {
    if (WIDE) return reinterpret_cast<const double2 *>(a)[k];
    return double2{a[2 * k], a[2 * k + 1]};
}
template <bool WIDE>
__device__ __forceinline__ void store_pair(double *a, long k, double2 v) {
    if (WIDE) {
        reinterpret_cast<double2 *>(a)[k] = v;
    } else {
        a[2 * k] = v.x;
        a[2 * k + 1] = v.y;
    }
}
static bool aligned16(const void *ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

// The combine kernels stream 2 P + 1 arrays in and P + 1 out (P = 3: eleven streams).  Two adjacent elements per lane (16-byte
// accesses: half as many, twice as large memory requests per stream) and the term count as a template parameter (pointers in
// scalar registers, every load of an iteration issued before the first use); PT = 0 is the run-time-P form for P > 4.
// Arithmetic and its order per element are the reference's (src/TVNDopt.cpp:212-227); the partial sums of the stopping value
// keep a fixed layout (kReduceBlocks blocks, a fixed element -> lane map): run-to-run deterministic.
template <int PT, bool WIDE>
__global__ __launch_bounds__(kThreads) void pd_combine_kernel(PtrPack p, PtrPack z, const double *x, double *xo, int Prt,
                                                                long n, double *partials) {
    const int P = PT > 0 ? PT : Prt;
    double acc = 0;
    const long pairs = n / 2;
    for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < pairs; k += (long)gridDim.x * kThreads) {
        const double2 xold = load_pair<WIDE>(x, k);
        double2 pv[PT > 0 ? PT : 1], zv[PT > 0 ? PT : 1];
        double2 xn{0, 0};
        if (PT > 0) {
#pragma unroll
            for (int i = 0; i < PT; i++) pv[i] = load_pair<WIDE>(p.v[i], k);
#pragma unroll
            for (int i = 0; i < PT; i++) zv[i] = load_pair<WIDE>(z.v[i], k);
#pragma unroll
            for (int i = 0; i < PT; i++) { xn.x += pv[i].x / P; xn.y += pv[i].y / P; }
#pragma unroll
            for (int i = 0; i < PT; i++) {
                zv[i].x += xn.x - pv[i].x;
                zv[i].y += xn.y - pv[i].y;
                store_pair<WIDE>(z.v[i], k, zv[i]);
            }
        } else {
            for (int i = 0; i < P; i++) {
                const double2 q = load_pair<WIDE>(p.v[i], k);
                xn.x += q.x / P; xn.y += q.y / P;
            }
            for (int i = 0; i < P; i++) {
                const double2 q = load_pair<WIDE>(p.v[i], k);
                double2 w = load_pair<WIDE>(z.v[i], k);
                w.x += xn.x - q.x; w.y += xn.y - q.y;
                store_pair<WIDE>(z.v[i], k, w);
            }
        }
        store_pair<WIDE>(xo, k, xn);
        acc += fabs(xn.x - xold.x);
        acc += fabs(xn.y - xold.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // odd length: the last element
        const long k = n - 1;
        const double xold = x[k];
        double xn = 0;
        for (int i = 0; i < P; i++) xn += p.v[i][k] / P;
        for (int i = 0; i < P; i++) z.v[i][k] += xn - p.v[i][k];
        xo[k] = xn;
        acc += fabs(xn - xold);
    }
    acc = block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ __launch_bounds__(kThreads) void pdr_combine_kernel(PtrPack p, PtrPack z, const double *x, double *xo, int P,
                                                                 long n, double *partials) {
    double acc = 0;
    for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < n; k += (long)gridDim.x * kThreads) {
        const double xold = x[k];
        double q = 0, xn = 0;
        for (int i = 0; i < P; i++) {
            q += p.v[i][k] / P;
            xn += z.v[i][k] / P;
        }
        for (int i = 0; i < P; i++) z.v[i][k] += 2 * q - xn - p.v[i][k];
        xo[k] = xn;
        acc += fabs(xn - xold);
    }
    acc = block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

template <int D, bool WIDE>
__global__ __launch_bounds__(kThreads) void yang_x_kernel(const double *Y, PtrPack U, PtrPack Z, double *X, double rho,
                                                            long n) {
    const long pairs = n / 2;   // two adjacent elements per lane (see pd_combine_kernel)
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < pairs; i += (long)gridDim.x * kThreads) {
        const double2 y = load_pair<WIDE>(Y, i);
        double2 u[D], zz[D];
#pragma unroll
        for (int k = 0; k < D; k++) u[k] = load_pair<WIDE>(U.v[k], i);
#pragma unroll
        for (int k = 0; k < D; k++) zz[k] = load_pair<WIDE>(Z.v[k], i);
        double2 su = y, sz = zz[0];
#pragma unroll
        for (int k = 0; k < D; k++) { su.x += u[k].x; su.y += u[k].y; }
#pragma unroll
        for (int k = 1; k < D; k++) { sz.x += zz[k].x; sz.y += zz[k].y; }
        store_pair<WIDE>(X, i, double2{(su.x + rho * sz.x) / (1 + D * rho), (su.y + rho * sz.y) / (1 + D * rho)});
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const long i = n - 1;
        double su = Y[i], sz = Z.v[0][i];
        for (int k = 0; k < D; k++) su += U.v[k][i];
        for (int k = 1; k < D; k++) sz += Z.v[k][i];
        X[i] = (su + rho * sz) / (1 + D * rho);
    }
}

__global__ __launch_bounds__(kThreads) void kolmo_dual_in_kernel(const double *D, double su, const double *X, const double *Xold,
                                                                   double sigma, double theta, double *V, long n,
                                                                   const int *gate, int *changed) {
    if (gate && *gate == 0) return;
    bool any = false;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const double x = X[i], xo = Xold[i];
        const double u = su * D[i];
        double v = u + sigma * (x + theta * (x - xo));
        v = v / sigma;
        V[i] = v;
        const double d = xo - x;
        any |= (d * d > 0);
    }
    if (changed && any) *changed = 1;
}

__global__ __launch_bounds__(kThreads) void kolmo_primal_in_kernel(const double *X, const double *D, double su, const double *Y,
                                                                     double tau, double c1, double c2, double *V, long n,
                                                                     const int *gate) {
    if (gate && *gate == 0) return;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const double u = su * D[i];
        const double v = X[i] - tau * u;
        V[i] = c1 * (Y[i] + c2 * v);
    }
}

__global__ __launch_bounds__(kThreads) void ccp_init_kernel(const double *Y, double *U1, double *U2, long M, long N) {
    const long n = M * N;
    for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < n; k += (long)gridDim.x * kThreads) {
        const long i = k % M, j = k / M;
        if (i < M - 1) U1[i + (M - 1) * j] = Y[k + 1] - Y[k];
        if (j < N - 1) U2[k] = Y[k + M] - Y[k];
    }
}

__device__ __forceinline__ double clip_to(double u, double lambda) {
    if (u < -lambda) u = -lambda;
    else if (u > lambda) u = lambda;
    return u;
}

template <bool FIRST>
__global__ __launch_bounds__(kThreads) void ccp_step_kernel(CcpArgs a) {
    if (a.gate && *a.gate == 0) return;
    const long M = a.M, N = a.N, n = M * N;
    bool any = false;
    for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < n; k += (long)gridDim.x * kThreads) {
        const long i = k % M, j = k / M;
        // the four duals around pixel (i, j): above / below (U1), left / right (U2)
        double ua = 0, ub = 0, ul = 0, ur = 0;
        if (FIRST) {
            if (i > 0) ua = a.U1o[i - 1 + (M - 1) * j];
            if (i < M - 1) ub = a.U1o[i + (M - 1) * j];
            if (j > 0) ul = a.U2o[k - M];
            if (j < N - 1) ur = a.U2o[k];
        } else {
            const double z = a.Zold[k];
            if (i > 0) ua = clip_to(a.U1o[i - 1 + (M - 1) * j] + a.sigma * (z - a.Zold[k - 1]), a.lambda);
            if (i < M - 1) {
                ub = clip_to(a.U1o[i + (M - 1) * j] + a.sigma * (a.Zold[k + 1] - z), a.lambda);
                a.U1n[i + (M - 1) * j] = ub;
            }
            if (j > 0) ul = clip_to(a.U2o[k - M] + a.sigma * (z - a.Zold[k - M]), a.lambda);
            if (j < N - 1) {
                ur = clip_to(a.U2o[k] + a.sigma * (a.Zold[k + M] - z), a.lambda);
                a.U2n[k] = ur;
            }
        }
        // adjoint of the difference operators: vertical part, then "+=" the horizontal part (reference order)
        double g = (i == 0) ? -ub : (i == M - 1) ? ua : ua - ub;
        if (j == 0) g += -ur;
        else if (j == N - 1) g += ul;
        else g += ul - ur;
        const double x = a.X[k], y = a.Y[k];
        double xt;
        if (a.alg == 0) {
            xt = x - a.tau * (x - y + g);
        } else {
            const double c = 1. / (1. + a.tau);
            xt = c * (x + a.tau * (y - g));
        }
        a.Xn[k] = xt;
        a.Zn[k] = xt + a.theta * (xt - x);
        const double d = xt - x;
        any |= (d * d > 0);
    }
    if (a.changed && any) *a.changed = 1;
}

inline unsigned grid_for(long n, unsigned cap) {
    long b = (n + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace

void sum_to(const double *a, long n, long segments, double *partials, double *out, hipStream_t s) {
    hipLaunchKernelGGL(sum_kernel, dim3(kReduceBlocks, (unsigned)segments), dim3(kThreads), 0, s, a, n, partials);
    hipLaunchKernelGGL(finish_kernel, dim3((unsigned)segments), dim3(kThreads), 0, s, partials, kReduceBlocks, out);
    PTV_HIP(hipGetLastError());
}

void dr_fill(double *t, long n, long segments, const double *sums, double sign, hipStream_t s) {
    if (transpose_cache().active) transpose_cache().forget(t);   // a writer inside a TransposeScope: copies of t are stale
    hipLaunchKernelGGL(dr_fill_kernel, dim3(grid_for(n, 1024), (unsigned)segments), dim3(kThreads), 0, s, t, n, sums, sign);
    PTV_HIP(hipGetLastError());
}

// more terms than a kernel-argument pack holds: pointers from a table in HBM ([0, P) p_i, [P, 2P) z_i)
template <bool DR>
__global__ __launch_bounds__(kThreads) void pd_combine_many_kernel(double *const *table, const double *x, double *xo, int P, long n,
                                                                     double *partials) {
    double acc = 0;
    for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < n; k += (long)gridDim.x * kThreads) {
        const double xold = x[k];
        double xn = 0, q = 0;
        if (DR) {
            for (int i = 0; i < P; i++) {
                q += table[i][k] / P;
                xn += table[P + i][k] / P;
            }
            for (int i = 0; i < P; i++) table[P + i][k] += 2 * q - xn - table[i][k];
        } else {
            for (int i = 0; i < P; i++) xn += table[i][k] / P;
            for (int i = 0; i < P; i++) table[P + i][k] += xn - table[i][k];
        }
        xo[k] = xn;
        acc += fabs(xn - xold);
    }
    acc = block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
void pd_combine_many(double *const *table, const double *x, double *xo, int P, long n, double *partials, double *out, bool dr_variant,
                     hipStream_t s) {
    if (dr_variant) hipLaunchKernelGGL((pd_combine_many_kernel<true>), dim3(kReduceBlocks), dim3(kThreads), 0, s, table, x, xo, P, n, partials);
    else            hipLaunchKernelGGL((pd_combine_many_kernel<false>), dim3(kReduceBlocks), dim3(kThreads), 0, s, table, x, xo, P, n, partials);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kThreads), 0, s, partials, kReduceBlocks, out);
    PTV_HIP(hipGetLastError());
}

void absdiff_to(const double *a, const double *b, long n, double *partials, double *out, hipStream_t s) {
    hipLaunchKernelGGL(absdiff_kernel, dim3(kReduceBlocks), dim3(kThreads), 0, s, a, b, n, partials);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kThreads), 0, s, partials, kReduceBlocks, out);
    PTV_HIP(hipGetLastError());
}

void pd_combine(const PtrPack &p, const PtrPack &z, const double *x, double *xo, int P, long n, double *partials,
                double *out, hipStream_t s) {
    const dim3 grid(kReduceBlocks), block(kThreads);
    bool wide = aligned16(x) && aligned16(xo);
    for (int i = 0; i < P; i++) wide = wide && aligned16(p.v[i]) && aligned16(z.v[i]);
#define PTV_COMBINE(PT)                                                                                                    \
    if (wide) hipLaunchKernelGGL((pd_combine_kernel<PT, true>), grid, block, 0, s, p, z, x, xo, P, n, partials);           \
    else      hipLaunchKernelGGL((pd_combine_kernel<PT, false>), grid, block, 0, s, p, z, x, xo, P, n, partials);
    switch (P) {
        case 1: PTV_COMBINE(1) break;
        case 2: PTV_COMBINE(2) break;
        case 3: PTV_COMBINE(3) break;
        case 4: PTV_COMBINE(4) break;
        default: PTV_COMBINE(0) break;
    }
#undef PTV_COMBINE
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kThreads), 0, s, partials, kReduceBlocks, out);
    PTV_HIP(hipGetLastError());
}

void pdr_combine(const PtrPack &p, const PtrPack &z, const double *x, double *xo, int P, long n, double *partials,
                 double *out, hipStream_t s) {
    hipLaunchKernelGGL(pdr_combine_kernel, dim3(kReduceBlocks), dim3(kThreads), 0, s, p, z, x, xo, P, n, partials);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kThreads), 0, s, partials, kReduceBlocks, out);
    PTV_HIP(hipGetLastError());
}

void calib_copy(const double *src, double *dst, long n, hipStream_t s) {
    hipLaunchKernelGGL(calib_copy_kernel, dim3(grid_for(n, 8192)), dim3(kThreads), 0, s, src, dst, n);
    PTV_HIP(hipGetLastError());
}

__global__ void lincomb_kernel(double *out, const double *a, double ca, const double *b, double cb, const double *c, double cc,
                               const double *d, double cd, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double v = ca * a[i];
        if (b) v += cb * b[i];
        if (c) v += cc * c[i];
        if (d) v += cd * d[i];
        out[i] = v;
    }
}

void lincomb(double *out, const double *a, double ca, const double *b, double cb, const double *c, double cc,
             const double *d, double cd, long n, hipStream_t s) {
    if (transpose_cache().active) transpose_cache().forget(out);
    if (n <= 0) return;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, out, a, ca, b, cb, c, cc, d,
                       cd, n);
    PTV_HIP(hipGetLastError());
}

void scale_to(const double *y, double *x, double divisor, long n, hipStream_t s) {
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 4096)), dim3(kThreads), 0, s, y, x, divisor, n);
    PTV_HIP(hipGetLastError());
}

void yang_x(const double *Y, const PtrPack &U, const PtrPack &Z, double *X, int D, double rho, long n, hipStream_t s) {
    const dim3 grid(grid_for(n, 4096)), block(kThreads);
    bool wide = aligned16(Y) && aligned16(X);
    for (int k = 0; k < D && k < 3; k++) wide = wide && aligned16(U.v[k]) && aligned16(Z.v[k]);
    if (D == 2 && wide)      hipLaunchKernelGGL((yang_x_kernel<2, true>), grid, block, 0, s, Y, U, Z, X, rho, n);
    else if (D == 2)         hipLaunchKernelGGL((yang_x_kernel<2, false>), grid, block, 0, s, Y, U, Z, X, rho, n);
    else if (D == 3 && wide) hipLaunchKernelGGL((yang_x_kernel<3, true>), grid, block, 0, s, Y, U, Z, X, rho, n);
    else if (D == 3)         hipLaunchKernelGGL((yang_x_kernel<3, false>), grid, block, 0, s, Y, U, Z, X, rho, n);
    else { set_error("yang_x: D must be 2 or 3"); throw HipFailure{hipErrorInvalidValue}; }
    PTV_HIP(hipGetLastError());
}

void kolmo_dual_in(const double *D, double su, const double *X, const double *Xold, double sigma, double theta, double *V,
                   long n, const int *gate, int *changed, hipStream_t s) {
    hipLaunchKernelGGL(kolmo_dual_in_kernel, dim3(grid_for(n, 4096)), dim3(kThreads), 0, s, D, su, X, Xold, sigma, theta, V, n,
                       gate, changed);
    PTV_HIP(hipGetLastError());
}

void kolmo_primal_in(const double *X, const double *D, double su, const double *Y, double tau, double c1, double c2,
                     double *V, long n, const int *gate, hipStream_t s) {
    hipLaunchKernelGGL(kolmo_primal_in_kernel, dim3(grid_for(n, 4096)), dim3(kThreads), 0, s, X, D, su, Y, tau, c1, c2, V, n,
                       gate);
    PTV_HIP(hipGetLastError());
}

void ccp_init(const double *Y, double *U1, double *U2, long M, long N, hipStream_t s) {
    hipLaunchKernelGGL(ccp_init_kernel, dim3(grid_for(M * N, 4096)), dim3(kThreads), 0, s, Y, U1, U2, M, N);
    PTV_HIP(hipGetLastError());
}

void ccp_step(const CcpArgs &a, bool first, hipStream_t s) {
    const dim3 grid(grid_for(a.M * a.N, 8192)), block(kThreads);
    if (first) hipLaunchKernelGGL(ccp_step_kernel<true>, grid, block, 0, s, a);
    else       hipLaunchKernelGGL(ccp_step_kernel<false>, grid, block, 0, s, a);
    PTV_HIP(hipGetLastError());
}

namespace {
// element (r, c) of a rows x cols block at in[r + ld_in c] -> out[c + ld_out r]; slab z of the grid is slab_in / slab_out elements on
__global__ __launch_bounds__(256) void slab_transpose_kernel(const double *in, double *out, long rows, long cols, long ld_in, long ld_out,
                                                             long slab_in, long slab_out, const int *gate) {
    __shared__ double tile[32][33];
    if (gate && *gate == 0) return;   // (the copies around a gated sweep: a no-op like the sweep itself)
    in += (long)blockIdx.z * slab_in;
    out += (long)blockIdx.z * slab_out;
    const long r0 = (long)blockIdx.x * 32, c0 = (long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const long r = r0 + tx, c = c0 + k;
        if (r < rows && c < cols) tile[k][tx] = in[r + ld_in * c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const long c = c0 + tx, r = r0 + k;
        if (r < rows && c < cols) out[c + ld_out * r] = tile[tx][k];
    }
}
}  // namespace

void slab_transpose(const double *in, double *out, long rows, long cols, long slabs, hipStream_t s, const int *gate) {
    if (rows <= 0 || cols <= 0 || slabs <= 0) return;
    const dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32), (unsigned)slabs);
    hipLaunchKernelGGL(slab_transpose_kernel, grid, dim3(256), 0, s, in, out, rows, cols, rows, cols, rows * cols, rows * cols, gate);
    PTV_HIP(hipGetLastError());
}

// a rows x cols block of a larger column-major array (leading dimension ld_in) into a block of another (ld_out)
void block_transpose(const double *in, double *out, long rows, long cols, long ld_in, long ld_out, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return;
    const dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32), 1u);
    hipLaunchKernelGGL(slab_transpose_kernel, grid, dim3(256), 0, s, in, out, rows, cols, ld_in, ld_out, 0L, 0L, (const int *)nullptr);
    PTV_HIP(hipGetLastError());
}

// ---- edge statistics (the geometry policy's seed) -------------------------------------------------------------------------------
namespace {
constexpr int kProbeRuns = 4096;        // runs of 64 consecutive elements sampled, at most
constexpr int kProbeRunsPerBlock = 16;

__device__ __forceinline__ int probe_key(double v) {
    long key = (long)(((unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull) >> 49) - ((long)kProbeLowExp << 3);
    return (int)(key < 0 ? 0 : (key >= kProbeBins ? kProbeBins - 1 : key));
}

// (y2 != null: the sampled array is y + c2 y2 -- the operand of a sweep whose input functor adds a second array: policy_reprobe)
__global__ __launch_bounds__(kThreads) void edge_hist_kernel(const double *y, const double *w, long n, long inc, int len, long runs,
                                                             long run_stride, unsigned *hist, const double *y2, double c2) {
    __shared__ unsigned bins[2 * (kProbeBins + 1)];
    for (int b = threadIdx.x; b < 2 * (kProbeBins + 1); b += kThreads) bins[b] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = wv; k < kProbeRunsPerBlock; k += kThreads / 64) {
        const long r = (long)blockIdx.x * kProbeRunsPerBlock + k;
        if (r >= runs) break;
        const long e = r * run_stride + lane;
        bool valid = false;
        double v = 0.0;
        long fibre = -1;
        if (e < n) {
            const long q = e / inc;
            const int pos = (int)(q % len);
            if (pos < len - 1) {
                valid = true;
                v = y2 ? fabs((y[e + inc] + c2 * y2[e + inc]) - (y[e] + c2 * y2[e])) : fabs(y[e + inc] - y[e]);
                if (w) {
                    const double we = w[(q / len) * inc * (len - 1) + (long)pos * inc + e % inc];
                    v = we > 0.0 ? v / we : (v > 0.0 ? 1e300 : 0.0);
                }
                // (which fibre the edge belongs to: dimension 0 -- inc == 1 -- runs ALONG a fibre, other dimensions across fibres)
                fibre = inc == 1 ? q / len : -2;
            }
        }
        if (valid) {
            atomicAdd(&bins[probe_key(v)], 1u);
            atomicAdd(&bins[kProbeBins], 1u);
        }
        // second histogram, dimension 0 only: the total variation of every stretch of 16 consecutive edges of one fibre
        // (sum over 16 lanes) -- a stretch whose variation is small against lambda has nothing for a speculative walk to
        // meet the true walk at, however lively the rest of the array is
        if (inc == 1) {
            double tv = valid ? v : 0.0;
            long fmin = fibre, fmax = fibre;
            bool all = valid;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                tv += __shfl_xor(tv, d);
                const long fo = __shfl_xor(fmin, d), fx = __shfl_xor(fmax, d);
                fmin = fo < fmin ? fo : fmin;
                fmax = fx > fmax ? fx : fmax;
                all = all && (__shfl_xor((int)all, d) != 0);
            }
            if ((lane & 15) == 0 && all && fmin == fmax) {
                atomicAdd(&bins[kProbeBins + 1 + probe_key(tv)], 1u);
                atomicAdd(&bins[2 * kProbeBins + 1], 1u);
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 2 * (kProbeBins + 1); b += kThreads)
        if (bins[b]) atomicAdd(&hist[b], bins[b]);
}
}  // namespace

namespace {
// The stretch statistic for strided dimensions: lanes run over 64 adjacent fibres (coalesced), each lane adds up 16
// consecutive edges of ITS fibre.  Sampled: up to kProbeRuns (fibre group, position) sites spread evenly.
__global__ __launch_bounds__(kThreads) void stretch_hist_kernel(const double *y, const double *w, long inc, int len, long slabs,
                                                                long sites, long groups_per_slab, int starts_per_fibre,
                                                                unsigned *hist, const double *y2, double c2) {
    __shared__ unsigned bins[kProbeBins + 1];
    for (int b = threadIdx.x; b <= kProbeBins; b += kThreads) bins[b] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long site = (long)blockIdx.x * (kThreads / 64) + wv;
    if (site < sites) {
        const long per_slab = groups_per_slab * starts_per_fibre;
        const long slab = site / per_slab, rem = site % per_slab;
        const long grp = rem / starts_per_fibre;
        const int st = (int)(rem % starts_per_fibre);
        const long off = grp * 64 + lane;                        // position inside the slab's inc
        const int pos0 = (int)(((long)st * (len - 17)) / (starts_per_fibre > 1 ? starts_per_fibre - 1 : 1));
        if (off < inc && slab < slabs) {
            const long base = slab * inc * len + off, wbase = slab * inc * (len - 1) + off;
            auto at = [&](long e) { return y2 ? y[e] + c2 * y2[e] : y[e]; };
            double tv = 0.0, prev = at(base + (long)pos0 * inc);
#pragma unroll 4
            for (int k = 1; k <= 16; k++) {
                const double cur = at(base + (long)(pos0 + k) * inc);
                double v = fabs(cur - prev);
                if (w) {
                    const double we = w[wbase + (long)(pos0 + k - 1) * inc];
                    v = we > 0.0 ? v / we : (v > 0.0 ? 1e300 : 0.0);
                }
                tv += v;
                prev = cur;
            }
            atomicAdd(&bins[probe_key(tv)], 1u);
            atomicAdd(&bins[kProbeBins], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= kProbeBins; b += kThreads)
        if (bins[b]) atomicAdd(&hist[kProbeBins + 1 + b], bins[b]);
}
}  // namespace

void edge_histogram(const double *y, const double *w, long n, long inc, int len, unsigned *hist, hipStream_t s, const double *y2, double c2) {
    if (n <= 0 || len < 2) return;
    if (inc > 1 && len >= 18) {
        const long slabs = n / (inc * (long)len), groups = (inc + 63) / 64;
        int starts = (int)(kProbeRuns / 4 / (slabs * groups > 0 ? slabs * groups : 1));
        starts = starts < 1 ? 1 : (starts > len / 16 ? len / 16 : starts);
        long sites = slabs * groups * starts;
        long slabs_used = slabs;
        if (sites > kProbeRuns) {   // (many slabs: the first ones of an evenly strided subset would need a stride; keep it simple -- cap)
            slabs_used = kProbeRuns / (groups * starts) > 0 ? kProbeRuns / (groups * starts) : 1;
            sites = slabs_used * groups * starts;
        }
        hipLaunchKernelGGL(stretch_hist_kernel, dim3((unsigned)((sites + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads), 0, s, y, w,
                           inc, len, slabs_used, sites, groups, starts, hist, y2, c2);
        PTV_HIP(hipGetLastError());
    }
    long runs = (n + 63) / 64;
    long stride = 64;
    if (runs > kProbeRuns) {
        stride = (n / kProbeRuns) & ~63L;
        runs = kProbeRuns;
    }
    const unsigned blocks = (unsigned)((runs + kProbeRunsPerBlock - 1) / kProbeRunsPerBlock);
    hipLaunchKernelGGL(edge_hist_kernel, dim3(blocks), dim3(kThreads), 0, s, y, w, n, inc, len, runs, stride, hist, y2, c2);
    PTV_HIP(hipGetLastError());
}

// ---- transposed operands (transposed.hpp) -----------------------------------------------------------------------------------
static thread_local TransposeCache g_transposed[kMaxDevices];
TransposeCache &transpose_cache() { return g_transposed[current_device()]; }

Scratch *TransposeCache::find(const double *src, const Shape &shape) {
    for (auto &e : entries)
        if (e.src == src && e.shape == shape) return e.copy.get();
    return nullptr;
}

void TransposeCache::remember(const double *src, const Shape &shape, std::unique_ptr<Scratch> copy) {
    for (auto &e : entries)
        if (e.src == src && e.shape == shape) {
            e.copy = std::move(copy);
            return;
        }
    entries.push_back(Entry{src, shape, std::move(copy)});
}

void TransposeCache::forget(const double *p) {
    if (!p) return;
    for (size_t k = entries.size(); k-- > 0;)
        if (entries[k].src == p) entries.erase(entries.begin() + (long)k);
}

TransposedOperands::TransposedOperands(const SweepArgs &args, unsigned in_mask, unsigned out_mask, const FibreGeom &g, hipStream_t s)
    : orig_(args), t_(args), g_(g), s_(s), out_mask_(out_mask), slabs_(g.count / g.inc),
      bytes_(sizeof(double) * (size_t)g.count * (size_t)g.len) {
    if (in_mask & 1u) t_.a = input(args.a, ia_, g.len);
    if (in_mask & 2u) t_.b = input(args.b, ib_, g.len);
    if (in_mask & 4u) t_.c = input(args.c, ic_, g.len);
    if (args.w && g.len > 1) t_.w = input(args.w, iw_, g.len - 1);
    if (out_mask & 1u) { o0_.reset(new Scratch(bytes_)); t_.o0 = o0_->d(); }
    if (out_mask & 2u) { o1_.reset(new Scratch(bytes_)); t_.o1 = o1_->d(); }
}

// A gated sweep (SweepArgs::gate: the loops that enqueue all their iterations and end on a device-side flag -- Kolmogorov2_TV,
// CondatChambollePock2_TV) is a no-op when its gate is closed, and so must the copies around it be: until round 6 the transposition BACK
// ran regardless and wrote the scratch array the skipped kernel never filled over the loop's result (a 2 x 96 image at lambda = 12.7: off by
// 2.75 from the third iteration on -- tools/fuzz.py, seed 701; tests/test_gpu_parity_2d.py).  Gated sweeps keep their copies to themselves
// (no cache: a copy made behind a closed gate is not a copy).
const double *TransposedOperands::input(const double *src, std::unique_ptr<Scratch> &own, int len) {
    TransposeCache &cache = transpose_cache();
    const bool cached = cache.active && !orig_.gate;
    if (cached)
        if (Scratch *c = cache.find(src, shape(len))) return c->d();
    std::unique_ptr<Scratch> copy(new Scratch(sizeof(double) * (size_t)g_.count * (size_t)len));
    slab_transpose(src, copy->d(), g_.inc, len, slabs_, s_, orig_.gate);
    const double *p = copy->d();
    if (cached) cache.remember(src, shape(len), std::move(copy));
    else own = std::move(copy);
    return p;
}

void TransposedOperands::finish() {
    TransposeCache &cache = transpose_cache();
    const bool cached = cache.active && !orig_.gate;
    if (out_mask_ & 1u) {
        slab_transpose(o0_->d(), orig_.o0, g_.len, g_.inc, slabs_, s_, orig_.gate);
        if (cached) cache.remember(orig_.o0, shape(g_.len), std::move(o0_));   // (what was just written, in the form the next strided sweep wants)
        else if (cache.active) cache.forget(orig_.o0);                           // (a copy of the array's old content must not outlive it)
    }
    if (out_mask_ & 2u) {
        slab_transpose(o1_->d(), orig_.o1, g_.len, g_.inc, slabs_, s_, orig_.gate);
        if (cached) cache.remember(orig_.o1, shape(g_.len), std::move(o1_));
        else if (cache.active) cache.forget(orig_.o1);
    }
}

void warm_pointwise() {
    hipFuncAttributes attr;
    PTV_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(calib_copy_kernel)));
}

}  // namespace ptv
