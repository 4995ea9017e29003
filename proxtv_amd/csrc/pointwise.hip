// pointwise.hip -- streaming pointwise updates and reductions of the splitting loops, gfx950.
// All HBM-bound: grid-stride loops, 256-thread blocks, fixed partial-sum layout so results are deterministic.
#include "pointwise.hpp"

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

__global__ __launch_bounds__(kThreads) void pd_combine_kernel(PtrPack p, PtrPack z, const double *x, double *xo, int P,
                                                                long n, double *partials) {
    double acc = 0;
    for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < n; k += (long)gridDim.x * kThreads) {
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

template <int D>
__global__ __launch_bounds__(kThreads) void yang_x_kernel(const double *Y, PtrPack U, PtrPack Z, double *X, double rho,
                                                            long n) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        double su = Y[i], sz = Z.v[0][i];
#pragma unroll
        for (int k = 0; k < D; k++) su += U.v[k][i];
#pragma unroll
        for (int k = 1; k < D; k++) sz += Z.v[k][i];
        X[i] = (su + rho * sz) / (1 + D * rho);
    }
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
    hipLaunchKernelGGL(dr_fill_kernel, dim3(grid_for(n, 1024), (unsigned)segments), dim3(kThreads), 0, s, t, n, sums, sign);
    PTV_HIP(hipGetLastError());
}

void absdiff_to(const double *a, const double *b, long n, double *partials, double *out, hipStream_t s) {
    hipLaunchKernelGGL(absdiff_kernel, dim3(kReduceBlocks), dim3(kThreads), 0, s, a, b, n, partials);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kThreads), 0, s, partials, kReduceBlocks, out);
    PTV_HIP(hipGetLastError());
}

void pd_combine(const PtrPack &p, const PtrPack &z, const double *x, double *xo, int P, long n, double *partials,
                double *out, hipStream_t s) {
    hipLaunchKernelGGL(pd_combine_kernel, dim3(kReduceBlocks), dim3(kThreads), 0, s, p, z, x, xo, P, n, partials);
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

void scale_to(const double *y, double *x, double divisor, long n, hipStream_t s) {
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 4096)), dim3(kThreads), 0, s, y, x, divisor, n);
    PTV_HIP(hipGetLastError());
}

void yang_x(const double *Y, const PtrPack &U, const PtrPack &Z, double *X, int D, double rho, long n, hipStream_t s) {
    const dim3 grid(grid_for(n, 4096)), block(kThreads);
    if (D == 2)      hipLaunchKernelGGL(yang_x_kernel<2>, grid, block, 0, s, Y, U, Z, X, rho, n);
    else if (D == 3) hipLaunchKernelGGL(yang_x_kernel<3>, grid, block, 0, s, Y, U, Z, X, rho, n);
    else { set_error("yang_x: D must be 2 or 3"); throw HipFailure{hipErrorInvalidValue}; }
    PTV_HIP(hipGetLastError());
}

}  // namespace ptv
