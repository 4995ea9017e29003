// ops.hpp -- what a fibre sweep reads and writes around the 1-D prox.
//
// Every outer algorithm of the hot path (DR / PD2 / PD / PDR / Yang) is "for every fibre along one dimension:
// build the fibre from a pointwise combination of arrays, prox it, scatter a pointwise combination back".
// The reference does the gather / scatter with per-thread copies and runs the remaining pointwise updates as
// separate (serial) loops (src/TV2Dopt.cpp:411,419,422).  Here both ends are fused into the sweep kernel:
//   y   = Op::load_y(args, idx)            what the walker sees at element idx
//   ext = Op::fetch(args, idx)             the operand values the epilogue needs (global loads only, so a kernel
//                                          can issue a batch of them before any dependent work)
//   Op::finish(args, idx, ext, y, x)       x = prox value at idx; computes and writes every output of the sweep
// The arithmetic inside each op follows the reference's operation order (cited per op) so results agree to the
// last ulps with the CPU path.
#pragma once

#include <hip/hip_runtime.h>

namespace ptv {

struct SweepArgs {
    const double *a = nullptr;  // operand arrays, meaning per op
    const double *b = nullptr;
    const double *c = nullptr;
    double *o0 = nullptr;       // outputs
    double *o1 = nullptr;
    double s0 = 0.0;            // scalar (rho for Yang)
    double lam = 0.0;           // uniform penalty
    const double *w = nullptr;  // per-edge penalties (weighted sweeps); same layout as the data with len-1 along the fibre
};

enum OpId : int {
    OP_PROX = 0,       // o0 = prox(a)
    OP_DR_COL,         // reflection through B_cols
    OP_DR_COL_FINAL,   // projection onto B_cols
    OP_DR_ROW,         // reflection through B_{-rows*} + combiner, unweighted sign convention
    OP_DR_ROW_FINAL,   // recovery step, unweighted
    OP_DRW_ROW,        // weighted sign convention (src/TV2DWopt.cpp)
    OP_DRW_ROW_FINAL,
    OP_PD2_A,          // first Dykstra term
    OP_PD2_B,          // second Dykstra term
    OP_YANG,           // Z/U update of Yang's ADMM
    OP_COUNT
};

struct Ext {
    double e0, e1, e2;
};

template <int ID> struct Op;

// o0 = prox(a)                                  (PD_TV :164-209, PDR_TV :405-458, batched tv1_1d)
template <> struct Op<OP_PROX> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.a[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &, long) { return Ext{0, 0, 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &, double, double x) { p.o0[idx] = x; }
};

// DR, columns (a = t): s = t - prox(t) ; s' = 2 s - t          (src/TV2Dopt.cpp:408-411, 539-547)
template <> struct Op<OP_DR_COL> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.a[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &, long) { return Ext{0, 0, 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &, double y, double x) {
        const double s = y - x;
        p.o0[idx] = 2 * s - y;
    }
};
// final projection: s = t - prox(t)                              (src/TV2Dopt.cpp:427)
template <> struct Op<OP_DR_COL_FINAL> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.a[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &, long) { return Ext{0, 0, 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &, double y, double x) { p.o0[idx] = y - x; }
};

// DR, rows (a = s', b = unary, c = t_old, o0 = t_new):
//   v = U - s' ; tb = U - (v - prox(v)) ; tb' = 2 tb - s' ; t = 0.5 (t + tb')      (src/TV2Dopt.cpp:417-422, 514-520)
template <> struct Op<OP_DR_ROW> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.b[idx] - p.a[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.b[idx], p.a[idx], p.c[idx]}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double y, double x) {
        double tb = e.e0 - (y - x);
        tb = 2 * tb - e.e1;
        p.o0[idx] = 0.5 * (e.e2 + tb);
    }
};
// recovery (a = s, b = unary): out = (U - (v - prox(v))) - s                         (src/TV2Dopt.cpp:429-430)
template <> struct Op<OP_DR_ROW_FINAL> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.b[idx] - p.a[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.b[idx], p.a[idx], 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double y, double x) {
        const double tb = e.e0 - (y - x);
        p.o0[idx] = tb - e.e1;
    }
};

// weighted DR rows: tbw = (v - prox(v)) - U ; tb' = -2 tbw - s' ; t = 0.5 (t + tb')   (src/TV2DWopt.cpp:114-119, 218)
template <> struct Op<OP_DRW_ROW> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.b[idx] - p.a[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.b[idx], p.a[idx], p.c[idx]}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double y, double x) {
        double tb = (y - x) - e.e0;
        tb = -2 * tb - e.e1;
        p.o0[idx] = 0.5 * (e.e2 + tb);
    }
};
// weighted recovery: out = -s - tbw                                                    (src/TV2DWopt.cpp:124-126)
template <> struct Op<OP_DRW_ROW_FINAL> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.b[idx] - p.a[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.b[idx], p.a[idx], 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double y, double x) {
        const double tb = (y - x) - e.e0;
        p.o0[idx] = -e.e1 - tb;
    }
};

// Dykstra term 1 (a = x, b = p_in, o0 = z, o1 = p_out): z = prox(x + p) ; p += x - z   (src/TV2Dopt.cpp:187-213)
template <> struct Op<OP_PD2_A> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.a[idx] + p.b[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], p.b[idx], 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double, double x) {
        p.o0[idx] = x;
        p.o1[idx] = e.e1 + (e.e0 - x);
    }
};
// Dykstra term 2 (a = z, b = q_in, o0 = x, o1 = q_out): x = prox(z + q) ; q += z - x    (src/TV2Dopt.cpp:234-263)
template <> struct Op<OP_PD2_B> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.a[idx] + p.b[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], p.b[idx], 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double, double x) {
        p.o0[idx] = x;
        p.o1[idx] = e.e1 + (e.e0 - x);
    }
};

// Yang ADMM (a = X, b = U_in, o0 = Z, o1 = U_out, s0 = rho):
//   Z = prox_{lambda/rho}(-1/rho U + X) ; U += rho (Z - X)          (src/TV2Dopt.cpp:836-862 ; src/TVNDopt.cpp:733-788)
template <> struct Op<OP_YANG> {
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) {
        return -1. / p.s0 * p.b[idx] + p.a[idx];
    }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], p.b[idx], 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double, double x) {
        p.o0[idx] = x;
        p.o1[idx] = e.e1 + p.s0 * (x - e.e0);
    }
};

}  // namespace ptv
