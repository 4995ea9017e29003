// ops.hpp -- what a fibre sweep reads and writes around the 1-D prox.
//
// Every outer algorithm of the hot path (DR / PD2 / PD / PDR / Yang) is "for every fibre along one dimension:
// build the fibre from a pointwise combination of arrays, prox it, scatter a pointwise combination back".
// The reference does the gather / scatter with per-thread copies and runs the remaining pointwise updates as
// separate (serial) loops (src/TV2Dopt.cpp:411,419,422).  Here both ends are fused into the sweep kernel:
//   y   = Op::load_y(args, idx)        what the walker sees at element idx; split as fetch_in (global loads only,
//                                      NIN operands) + y_of (arithmetic) so a kernel can batch the loads of a window
//   ext = Op::fetch(args, idx)         the operand values the epilogue needs (global loads only, so a kernel can
//                                      issue a batch of them before any dependent work)
//   Op::finish(args, idx, ext, x)      x = prox value at idx; computes and writes every output of the sweep
//   x   = Op::recover(args, idx, y, scale)   (option certify) the prox value at idx read back from what the sweep WROTE, for the
//                                      check of the optimality conditions behind the sweep; `scale` grows to the magnitude of
//                                      every operand the recovery touches (its rounding noise is a few ulps of that)
// The arithmetic follows the reference's operation order (cited per op) except where noted, so results agree to
// the last ulps with the CPU path.
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
    const int *gate = nullptr;  // device flag: the sweep is a no-op when *gate == 0 (loops whose exit test lives on the device)
};

enum OpId : int {
    OP_PROX = 0,       // o0 = prox(a)
    OP_DR_COL,         // reflection through B_cols
    OP_DR_COL_FINAL,   // projection onto B_cols
    OP_DR_ROW,         // reflection through B_{-rows*} + combiner (both sign conventions of the reference)
    OP_DR_ROW_FINAL,   // recovery step, unweighted sign convention
    OP_DRW_ROW_FINAL,  // recovery step, weighted sign convention (src/TV2DWopt.cpp)
    OP_PD2_A,          // first Dykstra term
    OP_PD2_B,          // second Dykstra term
    OP_YANG,           // Z/U update of Yang's ADMM
    OP_DR_COL_V,       // reflection through B_cols, leaving the row sweep's input and its epilogue operand (see Op<OP_DR_COL_V>)
    OP_DR_ROW_V,       // row sweep of that form
    OP_COUNT
};

struct Ext {
    double e0, e1;
};

// An op whose epilogue needs again an operand that was staged for the walk (second input of fetch_in) sets KEEP: the
// chunk kernel keeps that operand of the block's own rows in registers and asks only for the rest (fetch_rest).
struct NoKeep {
    static constexpr bool KEEP = false;
    __device__ static __forceinline__ Ext fetch_rest(const SweepArgs &, long, double) { return Ext{0, 0}; }
};

template <int ID> struct Op;

// Streams a sweep touches exactly once -- an operand only its epilogue reads, every output -- are marked non-temporal, so that
// what the sweep reads twice (a window operand the epilogue fetches again, the halo rows two workgroups share) has the XCD's L2
// to itself.  Measured on the DR row sweep (R s', R U | R s', R t, W t: U, t and the new t non-temporal): 126.3 -> 118.5 us per
// 4096^2 launch on the 32-fibre tile, 7.67 -> 7.36 ms per solve (profiles/r04_s1_ab_matrix.txt).  -DPTV_NO_NT_STREAMS: plain.
#ifndef PTV_NO_NT_STREAMS
__device__ __forceinline__ double ld_once(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_once(double *p, double v) { __builtin_nontemporal_store(v, p); }
#else
__device__ __forceinline__ double ld_once(const double *p) { return *p; }
__device__ __forceinline__ void st_once(double *p, double v) { *p = v; }
#endif
// The other ops' outputs and epilogue-only operands likewise (measured apart: 7.10 -> 7.01 ms at lambda = 0.1, 11.91 -> 11.68 at 0.5,
// weighted unchanged).  -DPTV_NO_NT_ALL: plain.
#ifndef PTV_NO_NT_ALL
__device__ __forceinline__ double ld_once2(const double *p) { return ld_once(p); }
__device__ __forceinline__ void st_once2(double *p, double v) { st_once(p, v); }
#else
__device__ __forceinline__ double ld_once2(const double *p) { return *p; }
__device__ __forceinline__ void st_once2(double *p, double v) { *p = v; }
#endif

template <int ID> struct Op;

// Ops whose epilogue fetches BOTH staged operands again (PD2_A / PD2_B / YANG: iterate + correction): the strided tiles keep the
// second one -- the correction -- in registers for PTV_KEEP_N = 8 of a thread's 16 rows (sweep.hip: one to three spilled registers;
// all 16 spill dozens), a twelfth of those sweeps' traffic: PD2 4096^2 4.40 -> 4.33 ms, Yang3 512 x 512 x 64 22.05 -> 21.47 ms
// (profiles/r04_s9_ab.txt).  -DPTV_NO_KEEP_OPS: fetch everything again.
#ifndef PTV_NO_KEEP_OPS
constexpr bool kKeepCorrection = true;
#else
constexpr bool kKeepCorrection = false;
#endif

// ---- one-operand inputs: y = a ---------------------------------------------------------------------------------------
struct InA : NoKeep {
    static constexpr int NIN = 1;
    __device__ static __forceinline__ void fetch_in(const SweepArgs &p, long idx, double &i0, double &i1) { i0 = p.a[idx]; i1 = 0.0; }
    __device__ static __forceinline__ double y_of(const SweepArgs &, double i0, double) { return i0; }
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.a[idx]; }
};
// ---- y = b - a  (DR rows: unary - s') ----------------------------------------------------------------------------------
struct InBminusA : NoKeep {
    static constexpr int NIN = 2;
    __device__ static __forceinline__ void fetch_in(const SweepArgs &p, long idx, double &i0, double &i1) { i0 = ld_once(p.b + idx); i1 = p.a[idx]; }
    // rows that TWO workgroups of a tile sweep stage (a block's zone and look-ahead rows are its neighbours' own): whoever comes first
    // must leave them in the L2 for the other -- no streaming hint on those (sweep_chunk_kernel: stage_as)
    __device__ static __forceinline__ void fetch_in_shared(const SweepArgs &p, long idx, double &i0, double &i1) { i0 = p.b[idx]; i1 = p.a[idx]; }
    __device__ static __forceinline__ double y_of(const SweepArgs &, double i0, double i1) { return i0 - i1; }
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.b[idx] - p.a[idx]; }
};
// ---- y = a + b  (Dykstra: iterate + correction) ------------------------------------------------------------------------
struct InAplusB : NoKeep {
    static constexpr int NIN = 2;
    __device__ static __forceinline__ void fetch_in(const SweepArgs &p, long idx, double &i0, double &i1) { i0 = p.a[idx]; i1 = p.b[idx]; }
    __device__ static __forceinline__ double y_of(const SweepArgs &, double i0, double i1) { return i0 + i1; }
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return p.a[idx] + p.b[idx]; }
};

// An op whose single output is a function of (y, x) alone is FUSED: the chunk kernel evaluates fuse(y, x) in LDS while
// it still holds y, and streams the result out with no operand fetch at all (store_fused).  The others get x and
// fetch what they need (fetch / finish).
// (USES_Y: fuse() really looks at y -- the chunk kernel's rebuild then has to hold on to the sample until the value is known)
struct NotFused {
    static constexpr bool FUSED = false;
    static constexpr bool USES_Y = false;
    __device__ static __forceinline__ double fuse(double, double x) { return x; }
    __device__ static __forceinline__ void store_fused(const SweepArgs &, long, double) {}
};

// o0 = prox(a)                                  (PD_TV :164-209, PDR_TV :405-458, batched tv1_1d)
template <> struct Op<OP_PROX> : InA {
    static constexpr unsigned IN_MASK = 1, OUT_MASK = 1;   // arrays the op reads (bit 0 a, 1 b, 2 c) / writes (bit 0 o0, 1 o1)
    static constexpr bool FUSED = true;
    static constexpr bool USES_Y = false;
    __device__ static __forceinline__ double fuse(double, double x) { return x; }
    __device__ static __forceinline__ void store_fused(const SweepArgs &p, long idx, double v) { st_once2(p.o0 + idx, v); }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &, long) { return Ext{0, 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &, double x) { st_once2(p.o0 + idx, x); }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        (void)y; (void)scale;
        return p.o0[idx];
    }
};

// DR, columns (a = t): s = t - prox(t) ; s' = 2 s - t          (src/TV2Dopt.cpp:408-411, 539-547)
template <> struct Op<OP_DR_COL> : InA {
    static constexpr unsigned IN_MASK = 1, OUT_MASK = 1;
    static constexpr bool FUSED = true;
    static constexpr bool USES_Y = true;
    __device__ static __forceinline__ double fuse(double y, double x) {
        const double s = y - x;
        return 2 * s - y;
    }
    __device__ static __forceinline__ void store_fused(const SweepArgs &p, long idx, double v) { st_once2(p.o0 + idx, v); }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) {
        st_once2(p.o0 + idx, fuse(e.e0, x));
    }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        const double o = p.o0[idx];   // o = 2 (y - x) - y
        scale = fmax(scale, fabs(o));
        return 0.5 * (y - o);
    }
};
// final projection: s = t - prox(t)                              (src/TV2Dopt.cpp:427)
template <> struct Op<OP_DR_COL_FINAL> : InA {
    static constexpr unsigned IN_MASK = 1, OUT_MASK = 1;
    static constexpr bool FUSED = true;
    static constexpr bool USES_Y = true;
    __device__ static __forceinline__ double fuse(double y, double x) { return y - x; }
    __device__ static __forceinline__ void store_fused(const SweepArgs &p, long idx, double v) { st_once2(p.o0 + idx, v); }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) { st_once2(p.o0 + idx, e.e0 - x); }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        const double o = p.o0[idx];   // o = y - x
        scale = fmax(scale, fabs(o));
        return y - o;
    }
};

// DR, rows (a = s', b = unary, c = t_old, o0 = t_new).  Reference (src/TV2Dopt.cpp:417-422, 514-520):
//   v = U - s' ; tb = U - (v - prox(v)) ; tb' = 2 tb - s' ; t = 0.5 (t + tb')
// and, weighted (src/TV2DWopt.cpp:114-119, 218): tbw = (v - prox(v)) - U ; tb' = -2 tbw - s' ; same t.
// Both are t = 0.5 (t + (s' + 2 prox(v))) once U - v is replaced by s' (it IS s' up to the rounding of v): evaluated
// in that form, which needs s' and t but not U at the epilogue -- a few ulps of |U| away from the reference's order.
template <> struct Op<OP_DR_ROW> : InBminusA, NotFused {
    static constexpr unsigned IN_MASK = 7, OUT_MASK = 1;
    // s' is staged for the walk (y = U - s') and needed again here: the strided tiles keep it in registers for PTV_KEEP_N = 8 of a
    // thread's 16 rows (sweep.hip) and fetch the rest again.  All 16 cost the kernel 32 VGPRs it does not have at four workgroups
    // per CU (28 spilled: 116.7 -> 144 us per launch); 8 cost three spilled registers and take a tenth off the sweep's HBM traffic
    // (116.7 -> 114.5 us; the second read is mostly served by the L2 since U, t and the new t are non-temporal).  -DPTV_NO_KEEP_STAGED:
    // fetch every row again.
#ifndef PTV_NO_KEEP_STAGED
    static constexpr bool KEEP = true;
#else
    static constexpr bool KEEP = false;
#endif
    __device__ static __forceinline__ Ext fetch_rest(const SweepArgs &p, long idx, double sp) { return Ext{sp, ld_once(p.c + idx)}; }
#ifdef PTV_EXP_NOREFETCH   // experiment only (wrong results): what would the sweep cost without the second read of s'?
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{0.0, p.c[idx]}; }
#else
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], ld_once(p.c + idx)}; }
#endif
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) {
        const double tb = e.e0 + 2 * x;
        st_once(p.o0 + idx, 0.5 * (e.e1 + tb));
    }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        const double tn = p.o0[idx], t = p.c[idx], sp = p.a[idx];   // tn = 0.5 (t + (sp + 2 x))
        scale = fmax(scale, fabs(tn));
        scale = fmax(scale, fabs(t));
        scale = fmax(scale, fabs(sp));
        return 0.5 * ((2 * tn - t) - sp);
    }
};
// The same iteration with the work split the other way round (round 3; dr2 in solvers.hip picks it when the row sweep runs
// on the 64-fibre tile).  The row tile is the kernel with barriers, halos and two workgroups per CU; the along-fibre column
// kernel is bound by its instruction count and has HBM bandwidth to spare.  So the column sweep does all the pointwise work
// (a = t, b = U, o0 = v, o1 = s):   s = t - prox(t) ; s' = 2 s - t ; v = U - s'      (:408-411 and the gather of :514)
// and the row sweep (a = v, b = s, o0 = t) is a one-operand walk with a one-operand epilogue:   t = s + prox(v)
// -- 0.5 (t + s' + 2 prox(v)) with t + s' = 2 s, a few ulps from OP_DR_ROW's order.  Array passes per iteration: column
// 2 R + 2 W, row 2 R + 1 W (7, against 6 + the second read of s' above), and the row sweep's halo is read for ONE array.
template <> struct Op<OP_DR_COL_V> : InA, NotFused {
    static constexpr unsigned IN_MASK = 3, OUT_MASK = 3;
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], ld_once2(p.b + idx)}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) {
        const double s = e.e0 - x;
        const double sp = 2 * s - e.e0;
        st_once2(p.o0 + idx, e.e1 - sp);
        st_once2(p.o1 + idx, s);
    }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        const double sv = p.o1[idx];   // s = y - x
        scale = fmax(scale, fabs(sv));
        return y - sv;
    }
};
template <> struct Op<OP_DR_ROW_V> : InA, NotFused {
    static constexpr unsigned IN_MASK = 3, OUT_MASK = 1;
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{ld_once2(p.b + idx), 0}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) { st_once2(p.o0 + idx, e.e0 + x); }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        const double tn = p.o0[idx], sv = p.b[idx];   // tn = s + x
        scale = fmax(scale, fabs(tn));
        scale = fmax(scale, fabs(sv));
        return tn - sv;
    }
};
// recovery (a = s, b = unary): out = (U - (v - prox(v))) - s                         (src/TV2Dopt.cpp:429-430)
template <> struct Op<OP_DR_ROW_FINAL> : InBminusA, NotFused {
    static constexpr unsigned IN_MASK = 3, OUT_MASK = 1;
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.b[idx], p.a[idx]}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) {
        const double y = e.e0 - e.e1;
        const double tb = e.e0 - (y - x);
        st_once2(p.o0 + idx, tb - e.e1);
    }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        const double out = p.o0[idx], U = p.b[idx], sv = p.a[idx];   // out = (U - (y - x)) - s
        scale = fmax(scale, fabs(out));
        scale = fmax(scale, fabs(U));
        scale = fmax(scale, fabs(sv));
        return y - (U - (out + sv));
    }
};
// weighted recovery: tbw = (v - prox(v)) - U ; out = -s - tbw                          (src/TV2DWopt.cpp:124-126, 218)
template <> struct Op<OP_DRW_ROW_FINAL> : InBminusA, NotFused {
    static constexpr unsigned IN_MASK = 3, OUT_MASK = 1;
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.b[idx], p.a[idx]}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) {
        const double y = e.e0 - e.e1;
        const double tb = (y - x) - e.e0;
        st_once2(p.o0 + idx, -e.e1 - tb);
    }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        const double out = p.o0[idx], U = p.b[idx], sv = p.a[idx];   // out = -s - ((y - x) - U)
        scale = fmax(scale, fabs(out));
        scale = fmax(scale, fabs(U));
        scale = fmax(scale, fabs(sv));
        return y - ((U - sv) - out);
    }
};

// Dykstra term 1 (a = x, b = p_in, o0 = z, o1 = p_out): z = prox(x + p) ; p += x - z   (src/TV2Dopt.cpp:187-213)
template <> struct Op<OP_PD2_A> : InAplusB, NotFused {
    static constexpr unsigned IN_MASK = 3, OUT_MASK = 3;
    static constexpr bool KEEP = kKeepCorrection;   // (the correction staged for the walk waits in registers for part of the thread's rows)
    __device__ static __forceinline__ Ext fetch_rest(const SweepArgs &p, long idx, double kept) { return Ext{p.a[idx], kept}; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], p.b[idx]}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) {
        st_once2(p.o0 + idx, x);
        st_once2(p.o1 + idx, e.e1 + (e.e0 - x));
    }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        scale = fmax(scale, fmax(fabs(p.a[idx]), fabs(p.b[idx])));   // (y = a + b)
        (void)y;
        return p.o0[idx];
    }
};
// Dykstra term 2 (a = z, b = q_in, o0 = x, o1 = q_out): x = prox(z + q) ; q += z - x    (src/TV2Dopt.cpp:234-263)
template <> struct Op<OP_PD2_B> : InAplusB, NotFused {
    static constexpr unsigned IN_MASK = 3, OUT_MASK = 3;
    static constexpr bool KEEP = kKeepCorrection;   // (the correction staged for the walk waits in registers for part of the thread's rows)
    __device__ static __forceinline__ Ext fetch_rest(const SweepArgs &p, long idx, double kept) { return Ext{p.a[idx], kept}; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], p.b[idx]}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) {
        st_once2(p.o0 + idx, x);
        st_once2(p.o1 + idx, e.e1 + (e.e0 - x));
    }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        scale = fmax(scale, fmax(fabs(p.a[idx]), fabs(p.b[idx])));   // (y = a + b)
        (void)y;
        return p.o0[idx];
    }
};

// Yang ADMM (a = X, b = U_in, o0 = Z, o1 = U_out, s0 = rho):
//   Z = prox_{lambda/rho}(-1/rho U + X) ; U += rho (Z - X)          (src/TV2Dopt.cpp:836-862 ; src/TVNDopt.cpp:733-788)
template <> struct Op<OP_YANG> : NotFused, NoKeep {
    static constexpr unsigned IN_MASK = 3, OUT_MASK = 3;
    static constexpr int NIN = 2;
    static constexpr bool KEEP = kKeepCorrection;
    __device__ static __forceinline__ Ext fetch_rest(const SweepArgs &p, long idx, double kept) { return Ext{p.a[idx], kept}; }
    __device__ static __forceinline__ void fetch_in(const SweepArgs &p, long idx, double &i0, double &i1) { i0 = p.a[idx]; i1 = p.b[idx]; }
    __device__ static __forceinline__ double y_of(const SweepArgs &p, double i0, double i1) { return -1. / p.s0 * i1 + i0; }
    __device__ static __forceinline__ double load_y(const SweepArgs &p, long idx) { return -1. / p.s0 * p.b[idx] + p.a[idx]; }
    __device__ static __forceinline__ Ext fetch(const SweepArgs &p, long idx) { return Ext{p.a[idx], p.b[idx]}; }
    __device__ static __forceinline__ void finish(const SweepArgs &p, long idx, const Ext &e, double x) {
        st_once2(p.o0 + idx, x);
        st_once2(p.o1 + idx, e.e1 + p.s0 * (x - e.e0));
    }
    __device__ static __forceinline__ double recover(const SweepArgs &p, long idx, double y, double &scale) {
        scale = fmax(scale, fmax(fabs(p.a[idx]), fabs(p.b[idx]) / p.s0));   // (y = -U / rho + X)
        (void)y;
        return p.o0[idx];
    }
};

}  // namespace ptv
