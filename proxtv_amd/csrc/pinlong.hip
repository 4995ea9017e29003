// pinlong.hip -- the pinning solver (pincore.hpp) for fibres too long for one workgroup's LDS: a 10^6-sample signal with long
// pieces used to fall through the ladder to ONE sequential lane (0.8 s at lambda = 30; the reference's CPU takes ~15 ms).
//
// Same algorithm, same per-lane code (PinLane), another home for the shared side of its protocol.  The fibre is cut into
// blocks of 4096 knots, one workgroup each, all resident at once (a cooperative launch); lane t of block b is lane
// 256 b + t of the fibre.  A block keeps ITS part of the running sums in LDS, laid out like pin.hip's plane (every knot a
// lane evaluates is its own); what a lane may need of another block -- the sum at a pin that just became its neighbour --
// is read from a copy of the sums in global memory.  The reduction slots (one per lane of the fibre) live in global
// memory too: atomicMax on the violation's bit pattern, atomicMin on a 64-bit claim key.  The three barriers of a level
// span the grid (a counter in global memory, release / acquire fences around it), so a level costs a few microseconds
// more than in LDS -- and there are ~20 of them for a million samples: the whole prox takes about a millisecond,
// whatever the pieces.
//
//   stage     block-local: samples into the LDS plane, the block's sum into global memory           | grid barrier
//   mean      every block adds up the block sums; centred block sums to global memory               | grid barrier
//   sums      block-local scan + the sum of the blocks before it; sums to LDS and to the global copy | grid barrier
//   levels    scan | barrier | claim | barrier | update, "did any lane gain a pin?" | barrier
//   values, stream: block-local, as in pin.hip
//
// Fibres must be contiguous; strided ones go through transposed copies (transposed.hpp).  Several long fibres share a
// launch (and its barriers): blocks per fibre x fibres workgroups, as long as all of them fit on the chip at once.
#include "pin.hpp"

#include <memory>

#include "pin_device.hpp"
#include "transposed.hpp"

namespace ptv {

namespace {

using namespace pin;

constexpr int kLongP = 16;                                // knots per lane
constexpr int kLongBlock = kPinThreads * kLongP;          // knots (samples) per workgroup

using LongKey = unsigned long long;
using LongLane = PinLane<kLongP, LongKey>;

struct LongCtl {             // zeroed before every launch
    unsigned barrier;        // arrivals, never reset: the k-th barrier waits for k * gridDim.x
    unsigned gained[3];      // "some lane gained a pin at level l": word l % 3
    unsigned capped;         // the level cap was hit: nothing was written
};

struct LongArgs {
    double *S;                   // [fibre][len + 1]  running sums, global copy
    unsigned long long *mx;      // [fibre][wall][slots]
    LongKey *arg;                // [fibre][wall][slots]
    double *bsum;                // [workgroup]       block sums, then centred block sums
    LongCtl *ctl;
    int bpf;                     // blocks per fibre
    int slots;                   // lanes per fibre + 1
};

__device__ __forceinline__ unsigned load_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long load_u64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_u64(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_f64(const double *p) {
    return __longlong_as_double((long long)load_u64(reinterpret_cast<const unsigned long long *>(p)));
}
__device__ __forceinline__ void store_f64(double *p, double v) {
    store_u64(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v));
}

// All workgroups of the launch are resident (cooperative launch).  `epoch` counts this thread's barriers.
__device__ __forceinline__ void grid_sync(LongCtl *ctl, unsigned &epoch) {
    __threadfence();   // release what this thread wrote
    __syncthreads();
    epoch++;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctl->barrier, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * gridDim.x;
        while (__hip_atomic_load(&ctl->barrier, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __threadfence();   // acquire what the others wrote
}

// the shared side of pincore.hpp's protocol for one lane of a fibre spread over the grid
template <bool WEIGHTED>
struct LongShared {
    static constexpr bool kWeighted = WEIGHTED;
    const double *Sg;            // the fibre's sums in global memory
    const double *wg;            // the fibre's penalties (weighted): half-width at knot j = wg[j - 1]
    double *ownS;                // the lane's own knots in the block's LDS planes (a settled pin's slot holds the string's height)
    const double *ownW;
    double lam;
    unsigned long long *mx;      // the fibre's slots: [wall][slots]
    LongKey *arg;
    int slots;
    __device__ __forceinline__ double S(int j) const { return load_f64(Sg + j); }
    __device__ __forceinline__ double r(int j) const { return WEIGHTED ? wg[j - 1] : lam; }
    __device__ __forceinline__ double own(int, int k) const { return ownS[k]; }
    __device__ __forceinline__ double own_at(int, int k) const { return ownS[k]; }
    __device__ __forceinline__ void set_own(int, int k, double v) { ownS[k] = v; }
    __device__ __forceinline__ double rown(int, int k) const { return WEIGHTED ? ownW[k] : lam; }
    __device__ __forceinline__ void post(int wall, int slot, double v) {
        __hip_atomic_fetch_max(&mx[(size_t)wall * slots + slot], (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ double best(int wall, int slot) const {
        return __longlong_as_double((long long)load_u64(&mx[(size_t)wall * slots + slot]));
    }
    __device__ __forceinline__ void claim(int wall, int slot, LongKey key) {
        __hip_atomic_fetch_min(&arg[(size_t)wall * slots + slot], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ int knot(int wall, int slot) const {
        const LongKey key = load_u64(&arg[(size_t)wall * slots + slot]);
        return key == ~0ull ? -1 : LongLane::claimed_knot(key);
    }
    __device__ __forceinline__ void clear_best(int slot) {
        store_u64(&mx[slot], 0ull);
        store_u64(&mx[(size_t)slots + slot], 0ull);
    }
    __device__ __forceinline__ void clear_knot(int slot) {
        store_u64(&arg[slot], ~0ull);
        store_u64(&arg[(size_t)slots + slot], ~0ull);
    }
};

template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(kPinThreads) void sweep_pin_long_kernel(SweepArgs p, FibreGeom g, LongArgs a, int *pieces) {
    using Geo = PinGeom<kLongP, kPinThreads, WEIGHTED>;
    constexpr int P = kLongP, UB = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Sp = reinterpret_cast<double *>(smem);            // the block's knots: local knot q (1 .. 4096) at Geo::sa(q)
    double *Wp = Sp + (WEIGHTED ? Geo::ROWS : 0);
    double *red = reinterpret_cast<double *>(smem + Geo::plane_bytes * (WEIGHTED ? 2 : 1));   // 8 doubles of scratch for the scans
    const int tid = threadIdx.x;
    const int fibre = blockIdx.x / a.bpf, blk = blockIdx.x % a.bpf;
    const int n = g.len;
    const long fbase = (long)fibre * n, wbase = (long)fibre * (n - 1);
    const int i0 = blk * kLongBlock;                           // first sample of the block
    const int bn = n - i0 < kLongBlock ? n - i0 : kLongBlock;  // samples in it
    const int T = blk * kPinThreads + tid;                     // lane of the fibre
    double *Sg = a.S + (size_t)fibre * (size_t)(n + 1);
    unsigned epoch = 0;

    // ---- stage: sample i0 + q at local knot q + 1 ----------------------------------------------------------------------------
#pragma unroll 1
    for (int u0 = 0; u0 < P; u0 += UB) {
        double s0[UB], s1[UB], sw[WEIGHTED ? UB : 1];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int q = (u0 + u) * kPinThreads + tid;
            s0[u] = s1[u] = 0.0;
            if (q < bn) Op<OP>::fetch_in(p, fbase + i0 + q, s0[u], s1[u]);
            if (WEIGHTED) sw[WEIGHTED ? u : 0] = (q < bn && i0 + q >= 1) ? p.w[wbase + i0 + q - 1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int q = (u0 + u) * kPinThreads + tid;
            if (q < bn) {
                Sp[Geo::sa(q + 1)] = Op<OP>::y_of(p, s0[u], s1[u]);
                if (WEIGHTED && q >= 1) Wp[Geo::sa(q)] = sw[WEIGHTED ? u : 0];   // half-width at global knot i0 + q = local knot q
            }
        }
    }
    if (WEIGHTED && tid == 0 && bn == kLongBlock && i0 + bn < n) Wp[Geo::sa(kLongBlock)] = p.w[wbase + i0 + bn - 1];   // the block's last knot
    __syncthreads();

    double *own = Sp + Geo::lane_base(tid);   // own[k]: sample i0 + tid P + k, then the sum at global knot i0 + tid P + k + 1
    const int cnt = bn - tid * P < 0 ? 0 : (bn - tid * P < P ? bn - tid * P : P);

    // ---- mean of the fibre -------------------------------------------------------------------------------------------------------
    double mean;
    {
        double ls = 0.0;
        for (int k = 0; k < cnt; k++) ls += own[k];
        double total;
        group_scan<kPinThreads>(ls, tid, red, total);
        if (tid == 0) store_f64(a.bsum + blockIdx.x, total);
        grid_sync(a.ctl, epoch);
        double fs = 0.0;
        for (int b = 0; b < a.bpf; b++) fs += load_f64(a.bsum + (size_t)fibre * a.bpf + b);   // (same order in every block)
        mean = fs / (double)n;
        grid_sync(a.ctl, epoch);                                                                 // everybody has read the block sums
    }
    // ---- centred running sums: block-local scan, then the blocks before ------------------------------------------------------------
    {
        double lc = 0.0;
        for (int k = 0; k < cnt; k++) lc += own[k] - mean;
        double btot;
        const double incl = group_scan<kPinThreads>(lc, tid, red, btot);
        if (tid == 0) store_f64(a.bsum + blockIdx.x, btot);
        grid_sync(a.ctl, epoch);
        double before = 0.0;
        for (int b = 0; b < blk; b++) before += load_f64(a.bsum + (size_t)fibre * a.bpf + b);
        double acc = before + (incl - lc);
        for (int k = 0; k < cnt; k++) {
            acc += own[k] - mean;
            own[k] = acc;
            store_f64(Sg + i0 + tid * P + k + 1, acc);
        }
        if (T == 0) store_f64(Sg, 0.0);
    }
    // reduction slots of this lane: empty
    unsigned long long *mx = a.mx + (size_t)fibre * 2 * a.slots;
    LongKey *arg = a.arg + (size_t)fibre * 2 * a.slots;
    for (int wall = 0; wall < 2; wall++) {
        store_u64(&mx[(size_t)wall * a.slots + T + 1], 0ull);
        store_u64(&arg[(size_t)wall * a.slots + T + 1], ~0ull);
        if (T == 0) {
            store_u64(&mx[(size_t)wall * a.slots], 0ull);
            store_u64(&arg[(size_t)wall * a.slots], ~0ull);
        }
    }
    grid_sync(a.ctl, epoch);

    // ---- levels ----------------------------------------------------------------------------------------------------------------------
    LongShared<WEIGHTED> sh{Sg, WEIGHTED ? p.w + wbase : nullptr, own, Wp + Geo::lane_base(tid), p.lam, mx, arg, a.slots};
    LongLane ln;
    ln.init(n, T, sh);
#pragma unroll 1
    for (unsigned level = 0;; level++) {
        if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(&a.ctl->gained[(level + 1) % 3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ln.scan(sh);
        grid_sync(a.ctl, epoch);
        ln.claim(sh);
        grid_sync(a.ctl, epoch);
        const bool gained = ln.update(sh);
        if (__syncthreads_or(gained ? 1 : 0) && tid == 0)
            __hip_atomic_fetch_or(&a.ctl->gained[level % 3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_sync(a.ctl, epoch);
        if (load_u32(&a.ctl->gained[level % 3]) == 0u) break;
        if (level + 1 >= (unsigned)kPinMaxLevels) {   // (uniform over the grid) periodic ties: the caller takes another rung
            if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(&a.ctl->capped, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }

    // ---- values, in place; then out through the op -------------------------------------------------------------------------------------
    ln.settle(sh);
    ln.values(sh, mean, [&](int, int k, double v) { own[k] = v; });
    if (pieces) {   // a measured launch: pieces of this sweep, for the geometry policy (one atomic per wave)
        int c = __popcll(ln.pinU | ln.pinL) + (T == 0 ? 1 : 0);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
        if ((tid & 63) == 0 && c > 0) atomicAdd(pieces, c);
    }
    __syncthreads();
#pragma unroll 1
    for (int u0 = 0; u0 < P; u0 += UB) {
        Ext ex[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int q = (u0 + u) * kPinThreads + tid;
            ex[u] = (q < bn) ? Op<OP>::fetch(p, fbase + i0 + q) : Ext{0, 0};
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int q = (u0 + u) * kPinThreads + tid;
            if (q < bn) Op<OP>::finish(p, fbase + i0 + q, ex[u], Sp[Geo::sa(q + 1)]);
        }
    }
}

// workgroups of `kern` that fit on the device at once
template <class K>
int resident_workgroups(K kern, size_t lds) {
    int per_cu = 0;
    PTV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kPinThreads, lds));
    hipDeviceProp_t prop;
    PTV_HIP(hipGetDeviceProperties(&prop, current_device()));
    return per_cu * prop.multiProcessorCount;
}

// Returns false when nothing was written: the instantiation does not fit the device at once after all (pin_supports()
// asked about the plain-prox instantiation; another op may hold fewer workgroups per unit), or the level cap was hit.
template <int OP, bool WEIGHTED>
bool launch_long(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *pieces) {
    using Geo = PinGeom<kLongP, kPinThreads, WEIGHTED>;
    constexpr size_t lds = Geo::plane_bytes * (WEIGHTED ? 2 : 1) + 64;
    auto kern = sweep_pin_long_kernel<OP, WEIGHTED>;
    static thread_local int resident[kMaxDevices] = {};
    int &cap = resident[current_device()];
    if (cap == 0) {
        if (lds > 64 * 1024)
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cap = resident_workgroups(kern, lds);
    }
    const int bpf = (g.len + kLongBlock - 1) / kLongBlock;
    const long wgs = (long)bpf * g.count;
    if (wgs > cap) {
        if (options().verbose)
            fprintf(stderr, "[proxtv_amd] pinlong: %ld workgroups do not fit this device at once (%d): next rung\n", wgs, cap);
        return false;
    }
    const int slots = bpf * kPinThreads + 1;
    Scratch S(sizeof(double) * (size_t)g.count * (size_t)(g.len + 1));
    Scratch mx(sizeof(unsigned long long) * (size_t)g.count * 2 * (size_t)slots);
    Scratch arg(sizeof(LongKey) * (size_t)g.count * 2 * (size_t)slots);
    Scratch bsum(sizeof(double) * (size_t)wgs);
    Scratch ctl(sizeof(LongCtl));
    PTV_HIP(hipMemsetAsync(ctl.as<void>(), 0, sizeof(LongCtl), stream));
    LongArgs la{S.d(), mx.as<unsigned long long>(), arg.as<LongKey>(), bsum.d(), ctl.as<LongCtl>(), bpf, slots};
    SweepArgs pa = args;
    FibreGeom ga = g;
    void *params[] = {&pa, &ga, &la, &pieces};
    PTV_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(kern), dim3((unsigned)wgs), dim3(kPinThreads), params, (unsigned)lds, stream));
    // Did it finish?  One word back per launch: these are sweeps of a few very long fibres (a millisecond each), and the
    // scratch above must outlive the kernel anyway.
    unsigned capped = 0;
    PTV_HIP(hipMemcpyAsync(&capped, &ctl.as<LongCtl>()->capped, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    PTV_HIP(hipStreamSynchronize(stream));
    if (capped && options().verbose)
        fprintf(stderr, "[proxtv_amd] pinlong: level cap (%d) hit on fibres of %d samples: next rung\n", kPinMaxLevels, g.len);
    return capped == 0;
}

template <int OP, bool WEIGHTED>
bool launch_long_op(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *pieces) {
    if (g.inc == 1) return launch_long<OP, WEIGHTED>(args, g, stream, pieces);
    TransposedOperands tr(args, Op<OP>::IN_MASK, Op<OP>::OUT_MASK, g, stream);
    if (!launch_long<OP, WEIGHTED>(tr.args(), tr.geom(), stream, pieces)) return false;
    tr.finish();
    return true;
}

}  // namespace

// What the current device holds of the grid-wide kernel (the plain-prox instantiation: the others are checked at launch).
long pin_long_capacity(bool weighted) {
    static thread_local long cap[kMaxDevices][2] = {};
    long &c = cap[current_device()][weighted ? 1 : 0];
    if (c == 0) {
        if (weighted) {
            using Geo = PinGeom<kLongP, kPinThreads, true>;
            constexpr size_t lds = Geo::plane_bytes * 2 + 64;
            auto kern = sweep_pin_long_kernel<OP_PROX, true>;
            if (lds > 64 * 1024)
                PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            c = resident_workgroups(kern, lds);
        } else {
            using Geo = PinGeom<kLongP, kPinThreads, false>;
            constexpr size_t lds = Geo::plane_bytes + 64;
            c = resident_workgroups(sweep_pin_long_kernel<OP_PROX, false>, lds);
        }
        if (c <= 0) c = -1;   // (asked and answered: nothing fits)
    }
    return c > 0 ? c : 0;
}

bool launch_pin_long(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *pieces) {
#define PTV_LONG_CASE(ID)                                                               \
    case ID:                                                                            \
        return weighted ? launch_long_op<ID, true>(args, g, stream, pieces)             \
                        : launch_long_op<ID, false>(args, g, stream, pieces);
#define PTV_LONG_CASE_U(ID) case ID: return launch_long_op<ID, false>(args, g, stream, pieces);
    switch (op) {
        PTV_LONG_CASE(OP_PROX)
        PTV_LONG_CASE(OP_DR_COL)
        PTV_LONG_CASE(OP_DR_COL_FINAL)
        PTV_LONG_CASE_U(OP_DR_ROW)
        PTV_LONG_CASE_U(OP_DR_ROW_FINAL)
        PTV_LONG_CASE_U(OP_PD2_A)
        PTV_LONG_CASE_U(OP_PD2_B)
        PTV_LONG_CASE_U(OP_YANG)
        PTV_LONG_CASE(OP_DR_COL_V)
        PTV_LONG_CASE_U(OP_DR_ROW_V)
        default:
            set_error("launch_pin_long: unsupported op %d", (int)op);
            throw HipFailure{hipErrorInvalidValue};
    }
#undef PTV_LONG_CASE
#undef PTV_LONG_CASE_U
}


void warm_pinlong() {
    hipFuncAttributes attr;
    PTV_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>((sweep_pin_long_kernel<OP_PROX, false>))));
}

}  // namespace ptv
