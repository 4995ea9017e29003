// pin_device.hpp -- device-side pieces shared by the pinning kernels: the LDS plane of one lane group (PinGeom), the
// shared-memory side of pincore.hpp's protocol on LDS (PinShared), group barriers and a group scan.
// pin.hip: a fibre per workgroup or wave.  pinlong.hip: a fibre spread over a grid of workgroups.
#pragma once

#include <hip/hip_runtime.h>

#include "pincore.hpp"

namespace ptv {
namespace pin {

constexpr int kPinThreads = 256;

template <int P, int G, bool WEIGHTED>
struct PinGeom {
    // One pad double per PADK knots: lane t's part starts at t P + (t P) / PADK + 1.  For P >= 32 the lanes of an access
    // are P + 1 doubles apart (odd); for P = 16 they alternate 16 / 17, which also puts the 32 lanes of a half-wave on 32
    // different bank pairs (0, 16, 1, 17, ...) at half the padding -- what lets four workgroups of the 4096-sample
    // geometry share a CU.
    static constexpr int PADK = P < 32 ? 32 : P;
    static constexpr int ROWS = G * P + (G * P) / PADK + 2;   // knots 0 .. G P
    static constexpr int NG = kPinThreads / G;           // fibres per workgroup
    static constexpr int SLOTS = G + 1;
    static constexpr size_t plane_bytes = sizeof(double) * ROWS;
    static constexpr size_t slot_bytes = (size_t)2 * SLOTS * (sizeof(unsigned long long) + sizeof(unsigned));   // [wall][slot]
    static constexpr size_t group_bytes = ((plane_bytes * (WEIGHTED ? 2 : 1) + slot_bytes + 15) / 16) * 16;
    static constexpr size_t lds = group_bytes * NG;
    static_assert(lds <= 160 * 1024, "pinning geometry does not fit the LDS of a CU");
    // address of knot j (and of sample j - 1, which lives there until the sums replace it)
    __device__ static __forceinline__ int sa(int j) { return j + (j > 0 ? (j - 1) / PADK : 0); }
    __device__ static __forceinline__ int lane_base(int t) { return 1 + t * P + (t * P) / PADK; }   // = sa(1 + t P); the lane's P knots follow contiguously
};

template <int P, int G, bool WEIGHTED>
struct PinShared {
    using Geo = PinGeom<P, G, WEIGHTED>;
    static constexpr bool kWeighted = WEIGHTED;
    static constexpr bool kCached = (P <= 16);   // the lane's sums stay in registers for all levels (32 VGPRs)
    double *Sp, *Wp;
    double *ownS, *ownW;      // the lane's own part of the two planes
    double cached[kCached ? P : 1];
    double lam;
    unsigned long long *mx;   // [wall][slot]
    unsigned *arg;
    __device__ __forceinline__ double S(int j) const { return Sp[Geo::sa(j)]; }
    __device__ __forceinline__ double r(int j) const { return WEIGHTED ? Wp[Geo::sa(j)] : lam; }
    __device__ __forceinline__ double own(int, int k) const { return kCached ? cached[kCached ? k : 0] : ownS[k]; }
    __device__ __forceinline__ double own_at(int, int k) const { return ownS[k]; }
    __device__ __forceinline__ void set_own(int, int k, double v) { ownS[k] = v; }
    __device__ __forceinline__ double rown(int, int k) const { return WEIGHTED ? ownW[k] : lam; }
    __device__ __forceinline__ void post(int wall, int slot, double v) {
        atomicMax(&mx[wall * Geo::SLOTS + slot], (unsigned long long)__double_as_longlong(v));   // positive doubles order like their bits
    }
    __device__ __forceinline__ double best(int wall, int slot) const { return __longlong_as_double((long long)mx[wall * Geo::SLOTS + slot]); }
    __device__ __forceinline__ void claim(int wall, int slot, unsigned key) { atomicMin(&arg[wall * Geo::SLOTS + slot], key); }
    __device__ __forceinline__ int knot(int wall, int slot) const {
        const unsigned key = arg[wall * Geo::SLOTS + slot];
        return key == ~0u ? -1 : PinLane<P>::claimed_knot(key);
    }
    __device__ __forceinline__ void clear_best(int slot) { mx[slot] = 0ull; mx[Geo::SLOTS + slot] = 0ull; }
    __device__ __forceinline__ void clear_knot(int slot) { arg[slot] = ~0u; arg[Geo::SLOTS + slot] = ~0u; }
};

// ---- group collectives --------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ void group_sync() {
    if constexpr (G == kPinThreads) {
        __syncthreads();
    } else {   // the group is one wave: order its LDS traffic, nothing to wait for
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int G>
__device__ __forceinline__ bool group_any(bool pred) {
    if constexpr (G == kPinThreads) {
        return __syncthreads_or(pred ? 1 : 0) != 0;
    } else {
        group_sync<G>();
        return __ballot(pred) != 0ull;
    }
}

// inclusive scan of v over the lanes of the group; `total` = the sum over the group.  red: G / 64 doubles of LDS scratch.
template <int G>
__device__ __forceinline__ double group_scan(double v, int t, double *red, double &total) {
    const int lane = t & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    if constexpr (G == 64) {
        total = __shfl(v, 63);
        return v;
    } else {
        const int wave = t >> 6;
        if (lane == 63) red[wave] = v;
        __syncthreads();
        double pre = 0.0, tot = 0.0;
#pragma unroll
        for (int w = 0; w < G / 64; w++) {
            const double x = red[w];
            if (w < wave) pre += x;
            tot += x;
        }
        __syncthreads();
        total = tot;
        return v + pre;
    }
}

// Nearest seeded knot on either side of every lane's range (PinLane::seed).  last / first: the lane's own last / first seeded
// knot as (knot << 1 | wall), 0 / INT_MAX when it has none.  Returns through `before` the largest `last` of the lanes below t
// (0: none -- the fibre start) and through `after` the smallest `first` of the lanes above (INT_MAX: none -- the fibre end).
// red: 2 * (G / 64) ints of LDS scratch.
template <int G>
__device__ __forceinline__ void group_neighbour_seeds(int last, int first, int t, int *red, int &before, int &after) {
    const int lane = t & 63;
    int up = last, dn = first;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int a = __shfl_up(up, d), b = __shfl_down(dn, d);
        if (lane >= d) up = max(up, a);
        if (lane + d < 64) dn = min(dn, b);
    }
    // exclusive: what the lanes strictly below / above hold
    int ex_up = __shfl_up(up, 1), ex_dn = __shfl_down(dn, 1);
    if (lane == 0) ex_up = 0;
    if (lane == 63) ex_dn = 0x7fffffff;
    if constexpr (G > 64) {
        const int wave = t >> 6;
        if (lane == 63) red[wave] = up;                  // the wave's largest
        if (lane == 0) red[G / 64 + wave] = dn;          // the wave's smallest
        __syncthreads();
#pragma unroll
        for (int w = 0; w < G / 64; w++) {
            if (w < wave) ex_up = max(ex_up, red[w]);
            if (w > wave) ex_dn = min(ex_dn, red[G / 64 + w]);
        }
        __syncthreads();
    }
    before = ex_up;
    after = ex_dn;
}

}  // namespace pin
}  // namespace ptv
