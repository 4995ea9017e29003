// common.hpp -- host-side plumbing shared by the HIP translation units of libproxtv_amd.so:
// error capture, the per-thread stream, the HBM scratch pool, fibre geometry.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/proxtv_amd.h"

namespace ptv {

// ---- error capture -------------------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
const char *last_error();

struct HipFailure {
    hipError_t code;
};

#define PTV_HIP(call)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            ::ptv::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            throw ::ptv::HipFailure{_e};                                                               \
        }                                                                                              \
    } while (0)

// ---- options (see proxtv_set_option) ---------------------------------------------------------------------------
struct Options {
    int runs = 1;             // rung 0, dimension-0 sweeps on data with many bends known a priori: 1 = interior segments are cut at those bends and
                              // solved run by run (sweep_along_kernel RUNS; exact by construction), 0 = speculative chunks everywhere
    int chunk_mode = -1;    // -1 = chosen per sweep (see `deterministic`); 0..5 pin the chunk geometry policy
    int deterministic = 1;  // 1: the rung of a sweep is a function of (input statistics, lambda) only -- reproducible to the last
                            // bit; 0: hill climb on measured sweep times, seeded by the same statistics (policy.hpp)
    int along = 1;            // dimension-0 sweeps: chunks along the fibre (sweep_along_kernel); 0 = the transposed 64-fibre tile
    int pin = 1;              // geometry rung 3: the pinning solver (pin.hip) where it applies; 0 = global-memory chunks
    int repair_jobs = 1;      // chunked sweeps: failed links across workgroups are repaired one lane per failure first (sweep_repair_jobs_kernel), what that
                              // leaves by the sequential repair kernel: 1 = where the sampled certain fraction says such links fail in numbers
                              // (policy.hpp: kSeedJobs), 2 = always, 0 = the sequential repair alone
    int pin_seed = 2;         // the pinning solver starts from the knots known a priori instead of the fibre ends alone: 1 = jumps above 4 lambda, 2 = windows as well
    int whole = 1;            // fibres of 16 .. chunk_min_len samples: 1 = by length and data (sequential up to 32 samples; one block of the
                              // chunk kernel on noisy data, else whole fibres in LDS), 2 = the whole-fibre-in-LDS kernel, 0 = the sequential kernel
    int chunk_min_len = 96;   // fibres shorter than this take the sequential kernel (measured crossover: 512x512xL volumes, L ~ 96)
    int xlink = 1;            // chunk kernels check the links across their workgroups themselves and tell the repair kernel, in one word,
                              // whether anything is left for it (0: the repair kernel checks every boundary after every sweep)
    int dr_form = 1;          // DR2 / DR2L1W: 1 = the column sweep leaves the row sweep's input and epilogue operand (OP_DR_COL_V / OP_DR_ROW_V)
                              // when the row sweep runs on the robust 64-fibre tile (rung 1); 2 = on rung 0 too; 0 = always the
                              // reference's split (OP_DR_COL / OP_DR_ROW)
    int tile = 1;             // strided sweeps on rungs 0 and 1 (unweighted and weighted): 1 = tiles of 32 fibres x 8 chunks in 4 waves (four
                              // workgroups per CU), 0 = the 64-fibre x 8-wave tile (two)
    int optimistic = 1;       // DR solves whose every sweep will run on rung 0 (decided from the sampled statistics) launch no repair kernels behind
                              // their sweeps: a sweep that leaves anything marks a sticky word, and a solve whose word is set is run again
                              // with the repairs (sweep.hpp: OptimisticScope); 0: a repair launch behind every chunked sweep
    int certify = 0;          // 1: every fibre sweep is followed by a check of the prox's optimality conditions on what it wrote, fibre by fibre;
                              // a fibre that fails is re-solved by the sequential walk and counted (kernel_certify.hpp)
    int verbose = 0;
    int profile = 0;    // per-kernel-family hipEvent timing
    int why = 0;        // tuning aid: count what marks sweeps dirty (proxtv_debug_why)
    int trace = 0;      // profiling aid: per-workgroup phase timestamps of the chunk kernel (proxtv_debug_trace)
    int ablate = 0;     // profiling aid, see ChunkPlan::ablate (results are WRONG when non-zero)
    int debug_legacy_rebuild = 0;   // test aid: the along-fibre kernel's rebuild with the semantics of rounds 1-4 (an unproven chunk values its first
                                    // piece from its own first row) -- results MAY BE WRONG; exists so that the suite can watch `certify` catch it
};
Options &options();
int *option_slot(const char *key);   // null for an unknown key

// ---- event counters: what ran, for tests and tools (process-wide, cumulative; proxtv_debug_counter) ------------------------------
enum Counter {
    CNT_SWEEP_LAUNCHES = 0,    // kernels launched to sweep fibres (chunk / along-fibre / sequential / pinning ... kernels)
    CNT_REPAIR_LAUNCHES,       // sweep_repair_kernel launches behind them
    CNT_REPAIR_JOBS_LAUNCHES,  // sweep_repair_jobs_kernel launches (option repair_jobs; gated by the sampled certain fraction)
    CNT_PIN_SWEEPS,            // sweeps the pinning solver took
    CNT_PIN_CAP_NEXT_RUNG,     // ... of which the grid-wide variant hit its level cap and handed the sweep to the next rung
    CNT_TV2_LONG_FIBRES,       // TV-L2 fibres solved parallel inside the fibre (tv2.hip: tv2_long_fibre)
    CNT_OPTIMISTIC_SOLVES,     // solves that ran without repair launches (option optimistic)
    CNT_OPTIMISTIC_REDONE,     // ... of which a sweep left something: run again with the repairs
    CNT_CERTIFY_SWEEPS,        // option certify: sweeps whose output was checked against the optimality conditions of the prox
    CNT_CERTIFY_FAILURES,      // ... fibres that failed the check and were re-solved by the sequential walk
    CNT_CERTIFY_SKIPPED,       // ... sweeps that could not be checked (an output aliases an operand; lambda <= 0)
    CNT_REPROBES,              // mid-solve samples of sweep operands (policy_reprobe: Dykstra / ADMM loops)
    CNT_COUNT
};
void count_event(Counter c, long n = 1);
long counter_value(const char *name);   // -1 for an unknown name

// ---- device / stream -------------------------------------------------------------------------------------------
// Throws HipFailure (with last_error set) when the current device is not a usable gfx950.  All library state is kept per
// host thread and per device (current_device(): the ordinal hipGetDevice reports, below kMaxDevices).
constexpr int kMaxDevices = 16;
int current_device();
void ensure_device();
// One-time costs belong to initialisation, not to the first solve: each translation unit's code object is uploaded to the
// device the first time one of its kernels is touched (milliseconds apiece); ensure_device() touches one kernel per unit
// the first time a device is used.  (Defined next to the kernels: sweep.hip, pin.hip, pinlong.hip, pointwise.hip, tv2.hip.)
void warm_sweep();
void warm_pin();
void warm_pinlong();
void warm_pointwise();
void warm_tv2();
hipStream_t thread_stream();

// ---- HBM scratch pool ------------------------------------------------------------------------------------------
// Per-thread, per-device cache of device allocations: solvers ask for a handful of image-sized arrays per call and
// hipMalloc/hipFree cost ~100 us each, so blocks are kept and reused (best fit, capped total: common.hip).
class Scratch {
  public:
    explicit Scratch(size_t bytes);
    ~Scratch();
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    template <class T> T *as() const { return static_cast<T *>(ptr_); }
    double *d() const { return static_cast<double *>(ptr_); }
    size_t bytes() const { return bytes_; }

  private:
    void *ptr_ = nullptr;
    size_t bytes_ = 0;
    int dev_ = 0;   // device the block lives on (it returns to that device's pool)
};
void release_scratch();

// ---- fibre geometry ---------------------------------------------------------------------------------------------
// Fibres of an N-D column-major array along one dimension (reference addressing: src/TV2Dopt.cpp:140-145,184):
// fibre j starts at (j / inc) * inc * len + (j % inc), its samples are `inc` elements apart.
struct FibreGeom {
    long inc;    // stride between consecutive samples of one fibre
    int len;     // samples per fibre
    long count;  // number of fibres
};

// q = a / b, r = a % b for non-negative a and positive b.  Fibre numbers and strides fit 32 bits in every array that fits
// HBM's address space many times over, and a 64-bit division costs a GPU lane ~170 instructions against ~25: the narrow form
// runs whenever both operands allow it (the wide one stays for exactness' sake).
__host__ __device__ __forceinline__ void divmod_nonneg(long a, long b, long &q, long &r) {
    if ((((unsigned long)a | (unsigned long)b) >> 32) == 0) {
        const unsigned qa = (unsigned)a / (unsigned)b;
        q = (long)qa;
        r = (long)((unsigned)a - qa * (unsigned)b);
    } else {
        q = a / b;
        r = a % b;
    }
}

inline FibreGeom fibres_along(const int *ns, int nds, int d) {
    long n = 1, inc = 1;
    for (int i = 0; i < nds; i++) n *= ns[i];
    for (int i = 0; i < d; i++) inc *= ns[i];
    return FibreGeom{inc, ns[d], ns[d] > 0 ? n / ns[d] : 0};
}

// ---- per-family kernel timing (bench.py roofline leg) ------------------------------------------------------------
enum KernelFamily { FAM_COL = 0, FAM_ROW = 1, FAM_OTHER = 2, FAM_COUNT = 3 };
struct FamilyTimer {
    // usage: { FamilyTimer t(fam, stream); launch...; }
    FamilyTimer(int fam, hipStream_t s);
    ~FamilyTimer();
    int fam;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
};
void timing_reset();
void timing_collect();  // resolves pending events (call after the stream is synchronised)
double timing_ms(int fam);
long timing_launches(int fam);

}  // namespace ptv
