// common.hip -- error capture, device selection, per-thread stream, HBM scratch pool, kernel-family timers.
#include "common.hpp"

#include <atomic>

#include <cctype>
#include <cstring>

namespace ptv {

// ---- error capture ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

// One table for both ways of setting a knob: proxtv_set_option("key", v) and the environment variable PROXTV_KEY read at
// load time (every knob has both: include/proxtv_amd.h).
namespace {
struct OptionEntry {
    const char *key;
    int Options::*field;
};
constexpr OptionEntry kOptionTable[] = {
    {"runs", &Options::runs},
    {"chunk_mode", &Options::chunk_mode},
    {"deterministic", &Options::deterministic},
    {"along", &Options::along},
    {"pin", &Options::pin},
    {"pin_seed", &Options::pin_seed},
    {"repair_jobs", &Options::repair_jobs},
    {"whole", &Options::whole},
    {"chunk_min_len", &Options::chunk_min_len},
    {"xlink", &Options::xlink},
    {"dr_form", &Options::dr_form},
    {"tile", &Options::tile},
    {"optimistic", &Options::optimistic},
    {"certify", &Options::certify},
    {"verbose", &Options::verbose},
    {"profile", &Options::profile},
    {"why", &Options::why},
    {"trace", &Options::trace},
    {"ablate", &Options::ablate},
    {"debug_legacy_rebuild", &Options::debug_legacy_rebuild},
};
}  // namespace

int *option_slot(const char *key) {
    if (!key) return nullptr;
    Options &o = options();
    for (const OptionEntry &e : kOptionTable)
        if (!strcmp(key, e.key)) return &(o.*(e.field));
    return nullptr;
}

Options &options() {
    static Options o = [] {
        Options v;
        for (const OptionEntry &e : kOptionTable) {
            char name[64] = "PROXTV_";
            size_t n = strlen(name);
            for (const char *c = e.key; *c && n + 1 < sizeof(name); c++) name[n++] = (char)toupper((unsigned char)*c);
            name[n] = 0;
            if (const char *val = getenv(name)) v.*(e.field) = atoi(val);
        }
        // the one knob that makes results WRONG (a profiling aid) must not arrive through the environment unnoticed
        if (v.ablate != 0) fprintf(stderr, "[proxtv_amd] WARNING: PROXTV_ABLATE=%d set in the environment: sweeps skip work and results are WRONG\n", v.ablate);
        return v;
    }();
    return o;
}

// ---- event counters (proxtv_debug_counter) -----------------------------------------------------------------------------
namespace {
std::atomic<long> g_counters[CNT_COUNT];
constexpr const char *kCounterNames[CNT_COUNT] = {"sweep_launches", "repair_launches", "repair_jobs_launches", "pin_sweeps",
                                                  "pin_cap_next_rung", "tv2_long_fibres", "optimistic_solves", "optimistic_redone", "certify_sweeps", "certify_failures",
                                                  "certify_skipped", "reprobes"};
}  // namespace
void count_event(Counter c, long n) { g_counters[c].fetch_add(n, std::memory_order_relaxed); }
long counter_value(const char *name) {
    if (!name) return -1;
    for (int c = 0; c < CNT_COUNT; c++)
        if (!strcmp(name, kCounterNames[c])) return g_counters[c].load(std::memory_order_relaxed);
    return -1;
}

// ---- device ------------------------------------------------------------------------------------------------------
// Everything the library keeps between calls -- stream, scratch pool, the chunk kernels' link buffers and geometry
// policies -- is per host thread AND per HIP device: a thread that moves to another GPU (hipSetDevice, proxtv_init,
// torch.cuda.device) gets that device's own state, never a stream or a cached block of the previous one.
static std::once_flag g_dev_once[kMaxDevices];
static bool g_dev_ok[kMaxDevices] = {};
static char g_dev_why[kMaxDevices][256] = {};

int current_device() {
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        set_error("no HIP device available (%s); libproxtv_amd has no CPU fallback", hipGetErrorString(e));
        throw HipFailure{hipErrorNoDevice};
    }
    if (dev < 0 || dev >= kMaxDevices) {
        set_error("device ordinal %d is outside the %d devices this library keeps state for", dev, kMaxDevices);
        throw HipFailure{hipErrorInvalidDevice};
    }
    return dev;
}

static void probe_device(int dev) {
    hipDeviceProp_t prop;
    const hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        snprintf(g_dev_why[dev], sizeof(g_dev_why[dev]), "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_dev_why[dev], sizeof(g_dev_why[dev]), "device %d is %s; this library carries gfx950 (MI355X) code objects only",
                 dev, prop.gcnArchName);
        return;
    }
    g_dev_ok[dev] = true;
    // load the code objects now (the device is current: probe_device runs under ensure_device on the calling thread)
    const std::string before = last_error();
    try {
        warm_sweep();
        warm_pin();
        warm_pinlong();
        warm_pointwise();
        warm_tv2();
    } catch (const HipFailure &) {
        (void)hipGetLastError();   // not fatal: the first launch will load (or report) what this could not
        set_error("%s", before.c_str());   // ... and PTV_HIP recorded a message before it threw: a solve that then succeeds must not report it
                                           // (what was there before the warm-up -- say, a failed call on another device -- stays)
    }
}

void ensure_device() {
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); libproxtv_amd has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        throw HipFailure{hipErrorNoDevice};
    }
    const int dev = current_device();
    std::call_once(g_dev_once[dev], probe_device, dev);
    if (!g_dev_ok[dev]) {
        set_error("%s", g_dev_why[dev]);
        throw HipFailure{hipErrorNoDevice};
    }
}

struct ThreadState {
    hipStream_t stream = nullptr;
    std::multimap<size_t, void *> free_blocks;
    size_t cached = 0;   // bytes in free_blocks
};
struct ThreadStates {
    ThreadState dev[kMaxDevices];
    bool alive = true;
    ~ThreadStates() {
        // thread exit: hand the cached blocks back (at process teardown the runtime may be gone: errors are ignored);
        // Scratch objects that outlive this one (other thread_locals) free their block directly from now on
        alive = false;
        for (auto &t : dev)
            for (auto &kv : t.free_blocks) (void)hipFree(kv.second);
    }
};
static thread_local ThreadStates g_ts;

static size_t pool_cap_bytes() {
    static const size_t cap = [] {
        const char *e = getenv("PROXTV_POOL_CAP_MB");
        return (size_t)(e ? atol(e) : 65536) << 20;   // 64 GiB of 288: a 64 x 2048^2 batch solve keeps ~11 GiB
    }();
    return cap;
}

hipStream_t thread_stream() {
    ensure_device();
    ThreadState &t = g_ts.dev[current_device()];
    if (!t.stream) PTV_HIP(hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking));
    return t.stream;
}

// ---- scratch pool --------------------------------------------------------------------------------------------------
// Best fit among the cached blocks of this thread and device (a block up to 25 % larger than asked for is taken), a cap
// on what stays cached (PROXTV_POOL_CAP_MB): workloads with varying shapes do not hoard HBM.
Scratch::Scratch(size_t bytes) : bytes_(bytes ? bytes : 8), dev_(current_device()) {
    ThreadState &t = g_ts.dev[dev_];
    auto it = t.free_blocks.lower_bound(bytes_);
    if (it != t.free_blocks.end() && it->first <= bytes_ + bytes_ / 4) {
        ptr_ = it->second;
        bytes_ = it->first;
        t.cached -= it->first;
        t.free_blocks.erase(it);
        return;
    }
    hipError_t e = hipMalloc(&ptr_, bytes_);
    if (e != hipSuccess) {
        // give cached blocks back to the driver and retry once
        release_scratch();
        e = hipMalloc(&ptr_, bytes_);
    }
    if (e != hipSuccess) {
        ptr_ = nullptr;
        set_error("out of memory (hipMalloc of %zu bytes: %s)", bytes_, hipGetErrorString(e));
        throw HipFailure{e};
    }
}

Scratch::~Scratch() {
    if (!ptr_) return;
    if (!g_ts.alive || g_ts.dev[dev_].cached + bytes_ > pool_cap_bytes()) {
        (void)hipFree(ptr_);
        return;
    }
    g_ts.dev[dev_].free_blocks.emplace(bytes_, ptr_);
    g_ts.dev[dev_].cached += bytes_;
}

void release_scratch() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return;
    ThreadState &t = g_ts.dev[dev];
    for (auto &kv : t.free_blocks) (void)hipFree(kv.second);
    t.free_blocks.clear();
    t.cached = 0;
}

// ---- kernel-family timers --------------------------------------------------------------------------------------------
struct Pending {
    int fam;
    hipEvent_t a, b;
};
static thread_local std::vector<Pending> g_pending;
static thread_local double g_ms[FAM_COUNT] = {0, 0, 0};
static thread_local long g_launches[FAM_COUNT] = {0, 0, 0};

FamilyTimer::FamilyTimer(int f, hipStream_t st) : fam(f), s(st) {
    if (!options().profile) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    (void)hipEventRecord(a, s);
}
FamilyTimer::~FamilyTimer() {
    if (!a || !b) return;
    (void)hipEventRecord(b, s);
    g_pending.push_back(Pending{fam, a, b});
}
void timing_reset() {
    for (auto &p : g_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    g_pending.clear();
    for (int i = 0; i < FAM_COUNT; i++) { g_ms[i] = 0; g_launches[i] = 0; }
}
void timing_collect() {
    for (auto &p : g_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { g_ms[p.fam] += ms; g_launches[p.fam]++; }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    g_pending.clear();
}
double timing_ms(int fam) { return (fam >= 0 && fam < FAM_COUNT) ? g_ms[fam] : 0.0; }
long timing_launches(int fam) { return (fam >= 0 && fam < FAM_COUNT) ? g_launches[fam] : 0; }

}  // namespace ptv
