// common.hip -- error capture, device selection, per-thread stream, HBM scratch pool, kernel-family timers.
#include "common.hpp"

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

Options &options() {
    static Options o = [] {
        Options v;
        if (const char *e = getenv("PROXTV_CHUNK")) v.chunk = atoi(e);
        if (const char *e = getenv("PROXTV_WARMUP")) v.warmup = atoi(e);
        if (const char *e = getenv("PROXTV_VERBOSE")) v.verbose = atoi(e);
        if (const char *e = getenv("PROXTV_ABLATE")) v.ablate = atoi(e);
        if (const char *e = getenv("PROXTV_BLOCKS_PER_WG")) v.blocks_per_wg = atoi(e);
        if (const char *e = getenv("PROXTV_CHUNK_MIN_LEN")) v.chunk_min_len = atoi(e);
        if (const char *e = getenv("PROXTV_ROUNDS")) v.rounds = atoi(e);
        if (const char *e = getenv("PROXTV_ALONG")) v.along = atoi(e);
        if (const char *e = getenv("PROXTV_ALONG_MIN_LEN")) v.along_min_len = atoi(e);
        if (const char *e = getenv("PROXTV_CHUNK_MODE")) v.chunk_mode = atoi(e);
        return v;
    }();
    return o;
}

// ---- device ------------------------------------------------------------------------------------------------------
static std::once_flag g_dev_once;
static bool g_dev_ok = false;
static char g_dev_why[256] = "";

static void probe_device() {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        snprintf(g_dev_why, sizeof(g_dev_why), "no HIP device available (%s); libproxtv_amd has no CPU fallback",
                 e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        snprintf(g_dev_why, sizeof(g_dev_why), "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_dev_why, sizeof(g_dev_why), "device %d is %s; this library carries gfx950 (MI355X) code objects only",
                 dev, prop.gcnArchName);
        return;
    }
    g_dev_ok = true;
}

void ensure_device() {
    std::call_once(g_dev_once, probe_device);
    if (!g_dev_ok) {
        set_error("%s", g_dev_why);
        throw HipFailure{hipErrorNoDevice};
    }
}

struct ThreadState {
    hipStream_t stream = nullptr;
    std::multimap<size_t, void *> free_blocks;
    ~ThreadState() {
        // process teardown: the HIP runtime may already be gone; leak rather than crash
    }
};
static thread_local ThreadState g_ts;

hipStream_t thread_stream() {
    ensure_device();
    if (!g_ts.stream) PTV_HIP(hipStreamCreateWithFlags(&g_ts.stream, hipStreamNonBlocking));
    return g_ts.stream;
}

// ---- scratch pool --------------------------------------------------------------------------------------------------
Scratch::Scratch(size_t bytes) : bytes_(bytes ? bytes : 8) {
    auto it = g_ts.free_blocks.find(bytes_);
    if (it != g_ts.free_blocks.end()) {
        ptr_ = it->second;
        g_ts.free_blocks.erase(it);
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
    if (ptr_) g_ts.free_blocks.emplace(bytes_, ptr_);
}

void release_scratch() {
    for (auto &kv : g_ts.free_blocks) (void)hipFree(kv.second);
    g_ts.free_blocks.clear();
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
