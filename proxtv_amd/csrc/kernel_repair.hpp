// kernel_repair.hpp -- kernels 3 / 3a: the repair of unproven stretches, sequential per fibre and one lane per failure.
// (One of the pieces of sweep_kernels.hpp, which includes them in order; not meant to be included on its own.)
#pragma once

namespace ptv {
namespace swp {

// ---- kernel 3: local repair of unproven stretches -----------------------------------------------------------------------
// One lane per fibre.  Fast path (the common case): every link is proven -> return.  Otherwise the lane scans its
// chunks in order keeping `cur` = the last bend of the TRUE walk (chunk 0 is true by construction; a chunk whose
// `mine` code equals `cur` continues the true walk, so its outputs and its `next` code are true).  At the first
// chunk that does not, a sequential walk restarts from `cur` -- the walker state after a bend is a function of the
// bend alone -- rewrites the outputs from that chunk on, and after every chunk boundary it crosses checks whether the
// chunk recorded there continues ITS walk (same last bend): if so the recorded outputs beyond are exact and the walk
// stops; the scan resumes there.  Cost: the unproven stretches only (plus the overhang of their last piece), not
// the fibre.  Data with pieces much longer than a chunk fail everywhere and degrade to one sequential walk per fibre.
// The scan does not visit the chunks in between two failures (round 4): the chunk kernels PROVED every link they did not flag, so
// only the flagged range and the first chunks of workgroups whose link in failed can be rejected, and the scan jumps from one of
// those to the next (one bit per boundary, found by the check of the links across workgroups anyway).  At the upper end of rung 1
// (lambda = 0.65 - 0.7 on unit noise: ~600 failed links across tile workgroups per row sweep) the lanes of a wave scan in lockstep
// between their walks, and the scan was a third of the kernel: 4096^2 DR 18.3 -> 17.7, 22.1 -> 21.4 ms.
constexpr link_t kFromStart = 1;   // "no bend yet: the true walk is still in its first piece" (real codes are >= 2)

// what a repair walk keeps track of, whatever it reads its samples from
struct RepairBook {
    const link_t *code_mine;   // code of (chunk c, fibre j) at [c * cstride + j * fstride]
    long cstride, fstride, j;
    int C, len;
    int wfrom = 0;             // outputs are (re)written from this sample on
    int boundary = 0;          // next chunk boundary whose chunk may take over
    link_t last = 0;           // last bend of this walk so far
    bool stop = false;
    int resume_chunk = 0;
    link_t resume_code = 0;

    // `cur`: the last bend of the true walk at or before the chunk (0: none, the walk starts at sample 0).  The chunk
    // kernels leave the rows of a piece to the lane in whose chunk it ends, and an unproven lane keeps to its own rows:
    // the rows between that bend and the chunk belong to the repair walk as well.
    __device__ __forceinline__ void begin(int chunk, link_t cur) {
        wfrom = cur ? (int)(cur >> 1) : 0;
        boundary = (chunk + 1) * C;
        last = cur;
        stop = false;
    }
    __device__ __forceinline__ void bend(int at, int type) {
        const link_t code = ((link_t)at << 1) | (link_t)type;
        while (!stop && boundary < len && at >= boundary) {
            const link_t here = (at == boundary) ? code : last;      // this walk's last bend at-or-before `boundary`
            const int c = boundary / C;
            link_t m = code_mine[(long)c * cstride + j * fstride];
            if (m != kLinkBad) m &= ~kLinkCertain;
            if (m != 0 && m == here) {
                stop = true;
                resume_chunk = c;
                resume_code = here;
            } else {
                boundary += C;
            }
        }
        last = code;
    }
    __device__ __forceinline__ bool keep_going(int) const { return !stop; }
    __device__ __forceinline__ int limit() const { return 1 << 30; }
};

// repair walk straight from global memory (the global-memory geometries: long stretches, pipelined walker)
template <int OP, bool WEIGHTED>
struct RepairSource : RepairBook {
    const SweepArgs &p;
    long base, inc, wbase;
    LazyRun<OP> run;
    __device__ __forceinline__ RepairSource(const RepairBook &b, const SweepArgs &p_, long base_, long inc_, long wbase_)
        : RepairBook(b), p(p_), base(base_), inc(inc_), wbase(wbase_) {}
    __device__ __forceinline__ double y(int i) const { return Op<OP>::load_y(p, base + (long)i * inc); }
    __device__ __forceinline__ double r(int i) const { return p.w[wbase + (long)i * inc]; }
    __device__ __forceinline__ void piece(int from, int to, double v) {
        if (to >= wfrom) run.queue(p, base, inc, max(from, wfrom), to, v);
    }
    __device__ __forceinline__ void pump() { run.pump(p, base, inc); }
    __device__ __forceinline__ void flush() { run.flush(p, base, inc); }
};

// Repair walk through a per-lane LDS window (the LDS geometries: short stretches of short pieces, where a dependent
// global access per sample AND per piece is all the cost -- 200 us for a 100-sample repair).  The lane fetches
// kRepairWindow samples of its fibre in batches of 32 independent loads, walks them out of LDS, parks the piece values
// in a second LDS plane and writes the outputs of the whole stretch at the end, 16 operand fetches in flight.
constexpr int kRepairWindow = 64;
constexpr int kRepairBack = 8;   // samples kept before the one that triggered a refill (short rewinds stay inside)

template <int OP, bool WEIGHTED>
struct WindowRepairSource : RepairBook {
    const SweepArgs &p;
    long base, inc, wbase;
    double *Yw, *Xw, *Rw;      // this lane's columns of the LDS planes: window slot s at [s * 64]
    int wlo = 0, whi = 0;      // samples in the window: [wlo, whi)
    int xlo = 0, xhi = 0;      // samples whose outputs wait in Xw: [xlo, xhi)
    __device__ __forceinline__ WindowRepairSource(const RepairBook &b, const SweepArgs &p_, long base_, long inc_,
                                                  long wbase_, double *lds, int lane)
        : RepairBook(b), p(p_), base(base_), inc(inc_), wbase(wbase_), Yw(lds + lane),
          Xw(lds + kRepairWindow * 64 + lane), Rw(lds + 2 * kRepairWindow * 64 + lane) {}

    // (this kernel is a handful of waves, each as slow as its slowest lane's chain of memory round trips: the batches are as
    // large as the registers of a wave that has the SIMD to itself allow)
    static constexpr int kFlushBatch = 16, kFillBatch = WEIGHTED ? 16 : 32;
    __device__ __forceinline__ void flush() {
        for (int k = xlo; k < xhi; k += kFlushBatch) {
            Ext e[kFlushBatch];
#pragma unroll
            for (int u = 0; u < kFlushBatch; u++)
                if (k + u < xhi) e[u] = Op<OP>::fetch(p, base + (long)(k + u) * inc);
#pragma unroll
            for (int u = 0; u < kFlushBatch; u++)
                if (k + u < xhi) Op<OP>::finish(p, base + (long)(k + u) * inc, e[u], Xw[(k + u - wlo) * 64]);
        }
        xlo = xhi = 0;
    }
    __device__ __forceinline__ void refill(int i) {
        flush();   // the parked outputs are addressed relative to the window
        wlo = max(0, i - kRepairBack);
        whi = min(len, wlo + kRepairWindow);
        for (int b = 0; b < kRepairWindow; b += kFillBatch) {
            double t[kFillBatch], rr[WEIGHTED ? kFillBatch : 1];
#pragma unroll
            for (int u = 0; u < kFillBatch; u++) {
                const int k = wlo + b + u;
                t[u] = (k < whi) ? Op<OP>::load_y(p, base + (long)k * inc) : 0.0;
                if (WEIGHTED) rr[WEIGHTED ? u : 0] = (k < whi && k < len - 1) ? p.w[wbase + (long)k * inc] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kFillBatch; u++) {
                Yw[(b + u) * 64] = t[u];
                if (WEIGHTED) Rw[(b + u) * 64] = rr[WEIGHTED ? u : 0];
            }
        }
    }
    __device__ __forceinline__ double y(int i) {
        if (i < wlo || i >= whi) refill(i);
        return Yw[(i - wlo) * 64];
    }
    __device__ __forceinline__ double r(int i) {
        if (i < wlo || i >= whi) refill(i);
        return Rw[(i - wlo) * 64];
    }
    __device__ __forceinline__ void piece(int from, int to, double v) {
        from = max(from, wfrom);
        if (from > to) return;
        if (from < wlo || to >= whi) {   // (partly) outside the window -- a piece longer than the look-back: write it directly
            const int a = (to >= whi) ? to : min(to, wlo - 1);
            if (to >= whi) flush();
            write_run<OP>(p, base, inc, from, a, v);
            from = a + 1;
            if (from > to) return;
        }
        for (int k = from; k <= to; k++) Xw[(k - wlo) * 64] = v;
        if (xlo == xhi) xlo = from;
        xhi = to + 1;
    }
};

// ---- kernel 3a (staged for round 5, option repair_jobs): one lane per FAILURE ---------------------------------------------------------
// At the upper end of rung 1 the repair kernel is a fifth of a solve, and a launch lasts as long as its worst fibre: 3-4 failed
// links across workgroups repaired one after the other by one lane.  Those failures are almost always independent -- a speculative
// walk that missed its link meets the true one within a chunk or two, far from the next failure 8+ chunks on -- so here every
// failing boundary gets its own lane ("job"): it starts from the bend its predecessor chunk recorded, walks through ONE window of
// its fibre and PARKS its values in LDS, writing nothing.  Then the (at most four) jobs of a fibre, four adjacent lanes, are
// looked at in order: job k is valid iff the last valid job before it re-synchronised at a chunk r <= X_k - 1 -- then chunk
// X_k - 1's recorded codes are true, which is all job k assumed; a job the previous valid walk ran through is discarded (that walk
// IS the truth there).  Valid jobs flush, and the fibre is marked handled for the sequential kernel behind.  Anything else -- a walk
// that leaves its window, more than four failures, links flagged inside a workgroup -- touches nothing and leaves the fibre to the
// sequential kernel: exactness never rests on this one.
#ifndef PTV_JOB_WINDOW
#define PTV_JOB_WINDOW 128
#endif
constexpr int kJobWindow = PTV_JOB_WINDOW;   // samples a job may see: from the bend it starts at
constexpr int kJobsPerFibre = 4;
constexpr int kJobAhead = 4;      // codes of the chunks behind the failed link that a job fetches before it walks

// A job is one wave-lane alone with the memory latency (a workgroup per CU, one wave): what it costs is the number of DEPENDENT
// round trips, ~2 us each.  So everything is fetched in as few, as wide batches as the registers allow: the fail flags with all the
// boundary codes (1), the codes around the failed link -- where the walk starts, where it may hand over -- (1), the first 64
// samples of the window (1; the second 64 only for the walk that gets that far), the operands of the outputs 32 at a time (1-2).
template <int OP, bool WEIGHTED>
struct JobSource : RepairBook {
    const SweepArgs &p;
    long base, inc, wbase;
    // The outputs take the place of the samples: a piece covers samples up to its bend, and the walk never looks at or before a bend again.
    double *Yw, *Rw;           // this lane's columns of the LDS planes: window slot s at [s * 64]
    int wlo = 0, whi = 0;      // samples of the window: [wlo, whi), of which the first `got` are in LDS
    int got = 0;
    int xlo = 0, xhi = 0;      // samples whose outputs wait in Yw: [xlo, xhi)
    bool abort = false;        // the walk needed something outside its window
    int ahead0 = 1 << 30;      // ahead[u] = code_mine of chunk ahead0 + u
    link_t ahead[kJobAhead] = {};
    __device__ __forceinline__ JobSource(const RepairBook &b, const SweepArgs &p_, long base_, long inc_, long wbase_, double *lds, int lane)
        : RepairBook(b), p(p_), base(base_), inc(inc_), wbase(wbase_), Yw(lds + lane), Rw(lds + kJobWindow * 64 + lane) {}
    static constexpr int kFillBatch = WEIGHTED ? 32 : 64;
    __device__ __forceinline__ void fill_more() {
        double t[kFillBatch], rr[WEIGHTED ? kFillBatch : 1];
#pragma unroll
        for (int u = 0; u < kFillBatch; u++) {
            const int k = wlo + got + u;
            t[u] = (k < whi) ? Op<OP>::load_y(p, base + (long)k * inc) : 0.0;
            if (WEIGHTED) rr[WEIGHTED ? u : 0] = (k < whi && k < len - 1) ? p.w[wbase + (long)k * inc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kFillBatch; u++) {
            Yw[(got + u) * 64] = t[u];
            if (WEIGHTED) Rw[(got + u) * 64] = rr[WEIGHTED ? u : 0];
        }
        got += kFillBatch;
    }
    __device__ __forceinline__ void fill(int from) {
        wlo = max(0, from - 1);   // (a weighted restart reads the edge before its bend)
        whi = min(len, wlo + kJobWindow);
        got = 0;
        fill_more();
        if (WEIGHTED) fill_more();
    }
    __device__ __forceinline__ double y(int i) {
        if (i < wlo || i >= whi) { abort = true; return 0.0; }
        while (i - wlo >= got) fill_more();
        return Yw[(i - wlo) * 64];
    }
    __device__ __forceinline__ double r(int i) {
        if (i < wlo || i >= whi) { abort = true; return 0.0; }
        while (i - wlo >= got) fill_more();
        return Rw[(i - wlo) * 64];
    }
    // (job_walk: samples known to be in LDS)
    __device__ __forceinline__ double win_y(int i) const { return Yw[(i - wlo) * 64]; }
    __device__ __forceinline__ double win_r(int i) const { return Rw[(i - wlo) * 64]; }
    __device__ __forceinline__ void piece(int from, int to, double v) {
        from = max(from, wfrom);
        if (from > to) return;
        if (from < wlo || to >= whi) { abort = true; return; }
        while (to - wlo >= got) fill_more();   // (the samples behind the piece's end must be in before outputs take their place)
        for (int k = from; k <= to; k++) Yw[(k - wlo) * 64] = v;
        if (xlo == xhi) xlo = from;
        xhi = to + 1;
    }
    // RepairBook::bend with the codes of the first chunks behind the failed link out of registers
    __device__ __forceinline__ void bend(int at, int type) {
        const link_t code = ((link_t)at << 1) | (link_t)type;
        while (!stop && boundary < len && at >= boundary) {
            const link_t here = (at == boundary) ? code : last;
            const int c = boundary / C;
            const int d = c - ahead0;
            link_t m;
            if (d >= 0 && d < kJobAhead) {
                m = ahead[0];
#pragma unroll
                for (int u = 1; u < kJobAhead; u++) m = (d == u) ? ahead[u] : m;
            } else {
                m = code_mine[(long)c * cstride + j * fstride];
            }
            if (m != kLinkBad) m &= ~kLinkCertain;
            if (m != 0 && m == here) {
                stop = true;
                resume_chunk = c;
                resume_code = here;
            } else {
                boundary += C;
            }
        }
        last = code;
    }
    __device__ __forceinline__ bool keep_going(int) const { return !stop && !abort; }
#ifdef PTV_JOB_TABDIV
    // a / span through a table of reciprocals in LDS (span <= kJobWindow) and one correction: the dozen dependent instructions of an IEEE
    // quotient are a tenth of a trip of a walk that has its SIMD to itself
    const double *tab = nullptr;
    __device__ __forceinline__ double over_span(double a, int span) const {
        const double s = (double)span, inv = tab[span];
        const double q = a * inv;
        return __builtin_fma(__builtin_fma(-q, s, a), inv, q);
    }
#endif
    __device__ __forceinline__ void flush() {
        constexpr int kFlushBatch = 32;
        for (int k = xlo; k < xhi; k += kFlushBatch) {
            Ext e[kFlushBatch];
#pragma unroll
            for (int u = 0; u < kFlushBatch; u++)
                if (k + u < xhi) e[u] = Op<OP>::fetch(p, base + (long)(k + u) * inc);
#pragma unroll
            for (int u = 0; u < kFlushBatch; u++)
                if (k + u < xhi) Op<OP>::finish(p, base + (long)(k + u) * inc, e[u], Yw[(k + u - wlo) * 64]);
        }
    }
};

// The walk of a job.  A wave with a SIMD to itself issues an instruction every ~8 cycles whatever it is, and a taken branch costs
// several of them: the trips of walker_run -- ~120 instructions and ~15 branches -- were 45 of the 57 us of a launch, whether the window
// was 64 or 128 samples, whether the quotients were IEEE or a table.  So the interior trips are chunkcore.hpp's walk_interior again:
// straight-line predicated code out of the LDS window, one branch for the books of a bend, SpanDiv quotients; the fibre's last sample
// keeps walker_run.  Same state machine and operation order as walker_run.
template <bool WEIGHTED, class S>
__device__ __forceinline__ void job_walk(Walker &w, S &src, int len, double lam) {
    for (;;) {
        const int lim = min(len - 1, src.wlo + src.got - 1);   // a trip looks one sample ahead
        if (w.i >= lim) {
            if (w.i >= len - 1) break;
            if (src.wlo + src.got >= src.whi) {   // the end of the window
                src.abort = true;
                return;
            }
            src.fill_more();
            continue;
        }
        double yi = src.win_y(w.i);
        while (!src.stop && w.i < lim) {
            const int i = w.i;
            const double ynx = src.win_y(i + 1);   // speculative: most trips advance by one
            const double r = WEIGHTED ? src.win_r(i) : lam;
            const double h1 = w.hlo + (w.lo - yi);
            const double h2 = w.hhi + (w.hi - yi);
            const bool cv = r < h1;
            const bool fv = !cv && (-r > h2);
            const bool bend = cv || fv;
            const int brk = cv ? w.klo : w.khi;
            const int at = brk + 1;                // wfrom < at <= i: an interior sample inside the window
            const double yat = src.win_y(at), yat1 = src.win_y(at + 1);

            // no bend: pull the pieces back inside the tube where they left it
            const SpanDiv over((double)(i - w.k0));
            const double d2 = ptv_min(r - h2, 0.0), d1 = ptv_max(-r - h1, 0.0);
            const double nhi = w.hi + over(d2), nlo = w.lo + over(d1);
            const double nhhi = ptv_min(h2, r), nhlo = ptv_max(h1, -r);
            const int nkhi = (h2 >= r) ? i : w.khi, nklo = (h1 <= -r) ? i : w.klo;

            // bend: closed-form first sample of the new piece (walker_restart_with, at < len - 1)
            double blo, bhi, bhhi, bhlo;
            if (WEIGHTED) {
                const double wp = src.win_r(brk), wc = src.win_r(at);
                const double a = cv ? yat + wp : yat - wp;
                blo = a - wc;
                bhi = a + wc;
                bhhi = wc;
                bhlo = -wc;
            } else {
                blo = cv ? yat : 2 * (-lam) + yat;
                bhi = cv ? 2 * lam + yat : yat;
                bhhi = lam;
                bhlo = -lam;
            }
            if (bend) {   // (every few trips: the finished piece's values into the window, the hand-over test at a chunk boundary)
                src.piece(w.k0 + 1, brk, cv ? w.lo : w.hi);
                src.bend(at, cv ? BEND_CEIL : BEND_FLOOR);
            }
            w.lo = bend ? blo : nlo;
            w.hi = bend ? bhi : nhi;
            w.hlo = bend ? bhlo : nhlo;
            w.hhi = bend ? bhhi : nhhi;
            w.k0 = bend ? brk : w.k0;
            w.klo = bend ? at : nklo;
            w.khi = bend ? at : nkhi;
            w.i = (bend ? at : i) + 1;
            yi = bend ? yat1 : ynx;
        }
        if (src.stop || src.abort) return;
    }
    walker_run<WEIGHTED>(w, src, len, lam);   // the fibre's last sample, with its own tests (and whatever a bend there rewinds to)
}

template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void sweep_repair_jobs_kernel(SweepArgs p, FibreGeom g, int C, int H, int chunks_per_wg,
                                                                const link_t *code_mine, const link_t *code_next, const int *failflags,
                                                                int *failcount, long cstride, long fstride, DirtyMark dirty,
                                                                unsigned *handled) {
    extern __shared__ __attribute__((aligned(16))) double repair_lds[];
    if (__hip_atomic_load(dirty.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != dirty.epoch) return;
    if (p.gate && *p.gate == 0) return;
    const int lane = threadIdx.x, slot = lane & (kJobsPerFibre - 1);
    const long j = (long)blockIdx.x * (64 / kJobsPerFibre) + (lane / kJobsPerFibre);
    const int len = g.len;
    const int NC = (len + C - 1) / C;
    const int nbound = (NC + chunks_per_wg - 1) / chunks_per_wg;
    const bool live = j < g.count && nbound <= 64;
    const long jj = live ? j : 0;
    // the fail flags and the links across workgroups of this fibre in one round trip (every lane of the fibre looks at all of them)
    const int f0 = failflags[2 * jj], f1 = failflags[2 * jj + 1];
    unsigned long long xbad = 0ull;
    {
        constexpr int UB = 32;
        for (int c0 = chunks_per_wg; c0 < NC; c0 += UB * chunks_per_wg) {
            link_t in[UB], out[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int c = min(c0 + u * chunks_per_wg, NC - 1);
                in[u] = code_mine[(long)c * cstride + jj * fstride];
                out[u] = code_next[(long)(c - 1) * cstride + jj * fstride];
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int c = c0 + u * chunks_per_wg;
                const bool certain = (in[u] & kLinkCertain) && in[u] != kLinkBad;
                if (c < NC && c * C - H > 0 && !certain && (in[u] == 0 || in[u] != out[u])) xbad |= 1ull << ((c / chunks_per_wg) & 63);
            }
        }
    }
    const int nb = __popcll(xbad);
    // (links flagged inside a workgroup, more failures than lanes: the sequential kernel's)
    const bool mine_to_do = live && f0 == 0 && f1 == 0 && nb >= 1 && nb <= kJobsPerFibre;
    const bool have_job = mine_to_do && slot < nb;
    // this lane's job: the slot-th failing boundary of its fibre
    int X = NC, r = NC;
    bool abort = false;
    if (have_job) {
        unsigned long long m = xbad;
        for (int k = 0; k < slot; k++) m &= m - 1;
        X = (int)__builtin_ctzll(m) * chunks_per_wg;
    }
    long base, wbase;
    {
        long blk, off;
        divmod_nonneg(jj, g.inc, blk, off);
        base = blk * g.inc * len + off;
        wbase = blk * g.inc * (len - 1) + off;
    }
    const RepairBook book{code_mine, cstride, fstride, jj, C, len};
    JobSource<OP, WEIGHTED> src(book, p, base, g.inc, wbase, repair_lds, lane);
#ifdef PTV_JOB_TABDIV
    {
        double *tab = repair_lds + (size_t)(1 + (WEIGHTED ? 1 : 0)) * kJobWindow * 64;
        for (int k = lane; k <= kJobWindow + 1; k += 64) tab[k] = k ? 1.0 / (double)k : 0.0;
        __syncthreads();
        src.tab = tab;
    }
#endif
    if (have_job) {
        // one round trip: the codes the walk starts from (the last bend before the link) and may hand over at
        constexpr int KB = 4;
        link_t back[KB];
#pragma unroll
        for (int u = 0; u < KB; u++) back[u] = (X - 1 - u >= 0) ? code_next[(long)(X - 1 - u) * cstride + j * fstride] : 0u;
        const link_t mine = code_mine[(long)X * cstride + j * fstride];
#pragma unroll
        for (int u = 0; u < kJobAhead; u++) src.ahead[u] = (X + 1 + u < NC) ? code_mine[(long)(X + 1 + u) * cstride + j * fstride] : 0u;
        src.ahead0 = X + 1;
        link_t cur = 0u;
#pragma unroll
        for (int u = KB - 1; u >= 0; u--) cur = (back[u] != 0) ? back[u] : cur;
        if (cur == 0u) {
            cur = kFromStart;
            for (int b = X - 1 - KB; b >= 0; b--) {
                const link_t nx = code_next[(long)b * cstride + j * fstride];
                if (nx != 0) {
                    cur = nx;
                    break;
                }
            }
        }
        if (mine != 0 && mine != kLinkBad && mine == cur) {
            r = X;   // (its predecessor had no bend of its own: the link holds after all -- what the sequential scan finds too)
        } else {
            const link_t from = (cur == kFromStart) ? 0u : cur;
            src.begin(X, from);
            src.fill(cur == kFromStart ? 0 : (int)(cur >> 1));
            Walker w;
            if (cur == kFromStart) walker_start<WEIGHTED>(w, src, 0, p.lam);
            else walker_restart<WEIGHTED>(w, src, (int)(cur >> 1), (int)(cur & 1u), len, p.lam);
#ifdef PTV_JOB_NOWALK   // (timing diagnostic: everything but the walk and the flush; every fibre goes on to the sequential kernel)
            src.abort = true;
#elif defined(PTV_JOB_PLAIN_WALK)
            walker_run<WEIGHTED>(w, src, len, p.lam);
#else
            job_walk<WEIGHTED>(w, src, len, p.lam);
#endif
            abort = src.abort;
            r = src.stop ? src.resume_chunk : NC;   // (not stopped and not aborted: it walked to the fibre end inside its window)
        }
    }
    // the jobs of a fibre in order (four adjacent lanes): which of them are valid, and is the fibre this kernel's at all
    const int lane0 = lane & ~(kJobsPerFibre - 1);
    bool fibre_ok = mine_to_do, valid = false;
    int lastr = -1;
#pragma unroll
    for (int k = 0; k < kJobsPerFibre; k++) {
        const int Xk = __shfl(X, lane0 + k), rk = __shfl(r, lane0 + k);
        const bool ak = __shfl((int)abort, lane0 + k) != 0;
        const bool jobk = k < nb;
        if (jobk && ak) fibre_ok = false;
        const bool vk = jobk && (lastr < 0 || lastr <= Xk - 1);
        if (vk) lastr = rk;
        if (k == slot) valid = vk;
    }
    if (fibre_ok && have_job && valid) {
        src.flush();
        atomicAdd(failcount + 1, r - X);   // chunks rewritten
    }
    if (fibre_ok && slot == 0) {
        handled[j] = dirty.epoch;
        atomicAdd(failcount, 1);           // fibres that needed a repair
    }
}

template <int OP, bool WEIGHTED>
__global__ __launch_bounds__(64) void sweep_repair_kernel(SweepArgs p, FibreGeom g, int C, int H, int chunks_per_wg,
                                                           const link_t *code_mine, const link_t *code_next,
                                                           int *failflags, int *failcount, long cstride, long fstride,
                                                           DirtyMark dirty, const unsigned *handled = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double repair_lds[];   // (2 + WEIGHTED) planes of kRepairWindow x 64 (LDS geometries only)
    // the common case: the chunk kernel proved every link itself, across its workgroups too, and said so by NOT marking the sweep
    if (dirty.word && __hip_atomic_load(dirty.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != dirty.epoch) return;
    const long j = (long)blockIdx.x * 64 + threadIdx.x;
    if (j >= g.count) return;
    if (p.gate && *p.gate == 0) return;
    if (handled && handled[j] == dirty.epoch) return;   // (sweep_repair_jobs_kernel repaired this fibre)
    const int len = g.len;
    const int NC = (len + C - 1) / C;
    // first / last chunk with an unproven link: what the chunk kernel flagged, widened below by the links between workgroups
    int first = NC, lastbad = -1;
    {
        const int f0 = failflags[2 * j], f1 = failflags[2 * j + 1];
        if (f0 > 0) first = NC - f0;
        if (f1 > 0) lastbad = f1 - 1;
    }
    // Which chunks may fail the scan below at all: those inside the range the chunk kernel flagged [ff, fl] (links inside a workgroup,
    // walks that ran off their window) and the first chunk of a workgroup whose link IN failed (xbad: one bit per boundary).  Every
    // other chunk was proven by the chunk kernel to continue its predecessor's walk -- mine == the predecessor's non-zero next code, or
    // a start at a bend known a priori -- so the scan accepts it whenever its predecessor is true: the scan may jump over them.
    const int ff = first, fl = lastbad;
    unsigned long long xbad = 0ull;
    const bool jump = (NC + chunks_per_wg - 1) / chunks_per_wg <= 64;
    // links between workgroups (inside a workgroup they were checked through LDS): 16 boundaries = 32 independent
    // loads in flight per lane -- the cost of the common case is the latency of these reads
    constexpr int UB = 16;
    for (int c0 = chunks_per_wg; c0 < NC; c0 += UB * chunks_per_wg) {
        link_t in[UB], out[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int c = min(c0 + u * chunks_per_wg, NC - 1);
            in[u] = code_mine[(long)c * cstride + j * fstride];
            out[u] = code_next[(long)(c - 1) * cstride + j * fstride];
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int c = c0 + u * chunks_per_wg;
            const bool certain = (in[u] & kLinkCertain) && in[u] != kLinkBad;
            if (c < NC && c * C - H > 0 && !certain && (in[u] == 0 || in[u] != out[u])) {
                first = min(first, c);
                lastbad = max(lastbad, c);
                if (jump) xbad |= 1ull << (c / chunks_per_wg);
            }
        }
    }
    if (lastbad < 0) return;
    failflags[2 * j] = failflags[2 * j + 1] = 0;
    atomicAdd(failcount, 1);       // fibres that needed a repair
    int walks = 0;

    long blk, off;
    divmod_nonneg(j, g.inc, blk, off);
    const long base = blk * g.inc * len + off, wbase = blk * g.inc * (len - 1) + off;
    const RepairBook book{code_mine, cstride, fstride, j, C, len};
    const bool windowed = H <= kWarmLong;
    RepairSource<OP, WEIGHTED> gsrc(book, p, base, g.inc, wbase);
    WindowRepairSource<OP, WEIGHTED> wsrc(book, p, base, g.inc, wbase, repair_lds, (int)threadIdx.x);
    // every chunk before `first` is proven: the true walk's last bend there is the last non-zero `next` code before it
    auto last_bend_before = [&](int chunk) {
        for (int b = chunk - 1; b >= 0; b--) {
            const link_t nx = code_next[(long)b * cstride + j * fstride];
            if (nx != 0) return nx;
        }
        return kFromStart;
    };
    // the first chunk at or after `chunk` that the scan could reject (NC: none)
    auto next_suspect = [&](int chunk) {
        if (!jump || (chunk >= ff && chunk <= fl)) return chunk;
        int best = chunk < ff ? ff : NC;
        const int b0 = (chunk + chunks_per_wg - 1) / chunks_per_wg;
        if (b0 < 64) {
            const unsigned long long m = xbad >> b0;
            if (m) best = min(best, (b0 + (int)__builtin_ctzll(m)) * chunks_per_wg);
        }
        return best;
    };
    link_t cur = last_bend_before(first);
    int c = first;
    // Two-phase loop so that the lanes of a wave repair TOGETHER: first every lane scans ahead to its next unproven
    // chunk, then all lanes that found one walk at the same time (a walk nested inside the scan would serialise the
    // lanes, each reaching its repair at a different trip).
    while (true) {
        // the scan reads the codes of UB chunks at a time (2 UB independent loads), then goes through them in registers:
        // one memory round trip per UB chunks instead of two per chunk
        bool rejected = false;
        while (c < NC && c <= lastbad && !rejected) {   // (everything after the last flagged chunk is proven)
            // jump over the chunks that cannot be rejected (after a repair walk: from the chunk that took its walk over, whose codes
            // and everything after it are true) -- one memory round trip per failure instead of one per UB chunks in between
            const int suspect = next_suspect(c);
            if (suspect > c) {
                if (suspect >= NC || suspect > lastbad) {
                    c = suspect;
                    break;
                }
                // The true walk's last bend before `suspect`: the latest one recorded by the chunks jumped over -- true records, of chunks
                // the chunk kernels proved.  The scan stops at the chunk the jump started from: the chunks BEFORE c may have been rewritten
                // by a repair walk since, and their records are the speculative walks' still.  (A proven chunk's `next` is never zero --
                // DESIGN "exactness" -- so the chunk just before `suspect` ends the scan in practice; the bound makes that argument unnecessary.)
                for (int b = suspect - 1; b >= c; b--) {
                    const link_t nx = code_next[(long)b * cstride + j * fstride];
                    if (nx != 0) {
                        cur = nx;
                        break;
                    }
                }
                c = suspect;
            }
            link_t mm[UB], nn[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int cc = min(c + u, NC - 1);
                mm[u] = code_mine[(long)cc * cstride + j * fstride];
                nn[u] = code_next[(long)cc * cstride + j * fstride];
            }
            const int c0 = c;
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (!rejected && c == c0 + u && c < NC) {
                    const link_t mraw = mm[u];
                    const bool certain = (mraw & kLinkCertain) && mraw != kLinkBad;
                    const link_t m = certain ? (mraw & ~kLinkCertain) : mraw;
                    // a chunk whose walk began at sample 0 (or at a bend known a priori) is the true walk unless it
                    // ran off its window (kLinkBad)
                    const bool accept = (c * C - H <= 0 || certain) ? (m != kLinkBad) : (m != 0 && m == cur);
                    if (accept) {
                        if (nn[u] != 0) cur = nn[u];
                        c++;
                    } else {
                        rejected = true;
                    }
                }
            }
        }
        if (c >= NC || !rejected) break;
        const link_t from = (cur == kFromStart) ? 0u : cur;
        bool stopped;
        int resume_chunk;
        link_t resume_code;
        Walker w;
        if (windowed) {
            wsrc.begin(c, from);
            if (cur == kFromStart) walker_start<WEIGHTED>(w, wsrc, 0, p.lam);
            else walker_restart<WEIGHTED>(w, wsrc, (int)(cur >> 1), (int)(cur & 1u), len, p.lam);
            walker_run<WEIGHTED>(w, wsrc, len, p.lam);
            wsrc.flush();
            stopped = wsrc.stop; resume_chunk = wsrc.resume_chunk; resume_code = wsrc.resume_code;
        } else {   // global-memory geometries: long pieces
            gsrc.begin(c, from);
            if (cur == kFromStart) walker_start<WEIGHTED>(w, gsrc, 0, p.lam);
            else walker_restart<WEIGHTED>(w, gsrc, (int)(cur >> 1), (int)(cur & 1u), len, p.lam);
            walker_run_blocked<WEIGHTED, kGlobalBlock>(w, gsrc, len, p.lam);
            stopped = gsrc.stop; resume_chunk = gsrc.resume_chunk; resume_code = gsrc.resume_code;
        }
        walks += stopped ? (resume_chunk - c) : (NC - c);   // chunks this walk had to rewrite
        if (!stopped) break;                    // walked to the fibre end: everything from chunk c on is rewritten
        c = resume_chunk;                       // that chunk continues this walk: accepted on the next trip
        cur = resume_code;
    }
    atomicAdd(failcount + 1, walks);   // chunks rewritten
}

}  // namespace swp
}  // namespace ptv
