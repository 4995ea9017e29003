// sweep_launch.hpp -- host side: per-thread state of the chunked path, the geometry ladder, the launchers.
// (One of the pieces of sweep_kernels.hpp, which includes them in order; not meant to be included on its own.)
#pragma once

namespace ptv {
namespace swp {

// ---- host side ---------------------------------------------------------------------------------------------------------
template <int OP, bool WEIGHTED>
void launch_seq(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool long_pieces, int *fibre_gate = nullptr) {
    const unsigned blocks = (unsigned)((g.count + 63) / 64);
    if (blocks == 0) return;
    if (long_pieces) hipLaunchKernelGGL((sweep_seq_kernel<OP, WEIGHTED, true>), dim3(blocks), dim3(64), 0, stream, args, g, fibre_gate);
    else             hipLaunchKernelGGL((sweep_seq_kernel<OP, WEIGHTED, false>), dim3(blocks), dim3(64), 0, stream, args, g, fibre_gate);
    count_event(CNT_SWEEP_LAUNCHES);
    PTV_HIP(hipGetLastError());
}

// persistent per-thread state of the chunked path: link codes, fail flags, repair counters and the geometry policy
struct ChunkScratch {
    std::unique_ptr<Scratch> links, flags;
    size_t link_bytes = 0, flag_count = 0;
    link_t *code_mine = nullptr, *code_next = nullptr;   // [chunk][fibre]
    int *failflags = nullptr;
    double sweep_seed_f = -1.0;             // certain fraction of the sweep being launched (-1: not sampled)
    std::unique_ptr<Scratch> handled_buf;   // option "repair_jobs": per fibre, the epoch of the launch whose failures sweep_repair_jobs_kernel repaired
    size_t handled_count = 0;
    unsigned *handled_for(size_t count, hipStream_t s) {
        if (count > handled_count) {
            handled_buf.reset(new Scratch(sizeof(unsigned) * count));
            handled_count = count;
            PTV_HIP(hipMemsetAsync(handled_buf->as<unsigned>(), 0, sizeof(unsigned) * count, s));
        }
        return handled_buf->as<unsigned>();
    }
    std::unique_ptr<Scratch> certify_buf;   // option "certify": [0] fibres that failed the last check, [1 ...] their flags (the sequential kernel's fibre gate)
    std::unique_ptr<Scratch> certify_notes; // ... and what the first few of them looked like (CertifyNote[kCertifyNotes])
    size_t certify_count = 0;
    int *certify_for(size_t count, hipStream_t s) {
        if (count > certify_count) {
            certify_buf.reset(new Scratch(sizeof(int) * (count + 1)));
            certify_count = count;
            PTV_HIP(hipMemsetAsync(certify_buf->as<int>(), 0, sizeof(int) * (count + 1), s));
        }
        return certify_buf->as<int>();
    }
    std::unique_ptr<Scratch> trace;   // option "trace": phase timestamps of the last chunk-kernel launch
    size_t trace_wgs = 0;
    unsigned long long *trace_buffer(size_t wgs) {
        if (!trace || trace->bytes() < wgs * 64) trace.reset(new Scratch(wgs * 64));
        trace_wgs = wgs;
        return trace->as<unsigned long long>();
    }
    int *failcount = nullptr;   // [family][2]: fibres that needed repair, chunks rewritten by repair walks (cumulative per solve)
    // one "dirty" word for all launches (each launch has its own epoch) and the cross-workgroup link words
    unsigned epoch = 0;
    std::unique_ptr<Scratch> dirty_word, xlink;
    size_t xlink_words = 0;
    DirtyMark next_dirty(hipStream_t s) {
        if (!options().xlink) return DirtyMark{nullptr, 0u, nullptr, nullptr};
        ensure_dirty(s);
        if (++epoch == 0u) ++epoch;   // (0 is what freshly allocated words hold)
        return DirtyMark{dirty_word->as<unsigned>(), epoch, options().why ? dirty_word->as<unsigned>() + 1 : nullptr,
                         optimistic ? dirty_word->as<unsigned>() + 9 : nullptr};
    }
    void ensure_dirty(hipStream_t s) {
        if (!dirty_word) {   // [0] the word, [1..8] option "why" counters, [9] the sticky word of an optimistic solve
            dirty_word.reset(new Scratch(sizeof(unsigned) * 10));
            PTV_HIP(hipMemsetAsync(dirty_word->as<unsigned>(), 0, sizeof(unsigned) * 10, s));
        }
    }
    // An optimistic solve: the chunked sweeps launch no repair kernels (an empty one still costs its dependent launch, 2.5-4.5 us behind
    // every sweep: 5 % of the headline solve, half of a 512^2 one); whatever a sweep leaves is recorded in the sticky word, read once at
    // the end of the solve.  Exactness rests on the run with the repairs that follows a solve whose word is set.
    bool optimistic = false;
    int optimistic_backoff = 0;   // eligible solves still to run WITH their repairs after one that had to be run again (sweep.hip)
    void begin_optimistic(hipStream_t s) {
        ensure_dirty(s);
        PTV_HIP(hipMemsetAsync(dirty_word->as<unsigned>() + 9, 0, sizeof(unsigned), s));
        optimistic = true;
    }
    // ends the optimistic stretch; true: every sweep was clean (synchronises the stream)
    bool end_optimistic(hipStream_t s) {
        optimistic = false;
        unsigned mark = 1;
        PTV_HIP(hipMemcpyAsync(&mark, dirty_word->as<unsigned>() + 9, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        PTV_HIP(hipStreamSynchronize(s));
        // (whoever marked the word also flagged chunks -- per fibre, in units of ITS sweep's geometry -- and no repair kernel came to
        //  take the flags back: the run that follows must not find them)
        if (mark != 0 && flags && flag_count > (size_t)kCounters)
            PTV_HIP(hipMemsetAsync(flags->as<int>() + kCounters, 0, sizeof(int) * (flag_count - (size_t)kCounters), s));
        return mark == 0;
    }
    unsigned long long *xlink_for(size_t words, hipStream_t s) {
        if (!options().xlink) return nullptr;
        if (words > xlink_words) {
            xlink.reset(new Scratch(sizeof(unsigned long long) * words));
            xlink_words = words;
            PTV_HIP(hipMemsetAsync(xlink->as<unsigned long long>(), 0, sizeof(unsigned long long) * words, s));
        }
        return xlink->as<unsigned long long>();
    }

    // Geometry policy (policy.hpp), one per sweep family -- fibres along dim 0 / along the other dims see different
    // data: in a DR solve at large lambda the column pieces are several times longer than the row pieces -- plus the
    // plumbing of its measurements: hipEvents around the measured launch, the repair counters read back behind it.
    struct Policy : GeometryPolicy {
        bool meas = false;       // a measurement is in flight
        int meas_mode = 0, meas_slot = -1;
        long meas_sweep = 0;
        hipEvent_t t0 = nullptr, t1 = nullptr;
        long chunks_done = 0;    // chunks processed (host-side count, cumulative per solve)
        long chunks_seen = 0;    // ... at the last evaluation
        int rewritten_seen = 0;
        double shown_f = -2.0;   // (option verbose: the certain fraction last printed)
    } pol[FAM_COUNT];

    // Edge statistics of this solve's input, one record per swept dimension (policy_probe): the seed of the policy.
    struct Probe {
        long inc = 0, count = 0;
        int len = 0;
        bool weighted = false;
        int iterate = 0;                   // sampled mid-solve from the operand of a sweep (policy_reprobe), not from the solve's input: 1 = Dykstra (x + p), 2 = ADMM (Yang)
        unsigned hist[kProbeWords] = {};   // edges | stretches (pointwise.hpp)
    };
    static constexpr int kMaxProbes = 8;
    Probe probes[kMaxProbes];
    int nprobes = 0;
    std::unique_ptr<Scratch> probe_dev;   // kMaxProbes histograms
    const Probe *find_probe(const FibreGeom &g, bool weighted) const {
        for (int k = 0; k < nprobes; k++)
            if (probes[k].inc == g.inc && probes[k].len == g.len && probes[k].count == g.count && probes[k].weighted == weighted)
                return &probes[k];
        return nullptr;
    }
    // fraction of the sampled edges above `mult` penalties (-1: this sweep's input was not sampled)
    double edge_fraction_above(const FibreGeom &g, double lam, bool weighted, double mult) const {
        const Probe *p = find_probe(g, weighted);
        if (!p || p->hist[kProbeBins] == 0) return -1.0;
        if (!weighted && !(lam > 0.0)) return 1.0;
        // (edges in the threshold's own bin do not count: an edge of exactly 4 lambda -- a checkerboard of +-2 lambda -- is not a
        // bend known a priori, and the kernels' test is strict)
        const int b = probe_bin(weighted ? mult : mult * lam);
        unsigned long above = 0;
        for (int k = b + 1; k < kProbeBins; k++) above += p->hist[k];
        return (double)above / (double)p->hist[kProbeBins];
    }
    // fraction of the sampled edges at which the string is known to bend at this penalty (-1: this sweep's input was not sampled)
    double certain_fraction(const FibreGeom &g, double lam, bool weighted) const { return edge_fraction_above(g, lam, weighted, 4.0); }
    // fraction of the sampled 16-edge stretches whose total variation is below 2 lambda: stretches the string crosses (all
    // but) flat -- nothing there for a speculative walk to meet the true one at (-1: not sampled)
    double flat_fraction(const FibreGeom &g, double lam, bool weighted) const {
        const Probe *p = find_probe(g, weighted);
        const unsigned *h = p ? p->hist + kProbeBins + 1 : nullptr;
        if (!p || h[kProbeBins] == 0) return -1.0;
        if (!weighted && !(lam > 0.0)) return 0.0;
        const int b = probe_bin(weighted ? 2.0 : 2.0 * lam);
        unsigned long below = 0;
        for (int k = 0; k < b; k++) below += h[k];
        return (double)below / (double)h[kProbeBins];
    }
    // rung the statistics ask for at this penalty (-1: not sampled)
    int seed(const FibreGeom &g, double lam, bool weighted) const {
        const double f = certain_fraction(g, lam, weighted);
        if (f < 0.0) return -1;
        // Spatially uneven data (half an image flat, sparse spikes on a constant background): whatever the average says, the
        // quiet stretches have pieces far longer than any zone and every chunk in them would go to the repair kernel.
        if (flat_fraction(g, lam, weighted) > kSeedFlat) return 3;
        const Probe *p = find_probe(g, weighted);
        return rung_from_certain_fraction(f, p && p->iterate == 1, (long)g.len * g.count <= kSmallSweep, weighted);
    }

    static constexpr int kSlots = 8, kCounters = 2 * FAM_COUNT;
    int *h_counts = nullptr;    // pinned: [slot][kCounters]
    hipEvent_t ev[kSlots] = {};
    bool pending[kSlots] = {};
    long pending_chunks[kSlots][FAM_COUNT] = {};
    int next_slot = 0;
    int latest_rewritten[FAM_COUNT] = {0, 0, 0};
    long latest_chunks[FAM_COUNT] = {0, 0, 0};

    void ensure_host() {
        if (h_counts) return;
        PTV_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_counts), sizeof(int) * kSlots * kCounters, hipHostMallocDefault));
        for (auto &e : ev) PTV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &pl : pol) {
            PTV_HIP(hipEventCreate(&pl.t0));
            PTV_HIP(hipEventCreate(&pl.t1));
        }
    }

    void ensure(long count, int NC, hipStream_t s) {
        const size_t need = sizeof(link_t) * (size_t)count * (size_t)NC * 2;
        if (need > link_bytes) {
            links.reset(new Scratch(need));
            link_bytes = need;
        }
        code_mine = links->as<link_t>();
        code_next = code_mine + (size_t)count * (size_t)NC;
        if (2 * (size_t)count + kCounters > flag_count) {
            poll(true);   // read-backs of the old counters must land before the buffer goes away
            flags.reset(new Scratch(sizeof(int) * (2 * (size_t)count + kCounters)));
            flag_count = 2 * (size_t)count + kCounters;
            PTV_HIP(hipMemsetAsync(flags->as<int>(), 0, sizeof(int) * flag_count, s));
            for (int f = 0; f < FAM_COUNT; f++) {
                pol[f].rewritten_seen = 0;
                latest_rewritten[f] = 0;
            }
        }
        failcount = flags->as<int>();
        failflags = failcount + kCounters;
        ensure_host();
    }

    // take in read-backs, oldest first: all of them (wait), or up to and including slot `until` (blocking), or those
    // that have completed
    void poll(bool wait, int until = -1) {
        for (int k = 0; k < kSlots; k++) {
            const int slot = (next_slot + k) % kSlots;
            if (!pending[slot]) continue;
            if (wait || until >= 0) PTV_HIP(hipEventSynchronize(ev[slot]));
            else if (hipEventQuery(ev[slot]) != hipSuccess) break;
            pending[slot] = false;
            for (int f = 0; f < FAM_COUNT; f++) {
                latest_rewritten[f] = h_counts[slot * kCounters + 2 * f + 1];
                latest_chunks[f] = pending_chunks[slot][f];
            }
            if (slot == until) break;
        }
    }

    int enqueue_readback(hipStream_t s) {
        const int slot = next_slot;
        if (pending[slot] || !failcount) return -1;   // ring full: skip this sample
        PTV_HIP(hipMemcpyAsync(h_counts + slot * kCounters, failcount, sizeof(int) * kCounters, hipMemcpyDeviceToHost, s));
        PTV_HIP(hipEventRecord(ev[slot], s));
        pending[slot] = true;
        for (int f = 0; f < FAM_COUNT; f++) pending_chunks[slot][f] = pol[f].chunks_done;
        next_slot = (slot + 1) % kSlots;
        return slot;
    }

    // the measurement in flight: sweep time in ms, and the fraction of the family's chunks that repair walks rewrote
    // since its last evaluation (-1: no counters, e.g. after a sequential sweep)
    void evaluate(int fam, double &t, double &f) {
        Policy &pl = pol[fam];
        PTV_HIP(hipEventSynchronize(pl.t1));
        float ms = 0.f;
        PTV_HIP(hipEventElapsedTime(&ms, pl.t0, pl.t1));
        t = ms;
        f = -1.0;
        if (pl.meas_slot >= 0) {
            poll(false, pl.meas_slot);
            const long d = latest_chunks[fam] - pl.chunks_seen;
            if (d > 0) {
                f = (double)(latest_rewritten[fam] - pl.rewritten_seen) / (double)d;
                pl.chunks_seen = latest_chunks[fam];
                pl.rewritten_seen = latest_rewritten[fam];
            }
        }
        pl.meas = false;
        pl.meas_slot = -1;
    }

    // hand the measurement in flight to the policy
    void settle(int fam, bool adaptive) {
        Policy &pl = pol[fam];
        const int r = pl.meas_mode;
        double t, f;
        evaluate(fam, t, f);
        if (!adaptive) return;
        if (options().verbose)
            fprintf(stderr, "[proxtv_amd] policy: family %d sweep %ld: mode %d %s took %.3f ms, rewrote %.5f of its chunks (incumbent %d: %.3f ms)\n",
                    fam, pl.sweeps, r, !pl.explore ? "(sample)" : pl.trial >= 0 ? "(trial)" : "(incumbent)", t, f, pl.mode, pl.t_mode);
        pl.measured(r, t, f);
    }
};
// per host thread and per device, like the stream and the pool (the one definition: sweep.hip)
ChunkScratch &chunk_state();

// option "repair_jobs": the failures across workgroups first, one lane each (sweep_repair_jobs_kernel); returns the per-fibre marks the
// sequential kernel behind it skips by (nullptr: not run)
template <int OP, bool WEIGHTED>
const unsigned *launch_repair_jobs(const SweepArgs &args, const FibreGeom &g, int C, int H, int chunks_per_wg, int *failcount, long cstride,
                                   long fstride, const DirtyMark &dirty, hipStream_t stream) {
    const int NC = (g.len + C - 1) / C;
    if (!options().repair_jobs || !dirty.word || H > kWarmLong || (NC + chunks_per_wg - 1) / chunks_per_wg > 64) return nullptr;
    // (1: only where links fail in numbers -- a launch that finds the sweep clean still costs its 2 us ; 2: always.  An UNSAMPLED input
    //  -- sweep_seed_f = -1: a pinned rung, a problem too small to sample -- counts as "in numbers": nothing says the sweep is clean,
    //  and the pinned-rung legs of the test suite run the jobs kernel through this door)
    if (options().repair_jobs == 1 && !(chunk_state().sweep_seed_f < kSeedJobs)) return nullptr;
    constexpr size_t lds = sizeof(double) * ((1 + (WEIGHTED ? 1 : 0)) * kJobWindow * 64 + kJobWindow + 2);   // (+ the table of -DPTV_JOB_TABDIV)
    auto kern = sweep_repair_jobs_kernel<OP, WEIGHTED>;
    {
        static thread_local bool attr_done[kMaxDevices] = {};
        bool &attr_set = attr_done[current_device()];
        if (!attr_set) {
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
    }
    unsigned *handled = chunk_state().handled_for((size_t)g.count, stream);
    constexpr int per_block = 64 / kJobsPerFibre;
    hipLaunchKernelGGL(kern, dim3((unsigned)((g.count + per_block - 1) / per_block)), dim3(64), lds, stream, args, g, C, H, chunks_per_wg,
                       chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags, failcount, cstride, fstride, dirty, handled);
    count_event(CNT_REPAIR_JOBS_LAUNCHES);
    return handled;
}

// Chunk geometry: C = 16 samples per chunk, 8 waves (chunks) per block of 128 samples.  LDS per workgroup = one window
// of H + 128 + 8 rows x 512 B: ~77 KiB for H = 16 -> two workgroups = 16 waves per CU; ~101 KiB for H = 64 -> one.
// Weighted sweeps carry a second (penalty) window and exist for H = 16 only.
// (Tried for the robust instantiation and not kept: chunks of 14 samples with the 16 rows that saves spent on look-ahead --
// H 16 / 8 x 14 / T 24, the same 152 rows -- so that the walks of a block's last chunk close their last piece inside the
// window instead of reading on from global memory: on DR iterates at lambda = 0.5 on unit noise 2 % of them need more than
// 8 rows, 0.02 % more than 16.  The row sweep got 13 % slower, 180 -> 203 us: 37 blocks per fibre instead of 32 cost more
// than the reads past the window did.  What makes these sweeps slow is the walk itself -- at lambda = 0.5 the linearized
// taut string re-walks every piece about once: 2.6 x the trips of the headline.)
template <int OP, bool WEIGHTED, bool TRANSPOSED, int H, bool ROBUST = false, int C = 16, int NW = 8,
          int T = tail_rows(H), bool SHORT = false, int FW = 64>
void launch_chunk_h(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam, int rounds_wanted) {
    constexpr int PITCH = TRANSPOSED ? 65 : FW;
    constexpr int NCH = NW * (64 / FW);   // chunks per block
    constexpr int ROWS = SHORT ? NCH * C : H + NCH * C + T;
    ChunkPlan plan{};
    plan.Q = (g.len + NCH * C - 1) / (NCH * C);
    // blocks per workgroup: enough workgroups to fill the chip a few times over
    const long groups = (g.count + FW - 1) / FW;
    int qpw = 8;
    {
        // (weighted strided sweeps run ONE workgroup per CU -- two LDS planes: as many blocks per workgroup as still gives
        // every CU one; 14.5 -> 14.05 ms on the 4096^2 weighted solve.  Keeping the next block's window share in registers
        // while the current one is processed -- the waves own 256 VGPRs there -- was tried and hid the staging phase, but the
        // sweep did not get faster: at 8 waves per CU it is the walk's dependent-instruction latency that bounds it)
        const long want = (WEIGHTED && !TRANSPOSED && !SHORT) ? (FW < 64 ? 512 : 256) : (FW < 64 ? 4096 : 2048);
        while (qpw > 1 && groups * ((plan.Q + qpw - 1) / qpw) < want) qpw >>= 1;
    }
    plan.qpw = qpw < plan.Q ? qpw : plan.Q;
    plan.ablate = options().ablate;
    plan.rounds = rounds_wanted;
    plan.trace = options().trace ? chunk_state().trace_buffer((size_t)groups * (size_t)((plan.Q + plan.qpw - 1) / plan.qpw)) : nullptr;
    const int WQ = (plan.Q + plan.qpw - 1) / plan.qpw;
    plan.dirty = chunk_state().next_dirty(stream);
    plan.xlink = chunk_state().xlink_for((size_t)WQ * (size_t)g.count, stream);
#ifndef PTV_NO_WALK_TABLE   // (the switch stays for A/B builds: the walk then divides with v_rcp_f64 + Newton + residual)
    constexpr size_t tab_bytes = ((WEIGHTED || !TRANSPOSED) && !SHORT && H <= kWarm && NW <= 8) ? sizeof(double) * (ROBUST ? kRecipTableRobust : kRecipTable) : 0;
#else
    constexpr size_t tab_bytes = 0;
#endif
    constexpr size_t lds = sizeof(double) * PITCH * (size_t)ROWS * (WEIGHTED ? 2 : 1) + sizeof(link_t) * ((NCH + 2) * FW + (TRANSPOSED ? 0 : FW) + 4 + 2 * NW) + tab_bytes;
    static_assert(WEIGHTED || H > kWarm || NW > 8 || 2 * lds <= 160 * 1024, "the short-zone geometry is meant to run two workgroups per CU");
    static_assert(FW == 64 || (WEIGHTED ? 2 : 4) * lds <= 160 * 1024, "the 32-fibre tile is meant to run four workgroups per CU (weighted: two)");
    if (SHORT && g.len > NCH * C) {
        set_error("launch_chunk_h: a fibre of %d samples does not fit the single-block geometry (%d)", g.len, NCH * C);
        throw HipFailure{hipErrorInvalidValue};
    }
    static_assert(lds <= 160 * 1024, "chunk geometry does not fit the LDS of a CU");
    const int NC = (g.len + C - 1) / C;
    chunk_state().ensure(g.count, NC, stream);
    // the second-chance rounds are a separate instantiation: their live state costs the plain kernel registers it
    // does not have (it sits at the 128-VGPR budget of two workgroups per CU)
    static_assert(!ROBUST || H <= kWarm, "second chances exist for the short-zone geometry");
    if (!ROBUST) plan.rounds = 0;
    auto kern = sweep_chunk_kernel<OP, WEIGHTED, TRANSPOSED, C, NW, H, ROBUST, T, SHORT, FW>;
    static thread_local bool attr_done[kMaxDevices] = {};   // function attributes are per device
    bool &attr_set = attr_done[current_device()];
    if (!attr_set) {
        PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr_set = true;
        if (options().verbose) {
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 64 * NW, lds);
            fprintf(stderr, "[proxtv_amd] tile kernel op %d%s%s: %d fibres x %d chunks of %d in %d waves, %zu B of LDS -> %d workgroups per CU\n", OP,
                    WEIGHTED ? " weighted" : "", ROBUST ? " robust" : "", FW, NCH, C, NW, lds, per_cu);
        }
    }
    const dim3 grid((unsigned)groups, (unsigned)WQ);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, args, g, plan, chunk_state().code_mine,
                       chunk_state().code_next, chunk_state().failflags);
    count_event(CNT_SWEEP_LAUNCHES);
    if (!plan.ablate && !plan.dirty.sticky) {   // (an optimistic solve: the sweep has marked the sticky word if it left anything)
        constexpr size_t rlds = sizeof(double) * (2 + (WEIGHTED ? 1 : 0)) * kRepairWindow * 64;
        auto rkern = sweep_repair_kernel<OP, WEIGHTED>;
        static thread_local bool rattr_done[kMaxDevices] = {};
        bool &rattr_set = rattr_done[current_device()];
        if (!rattr_set) {
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rkern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)rlds));
            rattr_set = true;
        }
        const unsigned *handled = launch_repair_jobs<OP, WEIGHTED>(args, g, C, H, plan.qpw * NCH, chunk_state().failcount + 2 * fam, (long)g.count, 1L, plan.dirty, stream);
        hipLaunchKernelGGL(rkern, dim3((unsigned)((g.count + 63) / 64)), dim3(64), rlds, stream, args, g, C, H, plan.qpw * NCH,
                           chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags, chunk_state().failcount + 2 * fam, (long)g.count, 1L,
                           plan.dirty, handled);
        count_event(CNT_REPAIR_LAUNCHES);
    }
    PTV_HIP(hipGetLastError());
    chunk_state().pol[fam].chunks_done += (long)NC * g.count;
}

// Chunks along the fibre (kernel 2a): dimension-0 sweeps, unweighted.  Codes are laid out [fibre][chunk] (a wave writes
// the codes of 64 consecutive chunks of one fibre).
template <int OP, bool WEIGHTED, int H, int G, bool ROBUST, bool ONESEG = false, bool RUNS = false>
void launch_along_g(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam, int rounds_wanted) {
    static_assert(!RUNS || (!WEIGHTED && !ROBUST && !ONESEG && G == 64 && H == kWarm), "known runs: the plain unweighted 64-lane instantiation");
    constexpr int C = along_chunk(ROBUST, WEIGHTED), SEG = G * C, ROWS = ONESEG ? SEG : along_zone_rows(H, ROBUST) + SEG + along_tail_rows(H, ROBUST), NG = 64 / G;
    const int nseg = (g.len + SEG - 1) / SEG;
    if (ONESEG && nseg != 1) {
        set_error("launch_along_g: a fibre of %d samples is more than one segment (%d)", g.len, SEG);
        throw HipFailure{hipErrorInvalidValue};
    }
    const int NC = (g.len + C - 1) / C;
    const long units = g.count * nseg;
    const long waves = (units + NG - 1) / NG;
    ChunkPlan plan{};
    plan.ablate = options().ablate;
    plan.rounds = ROBUST ? rounds_wanted : 0;
    chunk_state().ensure(g.count, NC, stream);
    plan.trace = options().trace ? chunk_state().trace_buffer((size_t)waves) : nullptr;
    plan.dirty = chunk_state().next_dirty(stream);
    plan.xlink = chunk_state().xlink_for((size_t)g.count * (size_t)nseg, stream);
    plan.legacy = options().debug_legacy_rebuild;
    constexpr size_t lds = sizeof(double) * (size_t)(ROWS + 2) * NG * kAlongWaves * (WEIGHTED ? 2 : 1) + (ROBUST ? 64 + sizeof(double) * kRecipTableRobust : sizeof(double) * kRecipTable) +
                           (RUNS ? sizeof(unsigned) * kRunsWords * kAlongWaves : 0);
    static_assert(lds <= 160 * 1024, "along-fibre geometry does not fit the LDS of a CU");
    static_assert(!RUNS || (16 / kAlongWaves) * lds <= 160 * 1024, "known runs: still sixteen waves per CU");
    auto kern = sweep_along_kernel<OP, WEIGHTED, H, G, ROBUST, ONESEG, RUNS>;
    if (lds > 64 * 1024) {   // above the default dynamic-LDS limit
        static thread_local bool attr_done[kMaxDevices] = {};
        bool &attr_set = attr_done[current_device()];
        if (!attr_set) {
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
    }
    // (One workgroup per kAlongWaves segments, dispatched as slots free up.  Tried in round 5 and dropped, profiles/NOTES_r05.md: as many
    // workgroups as the device holds, each taking its segments in static turns -- 76 -> 84 us, nothing rebalances the slow workgroups --
    // or drawing them from atomic counters -- the wave slots stay 98 % full instead of 70 % and the sweep takes as long: the vector
    // pipes, not the dispatcher, are what the waves wait for.)
    hipLaunchKernelGGL(kern, dim3((unsigned)((waves + kAlongWaves - 1) / kAlongWaves)), dim3(64 * kAlongWaves), lds, stream, args, g,
                       plan, chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags);
    count_event(CNT_SWEEP_LAUNCHES);
    if (!plan.ablate && !plan.dirty.sticky) {   // (an optimistic solve: the sweep has marked the sticky word if it left anything)
        constexpr size_t rlds = sizeof(double) * (2 + (WEIGHTED ? 1 : 0)) * kRepairWindow * 64;
        auto rkern = sweep_repair_kernel<OP, WEIGHTED>;
        static thread_local bool rattr_done[kMaxDevices] = {};
        bool &rattr_set = rattr_done[current_device()];
        if (!rattr_set) {
            PTV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rkern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
            rattr_set = true;
        }
        const unsigned *handled = launch_repair_jobs<OP, WEIGHTED>(args, g, C, H, G, chunk_state().failcount + 2 * fam, 1L, (long)NC, plan.dirty, stream);
        hipLaunchKernelGGL(rkern, dim3((unsigned)((g.count + 63) / 64)), dim3(64), rlds, stream, args, g, C, H, G, chunk_state().code_mine,
                           chunk_state().code_next, chunk_state().failflags, chunk_state().failcount + 2 * fam, 1L, (long)NC, plan.dirty, handled);
        count_event(CNT_REPAIR_LAUNCHES);
    }
    PTV_HIP(hipGetLastError());
    chunk_state().pol[fam].chunks_done += (long)NC * g.count;
}

// Chunks along the fibre (kernel 2a): dimension-0 sweeps, unweighted.  Codes are laid out [fibre][chunk] (a group writes
// the codes of consecutive chunks of one fibre).  Lanes per segment: a whole wave for long fibres; half or a quarter of
// one when the fibre fits 32 or 16 chunks.
template <int OP, bool WEIGHTED, int H, bool ROBUST, bool RUNS = false>
void launch_along(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam, int rounds) {
    constexpr int C = along_chunk(ROBUST, WEIGHTED);
    if constexpr (RUNS) {   // (fibres of more than a segment: the others have no interior segment to solve run by run)
        if (g.len > 64 * C) {
            launch_along_g<OP, WEIGHTED, H, 64, ROBUST, false, true>(args, g, stream, fam, rounds);
            return;
        }
    }
    // (fibres of one segment: the robust instantiation without its look-back / look-ahead rows)
    if (g.len <= 16 * C)      launch_along_g<OP, WEIGHTED, H, 16, ROBUST, true>(args, g, stream, fam, rounds);
    else if (g.len <= 32 * C) launch_along_g<OP, WEIGHTED, H, 32, ROBUST, true>(args, g, stream, fam, rounds);
    else if (ROBUST && H <= kWarm && g.len <= 64 * C) launch_along_g<OP, WEIGHTED, H, 64, ROBUST, ROBUST && H <= kWarm>(args, g, stream, fam, rounds);
    else                            launch_along_g<OP, WEIGHTED, H, 64, ROBUST>(args, g, stream, fam, rounds);
}

// Global-memory chunks (kernel 2b): chunk C and zone H are run-time values; every link is checked by the repair kernel.
template <int OP, bool WEIGHTED>
void launch_gchunk(const SweepArgs &args, const FibreGeom &g, int C, int H, hipStream_t stream, int fam) {
    const long groups = (g.count + 63) / 64;
    const int NC = (g.len + C - 1) / C;
    chunk_state().ensure(g.count, NC, stream);
    hipLaunchKernelGGL((sweep_gchunk_kernel<OP, WEIGHTED>), dim3((unsigned)groups, (unsigned)NC), dim3(64), 0, stream, args,
                       g, C, H, chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags);
    hipLaunchKernelGGL((sweep_repair_kernel<OP, WEIGHTED>), dim3((unsigned)groups), dim3(64), 0, stream, args, g, C, H, 1,
                       chunk_state().code_mine, chunk_state().code_next, chunk_state().failflags, chunk_state().failcount + 2 * fam, (long)g.count, 1L,
                       DirtyMark{nullptr, 0u, nullptr, nullptr});
    count_event(CNT_SWEEP_LAUNCHES);
    count_event(CNT_REPAIR_LAUNCHES);
    PTV_HIP(hipGetLastError());
    chunk_state().pol[fam].chunks_done += (long)NC * g.count;
}

// ---- strided sweeps through the along-fibre kernel: transpose, sweep, transpose back ---------------------------------------
// When the walks of a strided sweep need long zones (pieces of ~10 samples and more: the row family's modes 1 and 2) the
// 64-fibre tile pays for them in LDS -- one workgroup per CU, every zone staged again -- while the along-fibre kernel
// gets them for free: a lane's zone is its neighbours' chunks.  So the operands are transposed (fibres become
// contiguous; a tiled copy at HBM speed), the sweep runs as a dimension-0 sweep, and the outputs are transposed back.
// Fibre numbering is unchanged: fibre j = slab * inc + off sits at j * len after the transposition of every
// (inc x len) slab.

template <int OP, int H>
void launch_row_along(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam, int rounds) {
    TransposedOperands tr(args, Op<OP>::IN_MASK, Op<OP>::OUT_MASK, g, stream);
    launch_along<OP, false, H, true>(tr.args(), tr.geom(), stream, fam, 2 * rounds);
    tr.finish();
}

template <int OP, bool WEIGHTED, bool TRANSPOSED>
void launch_chunk(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int fam) {
    // The aliasing contract of sweep.hpp, enforced: a chunked sweep stages windows of its input operand(s) while other workgroups
    // write outputs, and an op with KEEP finishes rows with the operand value it captured at staging time -- an output array that
    // IS a staged operand would be read half-written / finished with stale values.  (Epilogue-only operands may alias an output
    // element for element: the thread that writes the element is the one that read it.)  The solvers ping-pong; this is the check.
    {
        const void *staged[2] = {args.a, Op<OP>::NIN > 1 ? args.b : nullptr};
        for (const void *in : staged)
            if (in && (in == args.o0 || in == args.o1)) {
                set_error("launch_sweep: op %d writes an array its chunked sweep stages as fibre samples (outputs must not alias window operands)", OP);
                throw HipFailure{hipErrorInvalidValue};
            }
    }
    ChunkScratch &st = chunk_state();
    ChunkScratch::Policy &pl = st.pol[fam];
    const bool pinned = options().chunk_mode >= 0;
    const bool pin_ok = options().pin && pin_supports((OpId)OP, WEIGHTED, g, args.lam);
    const int seed = st.seed(g, args.lam, WEIGHTED);
    const double seed_f = st.certain_fraction(g, args.lam, WEIGHTED);
    st.sweep_seed_f = seed_f;
    if (pl.workload(g.len, g.count, WEIGHTED, pin_ok, seed) && pl.meas) {   // a new workload: the measurement in flight is of the old one
        double t, f;
        st.evaluate(fam, t, f);
    }
    int mode;
    bool measure = false;
    if (pinned) {
        mode = pl.mode = options().chunk_mode < kModeSeq ? options().chunk_mode : kModeSeq;
        if (!pl.available(mode, true)) mode = pl.up(mode);
    } else if (options().deterministic) {
        // the rung is a function of the sweep's (sampled) input and penalty alone; unsampled inputs (tiny problems) take the
        // rung whose cost and exactness do not depend on the data
        mode = seed >= 0 ? seed : (pin_ok ? 3 : 0);
        if (mode == 1 && WEIGHTED && !pl.available(1, true)) mode = pl.up(mode);
        // Fibres beyond one workgroup's LDS: the pinning solver is then a cooperative grid with grid-wide barriers between its steps (25 ms
        // for 4 M samples whatever the data), and the along-fibre kernel with 64-sample zones a few thousand independent waves: 0.13-0.19 ms
        // on noise at lambda = 1-2 -- until walks stop meeting within a zone (lambda = 3: 31-36 ms, every chunk through the repair kernel).
        // Where a tenth of the edges still exceeds ONE penalty (unit noise: lambda <= 2.3) the zones have it.
        if (mode == 3 && pin_ok && !WEIGHTED && options().along && g.len >= kAlongMinLen && pin_is_long(false, g) &&
            st.edge_fraction_above(g, args.lam, false, 1.0) >= kSeedLongZones)
            mode = 2;
        if (options().verbose && (pl.sweeps == 0 || mode != pl.mode || seed_f != pl.shown_f))
            fprintf(stderr, "[proxtv_amd] policy: family %d sweep %ld (len %d x %ld fibres, lambda %g): certain fraction %.4f, seed %d -> mode %d\n", fam,
                    pl.sweeps, g.len, g.count, args.lam, seed_f, seed, mode);
        pl.shown_f = seed_f;
        pl.mode = mode;
    } else {
        st.ensure_host();
        if (pl.meas && (pl.explore || pl.sweeps - pl.meas_sweep >= kMonitorLag)) st.settle(fam, true);
        mode = pl.choose();
        measure = pl.wants_measurement(pl.meas);
        if (measure) PTV_HIP(hipEventRecord(pl.t0, stream));
    }
    const int rounds = (mode == 1 || mode == 2) ? kRounds : 0;
    // Geometry ladder.  Dimension 0 (chunks along the fibre once a fibre fills most of a lane group): 0 = 16-sample zones,
    // 1 / 2 = 64-sample zones, 3 = the pinning solver (pin.hip; where it does not apply: chunks from global memory, zone
    // 256), 4 = chunks from global memory (zone 1024), 5 = one sequential walk per fibre.
    // Strided sweeps: 0 / 1 = the 64-fibre tile (1: robust instantiation), 2 = transposed copies + the along-fibre kernel
    // with 64-sample zones (or the tile with 64-sample zones), 3 / 4 / 5 as above.
    const bool along_ok = options().along && g.len >= kAlongMinLen;
    bool pinned_done = false;
    if (mode == 3 && pin_ok) {
        int *pieces = nullptr;
        if (measure) {   // the policy's hint from this rung: pieces per sample (numerator and denominator of evaluate())
            st.ensure(g.count, 1, stream);
            pieces = st.failcount + 2 * fam + 1;
        }
        // false: the grid-wide variant wrote nothing (its instantiation does not fit this device at once after all, or it
        // hit the level cap on periodic data) -- the global-memory chunks below take the sweep
        // (knots known a priori: none to be had where the sampled input shows no edge above 4 lambda -- lambda = 3 on unit noise: the
        // search costs 3-6 % of such a sweep; unsampled inputs search)
        // (option pin_seed: 0 none; 1 jumps above 4 lambda, where the sampled input has any; 2, the default: the deepest knots of windows as well
        // -- pincore.hpp --, which find knots where no jump reaches 4 lambda.  No statistic of the input's EDGES says whether windows will: a
        // smooth image under a little noise has none above half a penalty and a knot in every 64-knot window (4096^2 DR, smooth field + 0.1 N(0,1)
        // at lambda = 0.5: 31.6 ms without them, 21.3 with) -- so they always run, and gate their stages themselves, wave by wave: 3 us a sweep
        // where the first stage finds nothing.  Unsampled inputs search both ways.)
        const int seeds = ((options().pin_seed >= 1 && (seed_f < 0.0 || seed_f >= kSeedPins)) ? 1 : 0) | (options().pin_seed >= 2 ? 2 : 0);
        pinned_done = launch_pin((OpId)OP, WEIGHTED, args, g, stream, pieces, seeds);
        count_event(pinned_done ? CNT_PIN_SWEEPS : CNT_PIN_CAP_NEXT_RUNG);
        if (pinned_done) count_event(CNT_SWEEP_LAUNCHES);
        if (pinned_done && measure) pl.chunks_done += (long)g.len * g.count;
    }
    if (pinned_done) {}
    else if (mode >= kModeSeq)  launch_seq<OP, WEIGHTED>(args, g, stream, true);
    else if (mode == 3)    launch_gchunk<OP, WEIGHTED>(args, g, 64, 256, stream, fam);
    else if (mode == 4)    launch_gchunk<OP, WEIGHTED>(args, g, 256, 1024, stream, fam);
    else if (TRANSPOSED && along_ok) {
        // chunks along the fibre: 0 = 16-sample zones ; 1 = the same, robust (second chances inside the wave and across the
        // waves of a workgroup, walks past the look-ahead rows) ; 2 = 64-sample zones, robust
        // (second chances cost the along-fibre kernel a wave's re-walk, and only in waves that need one: twice the rounds of the tile)
        // (rung 0 on data most of whose edges are bends known a priori: interior segments are cut there and solved run by run)
        if (mode == 0 && !WEIGHTED && options().runs && seed_f >= kSeedRuns) {
            if constexpr (!WEIGHTED) launch_along<OP, false, kWarm, false, true>(args, g, stream, fam, 0);
        }
        else if (mode == 0) launch_along<OP, WEIGHTED, kWarm, false>(args, g, stream, fam, 0);
        else if (mode == 1) launch_along<OP, WEIGHTED, kWarm, true>(args, g, stream, fam, 2 * rounds);
        else                launch_along<OP, WEIGHTED, kWarmLong, true>(args, g, stream, fam, 2 * rounds);
    }
    else if (!TRANSPOSED && !WEIGHTED && along_ok && mode == 2) {
        if constexpr (!WEIGHTED) launch_row_along<OP, kWarmLong>(args, g, stream, fam, rounds);
    }
    else if constexpr (!WEIGHTED) {
        if (mode == 2)      launch_chunk_h<OP, false, TRANSPOSED, kWarmLong>(args, g, stream, fam, 0);
        else if (mode == 1 && !TRANSPOSED && options().tile == 1) launch_chunk_h<OP, false, false, kWarm, true, 16, 4, kTail, false, 32>(args, g, stream, fam, rounds);
        else if (mode == 1) launch_chunk_h<OP, false, TRANSPOSED, kWarm, true>(args, g, stream, fam, rounds);
        else if (!TRANSPOSED && options().tile == 1) launch_chunk_h<OP, false, false, kWarm, false, 16, 4, kTail, false, 32>(args, g, stream, fam, 0);
        else                launch_chunk_h<OP, false, TRANSPOSED, kWarm>(args, g, stream, fam, 0);
    } else {
        if (mode == 1 && !TRANSPOSED && options().tile == 1) launch_chunk_h<OP, true, false, kWarm, true, 16, 4, kTail, false, 32>(args, g, stream, fam, rounds);
        else if (mode == 1) launch_chunk_h<OP, true, TRANSPOSED, kWarm, true>(args, g, stream, fam, rounds);
        else if (!TRANSPOSED && options().tile == 1) launch_chunk_h<OP, true, false, kWarm, false, 16, 4, kTail, false, 32>(args, g, stream, fam, 0);
        else           launch_chunk_h<OP, true, TRANSPOSED, kWarm>(args, g, stream, fam, 0);
    }
    pl.sweeps++;
    if (measure) {
        PTV_HIP(hipEventRecord(pl.t1, stream));
        pl.meas = true;
        pl.meas_mode = mode;
        pl.meas_sweep = pl.sweeps;
        pl.meas_slot = (mode < kModeSeq) ? st.enqueue_readback(stream) : -1;
    }
}

template <int OP, bool WEIGHTED>
void launch_op_w(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool allow_chunked, int fam) {
    // chunking pays once a fibre spans several blocks; short fibres stay sequential
    const bool chunked = allow_chunked && g.len >= options().chunk_min_len;
    if (!chunked && !WEIGHTED && options().whole && g.len >= 16 && g.len <= kWholeMax && args.lam >= 0.0) {   // (negative
        // penalties -- tvgen lets them through -- keep the sequential kernel, whose reads past the fibre mirror the reference's;
        // so do fibres of a handful of samples: that kernel divides like the CPU, bit for bit, and loops that end at a
        // bitwise fixed point -- Kolmogorov2_TV -- count their iterations on the last bit)
        // Three kernels for short fibres.  Measured on 512 x 512 x L volumes, unit noise (tools/short_probe.py): up to 32
        // samples the sequential walk wins (0.084 ms per sweep at L = 32 against 0.093 / 0.120); beyond, with pieces of a few
        // samples (the policy's seed says rung 0), ONE block of the chunk kernel -- 4 or 6 chunks of 16 samples walk in
        // parallel -- beats the whole-fibre kernel, which walks 64-96 samples in sequence at five waves per CU (L = 64:
        // 0.136 against 0.206 ms); with longer pieces the whole-fibre kernel, which has no links to lose (0.300 against 0.356).
        if (options().whole == 1 && g.len <= 32) {
            launch_seq<OP, WEIGHTED>(args, g, stream, false);
            return;
        }
        if (options().whole == 1 && chunk_state().seed(g, args.lam, false) == 0) {
            chunk_state().sweep_seed_f = chunk_state().certain_fraction(g, args.lam, false);   // (this sweep's, not the last chunked sweep's: the jobs gate reads it)
            if (g.len <= 64) {
                if (g.inc == 1) launch_chunk_h<OP, false, true, kWarm, false, 16, 4, kTail, true>(args, g, stream, fam, 0);
                else            launch_chunk_h<OP, false, false, kWarm, false, 16, 4, kTail, true>(args, g, stream, fam, 0);
            } else {
                if (g.inc == 1) launch_chunk_h<OP, false, true, kWarm, false, 16, 6, kTail, true>(args, g, stream, fam, 0);
                else            launch_chunk_h<OP, false, false, kWarm, false, 16, 6, kTail, true>(args, g, stream, fam, 0);
            }
            return;
        }
        // short fibres whole in LDS, one lane per fibre (kernel 1b)
        const unsigned blocks = (unsigned)((g.count + 63) / 64);
        if (g.inc == 1) {
            hipLaunchKernelGGL((sweep_whole_kernel<OP, true>), dim3(blocks), dim3(64), sizeof(double) * 65 * (size_t)g.len, stream, args, g);
        } else {
            hipLaunchKernelGGL((sweep_whole_kernel<OP, false>), dim3(blocks), dim3(64), sizeof(double) * 64 * (size_t)g.len, stream, args, g);
        }
        count_event(CNT_SWEEP_LAUNCHES);
        PTV_HIP(hipGetLastError());
    }
    else if (!chunked) launch_seq<OP, WEIGHTED>(args, g, stream, false);
    else if (g.inc == 1) launch_chunk<OP, WEIGHTED, true>(args, g, stream, fam);
    else launch_chunk<OP, WEIGHTED, false>(args, g, stream, fam);
}

// option certify: check what the sweep just wrote (kernel 4), re-solve the fibres that fail, count them.  One small read-back per
// sweep: a validation mode, not a fast path.
// (returns the number of fibres that failed -- their flags are set in `*flags_out` --, or -1 when the sweep cannot be checked)
template <int OP, bool WEIGHTED>
long certify_count(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int **flags_out) {
    if (g.count <= 0 || g.len <= 0) return 0;
    // what cannot be checked: no minimisation behind the sweep (lambda <= 0 -- the identity, or the reference's code as it stands),
    // or an output that IS an operand (in-place sweeps of the sequential kernels: the inputs are gone)
    const void *ins[3] = {(Op<OP>::IN_MASK & 1u) ? args.a : nullptr, (Op<OP>::IN_MASK & 2u) ? args.b : nullptr,
                          (Op<OP>::IN_MASK & 4u) ? args.c : nullptr};
    bool aliased = false;
    for (const void *in : ins) aliased = aliased || (in && (in == args.o0 || in == args.o1));
    if (aliased || (!WEIGHTED && !(args.lam > 0.0))) {
        count_event(CNT_CERTIFY_SKIPPED);
        return -1;
    }
    int *buf = chunk_state().certify_for((size_t)g.count, stream);
    unsigned *count = reinterpret_cast<unsigned *>(buf);
    int *flags = buf + 1;
    if (!chunk_state().certify_notes) chunk_state().certify_notes.reset(new Scratch(sizeof(CertifyNote) * kCertifyNotes));
    CertifyNote *notes = chunk_state().certify_notes->as<CertifyNote>();
    if (g.inc == 1 && g.len >= 64)
        hipLaunchKernelGGL((certify_along_kernel<OP, WEIGHTED>), dim3((unsigned)((g.count + 3) / 4)), dim3(256), 0, stream, args, g, flags, count, notes);
    else
        hipLaunchKernelGGL((certify_strided_kernel<OP, WEIGHTED>), dim3((unsigned)((g.count + 63) / 64)), dim3(64), 0, stream, args, g, flags, count, notes);
    PTV_HIP(hipGetLastError());
    unsigned failed = 0;
    PTV_HIP(hipMemcpyAsync(&failed, count, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    PTV_HIP(hipStreamSynchronize(stream));
    count_event(CNT_CERTIFY_SWEEPS);
    if (failed) PTV_HIP(hipMemsetAsync(count, 0, sizeof(unsigned), stream));
    if (failed && options().verbose) {
        CertifyNote h[kCertifyNotes];
        PTV_HIP(hipMemcpyAsync(h, notes, sizeof(h), hipMemcpyDeviceToHost, stream));
        PTV_HIP(hipStreamSynchronize(stream));
        static const char *const kinds[4] = {"|u| above the penalty", "a step up off the floor wall", "a step down off the ceiling wall", "the total"};
        for (unsigned k = 0; k < failed && k < (unsigned)kCertifyNotes; k++)
            fprintf(stderr, "[proxtv_amd] certify: op %d%s, fibre %ld of %ld (%d samples, stride %ld, lambda %g): %s at sample %d: %.3e against a tolerance of %.3e\n",
                    OP, WEIGHTED ? " weighted" : "", h[k].fibre, g.count, g.len, g.inc, args.lam, kinds[h[k].kind & 3], h[k].where, h[k].viol, h[k].tol);
    }
    if (flags_out) *flags_out = flags;
    return (long)failed;
}

template <int OP, bool WEIGHTED>
void launch_certify(const SweepArgs &args, const FibreGeom &g, hipStream_t stream) {
    int *flags = nullptr;
    const long failed = certify_count<OP, WEIGHTED>(args, g, stream, &flags);
    if (failed > 0) {
        count_event(CNT_CERTIFY_FAILURES, failed);
        if (options().verbose)
            fprintf(stderr, "[proxtv_amd] certify: op %d, %ld fibres of %d samples (stride %ld): %ld failed the optimality conditions -- re-solved sequentially\n",
                    OP, g.count, g.len, g.inc, failed);
        launch_seq<OP, WEIGHTED>(args, g, stream, true, flags);   // (walks the flagged fibres only, and clears their flags)
    }
}

}  // namespace swp
}  // namespace ptv
