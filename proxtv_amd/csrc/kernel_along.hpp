// kernel_along.hpp -- kernel 2a: speculative chunks ALONG contiguous fibres, 64 chunks of one fibre per wave; the known-runs path.
// (One of the pieces of sweep_kernels.hpp, which includes them in order; not meant to be included on its own.)
#pragma once

namespace ptv {
namespace swp {

// ---- kernel 2a: speculative chunks ALONG the fibre (dimension-0 sweeps, unweighted) -------------------------------------
// Fibres of dimension 0 are contiguous in memory, so 64 consecutive chunks of ONE fibre can share a wavefront: lane l
// owns chunk l of a 64-chunk segment.  The segment (+ zone and look-ahead rows) is copied into LDS as it lies in
// memory -- 512-byte coalesced loads, no transposition -- and everything after that stays inside the wave: the walks
// read the shared copy (a lane's zone IS its neighbour's chunk: no row is staged twice but the H + T rows at the two
// ends of a 1088-sample segment, 2 %), a link is proven with one lane shuffle, the rebuild follows the same ownership
// rule, the outputs leave as 512-byte rows.  No workgroup barrier anywhere: the four waves of a workgroup are only
// scheduled together, so the memory phases of one wave overlap the walks of the others on the same CU.
// Chunks are 17 samples long, not 16: lane l walks rows 17 l + t of the linear LDS copy, and 17 is odd, so the 64 lanes
// of a read hit 32 different bank pairs -- the floor for 8-byte accesses -- without any padding (with 16 they would
// hit two).
#ifndef PTV_ALONG_C
#define PTV_ALONG_C 17
#endif
// Weighted sweeps hold two LDS planes per wave (samples, penalties): with 17-sample chunks 17.8 KB, i.e. 8 waves per CU, and the kernel
// idles -- vector pipes 45 % busy, HBM at a third of its rate (profiles/r05_s1_kernel_counters.txt).  Chunks of 9 samples (odd: no bank
// conflicts) halve the segment and with it the LDS: 16 waves per CU.  The walk costs more per sample (a zone per 9 samples instead of
// per 17) and still wins: weighted column sweep 143.3 -> 111.2 us, weighted 4096^2 solve 11.64 -> 10.53 ms; 11 samples (12 waves):
// 117.8 us (profiles/r05_s3_ab_weighted.txt).  (Unweighted: 15 / 13 samples -- 20 waves per CU -- move the column sweep 75.0 ->
// 75.6 / 72.1 us and cost 512-sample fibres their one-segment instantiation: 17 stays.)
#ifndef PTV_ALONG_W_C
#define PTV_ALONG_W_C 9
#endif
constexpr int kAlongC = PTV_ALONG_C;
// Chunk length of the robust instantiation (rungs 1 / 2: pieces of a few samples).  Longer chunks walk the zone less often and --
// what matters more -- shrink the spread between the lanes of a wave, whose walk lasts as long as its slowest lane's: with 31
// samples a wave walks 2.70 -> 2.03 trips per sample at lambda = 0.5 and 4.24 -> 3.03 at 0.7 on the inputs of DR sweeps (host model:
// tools/study/links_study.py).  The price is LDS: 16.9 KB per wave instead of 9.7, eight waves per CU instead of sixteen -- and
// measured (end of round 3) that price is too high: 4096^2 DR at lambda = 0.5 / 0.7 12.7 -> 16.1, 23.7 -> 33.3 ms with 31, 13.7 / 25.0
// with 23.  So: 17, like the plain instantiation.  (An odd number, see above; a chunk's piece ends fit the 32-bit masks of ChunkRec.)
#ifndef PTV_ALONG_ROBUST_C
#define PTV_ALONG_ROBUST_C PTV_ALONG_C
#endif
constexpr int along_chunk(bool robust, bool weighted) { return weighted ? PTV_ALONG_W_C : (robust ? PTV_ALONG_ROBUST_C : kAlongC); }
#ifndef PTV_ALONG_WAVES
#define PTV_ALONG_WAVES 4
#endif
#ifndef PTV_ALONG_UNROLL
#define PTV_ALONG_UNROLL 4
#endif
constexpr int kAlongWaves = PTV_ALONG_WAVES;
// Look-ahead rows after a segment.  Only the segment's LAST lane reads them -- to close the piece that covers its last
// sample -- and a segment is 1088 samples, so they cost next to nothing here: the robust instantiation takes 64 (on DR
// iterates at lambda = 0.5 / 0.7 / 1 on unit noise a walk needs more than 8 rows past its chunk in 2 % / 20 % / 70 % of
// the cases, more than 32 in 0 / 0.01 % / 7 %), where every further sample would be a dependent global read.
constexpr int along_tail_rows(int H, bool robust) { return robust ? 64 : tail_rows(H); }
// Rows kept BEFORE a segment.  The walks start H samples early whatever this is; the robust instantiation keeps 64 so that
// a second chance can start from a bend that lies further back than the zone (the predecessor's last bend sits more than
// 16 samples before the boundary in ~1 % of the cases at lambda = 0.7: a third of the links it still left to the repair kernel).
constexpr int along_zone_rows(int H, bool robust) { return robust && H < 64 ? 64 : H; }

// G lanes share one segment of G chunks: 64 for long fibres; 32 or 16 pack two or four shorter fibres into a wave.
// ROBUST (geometry mode 1: pieces of a few samples, walks that need the whole zone -- or more -- to meet): like the tile
// kernel's robust instantiation, nothing a failed link needs leaves the kernel if it can be helped:
//   * a walk may run past the segment's look-ahead rows (global reads, a few samples, only the lanes that need it);
//   * second chances inside the wave: a lane whose link fails while its predecessor's holds walks its chunk again from the
//     predecessor's last bend (one lane shuffle tells it which); links are re-examined after every round, plan.rounds rounds;
//   * the first lane of a segment, whose predecessor is the last lane of ANOTHER wave, looks that wave's final code up
//     through LDS when both waves sit in the same workgroup (4096-sample fibres: 4 segments = the 4 waves of a workgroup)
//     and gets its second chance from it.
// What is still unproven goes to the repair kernel as before, which also re-checks every link between SEGMENTS (and tile workgroups)
// from the codes the waves finally publish.  Links INSIDE a segment / workgroup that no lane flagged are not re-checked there (the
// repair kernel jumps from flagged chunk to flagged chunk): for those the in-kernel proofs -- code equality, second chances, the
// hand-over across waves -- ARE what exactness rests on.
// ONESEG: fibres of at most one segment (G chunks) -- there is no row before the segment and none after it, so the robust
// instantiation's 64 + 64 rows of look-back / look-ahead are not allocated: a third of its LDS for 512-sample fibres, and with it
// twelve waves per CU become sixteen.
constexpr int kRunsWords = 196;   // RUNS: LDS words per wave (64 runs, 64 + 64 masks, the bend before the segment; padded)
template <int OP, bool WEIGHTED, int H, int G, bool ROBUST, bool ONESEG = false, bool RUNS = false>
__global__ __launch_bounds__(64 * kAlongWaves) void sweep_along_kernel(SweepArgs p, FibreGeom g, ChunkPlan plan, link_t *code_mine,
                                                                        link_t *code_next, int *failflags) {
    constexpr int C = along_chunk(ROBUST, WEIGHTED), SEG = G * C, T = ONESEG ? 0 : along_tail_rows(H, ROBUST), HZ = ONESEG ? 0 : along_zone_rows(H, ROBUST), ROWS = HZ + SEG + T, NG = 64 / G;
    constexpr int NU = (ROWS + G - 1) / G;   // staged elements per lane
    constexpr int UL = (C + 1) / 2;          // epilogue operand fetches in flight per lane (C = 17 rows per lane: 9 + 8)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.gate && *p.gate == 0) return;
    // (the wave number is the same in every lane: said so, everything derived from it -- fibre, segment, base addresses, window
    //  bounds -- lives in scalar registers and the address arithmetic of the memory phases runs on the scalar unit)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gi = lane / G, gl = lane % G;
    double *Yp = reinterpret_cast<double *>(smem) + (size_t)(wave * NG + gi) * (ROWS + 2) * (WEIGHTED ? 2 : 1);
    double *Wp = Yp + (WEIGHTED ? ROWS + 2 : 0);   // per-edge penalties, same rows (weighted sweeps)
    // ROBUST: what a wave's last lane ends up with, for the first lane of the next wave: [kAlongWaves] codes, [kAlongWaves] "ready"
    unsigned *xwave = reinterpret_cast<unsigned *>(reinterpret_cast<double *>(smem) + (size_t)kAlongWaves * NG * (ROWS + 2) * (WEIGHTED ? 2 : 1));
    // plain instantiation: the pull-backs of the walk divide by table (walk_asm.hpp: walk_interior_asm_tab); one table per workgroup
#ifndef PTV_NO_WALK_TABLE   // (the switch stays for A/B builds: the walk then divides with v_rcp_f64 + Newton + residual)
    constexpr bool TAB = H <= kWarm && (ROBUST || H + kAlongC + T < kRecipTable);
#else
    constexpr bool TAB = false;
#endif
    constexpr int TS = ROBUST ? kRecipTableRobust : kRecipTable;
    double *rtab = reinterpret_cast<double *>(xwave + (ROBUST ? 16 : 0));
    if constexpr (TAB) {
        if (threadIdx.x < TS) rtab[threadIdx.x] = threadIdx.x ? 1.0 / (double)threadIdx.x : 0.0;
        if (!ROBUST || G != 64) __syncthreads();   // (before anything else happens: every wave of the workgroup is here; the robust
                                                   //  64-lane instantiation has its own barrier right below)
    }
    if (ROBUST && G == 64) {
        if (lane == 0) xwave[kAlongWaves + wave] = 0u;
        __syncthreads();   // (the only workgroup barrier of the kernel: before anything else happens)
    }
    const int len = g.len;
    const int nseg = (len + SEG - 1) / SEG, NC = (len + C - 1) / C;
    const long wid = (long)blockIdx.x * kAlongWaves + wave;
    const long unit = wid * NG + gi;         // one group of G lanes = one segment of one fibre
    long j, sg_l;
    divmod_nonneg(unit, (long)nseg, j, sg_l);
    const int sg = (int)sg_l;
    const bool live = j < g.count;           // (nothing in this kernel synchronises across waves; a group past the end idles)
    if (plan.trace && lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        plan.trace[8 * (size_t)wid] = ((unsigned long long)xcc << 32) | hwid;
        plan.trace[8 * (size_t)wid + 1] = wall_clock64();
    }
    const long fbase = live ? j * len : 0, wbase = live ? j * (len - 1) : 0;
    const int seg_s = sg * SEG, seg_e = live ? min(len, seg_s + SEG) : seg_s;
    const int lo = seg_s - HZ, hi = min(len, seg_s + SEG + T);

    // ---- stage: the segment as it lies in memory ---------------------------------------------------------------------------
    // `interior`: every row of the window but (first segment) the zone before sample 0 exists -- three segments out of four of a
    // 4096-sample fibre.  Then nothing is tested per element: the loads are scalar base + lane + immediate, the LDS stores lane base +
    // immediate (the first segment's zone rows, which nothing ever reads, take copies of sample 0: a clamped address instead of a mask).
    // Uniform over the wave for whole-wave segments (G = 64: fibre and segment are scalar values there).
    const bool interior = G == 64 && !ONESEG && live && seg_s + SEG + T <= len - 1;
    const unsigned ul = (unsigned)gl;
    auto stage_interior = [&]() {
        constexpr int NB = NU <= 20 ? NU : (NU + 1) / 2;
        const long row0 = fbase + lo, wrow0 = wbase + lo;   // (scalar; the first segment: lo = -HZ, element 0 is clamped below)
#pragma unroll
        for (int b0 = 0; b0 < NU; b0 += NB) {
            double s0[NB], s1[NB], sw[WEIGHTED ? NB : 1];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int rel = G * (b0 + u);
                s0[u] = s1[u] = 0.0;
                if (b0 + u >= NU) continue;
                long idx = (row0 + rel) + (long)ul, widx = (wrow0 + rel) + (long)ul;
                if (rel < HZ) {   // (compile time: the element that holds the zone rows)
                    const int r = max(lo + rel + gl, 0);
                    idx = fbase + r;
                    widx = wbase + r;
                }
                if (rel + G <= ROWS) {
                    Op<OP>::fetch_in(p, idx, s0[u], s1[u]);
                    if (WEIGHTED) sw[WEIGHTED ? u : 0] = p.w[widx];
                } else if (rel + gl < ROWS) {   // (the window's last, partial group of rows)
                    Op<OP>::fetch_in(p, idx, s0[u], s1[u]);
                    if (WEIGHTED) sw[WEIGHTED ? u : 0] = p.w[widx];
                }
            }
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int rel = G * (b0 + u);
                if (b0 + u >= NU) continue;
                if (rel + G <= ROWS || rel + gl < ROWS) {
                    Yp[rel + ul] = Op<OP>::y_of(p, s0[u], s1[u]);
                    if (WEIGHTED) Wp[rel + ul] = sw[WEIGHTED ? u : 0];
                }
            }
        }
    };
    if (interior && !(plan.ablate & 4)) {
        stage_interior();
    } else
    if (live && !(plan.ablate & 4)) {
        // every load of a batch is issued before the first is waited for; NB rows per lane and batch (the 31-sample chunks stage
        // 34 rows per lane: in one batch a two-operand op would hold 136 registers)
        constexpr int NB = NU <= 20 ? NU : (NU + 1) / 2;
#pragma unroll
        for (int b0 = 0; b0 < NU; b0 += NB) {
            double s0[NB], s1[NB], sw[WEIGHTED ? NB : 1];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int r = lo + G * (b0 + u) + gl;
                s0[u] = s1[u] = 0.0;
                if (b0 + u < NU && r >= 0 && r < hi) Op<OP>::fetch_in(p, fbase + r, s0[u], s1[u]);
                if (WEIGHTED) sw[WEIGHTED ? u : 0] = (b0 + u < NU && r >= 0 && r < hi && r < len - 1) ? p.w[wbase + r] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int r = lo + G * (b0 + u) + gl;
                if (b0 + u < NU && r >= 0 && r < hi) {
                    Yp[r - lo] = Op<OP>::y_of(p, s0[u], s1[u]);
                    if (WEIGHTED) Wp[r - lo] = sw[WEIGHTED ? u : 0];
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (plan.trace && lane == 0) plan.trace[8 * (size_t)wid + 2] = wall_clock64();

    // ---- speculative walk of this lane's chunk ---------------------------------------------------------------------------------
    const int cs = seg_s + gl * C;
    const int ce = min(cs + C, len);
    const bool has_chunk = cs < seg_e;
    const int start = max(0, cs - H);
    const LdsWin<WEIGHTED, 1> win{(lds_double *)Yp, (lds_double *)Wp, lo};
    const FarFibre<OP> far{p, fbase, 1, wbase};
    ChunkRec rec;
    bool certain = false;
    // ---- RUNS: the segment cut at the bends known a priori, run by run (chunkcore.hpp "known runs") ------------------------------------
    // Interior segments only (every row of the window exists; the fibre's last sample, with its own tests, is another segment's).  Four
    // phases, all inside the wave: (1) every lane looks at the edges of its chunk -- bends known a priori, and what decides a run of two
    // samples -- and borrows its neighbours' for the runs that cross into and out of its chunk; (2) runs of three and more samples are
    // listed; (3) lane i walks run i from its first bend to the closing one; what it finds goes to the chunks it belongs to; (4) every
    // lane puts its chunk's record together: piece ends and types, the bend before its chunk, the one that closes the segment.
    // Nothing is written to the window before phase 4 is through, and whatever does not fit -- a run longer than a lane takes, more
    // than 64 runs, no known bend within two samples of the segment start or eight of its end -- sends the whole wave to the
    // speculative walk below as if nothing had happened.
    bool solved = false;
    if constexpr (RUNS) {
        if (interior && p.lam > 0.0 && !(plan.ablate & 1)) {
            typedef __attribute__((address_space(3))) unsigned lds_uint;   // (LDS instructions, not flat ones: the atomics below are ds_or / ds_max)
            lds_uint *rl = (lds_uint *)(reinterpret_cast<unsigned *>(rtab + TS) + wave * kRunsWords);   // [0, 64) runs ; [64, 128) ends ; [128, 192) types ; [192] the bend before the segment
            constexpr unsigned CM = (1u << C) - 1u;
            const bool free0 = sg == 0 && lane == 0;   // the fibre starts here: no bend, height 0
            // (1) edges
            const EdgeMasks own = own_edges<C>(win, cs, p.lam);
            EdgeMasks pv, nx;
            pv.K = (unsigned)__shfl_up((int)own.K, 1); pv.P = (unsigned)__shfl_up((int)own.P, 1);
            pv.N = (unsigned)__shfl_up((int)own.N, 1); pv.B = (unsigned)__shfl_up((int)own.B, 1);
            nx.K = (unsigned)__shfl_down((int)own.K, 1); nx.P = (unsigned)__shfl_down((int)own.P, 1);
            nx.N = (unsigned)__shfl_down((int)own.N, 1); nx.B = (unsigned)__shfl_down((int)own.B, 1);
            {   // the edges no chunk of the segment owns: the T = 8 behind it (lanes 0 .. 7), the two before it (lanes 8, 9)
                unsigned k = 0, pp = 0, nn = 0, bb = 0;
                if (lane < 8) one_edge(win.y(seg_e - 1 + lane), win.y(seg_e + lane), p.lam, k, pp, nn, bb);
                else if (lane < 10 && sg > 0) one_edge(win.y(seg_s - 11 + lane), win.y(seg_s - 10 + lane), p.lam, k, pp, nn, bb);
                const unsigned xk = (unsigned)__ballot(k != 0u), xp = (unsigned)__ballot(pp != 0u), xn = (unsigned)__ballot(nn != 0u),
                               xb = (unsigned)__ballot(bb != 0u);
                if (lane == 63) { nx.K = (xk & 0xffu) << kEdgeBias; nx.P = (xp & 0xffu) << kEdgeBias; nx.N = (xn & 0xffu) << kEdgeBias; nx.B = (xb & 0xffu) << kEdgeBias; }
                if (lane == 0)  { pv.K = ((xk >> 8) & 3u) << C; pv.P = ((xp >> 8) & 3u) << C; pv.N = ((xn >> 8) & 3u) << C; pv.B = ((xb >> 8) & 3u) << C; }
            }
            EdgeMasks m;
            m.K = edge_ext<C>(own.K, pv.K, nx.K); m.P = edge_ext<C>(own.P, pv.P, nx.P);
            m.N = edge_ext<C>(own.N, pv.N, nx.N); m.B = edge_ext<C>(own.B, pv.B, nx.B);
            if (free0) m.K = (m.K & ~7u) | 4u;   // (the fibre start delimits the first run like a bend; nothing lies before it)
            unsigned BE, BT, WS;
            settle_short_runs(m, BE, BT, WS);
            if (free0) {   // no rule of thumb across the free end: the first run is walked unless it is one sample long
                BE = (BE & ~8u) | (m.K & 8u);
                BT = (BT & ~8u) | (m.P & m.K & 8u);
                WS = (WS & ~7u) | ((m.K & 8u) ? 0u : 4u);
            }
            // (2) this lane's runs: those whose first sample is its own; lane 0 also the one that comes in from before the segment
            unsigned dom = WS & ((1u << (C + kEdgeBias)) - 1u);
            if (lane > 0) dom &= ~3u;
            bool fail = lane == 0 && !free0 && (m.K & 7u) == 0u;   // (no bend known within two samples of the segment start)
            const int nruns = __popc(dom);
            int pos = nruns;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(pos, o);
                if (lane >= o) pos += t;
            }
            const int total = __shfl(pos, 63);
            pos -= nruns;
            rl[64 + lane] = 0u;
            rl[128 + lane] = 0u;
            if (lane == 0) rl[192] = 0u;
            while (dom) {
                const int b = __ffs((int)dom) - 1;
                dom &= dom - 1u;
                const int e = run_end(m.K, b);
                if (e < 0) fail = true;
                else if (pos < 64) rl[pos] = RunEntry::make(lane, b, e, (int)((m.P >> b) & 1u), free0 && b == kEdgeBias).word;
                pos++;
            }
            bool go = __ballot(fail) == 0ull && total <= 64;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (3) one run per lane
            if (go) {
                bool walked = true;
                if (lane < total) {
                    const RunEntry en{rl[lane]};
                    const int c0 = seg_s + C * en.lane() - kEdgeBias, a = c0 + en.b(), ee = c0 + en.e();
                    ChunkRec rr;
                    Walker w;
                    if (en.free_start()) {
                        walker_start<false>(w, win, 0, p.lam);
                    } else {
                        walker_restart_with<false>(w, a, en.type(), len, p.lam, win.y(a), 0.0, 0.0);
                        rr.mine = rr.next = rr.last = ((link_t)a << 1) | (link_t)en.type();
                    }
                    walk_chunk<OP, false, 1, false, TAB>(w, rr, win, far, hi, a, ee, len, p.lam, (unsigned)(unsigned long long)rtab);
                    walked = rr.done && !rr.failed;
                    if (walked) {
                        const int o = en.b() - kEdgeBias;
                        if (o >= 0) {
                            __hip_atomic_fetch_or(&rl[64 + en.lane()], rr.ends << o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_or(&rl[128 + en.lane()], rr.types << o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {   // (lane 0's run from before the segment: the pieces that end before it only say where the segment hangs)
                            __hip_atomic_fetch_or(&rl[64], rr.ends >> (-o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_or(&rl[128], rr.types >> (-o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const unsigned low = rr.ends & ((1u << (-o)) - 1u);
                            if (low) {
                                const int j = 31 - __clz((int)low);
                                __hip_atomic_fetch_max(&rl[192], ((unsigned)(en.b() + j + 2) << 1) | ((rr.types >> j) & 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
                go = __ballot(!walked) == 0ull;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (4) the chunk's record
            if (go) {
                const unsigned e_own = rl[64 + lane], t_own = rl[128 + lane];
                const unsigned e_prev = lane ? rl[64 + lane - 1] : 0u, t_prev = lane ? rl[128 + lane - 1] : 0u;
                const unsigned ends = ((BE >> (kEdgeBias + 1)) | e_own | (e_prev >> C)) & CM;
                const unsigned types = ((BT >> (kEdgeBias + 1)) | t_own | (t_prev >> C)) & CM & ends;
                const int lastb = ends ? 31 - __clz((int)ends) : -1;
                const link_t lcode = lastb >= 0 ? ((((link_t)(cs + lastb + 1)) << 1) | ((types >> lastb) & 1u)) : 0u;
                // the bend the segment hangs on (lane 0 knows): the last one among the edges before its first sample -- known a priori,
                // settled by rule, or found by the walk that came in from before the segment
                link_t hang = 0u;
                if (lane == 0 && !free0) {
                    unsigned best = rl[192];
                    const unsigned kb = BE & 7u;
                    if (kb) {
                        const int bi = 31 - __clz((int)kb);
                        const unsigned cand = ((unsigned)(bi + 1) << 1) | ((BT >> bi) & 1u);   // (edge + 1: zero means none)
                        best = cand > best ? cand : best;   // (ordered by the edge: the later bend wins; a bend has one type)
                    }
                    if (best) hang = (((link_t)(seg_s + (int)(best >> 1) - 1 - kEdgeBias)) << 1) | (best & 1u);
                }
                hang = (link_t)__shfl((int)hang, 0);
                const unsigned long long has = __ballot(lastb >= 0);
                const unsigned long long lower = has & ((1ull << lane) - 1ull);
                const int src = lower ? 63 - __clzll((long long)lower) : 0;
                const link_t from_lower = (link_t)__shfl((int)lcode, src);
                const link_t mine = lower ? from_lower : hang;
                bool bad_rec = mine == 0u && !(sg == 0 && lane == 0);   // (a first piece longer than a chunk at the fibre start: the walk's)
                link_t tail = 0u;
                if (lane == 63 && !((ends >> (C - 1)) & 1u)) {
                    // the piece that covers the segment's last sample ends behind it: at the first bend among the T edges there
                    const unsigned beyond = (BE >> (kEdgeBias + 1 + C)) | (e_own >> C), tbeyond = (BT >> (kEdgeBias + 1 + C)) | (t_own >> C);
                    if (beyond) {
                        const int j0 = __ffs((int)beyond) - 1;
                        tail = (((link_t)(seg_e + j0 + 1)) << 1) | ((tbeyond >> j0) & 1u);
                    } else {
                        bad_rec = true;
                    }
                }
                if (__ballot(bad_rec) == 0ull) {
                    rec.ends = ends;
                    rec.types = types;
                    rec.mine = mine;
                    rec.next = lastb >= 0 ? lcode : mine;
                    rec.last = lane == 63 && tail ? tail : rec.next;
                    rec.done = true;
                    certain = true;
                    solved = true;
                    if (lane == 0) plan.dirty.note(5);   // (option "why": waves solved run by run)
                } else if (lane == 0) {
                    plan.dirty.note(6);                  // (... that went to the speculative walk after all)
                }
            } else if (lane == 0) {
                plan.dirty.note(6);
            }
        }
    }
    if (has_chunk && !solved && !(plan.ablate & 1)) {
        Walker w;
        // (robust: the whole zone is searched -- a lane that starts at a bend known a priori has no link that could fail)
        constexpr int kLook = ROBUST ? kWarm - 2 : 8;
        int cat = -1, ctype = 0;
        if (start > 0 && H <= kWarm && (WEIGHTED || p.lam > 0.0)) cat = certain_bend_before<WEIGHTED, kLook>(win, cs, len, p.lam, ctype);
        if (cat >= 0) {
            certain = true;
            walker_restart_with<WEIGHTED>(w, cat, ctype, len, p.lam, win.y(cat), WEIGHTED ? win.r(cat - 1) : 0.0,
                                          (WEIGHTED && cat < len - 1) ? win.r(cat) : 0.0);
            rec.mine = rec.next = rec.last = ((link_t)cat << 1) | (link_t)ctype;
        } else {
            walker_start<WEIGHTED>(w, win, start, p.lam);
        }
        walk_chunk<OP, WEIGHTED, 1, ROBUST, TAB>(w, rec, win, far, hi, cs, ce, len, p.lam, (unsigned)(unsigned long long)rtab);
    }
    if (plan.trace && lane == 0) plan.trace[8 * (size_t)wid + 3] = wall_clock64();

    // ---- links: the predecessor is the lane before (a group's first lane: in another group or wave, left to the repair kernel) ----
    bool bad;
    bool head_linked = false, head_bad = false;   // ROBUST: the group's first lane hangs on another wave's last lane / and that link failed
    auto examine = [&]() {
        const link_t prev_next = (link_t)__shfl_up((int)rec.next, 1);
        const bool linked = has_chunk && !(start == 0 || certain) && gl > 0;
        bad = has_chunk && (rec.failed || (linked && (rec.mine == 0 || rec.mine != prev_next)) || (gl == 0 && head_bad));
        return prev_next;
    };
    // one more walk of this lane's chunk from a bend of its predecessor's walk (a bend of the true walk if the predecessor is true)
    auto second_chance = [&](link_t from) {
        const int at = (int)(from >> 1);
        if (from == 0 || at <= max(lo, 0)) return;
        ChunkRec again;
        Walker w;
        walker_restart_with<WEIGHTED>(w, at, (int)(from & 1u), len, p.lam, win.y(at), WEIGHTED ? win.r(at - 1) : 0.0,
                                      (WEIGHTED && at < len - 1) ? win.r(at) : 0.0);
        again.mine = again.next = again.last = from;
        walk_chunk<OP, WEIGHTED, 1, ROBUST, TAB>(w, again, win, far, hi, cs, ce, len, p.lam, (unsigned)(unsigned long long)rtab);
        if (!again.failed) {
            rec = again;
            certain = false;   // from now on the chunk hangs on its predecessor like any other
        }
    };
    auto rounds = [&]() {
        for (int round = 0; round < plan.rounds; round++) {
            const link_t prev_next = examine();
            if (__ballot(bad) == 0ull) break;
            const bool prev_bad = __shfl_up((int)bad, 1) != 0;
            if (bad && gl > 0 && !prev_bad) second_chance(prev_next);
        }
    };
    if constexpr (ROBUST) {
        rounds();
        if constexpr (G == 64) {
            examine();
            // the wave's last lane, as it stands now, for the next wave's first lane
            if (lane == 63) xwave[wave] = (bad || !has_chunk) ? rec.next : (rec.next | kLinkCertain);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 63) __hip_atomic_store(&xwave[kAlongWaves + wave], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            head_linked = has_chunk && lane == 0 && wave > 0 && sg > 0 && !(certain || rec.failed);
            if (head_linked) {
                while (__hip_atomic_load(&xwave[kAlongWaves + wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u)
                    __builtin_amdgcn_s_sleep(1);
                const link_t praw = xwave[wave - 1];
                const link_t prev = praw & ~kLinkCertain;
                head_bad = rec.mine == 0 || rec.mine != prev;
                if (head_bad && (praw & kLinkCertain)) {
                    second_chance(prev);
                    head_bad = rec.failed || rec.mine == 0 || rec.mine != prev;
                }
            }
            if (__ballot(head_linked) != 0ull) rounds();   // (the first lane may have walked again: its successors' links are looked at afresh)
        }
    }
    examine();
    if (gl == 0 && head_bad && !rec.failed) bad = false;   // (not this kernel's to flag: the repair kernel checks the links between segments)
    if (has_chunk) {
        if (rec.failed) {
            rec.mine = kLinkBad;
            rec.next = 0;
        }
        const int chunk = sg * G + gl;
        if (bad) flag_chunk(failflags, j, chunk, NC, plan.dirty, rec.failed);
        code_mine[j * NC + chunk] = (certain && rec.mine != kLinkBad) ? (rec.mine | kLinkCertain) : rec.mine;
        code_next[j * NC + chunk] = rec.next;
        // the segment's last chunk: what the next segment's first chunk must have begun with (checked by that segment at its end)
        if (plan.xlink && gl == G - 1 && sg + 1 < nseg) xlink_publish(plan.xlink + (size_t)j * nseg + sg, plan.dirty.epoch, rec.next);
    }
    // a lane's writes stop at the nearest unproven chunk before it (see GUARD in sweep_chunk_kernel; needed for H > C)
    int wlo = seg_s;
    if (H > C || ROBUST) {
        const unsigned long long all = __ballot(bad || (gl == 0 && head_bad));
        const unsigned long long grp = (G == 64) ? all : ((all >> (gi * G)) & ((1ull << (G & 63)) - 1ull));
        const unsigned long long below = grp & ((1ull << gl) - 1ull);
        if (below) wlo = seg_s + (63 - __clzll((long long)below)) * C;
    }
    if (!WEIGHTED && interior && !(plan.ablate & 1))   // (every lane of the wave holds a whole chunk: the form that keeps it in registers)
        rebuild_owned<Op<OP>, WEIGHTED, C, PTV_ALONG_UNROLL, TAB, lds_double *, (ROBUST ? TS : 0), 2>(win, rec, cs, ce, len, start, !bad, wlo, gl == G - 1, p.lam,
                                                                                      (lds_double *)rtab, nullptr, plan.legacy != 0);
    else if (has_chunk && !(plan.ablate & 1))
        rebuild_owned<Op<OP>, WEIGHTED, C, PTV_ALONG_UNROLL, TAB, lds_double *, (ROBUST ? TS : 0)>(win, rec, cs, ce, len, start, !bad, wlo, gl == G - 1 || ce == len, p.lam,
                                                                                (lds_double *)rtab, nullptr, plan.legacy != 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (plan.trace && lane == 0) plan.trace[8 * (size_t)wid + 4] = wall_clock64();

    // ---- stream the segment out: rows seg_s + G t + gl, t < C ----------------------------------------------------------------------
    if (interior && !(plan.ablate & 2)) {   // (the whole segment exists: scalar base + lane + immediate, nothing tested)
        const long out0 = fbase + seg_s;
#pragma unroll
        for (int t0 = 0; t0 < C; t0 += UL) {
            Ext ex[UL];
#pragma unroll
            for (int u = 0; u < UL; u++)
                ex[u] = (t0 + u < C && !Op<OP>::FUSED) ? Op<OP>::fetch(p, (out0 + G * (t0 + u)) + (long)ul) : Ext{0, 0};
#pragma unroll
            for (int u = 0; u < UL; u++) {
                if (t0 + u >= C) continue;
                const long idx = (out0 + G * (t0 + u)) + (long)ul;
                const double v = Yp[HZ + G * (t0 + u) + ul];
                if (Op<OP>::FUSED) Op<OP>::store_fused(p, idx, v);
                else               Op<OP>::finish(p, idx, ex[u], v);
            }
        }
    } else
    if (live && !(plan.ablate & 2)) {
#pragma unroll
        for (int t0 = 0; t0 < C; t0 += UL) {
            Ext ex[UL];
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const int k = seg_s + G * (t0 + u) + gl;
                ex[u] = (t0 + u < C && k < seg_e && !Op<OP>::FUSED) ? Op<OP>::fetch(p, fbase + k) : Ext{0, 0};
            }
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const int k = seg_s + G * (t0 + u) + gl;
                if (t0 + u < C && k < seg_e) {
                    const double v = Yp[k - lo];
                    if (Op<OP>::FUSED) Op<OP>::store_fused(p, fbase + k, v);
                    else               Op<OP>::finish(p, fbase + k, ex[u], v);
                }
            }
        }
    }
    // the link into this segment, against what the segment before published (it was dispatched earlier and published before its
    // rebuild: almost always there by now -- else the sweep is marked dirty and the repair kernel checks every boundary itself)
    if (plan.xlink && has_chunk && gl == 0 && sg > 0 && !certain) {
        const int why = xlink_check(plan.xlink + (size_t)j * nseg + sg - 1, plan.dirty.epoch, rec.mine);
        if (why) plan.dirty.set(why);
    }
    if (plan.trace && lane == 0) plan.trace[8 * (size_t)wid + 5] = wall_clock64();
}

}  // namespace swp
}  // namespace ptv
