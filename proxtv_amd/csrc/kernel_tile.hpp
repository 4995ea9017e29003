// kernel_tile.hpp -- kernel 2: speculative chunks over an LDS window, strided fibres (tiles of 64 or 32 fibres) and transposed dimension-0 tiles.
// (One of the pieces of sweep_kernels.hpp, which includes them in order; not meant to be included on its own.)
#pragma once

namespace ptv {
namespace swp {

// One workgroup = NW waves = NW consecutive chunks (a "block" of NW*C samples) of the same 64 fibres; it processes
// plan.qpw consecutive blocks of those fibres.  Per block:
//   1. stage the window [block start - H, block end + T) into LDS through the op's input functor: all loads of a
//      thread are issued before the first is waited for; for dimension-0 sweeps the tile is transposed on the way;
//   2. every wave walks its chunk speculatively (LDS only), recording piece ends, bend types and link codes;
//   3. links between consecutive chunks are proven through LDS (and, across workgroups, by sweep_repair_kernel);
//   4. piece values are rebuilt in place (rebuild_owned: a piece is rewritten by the lane in whose chunk it ends), then
//      the block's rows are streamed out: straight from LDS for fused ops, otherwise through the op's output functor
//      (an operand that was staged for the walk and is needed again stays in registers: Op::KEEP).
// LDS carve (dynamic, 16-byte aligned base): Y window | Wt window (weighted) | link codes.
// SHORT: fibres no longer than one block (len <= NW * C: the 64-sample dimension of a 512 x 512 x 64 volume).  The window is
// the fibre itself -- no zone rows before it, no look-ahead rows after it are allocated (HA = TA = 0: chunks still start
// their walks H samples early, inside the block) -- so a workgroup of NW = 4 waves holds 32 KB of LDS and four or five of
// them share a CU; there are no links between workgroups, and the HBM traffic is exactly the algorithmic one.
// FW (fibres per tile, 64 or 32): with FW = 32 a wave carries TWO consecutive chunks of the same 32 fibres (lanes 0-31 the
// first, 32-63 the second), so a workgroup of NW = 4 waves covers the same 8-chunk block over half the fibres: half the LDS,
// FOUR independent workgroups per CU instead of two -- the stage / stream-out phases of one (memory latency) overlap the walks
// of three others -- and rows of 256 bytes towards HBM (two full 128-byte lines).  Strided plain tiles only.
template <int OP, bool WEIGHTED, bool TRANSPOSED, int C, int NW, int H, bool ROUNDS, int T = tail_rows(H), bool SHORT = false, int FW = 64>
__global__ __launch_bounds__(64 * NW, SHORT ? 16 / NW : (FW < 64 ? ((WEIGHTED ? 8 : 16) / NW) : ((WEIGHTED || H > 16 || NW > 8) ? NW / 4 : NW / 2))) void sweep_chunk_kernel(SweepArgs p, FibreGeom g, ChunkPlan plan,
                                                                                   link_t *code_mine, link_t *code_next,
                                                                                   int *failflags) {
    static_assert(FW == 64 || (FW == 32 && !TRANSPOSED && !SHORT && H <= C), "the 32-fibre tile is a strided short-zone tile");
    constexpr int CPW = 64 / FW;            // chunks per wave
    constexpr int NCH = NW * CPW;           // chunks per block
    constexpr int PITCH = TRANSPOSED ? 65 : FW;
    constexpr int HA = SHORT ? 0 : H, TA = SHORT ? 0 : T;   // zone / look-ahead rows the window really has
    constexpr int ROWS = HA + NCH * C + TA;
    static_assert(!(!TRANSPOSED && Op<OP>::KEEP) || (HA % NCH == 0 && TA % NCH == 0), "Op::KEEP relies on whole staging shares");
    constexpr int RB = (ROWS + 63) / 64;                                      // transposed: 64-row blocks per fibre
    constexpr int FPW = (64 + NW - 1) / NW;                                   // transposed: fibres per wave (the last wave's share may be short)
    constexpr int NST = TRANSPOSED ? FPW * RB : (ROWS + NCH - 1) / NCH;       // staged window elements per thread (the last may fall past the window)
    constexpr int UL = 8;                                                     // epilogue rows in flight per lane
    constexpr bool KEEP = !TRANSPOSED && Op<OP>::KEEP;                        // a staged operand is reused by the epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Yp = reinterpret_cast<double *>(smem);
    double *Wp = Yp + (WEIGHTED ? (size_t)ROWS * PITCH : 0);
    // (the walk's look-ahead read of row `hi` lands in whatever follows the Y window -- allocated LDS, value never used)
    link_t *codes = reinterpret_cast<link_t *>(Wp + (size_t)ROWS * PITCH);   // [NCH + 2][FW]; slots NCH, NCH + 1 carry over blocks
    // (bit 31 of a slot -- never part of a code: restart indices are below 2^30 -- says "this lane's link is proven")
    int *anybad = reinterpret_cast<int *>(codes + (NCH + 2) * FW);           // [2], by round parity: some lane of the block has an unproven link
    // A lane rewrites the rows of a piece that ends in its chunk even where they lie in earlier chunks; its walk reaches
    // back H rows (second-chance walks: anywhere in the block).  With H <= C that is the chunk before at most, and if
    // that chunk's lane is unproven its rows are rewritten by the repair kernel anyway.  Further back there may be
    // PROVEN chunks before an unproven one -- rows the repair kernel will not touch -- so those instantiations stop a
    // lane's writes at the nearest unproven chunk before it (GUARD: one flag per lane through LDS, one more barrier).
    constexpr bool GUARD = ROUNDS || H > C;
    unsigned long long *unproven = reinterpret_cast<unsigned long long *>(anybad + 2);   // [NW] lane masks (GUARD)
    // what the first chunk of the workgroup's first block began with, kept for the check at the kernel's end: an LDS row ([64];
    // the pitch-65 tile has no room left for one at two workgroups per CU and keeps it in a register)
    link_t *stash = reinterpret_cast<link_t *>(unproven + NW);
    constexpr link_t kNoCheck = 0xffffffffu;
    link_t began_reg = kNoCheck;
    // the walk's reciprocal table (walk_asm.hpp: walk_interior_asm_tab), after the stash row: the strided short-zone tiles only
    // (the pitch-65 tile has no LDS left for it at two workgroups per CU)
#ifndef PTV_NO_WALK_TABLE   // (the switch stays for A/B builds: the walk then divides with v_rcp_f64 + Newton + residual)
    constexpr bool TAB = (WEIGHTED || !TRANSPOSED) && !SHORT && H <= kWarm && NW <= 8 && (ROUNDS || H + C + T < kRecipTable);
#else
    constexpr bool TAB = false;
#endif
    constexpr int TS = ROUNDS ? kRecipTableRobust : kRecipTable;
    double *rtab = reinterpret_cast<double *>(stash + (TRANSPOSED ? 0 : FW));   // (the pitch-65 tile has no stash row: launch_chunk_h's LDS size)
    if constexpr (TAB) {
        if (threadIdx.x < TS) rtab[threadIdx.x] = threadIdx.x ? 1.0 / (double)threadIdx.x : 0.0;   // (visible after the staging barrier)
    }

    if (p.gate && *p.gate == 0) return;   // uniform over the grid
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform over the wave: scalar)
    // fl: this lane's fibre within the tile ; ch: its chunk within the block (FW = 64: the lane and the wave)
    const int fl = FW == 64 ? lane : (lane & (FW - 1)), ch = FW == 64 ? wave : wave * CPW + lane / FW;
    if (plan.trace && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        plan.trace[8 * (size_t)(blockIdx.x + gridDim.x * blockIdx.y)] = ((unsigned long long)xcc << 32) | hwid;
    }
    trace_mark(plan, 1);
    const int len = g.len;
    const long j0 = (long)blockIdx.x * FW;
    const long j = j0 + fl;
    const bool active = j < g.count;
    long base = 0, wbase = 0;
    if (active) {
        long blk, off;
        divmod_nonneg(j, g.inc, blk, off);
        base = blk * g.inc * len + off;
        wbase = blk * g.inc * (len - 1) + off;
    }
    const FarFibre<OP> far{p, base, g.inc, wbase};

    // Window rows are addressed relative to lo = block start - H (negative for the first block: those rows do not
    // exist and are never touched), so that the share of a thread is the same set of slots in every block.
    // Strided sweeps: element u of a thread is row lo + wave + NW*u of its own fibre (each wave instruction = one
    // coalesced 512-byte row); elements H/NW .. H/NW + C - 1 are rows of the block itself -- the ones the same thread
    // streams out at the end, so an operand staged here can wait in registers for the epilogue (Op::KEEP).
    // Dimension-0 sweeps (fibres contiguous): lanes run ALONG the fibre, element u is row lo + 64*(u % RB) + lane of
    // fibre wave + NW*(u / RB), and the tile is transposed on its way into LDS (pitch 65).
    // The loads of a batch of NB elements are all issued before the first is waited for.  NB = NST (the whole window
    // share of the thread) unless that would not fit the register budget: transposed sweeps stage in two batches (three
    // for two-operand inputs: 48 live doubles spill otherwise), two-operand strided sweeps in two.
    constexpr int NB = TRANSPOSED ? (Op<OP>::NIN > 1 ? (NST + 2) / 3 : (NST + 1) / 2) : (Op<OP>::NIN > 1 ? (NST + 1) / 2 : NST);
    // (PTV_KEEP_N: how many of a thread's C own rows keep the operand -- the rest is fetched again.  Round 4: all 16 spilled 28 registers
    // at the 128-VGPR budget (DR row sweep 116.7 -> 115.1 / 114.5 / 144 us at 4 / 8 / 16 rows), so 8 were kept.  Round 5: with the staging
    // addresses in scalar registers all 16 fit but for eight spilled dwords outside the walk: 108.2 -> 105.1 / 104.0 us at 12 / 16, the
    // second read of s' is gone and with it a tenth of the row sweep's traffic (profiles/r05_s4_ab_keep.txt).  The 64-fibre x 8-wave
    // tile (option tile = 0) stays at 8.)
#ifndef PTV_KEEP_N
#define PTV_KEEP_N 16
#endif
    constexpr int KNW = (FW < 64 || WEIGHTED) ? PTV_KEEP_N : (PTV_KEEP_N < 8 ? PTV_KEEP_N : 8);
    constexpr int KN = KEEP ? (KNW < C ? KNW : C) : 0;
    double kept[KEEP ? KN : 1];
    // `inner` blocks (strided tiles): all FW fibres of the tile exist, the whole window lies inside the fibre (but, first block, the
    // zone before sample 0) and the fibre's last sample beyond it -- nearly every block of a large image.  Uniform over the workgroup,
    // so nothing is tested per element there: the window loads run down the fibre from one address (the zone rows before sample 0,
    // which nothing ever reads, take copies of sample 0: a clamped row instead of a mask), the rebuild takes its FULL form, the
    // stream-out its rows as they come.
    const bool tile_whole = !TRANSPOSED && !SHORT && (long)blockIdx.x * FW + FW <= g.count;
    auto inner_block = [&](int q) { return tile_whole && q * NCH * C + NCH * C + TA <= len - 1; };
    auto stage_as = [&](int q, auto inner_tag) {
        constexpr bool inner = decltype(inner_tag)::value;
        const int cs_wg = q * NCH * C;
        const int lo = cs_wg - HA, hi = min(len, cs_wg + NCH * C + TA);
#pragma unroll
        for (int u0 = 0; u0 < NST; u0 += NB) {
            double s0[NB], s1[NB], sw[NB];
#pragma unroll
            for (int v = 0; v < NB; v++) {
                const int u = u0 + v;
                int r;
                long idx, widx;
                bool ok;
                if (!TRANSPOSED) {
                    r = lo + ch + NCH * u;
                    ok = active && r >= 0 && r < hi && r - lo < ROWS;
                    if constexpr (inner) {
                        ok = NCH * u + NCH <= ROWS || r - lo < ROWS;          // (compile time for all but a ragged last share)
                        if (NCH * u < HA) r = max(r, 0);                        // (compile time: the shares that hold zone rows)
                    }
                    idx = base + (long)r * g.inc;
                    widx = wbase + (long)r * g.inc;
                } else {
                    const long jf = j0 + wave + NW * (u / RB);
                    r = lo + (u % RB) * 64 + lane;
                    ok = wave + NW * (u / RB) < 64 && jf < g.count && r >= 0 && r < hi;
                    idx = jf * len + r;
                    widx = jf * (len - 1) + r;
                }
                ok = ok && u < NST;
                s0[v] = s1[v] = 0.0;
                // (strided tiles: the shares that hold rows two workgroups stage -- the zone and look-ahead rows, and the own rows that are
                //  a neighbour's: the block's first TA and last HA -- are loaded without the streaming hint an op may put on a window operand)
                constexpr int kOwn0 = HA / NCH, kOwn1 = HA / NCH + C;
#ifndef PTV_NO_SHARED_HALO   // (A/B switch)
                // (measured, profiles/r06_s13_ab_halo.txt: DR row sweep 105.0 -> 103.9 us; the weighted tile, one or two workgroups per CU, lost
                //  1 % and keeps the hint everywhere)
                const bool shared_rows = !TRANSPOSED && !SHORT && !WEIGHTED && (u < kOwn0 + (TA + NCH - 1) / NCH || u >= kOwn1 - (HA + NCH - 1) / NCH);
#else
                const bool shared_rows = false;
#endif
                if (ok) {
                    if (shared_rows) fetch_in_shared_or_plain<OP>(p, idx, s0[v], s1[v]);
                    else             Op<OP>::fetch_in(p, idx, s0[v], s1[v]);
                }
                if (WEIGHTED) sw[v] = (ok && (inner || r < len - 1)) ? p.w[widx] : 0.0;
            }
#pragma unroll
            for (int v = 0; v < NB; v++) {
                const int u = u0 + v;
                int r, col;
                bool ok;
                if (!TRANSPOSED) {
                    r = lo + ch + NCH * u;
                    col = fl;
                    ok = active && r >= 0 && r < hi && r - lo < ROWS;
                    if constexpr (inner) ok = NCH * u + NCH <= ROWS || r - lo < ROWS;
                } else {
                    col = wave + NW * (u / RB);
                    r = lo + (u % RB) * 64 + lane;
                    ok = col < 64 && j0 + col < g.count && r >= 0 && r < hi;
                }
                if (ok && u < NST) {
                    Yp[(r - lo) * PITCH + col] = Op<OP>::y_of(p, s0[v], s1[v]);
                    if (WEIGHTED) Wp[(r - lo) * PITCH + col] = sw[v];
                }
                if (KEEP && u >= HA / NCH && u < HA / NCH + KN) kept[(KEEP && u >= HA / NCH && u < HA / NCH + KN) ? u - HA / NCH : 0] = s1[v];
            }
        }
    };
    auto stage = [&](int q) {
        if (inner_block(q)) stage_as(q, std::true_type{});
        else                stage_as(q, std::false_type{});
    };

    const int q_first = blockIdx.y * plan.qpw;
    const int nblk = min(plan.qpw, plan.Q - q_first);

    for (int kb = 0; kb < nblk; kb++) {
        const int q = q_first + kb;
        if (plan.ablate & 4) {
            if (kb == 0)
                for (int e = tid; e < ROWS * PITCH; e += 64 * NW) Yp[e] = (double)((e * 2654435761u) >> 20) * 1e-3;
        } else {
            stage(q);
        }
        __syncthreads();
        if (kb == 0) trace_mark(plan, 2);

        const int cs_wg = q * NCH * C;
        const int lo = cs_wg - HA;
        const int hi = min(len, cs_wg + NCH * C + TA);

        // ---- speculative walk of this wave's chunk(s) -----------------------------------------------------------------
        const int cs = cs_wg + ch * C;
        const int ce = min(cs + C, len);
        const bool has_chunk = active && cs < len;
        const int start = max(0, cs - H);
        const LdsWin<WEIGHTED, PITCH> win{(lds_double *)Yp + fl, (lds_double *)Wp + fl, lo};
        ChunkRec rec;
        PiecePrefix head;
        bool certain = false;
        if (has_chunk && !(plan.ablate & 1)) {
            Walker w;
            // A lane that finds a bend known a priori (chunkcore.hpp) among the kLook edges before its chunk starts its
            // walk AT it -- exact by construction, no warm-up zone to walk, no link to prove.  On noisy data with small
            // lambda (the headline: 78 % of all edges qualify) every lane of a wave does; otherwise the lane falls back to
            // the speculative start.
            // (robust instantiation: the whole zone is searched -- a lane that starts at a bend known a priori has no link that could
            // fail, and failed links across workgroups are what the repair kernel is left with at the upper end of rung 1)
#ifndef PTV_TILE_ROBUST_LOOK
#define PTV_TILE_ROBUST_LOOK 14   // (against 8: 4096^2 DR at lambda = 0.4 / 0.5: 10.01 -> 9.89, 11.47 -> 11.33 ms; nothing from 0.6 on)
#endif
            constexpr int kLook = ROUNDS ? PTV_TILE_ROBUST_LOOK : 8;
            static_assert(H >= kLook + 2, "the certain-bend search reads rows of the warm-up zone");
            int cat = -1, ctype = 0;
            if (start > 0 && H <= kWarm && p.lam > 0.0) cat = certain_bend_before<WEIGHTED, kLook>(win, cs, len, p.lam, ctype);
            if (cat >= 0) {
                certain = true;
                walker_restart_with<WEIGHTED>(w, cat, ctype, len, p.lam, win.y(cat), WEIGHTED ? win.r(cat - 1) : 0.0,
                                              (WEIGHTED && cat < len - 1) ? win.r(cat) : 0.0);
                rec.mine = rec.next = rec.last = ((link_t)cat << 1) | (link_t)ctype;
            } else {
                walker_start<WEIGHTED>(w, win, start, p.lam);
            }
            walk_chunk<OP, WEIGHTED, PITCH, ROUNDS, TAB>(w, rec, win, far, hi, cs, ce, len, p.lam, (unsigned)(unsigned long long)rtab);
            // (the rows before the chunk that belong to its first piece, summed while the window holds samples only: an unproven lane's
            //  are not its own to rely on once the rebuild has begun in other waves -- chunkcore.hpp first_piece_prefix)
            if (!GUARD) head = first_piece_prefix(win, rec, cs, start);
        }
        // ---- prove the links between consecutive chunks ------------------------------------------------------------------
        codes[ch * FW + fl] = rec.next;
        if (ROUNDS && tid == 0) anybad[0] = anybad[1] = 0;
        __syncthreads();   // all walks done: link codes visible, window rows no longer read as walk input
        if (kb == 0) trace_mark(plan, 3);
        const int prev_slot = (ch > 0) ? (ch - 1) * FW + fl : (NCH + ((kb + 1) & 1)) * FW + fl;
        bool bad = false;
        // Second chances inside the block (plan.rounds > 0; data whose walks need more than the zone to meet): a lane
        // whose link fails, while its predecessor's holds, walks its chunk again from the predecessor's last bend -- a
        // bend of the true walk if the predecessor is true.  Every round moves the proven frontier of a failing run one
        // chunk on; links are re-examined after every round (a predecessor that walked again may have changed its
        // code), and what is still unproven after the last round goes to the repair kernel as usual.
        for (int round = 0; ; round++) {
            const bool linked = has_chunk && !(start == 0 || certain) && (ch > 0 || kb > 0);   // hangs on its predecessor
            bad = has_chunk && (rec.failed || (linked && (rec.mine == 0 || rec.mine != (codes[prev_slot] & ~kLinkCertain))));
            if (!ROUNDS || round >= plan.rounds) break;
            if (has_chunk) codes[ch * FW + fl] = bad ? rec.next : (rec.next | kLinkCertain);   // same code, plus the flag
            if (bad) anybad[round & 1] = 1;
            __syncthreads();
            if (!anybad[round & 1]) break;               // uniform
            if (tid == 0) anybad[(round + 1) & 1] = 0;   // set again only after the barrier below
            if (bad && (ch > 0 || kb > 0)) {
                const link_t praw = codes[prev_slot];
                const link_t prev = praw & ~kLinkCertain;
                const int at = (int)(prev >> 1);
                if ((praw & kLinkCertain) && prev != 0 && at > max(lo, 0)) {
                    ChunkRec again;
                    Walker w;
                    walker_restart_with<WEIGHTED>(w, at, (int)(prev & 1u), len, p.lam, win.y(at), WEIGHTED ? win.r(at - 1) : 0.0,
                                                  (WEIGHTED && at < len - 1) ? win.r(at) : 0.0);
                    again.mine = again.next = again.last = prev;
                    walk_chunk<OP, WEIGHTED, PITCH, ROUNDS, TAB>(w, again, win, far, hi, cs, ce, len, p.lam, (unsigned)(unsigned long long)rtab);
                    if (!again.failed) {
                        rec = again;
                        certain = false;   // from now on the chunk hangs on its predecessor like any other
                        codes[ch * FW + fl] = rec.next;
                    }
                }
            }
            __syncthreads();
        }
        if (has_chunk) {
            if (rec.failed) {
                rec.mine = kLinkBad;
                rec.next = 0;
            }
            if (bad) flag_chunk(failflags, j, q * NCH + ch, (len + C - 1) / C, plan.dirty, rec.failed);
            // Every chunk publishes its two codes: sweep_repair_kernel proves the links between workgroups with them
            // and, for a fibre with an unproven link, finds where a repair walk may stop.
            const long slot = (long)(q * NCH + ch) * g.count + j;
            code_mine[slot] = (certain && rec.mine != kLinkBad) ? (rec.mine | kLinkCertain) : rec.mine;
            code_next[slot] = rec.next;
            // ... and the workgroup's last chunk, right now, what the next workgroup's first chunk must have begun with
            if (plan.xlink && kb == nblk - 1 && ch == NCH - 1)
                xlink_publish(plan.xlink + (size_t)blockIdx.y * g.count + j, plan.dirty.epoch, rec.next);
        }
        // (the link INTO this workgroup is checked at the very end, when the workgroup before has surely published)
        if (kb == 0 && ch == 0) {
            const link_t began = (has_chunk && !certain) ? rec.mine : kNoCheck;
            if (TRANSPOSED) began_reg = began;
            else stash[fl] = began;
        }
        // carried to the next block's first chunk; two slots in turn, so that no barrier is needed before the write
        if (ch == NCH - 1) codes[(NCH + (kb & 1)) * FW + fl] = (bad || !has_chunk) ? rec.next : (rec.next | kLinkCertain);
        // Non-fused ops: the operand fetches of the epilogue's first batch of rows go out now and fly while the rebuild
        // runs (the walk's registers are free by now); the second batch is fetched while the first is stored.
        const int ce_wg = min(len, cs_wg + NCH * C);
#ifdef PTV_PREFETCH_EPILOGUE   // measured: the 32 VGPRs it holds across the rebuild spill at two workgroups per CU, 14 % slower
        constexpr bool PREFETCH = !TRANSPOSED && !Op<OP>::FUSED && !KEEP;
#else
        constexpr bool PREFETCH = false;
#endif
        constexpr int NPRE = UL;   // (all C rows would not fit the register budget next to the rebuild)
        Ext pre[PREFETCH ? NPRE : 1];
        if (PREFETCH && active && !(plan.ablate & 2)) {
#pragma unroll
            for (int m = 0; m < NPRE; m++) {
                const int k = min(cs_wg + ch + NCH * m, ce_wg - 1);
                pre[PREFETCH ? m : 0] = Op<OP>::fetch(p, base + (long)k * g.inc);
            }
        }
        int wlo = cs_wg;   // first row this lane may write
        if (GUARD) {
            // (second chances may have replaced the record: the sums of first_piece_prefix now, before the barrier every rebuild waits behind)
            if (has_chunk && !(plan.ablate & 1)) head = first_piece_prefix(win, rec, cs, start);
            // one lane mask per wave: FW = 64 -> the wave's chunk ; FW = 32 -> its two chunks, the later one in the high half
            const unsigned long long mask = __ballot(bad);
            if (lane == 0) unproven[wave] = mask;
            __syncthreads();
            for (int k = ch - 1; k >= 0; k--)
                if ((unproven[k / CPW] >> ((k % CPW) * FW + fl)) & 1ull) {
                    wlo = cs_wg + k * C;
                    break;
                }
        }
        const bool inner = inner_block(q);   // (uniform over the workgroup)
        if (inner && !(plan.ablate & 1))
            rebuild_owned<Op<OP>, WEIGHTED, C, PTV_TILE_UNROLL, TAB, lds_double *, (ROUNDS ? TS : 0), 1>(win, rec, cs, ce, len, start, !bad, wlo,
                                                                                                       ch == NCH - 1, p.lam, (lds_double *)rtab, &head);
        else if (has_chunk && !(plan.ablate & 1))
            rebuild_owned<Op<OP>, WEIGHTED, C, PTV_TILE_UNROLL, TAB, lds_double *, (ROUNDS ? TS : 0)>(win, rec, cs, ce, len, start, !bad, wlo,
                                                                                                    ch == NCH - 1 || ce == len, p.lam, (lds_double *)rtab, &head);
        __syncthreads();
        if (kb == 0) trace_mark(plan, 4);

        // ---- stream the block's NW*C rows out: coalesced 512-byte rows, UL operand fetches in flight per lane ------------------
        if (!(plan.ablate & 2)) {
            if (!TRANSPOSED && inner) {
                // every row of the block exists for every fibre of the tile: nothing is tested
#pragma unroll
                for (int m0 = 0; m0 < C; m0 += UL) {
                    Ext ex[UL];
#pragma unroll
                    for (int u = 0; u < UL; u++) {
                        const long idx = base + (long)(cs_wg + ch + NCH * (m0 + u)) * g.inc;
                        if (KEEP && m0 + u < KN) ex[u] = Op<OP>::fetch_rest(p, idx, kept[(KEEP && m0 + u < KN) ? m0 + u : 0]);
                        else if (!Op<OP>::FUSED) ex[u] = Op<OP>::fetch(p, idx);
                    }
#pragma unroll
                    for (int u = 0; u < UL; u++) {
                        const int k = cs_wg + ch + NCH * (m0 + u);
                        const double v = Yp[(k - lo) * PITCH + fl];
                        if (Op<OP>::FUSED) Op<OP>::store_fused(p, base + (long)k * g.inc, v);
                        else               Op<OP>::finish(p, base + (long)k * g.inc, ex[u], v);
                    }
                }
            } else if (!TRANSPOSED) {
                if (active) {
                    // the thread that staged rows cs_wg + ch + NCH*m streams them out (Op::KEEP: with the staged operand)
#pragma unroll
                    for (int m0 = 0; m0 < C; m0 += UL) {
                        Ext ex[UL];
#pragma unroll
                        for (int u = 0; u < UL; u++) {
                            const int k = min(cs_wg + ch + NCH * (m0 + u), ce_wg - 1);
                            if (KEEP && m0 + u < KN) ex[u] = Op<OP>::fetch_rest(p, base + (long)k * g.inc, kept[(KEEP && m0 + u < KN) ? m0 + u : 0]);
                            else if (PREFETCH && m0 == 0) ex[u] = pre[PREFETCH ? u : 0];
                            else if (!Op<OP>::FUSED) ex[u] = Op<OP>::fetch(p, base + (long)k * g.inc);
                        }
#pragma unroll
                        for (int u = 0; u < UL; u++) {
                            const int k = cs_wg + ch + NCH * (m0 + u);
                            if (k < ce_wg) {
                                const double v = Yp[(k - lo) * PITCH + fl];
                                if (Op<OP>::FUSED) Op<OP>::store_fused(p, base + (long)k * g.inc, v);
                                else               Op<OP>::finish(p, base + (long)k * g.inc, ex[u], v);
                            }
                        }
                    }
                }
            } else {
                constexpr int ERB = (NW * C + 63) / 64;
                constexpr int items = FPW * ERB;
#pragma unroll
                for (int t0 = 0; t0 < items; t0 += UL) {
                    Ext ex[UL];
#pragma unroll
                    for (int u = 0; u < UL; u++) {
                        const int t = t0 + u;
                        const long jf = j0 + wave + NW * (t / ERB);
                        const int k = cs_wg + (t % ERB) * 64 + lane;
                        const bool ok = t < items && wave + NW * (t / ERB) < 64 && jf < g.count && k < ce_wg;
                        ex[u] = (ok && !Op<OP>::FUSED) ? Op<OP>::fetch(p, jf * len + k) : Ext{0, 0};
                    }
#pragma unroll
                    for (int u = 0; u < UL; u++) {
                        const int t = t0 + u;
                        const int f = wave + NW * (t / ERB);
                        const int k = cs_wg + (t % ERB) * 64 + lane;
                        if (t < items && f < 64 && j0 + f < g.count && k < ce_wg) {
                            const double v = Yp[(k - lo) * PITCH + f];
                            if (Op<OP>::FUSED) Op<OP>::store_fused(p, (j0 + f) * len + k, v);
                            else               Op<OP>::finish(p, (j0 + f) * len + k, ex[u], v);
                        }
                    }
                }
            }
        }
        if (kb + 1 < nblk) __syncthreads();   // every wave is done reading this block's window
    }
    if (plan.xlink && blockIdx.y > 0 && ch == 0 && active) {
        const link_t began = TRANSPOSED ? began_reg : stash[fl];   // (written by this very thread)
        if (began != kNoCheck) {
            const int why = xlink_check(plan.xlink + (size_t)(blockIdx.y - 1) * g.count + j, plan.dirty.epoch, began);
            if (why) plan.dirty.set(why);
        }
    }
    trace_mark(plan, 5);
}

}  // namespace swp
}  // namespace ptv
