// walk_asm.hpp -- the interior walk of a chunk (chunkcore.hpp: walk_interior) written out in gfx950 assembly: four loops,
// unweighted / weighted (per-edge penalties from a second LDS plane), each dividing by v_rcp_f64 + Newton or by table
// (walk_interior_asm, _tab, _w, _w_tab; sweep_window.hpp: walk_chunk picks).  Same state machine, same arithmetic and operation
// order as the C++ form, which stays the specification: the host harness runs it, tests/test_walk_asm_emulated.py interprets these
// loops against it, and -DPTV_NO_ASM_WALK substitutes it on the device.  What the hand version buys is the schedule --
//   * the three LDS reads of a trip (next sample; the two samples a rewinding bend restarts from) are issued before the
//     arithmetic that hides them and waited for once;
//   * no selects: the no-bend update, the two pull-backs, the bend bookkeeping and the two post-bend states each run
//     under their own exec mask (68 vector instructions a trip when every path is taken, against ~100 compiled);
//   * no branches but the loop's own.
// Indices are relative to the window start inside the loop (24-bit multiplies for the LDS addresses).
// Hazards the assembler does not see in inline code (gfx940 family): a VALU-written SGPR needs two wait states before a
// VALU reads it as a mask; a transcendental result one before a VALU uses it -- the instruction order provides both.
#pragma once

#include "chunkcore.hpp"

namespace ptv {

template <int PITCH, class Win>
__device__ __forceinline__ void walk_interior_asm(Walker &w, ChunkRec &rec, const Win &win, int lim, int cs, int ce, double lam) {
    if (w.i >= lim || rec.done) return;
    constexpr int PB = PITCH * 8;
    const int wlo = win.lo;
    const unsigned abase = (unsigned)(unsigned long long)win.Y;   // LDS byte address of window row 0 in this lane's column
    int i = w.i - wlo, k0 = w.k0 - wlo, klo = w.klo - wlo, khi = w.khi - wlo;
    unsigned ai = abase + (unsigned)i * PB;
    double lo = w.lo, hi = w.hi, hlo = w.hlo, hhi = w.hhi;
    double yi = win.y(w.i);
    unsigned ends = rec.ends, types = rec.types, mine = rec.mine, next = rec.next, last = rec.last;
    int doneflag = 0;
    const double nlam = -lam, lam2 = 2 * lam, nlam2 = 2 * (-lam);
    const int lim_r = lim - wlo, cs_r = cs - wlo, ce_r = ce - wlo, cem1_r = ce - 1 - wlo, span = ce - cs, pbs = PB;
    double ynx, t0, h1, h2, sd, inv, e, q;   // q doubles as the restart sample of a bend (different lanes)
    int brk, at, aat, sp, code, sh, bit;
    unsigned long long msave, mlive, mcv, mfv, mb, mth, mtl, mdone, m1, m2, m3;
    asm volatile(
        "s_mov_b64 %[msave], exec\n"
        "s_mov_b64 %[mlive], exec\n"
        "s_mov_b64 %[mdone], 0\n"
        ".Lptv_walk_%=:\n"
        "ds_read_b64 %[ynx], %[ai] offset:%[pb]\n"
        "v_add_f64 %[h1], %[lo], -%[yi]\n"
        "v_add_f64 %[h2], %[hi], -%[yi]\n"
        "v_add_f64 %[h1], %[hlo], %[h1]\n"
        "v_add_f64 %[h2], %[hhi], %[h2]\n"
        "v_cmp_lt_f64 %[mcv], %[lam], %[h1]\n"          // low piece through the ceiling
        "v_cmp_gt_f64 %[mfv], %[nlam], %[h2]\n"         // high piece through the floor
        "v_sub_u32 %[sp], %[i], %[k0]\n"
        "s_andn2_b64 %[mfv], %[mfv], %[mcv]\n"
        "s_or_b64 %[mb], %[mcv], %[mfv]\n"
        "v_cndmask_b32_e64 %[brk], %[khi], %[klo], %[mcv]\n"
        "v_cvt_f64_i32 %[sd], %[sp]\n"
        "v_add_u32 %[at], 1, %[brk]\n"
        "v_mad_u32_u24 %[aat], %[at], %[pbs], %[abase]\n"
        // bend lanes: the sample the new piece starts at, and the one after it
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "ds_read_b64 %[q], %[aat]\n"
        "ds_read_b64 %[ynx], %[aat] offset:%[pb]\n"
        // no bend: pull the pieces back inside the tube where they left it
        "s_andn2_b64 exec, %[mlive], %[mb]\n"
        "v_rcp_f64 %[inv], %[sd]\n"
        "v_cmp_ge_f64 %[mth], %[h2], %[lam]\n"
        "v_cmp_le_f64 %[mtl], %[h1], %[nlam]\n"
        "v_fma_f64 %[e], -%[sd], %[inv], 1.0\n"
        "v_min_f64 %[hhi], %[h2], %[lam]\n"
        "v_max_f64 %[hlo], %[h1], %[nlam]\n"
        "v_fma_f64 %[inv], %[e], %[inv], %[inv]\n"
        "s_mov_b64 exec, %[mth]\n"
        "v_add_f64 %[t0], %[lam], -%[h2]\n"
        "v_mul_f64 %[q], %[t0], %[inv]\n"
        "v_fma_f64 %[e], -%[q], %[sd], %[t0]\n"
        "v_fma_f64 %[q], %[e], %[inv], %[q]\n"
        "v_add_f64 %[hi], %[hi], %[q]\n"
        "v_mov_b32 %[khi], %[i]\n"
        "s_mov_b64 exec, %[mtl]\n"
        "v_add_f64 %[t0], %[nlam], -%[h1]\n"
        "v_mul_f64 %[q], %[t0], %[inv]\n"
        "v_fma_f64 %[e], -%[q], %[sd], %[t0]\n"
        "v_fma_f64 %[q], %[e], %[inv], %[q]\n"
        "v_add_f64 %[lo], %[lo], %[q]\n"
        "v_mov_b32 %[klo], %[i]\n"
        // bend: what it leaves behind
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "v_cndmask_b32_e64 %[code], 0, 1, %[mfv]\n"
        "v_add_u32 %[bit], %[wlo], %[at]\n"
        "v_subrev_u32 %[sh], %[csr], %[brk]\n"
        "v_cmp_gt_u32 %[m1], %[span], %[sh]\n"
        "v_cmp_ge_i32 %[m2], %[csr], %[at]\n"
        "v_cmp_ge_i32 %[m3], %[cer], %[at]\n"
        "v_lshl_or_b32 %[code], %[bit], 1, %[code]\n"
        "v_lshlrev_b32_e64 %[bit], %[sh], 1\n"
        "v_mov_b32 %[last], %[code]\n"
        "v_cndmask_b32_e64 %[mine], %[mine], %[code], %[m2]\n"
        "v_cndmask_b32_e64 %[next], %[next], %[code], %[m3]\n"
        "v_cndmask_b32_e64 %[sp], 0, %[bit], %[m1]\n"
        "s_and_b64 %[m1], %[m1], %[mfv]\n"
        "v_or_b32 %[ends], %[ends], %[sp]\n"
        "v_cmp_le_i32 vcc, %[cem1r], %[brk]\n"
        "v_cndmask_b32_e64 %[sp], 0, %[bit], %[m1]\n"
        "s_or_b64 %[mdone], %[mdone], vcc\n"
        "v_or_b32 %[types], %[types], %[sp]\n"
        // ... and the state right after it (closed form, the piece restarts at `at`)
        "v_mov_b32 %[k0], %[brk]\n"
        "v_mov_b32 %[klo], %[at]\n"
        "v_mov_b32 %[khi], %[at]\n"
        "v_mov_b32 %[i], %[at]\n"
        "v_mov_b32 %[ai], %[aat]\n"
        "v_mov_b64 %[hhi], %[lam]\n"
        "v_mov_b64 %[hlo], %[nlam]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, %[mcv]\n"
        "v_mov_b64 %[lo], %[q]\n"
        "v_add_f64 %[hi], %[lam2], %[q]\n"
        "s_mov_b64 exec, %[mfv]\n"
        "v_mov_b64 %[hi], %[q]\n"
        "v_add_f64 %[lo], %[nlam2], %[q]\n"
        // every live lane steps to its next sample
        "s_mov_b64 exec, %[mlive]\n"
        "v_add_u32 %[i], 1, %[i]\n"
        "v_add_u32 %[ai], %[pbs], %[ai]\n"
        "v_mov_b64 %[yi], %[ynx]\n"
        "v_cmp_gt_i32 vcc, %[limr], %[i]\n"
        "s_andn2_b64 vcc, vcc, %[mdone]\n"
        "s_and_b64 exec, %[mlive], vcc\n"
        "s_mov_b64 %[mlive], exec\n"
        "s_cbranch_execnz .Lptv_walk_%=\n"
        "s_mov_b64 exec, %[msave]\n"
        "s_nop 1\n"
        "v_cndmask_b32_e64 %[doneflag], 0, 1, %[mdone]\n"
        : [lo] "+v"(lo), [hi] "+v"(hi), [hlo] "+v"(hlo), [hhi] "+v"(hhi), [yi] "+v"(yi), [i] "+v"(i), [k0] "+v"(k0),
          [klo] "+v"(klo), [khi] "+v"(khi), [ai] "+v"(ai), [ends] "+v"(ends), [types] "+v"(types), [mine] "+v"(mine),
          [next] "+v"(next), [last] "+v"(last), [doneflag] "+v"(doneflag),
          [ynx] "=&v"(ynx), [t0] "=&v"(t0), [h1] "=&v"(h1), [h2] "=&v"(h2),
          [sd] "=&v"(sd), [inv] "=&v"(inv), [e] "=&v"(e), [q] "=&v"(q), [brk] "=&v"(brk), [at] "=&v"(at), [aat] "=&v"(aat),
          [sp] "=&v"(sp), [code] "=&v"(code), [sh] "=&v"(sh), [bit] "=&v"(bit),
          [msave] "=&s"(msave), [mlive] "=&s"(mlive), [mcv] "=&s"(mcv), [mfv] "=&s"(mfv), [mb] "=&s"(mb), [mth] "=&s"(mth),
          [mtl] "=&s"(mtl), [mdone] "=&s"(mdone), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
        : [lam] "s"(lam), [nlam] "s"(nlam), [lam2] "s"(lam2), [nlam2] "s"(nlam2), [pbs] "s"(pbs), [limr] "s"(lim_r),
          [csr] "s"(cs_r), [cer] "s"(ce_r), [cem1r] "s"(cem1_r), [span] "s"(span), [wlo] "s"(wlo), [abase] "v"(abase),
          [pb] "n"(PB)
        : "vcc", "scc", "memory");
    w.lo = lo;
    w.hi = hi;
    w.hlo = hlo;
    w.hhi = hhi;
    w.i = i + wlo;
    w.k0 = k0 + wlo;
    w.klo = klo + wlo;
    w.khi = khi + wlo;
    rec.ends = ends;
    rec.types = types;
    rec.mine = mine;
    rec.next = next;
    rec.last = last;
    rec.done = doneflag != 0;
}

// The same loop with the division of the two pull-backs taken from a table: the span i - k0 of a piece is a small integer
// (at most the rows a walk can cover: zone + chunk + look-ahead = 41 in the plain geometries), and a quotient a / s is ONE
// product with the correctly rounded reciprocal of s -- within one ulp of the division, like the v_rcp_f64 + Newton +
// residual sequence above (host study, -DPTV_TABLE_RECIP in walker.hpp: worst deviation from the oracle 4.9e-16 against
// 4.0e-16, same links proven), for 9 vector instructions instead of 16 and no transcendental.  `rtab`: LDS byte address of a
// table with rtab[s] = 1.0 / s for 1 <= s < TS, filled by the kernel before any walk.  In the plain instantiations a walk never
// leaves its window, so the span is bounded by construction (TS = kRecipTable); the robust ones (SPAN_EXIT) leave the loop when
// a lane's span reaches TS -- a piece of 64 samples on the rung for pieces of a few -- and that lane goes on in the slow tail
// (walker_run through TailSource, which divides), exactly as a lane that reaches the window's end does.
// The read is issued first in the trip and used last: the corrections moved behind the bend's bookkeeping (different lanes:
// a lane either bends or is pulled back), so one wait covers all four LDS reads of a trip.
// Round 4, measured (same box, alternating processes; profiles/r04_s1_ab_matrix.txt): DR column sweep 88.3 -> 83.3 us, plain
// sweeps 91.5 -> 86.2 (columns) / 93.4 -> 87.6 us (rows), 4096^2 DR solve 7.67 -> 7.49 ms at lambda = 0.1, 12.9 -> 12.45 ms at 0.5.
constexpr int kRecipTable = 48;        // plain geometries: zone 16 + chunk 17 + look-ahead 8 = 41 rows at most
constexpr int kRecipTableRobust = 64;

template <int PITCH, bool SPAN_EXIT, class Win>
__device__ __forceinline__ void walk_interior_asm_tab(Walker &w, ChunkRec &rec, const Win &win, int lim, int cs, int ce, double lam,
                                                      unsigned rtab) {
    if (w.i >= lim || rec.done) return;
    constexpr int TS = SPAN_EXIT ? kRecipTableRobust : kRecipTable;
    if (SPAN_EXIT && w.i - w.k0 >= TS) return;   // (the slow tail takes it)
    constexpr int PB = PITCH * 8;
    const int wlo = win.lo;
    const unsigned abase = (unsigned)(unsigned long long)win.Y;
    int i = w.i - wlo, k0 = w.k0 - wlo, klo = w.klo - wlo, khi = w.khi - wlo;
    unsigned ai = abase + (unsigned)i * PB;
    double lo = w.lo, hi = w.hi, hlo = w.hlo, hhi = w.hhi;
    double yi = win.y(w.i);
    unsigned ends = rec.ends, types = rec.types, mine = rec.mine, next = rec.next, last = rec.last;
    int doneflag = 0;
    const double nlam = -lam, lam2 = 2 * lam, nlam2 = 2 * (-lam);
    const int lim_r = lim - wlo, cs_r = cs - wlo, ce_r = ce - wlo, cem1_r = ce - 1 - wlo, span = ce - cs, pbs = PB;
    double ynx, t0, h1, h2, inv, q;
    int brk, at, aat, code, sh, bit;
    int sp = i - k0;
    const int tsz = TS;
    unsigned atab;
    unsigned long long msave, mlive, mcv, mfv, mb, mth, mtl, mdone, m1, m2, m3;
    asm volatile(
        "s_mov_b64 %[msave], exec\n"
        "s_mov_b64 %[mlive], exec\n"
        "s_mov_b64 %[mdone], 0\n"
        ".Lptv_walkt_%=:\n"
        "ds_read_b64 %[ynx], %[ai] offset:%[pb]\n"
        "v_lshl_add_u32 %[atab], %[sp], 3, %[rtab]\n"
        "v_add_f64 %[h1], %[lo], -%[yi]\n"
        "v_add_f64 %[h2], %[hi], -%[yi]\n"
        "ds_read_b64 %[inv], %[atab]\n"                 // 1 / (i - k0)
        "v_add_f64 %[h1], %[hlo], %[h1]\n"
        "v_add_f64 %[h2], %[hhi], %[h2]\n"
        "v_cmp_lt_f64 %[mcv], %[lam], %[h1]\n"          // low piece through the ceiling
        "v_cmp_gt_f64 %[mfv], %[nlam], %[h2]\n"         // high piece through the floor
        "s_nop 0\n"
        "s_andn2_b64 %[mfv], %[mfv], %[mcv]\n"
        "s_or_b64 %[mb], %[mcv], %[mfv]\n"
        "v_cndmask_b32_e64 %[brk], %[khi], %[klo], %[mcv]\n"
        "v_add_u32 %[at], 1, %[brk]\n"
        "v_mad_u32_u24 %[aat], %[at], %[pbs], %[abase]\n"
        // bend lanes: the sample the new piece starts at, and the one after it
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "ds_read_b64 %[q], %[aat]\n"
        "ds_read_b64 %[ynx], %[aat] offset:%[pb]\n"
        // no bend: the new heights, and which of the two pieces left the tube
        "s_andn2_b64 exec, %[mlive], %[mb]\n"
        "v_cmp_ge_f64 %[mth], %[h2], %[lam]\n"
        "v_cmp_le_f64 %[mtl], %[h1], %[nlam]\n"
        "v_min_f64 %[hhi], %[h2], %[lam]\n"
        "v_max_f64 %[hlo], %[h1], %[nlam]\n"
        // bend: what it leaves behind
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "v_cndmask_b32_e64 %[code], 0, 1, %[mfv]\n"
        "v_add_u32 %[bit], %[wlo], %[at]\n"
        "v_subrev_u32 %[sh], %[csr], %[brk]\n"
        "v_cmp_gt_u32 %[m1], %[span], %[sh]\n"
        "v_cmp_ge_i32 %[m2], %[csr], %[at]\n"
        "v_cmp_ge_i32 %[m3], %[cer], %[at]\n"
        "v_lshl_or_b32 %[code], %[bit], 1, %[code]\n"
        "v_lshlrev_b32_e64 %[bit], %[sh], 1\n"
        "v_mov_b32 %[last], %[code]\n"
        "v_cndmask_b32_e64 %[mine], %[mine], %[code], %[m2]\n"
        "v_cndmask_b32_e64 %[next], %[next], %[code], %[m3]\n"
        "v_cndmask_b32_e64 %[sp], 0, %[bit], %[m1]\n"
        "s_and_b64 %[m1], %[m1], %[mfv]\n"
        "v_or_b32 %[ends], %[ends], %[sp]\n"
        "v_cmp_le_i32 vcc, %[cem1r], %[brk]\n"
        "v_cndmask_b32_e64 %[sp], 0, %[bit], %[m1]\n"
        "s_or_b64 %[mdone], %[mdone], vcc\n"
        "v_or_b32 %[types], %[types], %[sp]\n"
        // ... and the state right after it (closed form, the piece restarts at `at`)
        "v_mov_b32 %[k0], %[brk]\n"
        "v_mov_b32 %[klo], %[at]\n"
        "v_mov_b32 %[khi], %[at]\n"
        "v_mov_b32 %[i], %[at]\n"
        "v_mov_b32 %[ai], %[aat]\n"
        "v_mov_b64 %[hhi], %[lam]\n"
        "v_mov_b64 %[hlo], %[nlam]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, %[mcv]\n"
        "v_mov_b64 %[lo], %[q]\n"
        "v_add_f64 %[hi], %[lam2], %[q]\n"
        "s_mov_b64 exec, %[mfv]\n"
        "v_mov_b64 %[hi], %[q]\n"
        "v_add_f64 %[lo], %[nlam2], %[q]\n"
        // no bend: pull the pieces back inside the tube where they left it (mth / mtl hold no bending lane)
        "s_mov_b64 exec, %[mth]\n"
        "v_add_f64 %[t0], %[lam], -%[h2]\n"
        "v_mul_f64 %[t0], %[t0], %[inv]\n"
        "v_mov_b32 %[khi], %[i]\n"
        "v_add_f64 %[hi], %[hi], %[t0]\n"
        "s_mov_b64 exec, %[mtl]\n"
        "v_add_f64 %[t0], %[nlam], -%[h1]\n"
        "v_mul_f64 %[t0], %[t0], %[inv]\n"
        "v_mov_b32 %[klo], %[i]\n"
        "v_add_f64 %[lo], %[lo], %[t0]\n"
        // every live lane steps to its next sample
        "s_mov_b64 exec, %[mlive]\n"
        "v_add_u32 %[i], 1, %[i]\n"
        "v_add_u32 %[ai], %[pbs], %[ai]\n"
        "v_mov_b64 %[yi], %[ynx]\n"
        "v_sub_u32 %[sp], %[i], %[k0]\n"               // the span of the next trip: its table index ...
        "v_cmp_gt_i32 vcc, %[limr], %[i]\n"
        "v_cmp_gt_u32 %[m1], %[tsz], %[sp]\n"          // ... and the lane leaves the loop where the table ends (robust instantiations;
        "s_and_b64 vcc, vcc, %[m1]\n"                  //     never true in the plain ones)
        "s_andn2_b64 vcc, vcc, %[mdone]\n"
        "s_and_b64 exec, %[mlive], vcc\n"
        "s_mov_b64 %[mlive], exec\n"
        "s_cbranch_execnz .Lptv_walkt_%=\n"
        "s_mov_b64 exec, %[msave]\n"
        "s_nop 1\n"
        "v_cndmask_b32_e64 %[doneflag], 0, 1, %[mdone]\n"
        : [lo] "+v"(lo), [hi] "+v"(hi), [hlo] "+v"(hlo), [hhi] "+v"(hhi), [yi] "+v"(yi), [i] "+v"(i), [k0] "+v"(k0),
          [klo] "+v"(klo), [khi] "+v"(khi), [ai] "+v"(ai), [ends] "+v"(ends), [types] "+v"(types), [mine] "+v"(mine),
          [next] "+v"(next), [last] "+v"(last), [doneflag] "+v"(doneflag), [sp] "+v"(sp),
          [ynx] "=&v"(ynx), [t0] "=&v"(t0), [h1] "=&v"(h1), [h2] "=&v"(h2),
          [inv] "=&v"(inv), [q] "=&v"(q), [brk] "=&v"(brk), [at] "=&v"(at), [aat] "=&v"(aat),
          [code] "=&v"(code), [sh] "=&v"(sh), [bit] "=&v"(bit), [atab] "=&v"(atab),
          [msave] "=&s"(msave), [mlive] "=&s"(mlive), [mcv] "=&s"(mcv), [mfv] "=&s"(mfv), [mb] "=&s"(mb), [mth] "=&s"(mth),
          [mtl] "=&s"(mtl), [mdone] "=&s"(mdone), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
        : [lam] "s"(lam), [nlam] "s"(nlam), [lam2] "s"(lam2), [nlam2] "s"(nlam2), [pbs] "s"(pbs), [limr] "s"(lim_r),
          [csr] "s"(cs_r), [cer] "s"(ce_r), [cem1r] "s"(cem1_r), [span] "s"(span), [wlo] "s"(wlo), [abase] "v"(abase),
          [rtab] "s"(rtab), [tsz] "s"(tsz), [pb] "n"(PB)
        : "vcc", "scc", "memory");
    w.lo = lo;
    w.hi = hi;
    w.hlo = hlo;
    w.hhi = hhi;
    w.i = i + wlo;
    w.k0 = k0 + wlo;
    w.klo = klo + wlo;
    w.khi = khi + wlo;
    rec.ends = ends;
    rec.types = types;
    rec.mine = mine;
    rec.next = next;
    rec.last = last;
    rec.done = doneflag != 0;
}

// The weighted walk (per-edge penalties in a second window plane, `wdelta` bytes after the sample plane): the same loop
// with the penalty of the current edge in a register instead of lambda in a scalar, and three more reads on a bend (the
// penalties either side of the restart sample, and the one the next trip needs).  Same arithmetic and order as
// walk_interior<true> (reference: src/TVL1Wopt.cpp:364-567).
template <int PITCH, class Win>
__device__ __forceinline__ void walk_interior_asm_w(Walker &w, ChunkRec &rec, const Win &win, int lim, int cs, int ce) {
    if (w.i >= lim || rec.done) return;
    constexpr int PB = PITCH * 8;
    const int wlo = win.lo;
    const unsigned abase = (unsigned)(unsigned long long)win.Y;
    const int wdelta = (int)((unsigned)(unsigned long long)win.Wt - abase);
    int i = w.i - wlo, k0 = w.k0 - wlo, klo = w.klo - wlo, khi = w.khi - wlo;
    unsigned ai = abase + (unsigned)i * PB;
    unsigned awi = ai + (unsigned)wdelta;
    double lo = w.lo, hi = w.hi, hlo = w.hlo, hhi = w.hhi;
    double yi = win.y(w.i), r = win.r(w.i);
    unsigned ends = rec.ends, types = rec.types, mine = rec.mine, next = rec.next, last = rec.last;
    int doneflag = 0;
    const int lim_r = lim - wlo, cs_r = cs - wlo, ce_r = ce - wlo, cem1_r = ce - 1 - wlo, span = ce - cs, pbs = PB;
    double ynx, rnx, wp, wc, t0, h1, h2, sd, inv, e, q;
    int brk, at, aat, abw, sp, code, sh, bit;
    unsigned long long msave, mlive, mcv, mfv, mb, mth, mtl, mdone, m1, m2, m3;
    asm volatile(
        "s_mov_b64 %[msave], exec\n"
        "s_mov_b64 %[mlive], exec\n"
        "s_mov_b64 %[mdone], 0\n"
        ".Lptv_walkw_%=:\n"
        "ds_read_b64 %[ynx], %[ai] offset:%[pb]\n"
        "ds_read_b64 %[rnx], %[awi] offset:%[pb]\n"
        "v_add_f64 %[h1], %[lo], -%[yi]\n"
        "v_add_f64 %[h2], %[hi], -%[yi]\n"
        "v_add_f64 %[h1], %[hlo], %[h1]\n"
        "v_add_f64 %[h2], %[hhi], %[h2]\n"
        "v_cmp_lt_f64 %[mcv], %[r], %[h1]\n"
        "v_cmp_gt_f64 %[mfv], -%[r], %[h2]\n"
        "v_sub_u32 %[sp], %[i], %[k0]\n"
        "s_andn2_b64 %[mfv], %[mfv], %[mcv]\n"
        "s_or_b64 %[mb], %[mcv], %[mfv]\n"
        "v_cndmask_b32_e64 %[brk], %[khi], %[klo], %[mcv]\n"
        "v_cvt_f64_i32 %[sd], %[sp]\n"
        "v_add_u32 %[at], 1, %[brk]\n"
        "v_mad_u32_u24 %[aat], %[at], %[pbs], %[abase]\n"
        "v_mad_u32_u24 %[abw], %[brk], %[pbs], %[abase]\n"
        "v_add_u32 %[abw], %[wd], %[abw]\n"                       // penalty plane, row brk
        // bend lanes: the restart sample, the one after it, the penalties of edges brk, at, at + 1
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "ds_read_b64 %[q], %[aat]\n"
        "ds_read_b64 %[ynx], %[aat] offset:%[pb]\n"
        "ds_read_b64 %[wp], %[abw]\n"
        "ds_read_b64 %[wc], %[abw] offset:%[pb]\n"
        "ds_read_b64 %[rnx], %[abw] offset:%[pb2]\n"
        // no bend
        "s_andn2_b64 exec, %[mlive], %[mb]\n"
        "v_rcp_f64 %[inv], %[sd]\n"
        "v_cmp_ge_f64 %[mth], %[h2], %[r]\n"
        "v_cmp_le_f64 %[mtl], %[h1], -%[r]\n"
        "v_fma_f64 %[e], -%[sd], %[inv], 1.0\n"
        "v_min_f64 %[hhi], %[h2], %[r]\n"
        "v_max_f64 %[hlo], %[h1], -%[r]\n"
        "v_fma_f64 %[inv], %[e], %[inv], %[inv]\n"
        "s_mov_b64 exec, %[mth]\n"
        "v_add_f64 %[t0], %[r], -%[h2]\n"
        "v_mul_f64 %[q], %[t0], %[inv]\n"
        "v_fma_f64 %[e], -%[q], %[sd], %[t0]\n"
        "v_fma_f64 %[q], %[e], %[inv], %[q]\n"
        "v_add_f64 %[hi], %[hi], %[q]\n"
        "v_mov_b32 %[khi], %[i]\n"
        "s_mov_b64 exec, %[mtl]\n"
        "v_add_f64 %[t0], -%[r], -%[h1]\n"
        "v_mul_f64 %[q], %[t0], %[inv]\n"
        "v_fma_f64 %[e], -%[q], %[sd], %[t0]\n"
        "v_fma_f64 %[q], %[e], %[inv], %[q]\n"
        "v_add_f64 %[lo], %[lo], %[q]\n"
        "v_mov_b32 %[klo], %[i]\n"
        // bend: what it leaves behind
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "v_cndmask_b32_e64 %[code], 0, 1, %[mfv]\n"
        "v_add_u32 %[bit], %[wlo], %[at]\n"
        "v_subrev_u32 %[sh], %[csr], %[brk]\n"
        "v_cmp_gt_u32 %[m1], %[span], %[sh]\n"
        "v_cmp_ge_i32 %[m2], %[csr], %[at]\n"
        "v_cmp_ge_i32 %[m3], %[cer], %[at]\n"
        "v_lshl_or_b32 %[code], %[bit], 1, %[code]\n"
        "v_lshlrev_b32_e64 %[bit], %[sh], 1\n"
        "v_mov_b32 %[last], %[code]\n"
        "v_cndmask_b32_e64 %[mine], %[mine], %[code], %[m2]\n"
        "v_cndmask_b32_e64 %[next], %[next], %[code], %[m3]\n"
        "v_cndmask_b32_e64 %[sp], 0, %[bit], %[m1]\n"
        "s_and_b64 %[m1], %[m1], %[mfv]\n"
        "v_or_b32 %[ends], %[ends], %[sp]\n"
        "v_cmp_le_i32 vcc, %[cem1r], %[brk]\n"
        "v_cndmask_b32_e64 %[sp], 0, %[bit], %[m1]\n"
        "s_or_b64 %[mdone], %[mdone], vcc\n"
        "v_or_b32 %[types], %[types], %[sp]\n"
        // ... and the state right after it:  a = yn +- wp ; lo = a - wc ; hi = a + wc ; heights -+ wc
        "v_mov_b32 %[k0], %[brk]\n"
        "v_mov_b32 %[klo], %[at]\n"
        "v_mov_b32 %[khi], %[at]\n"
        "v_mov_b32 %[i], %[at]\n"
        "v_mov_b32 %[ai], %[aat]\n"
        "v_add_u32 %[awi], %[pbs], %[abw]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, %[mcv]\n"
        "v_add_f64 %[q], %[q], %[wp]\n"
        "s_mov_b64 exec, %[mfv]\n"
        "v_add_f64 %[q], %[q], -%[wp]\n"
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "v_add_f64 %[lo], %[q], -%[wc]\n"
        "v_add_f64 %[hi], %[q], %[wc]\n"
        "v_mov_b64 %[hhi], %[wc]\n"
        "v_add_f64 %[hlo], 0, -%[wc]\n"
        // every live lane steps to its next sample
        "s_mov_b64 exec, %[mlive]\n"
        "v_add_u32 %[i], 1, %[i]\n"
        "v_add_u32 %[ai], %[pbs], %[ai]\n"
        "v_add_u32 %[awi], %[pbs], %[awi]\n"
        "v_mov_b64 %[yi], %[ynx]\n"
        "v_mov_b64 %[r], %[rnx]\n"
        "v_cmp_gt_i32 vcc, %[limr], %[i]\n"
        "s_andn2_b64 vcc, vcc, %[mdone]\n"
        "s_and_b64 exec, %[mlive], vcc\n"
        "s_mov_b64 %[mlive], exec\n"
        "s_cbranch_execnz .Lptv_walkw_%=\n"
        "s_mov_b64 exec, %[msave]\n"
        "s_nop 1\n"
        "v_cndmask_b32_e64 %[doneflag], 0, 1, %[mdone]\n"
        : [lo] "+v"(lo), [hi] "+v"(hi), [hlo] "+v"(hlo), [hhi] "+v"(hhi), [yi] "+v"(yi), [r] "+v"(r), [i] "+v"(i), [k0] "+v"(k0),
          [klo] "+v"(klo), [khi] "+v"(khi), [ai] "+v"(ai), [awi] "+v"(awi), [ends] "+v"(ends), [types] "+v"(types),
          [mine] "+v"(mine), [next] "+v"(next), [last] "+v"(last), [doneflag] "+v"(doneflag),
          [ynx] "=&v"(ynx), [rnx] "=&v"(rnx), [wp] "=&v"(wp), [wc] "=&v"(wc), [t0] "=&v"(t0), [h1] "=&v"(h1), [h2] "=&v"(h2),
          [sd] "=&v"(sd), [inv] "=&v"(inv), [e] "=&v"(e), [q] "=&v"(q), [brk] "=&v"(brk), [at] "=&v"(at), [aat] "=&v"(aat),
          [abw] "=&v"(abw), [sp] "=&v"(sp), [code] "=&v"(code), [sh] "=&v"(sh), [bit] "=&v"(bit),
          [msave] "=&s"(msave), [mlive] "=&s"(mlive), [mcv] "=&s"(mcv), [mfv] "=&s"(mfv), [mb] "=&s"(mb), [mth] "=&s"(mth),
          [mtl] "=&s"(mtl), [mdone] "=&s"(mdone), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
        : [pbs] "s"(pbs), [limr] "s"(lim_r), [csr] "s"(cs_r), [cer] "s"(ce_r), [cem1r] "s"(cem1_r), [span] "s"(span),
          [wlo] "s"(wlo), [wd] "s"(wdelta), [abase] "v"(abase), [pb] "n"(PB), [pb2] "n"(2 * PB)
        : "vcc", "scc", "memory");
    w.lo = lo;
    w.hi = hi;
    w.hlo = hlo;
    w.hhi = hhi;
    w.i = i + wlo;
    w.k0 = k0 + wlo;
    w.klo = klo + wlo;
    w.khi = khi + wlo;
    rec.ends = ends;
    rec.types = types;
    rec.mine = mine;
    rec.next = next;
    rec.last = last;
    rec.done = doneflag != 0;
}

// The weighted walk with the table (see walk_interior_asm_tab): same reordering, same exit where the table ends.
template <int PITCH, bool SPAN_EXIT, class Win>
__device__ __forceinline__ void walk_interior_asm_w_tab(Walker &w, ChunkRec &rec, const Win &win, int lim, int cs, int ce, unsigned rtab) {
    if (w.i >= lim || rec.done) return;
    constexpr int TS = SPAN_EXIT ? kRecipTableRobust : kRecipTable;
    if (SPAN_EXIT && w.i - w.k0 >= TS) return;   // (the slow tail takes it)
    constexpr int PB = PITCH * 8;
    const int wlo = win.lo;
    const unsigned abase = (unsigned)(unsigned long long)win.Y;
    const int wdelta = (int)((unsigned)(unsigned long long)win.Wt - abase);
    int i = w.i - wlo, k0 = w.k0 - wlo, klo = w.klo - wlo, khi = w.khi - wlo;
    unsigned ai = abase + (unsigned)i * PB;
    unsigned awi = ai + (unsigned)wdelta;
    double lo = w.lo, hi = w.hi, hlo = w.hlo, hhi = w.hhi;
    double yi = win.y(w.i), r = win.r(w.i);
    unsigned ends = rec.ends, types = rec.types, mine = rec.mine, next = rec.next, last = rec.last;
    int doneflag = 0;
    const int lim_r = lim - wlo, cs_r = cs - wlo, ce_r = ce - wlo, cem1_r = ce - 1 - wlo, span = ce - cs, pbs = PB;
    double ynx, rnx, wp, wc, t0, h1, h2, inv, q;
    int brk, at, aat, abw, code, sh, bit;
    int sp = i - k0;
    const int tsz = TS;
    unsigned atab;
    unsigned long long msave, mlive, mcv, mfv, mb, mth, mtl, mdone, m1, m2, m3;
    asm volatile(
        "s_mov_b64 %[msave], exec\n"
        "s_mov_b64 %[mlive], exec\n"
        "s_mov_b64 %[mdone], 0\n"
        ".Lptv_walkwt_%=:\n"
        "ds_read_b64 %[ynx], %[ai] offset:%[pb]\n"
        "ds_read_b64 %[rnx], %[awi] offset:%[pb]\n"
        "v_lshl_add_u32 %[atab], %[sp], 3, %[rtab]\n"
        "ds_read_b64 %[inv], %[atab]\n"                 // 1 / (i - k0)
        "v_add_f64 %[h1], %[lo], -%[yi]\n"
        "v_add_f64 %[h2], %[hi], -%[yi]\n"
        "v_add_f64 %[h1], %[hlo], %[h1]\n"
        "v_add_f64 %[h2], %[hhi], %[h2]\n"
        "v_cmp_lt_f64 %[mcv], %[r], %[h1]\n"
        "v_cmp_gt_f64 %[mfv], -%[r], %[h2]\n"
        "s_nop 0\n"
        "s_andn2_b64 %[mfv], %[mfv], %[mcv]\n"
        "s_or_b64 %[mb], %[mcv], %[mfv]\n"
        "v_cndmask_b32_e64 %[brk], %[khi], %[klo], %[mcv]\n"
        "v_add_u32 %[at], 1, %[brk]\n"
        "v_mad_u32_u24 %[aat], %[at], %[pbs], %[abase]\n"
        "v_mad_u32_u24 %[abw], %[brk], %[pbs], %[abase]\n"
        "v_add_u32 %[abw], %[wd], %[abw]\n"                       // penalty plane, row brk
        // bend lanes: the restart sample, the one after it, the penalties of edges brk, at, at + 1
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "ds_read_b64 %[q], %[aat]\n"
        "ds_read_b64 %[ynx], %[aat] offset:%[pb]\n"
        "ds_read_b64 %[wp], %[abw]\n"
        "ds_read_b64 %[wc], %[abw] offset:%[pb]\n"
        "ds_read_b64 %[rnx], %[abw] offset:%[pb2]\n"
        // no bend: the new heights, and which of the two pieces left the tube
        "s_andn2_b64 exec, %[mlive], %[mb]\n"
        "v_cmp_ge_f64 %[mth], %[h2], %[r]\n"
        "v_cmp_le_f64 %[mtl], %[h1], -%[r]\n"
        "v_min_f64 %[hhi], %[h2], %[r]\n"
        "v_max_f64 %[hlo], %[h1], -%[r]\n"
        // bend: what it leaves behind
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "v_cndmask_b32_e64 %[code], 0, 1, %[mfv]\n"
        "v_add_u32 %[bit], %[wlo], %[at]\n"
        "v_subrev_u32 %[sh], %[csr], %[brk]\n"
        "v_cmp_gt_u32 %[m1], %[span], %[sh]\n"
        "v_cmp_ge_i32 %[m2], %[csr], %[at]\n"
        "v_cmp_ge_i32 %[m3], %[cer], %[at]\n"
        "v_lshl_or_b32 %[code], %[bit], 1, %[code]\n"
        "v_lshlrev_b32_e64 %[bit], %[sh], 1\n"
        "v_mov_b32 %[last], %[code]\n"
        "v_cndmask_b32_e64 %[mine], %[mine], %[code], %[m2]\n"
        "v_cndmask_b32_e64 %[next], %[next], %[code], %[m3]\n"
        "v_cndmask_b32_e64 %[sp], 0, %[bit], %[m1]\n"
        "s_and_b64 %[m1], %[m1], %[mfv]\n"
        "v_or_b32 %[ends], %[ends], %[sp]\n"
        "v_cmp_le_i32 vcc, %[cem1r], %[brk]\n"
        "v_cndmask_b32_e64 %[sp], 0, %[bit], %[m1]\n"
        "s_or_b64 %[mdone], %[mdone], vcc\n"
        "v_or_b32 %[types], %[types], %[sp]\n"
        // ... and the state right after it:  a = yn +- wp ; lo = a - wc ; hi = a + wc ; heights -+ wc
        "v_mov_b32 %[k0], %[brk]\n"
        "v_mov_b32 %[klo], %[at]\n"
        "v_mov_b32 %[khi], %[at]\n"
        "v_mov_b32 %[i], %[at]\n"
        "v_mov_b32 %[ai], %[aat]\n"
        "v_add_u32 %[awi], %[pbs], %[abw]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, %[mcv]\n"
        "v_add_f64 %[q], %[q], %[wp]\n"
        "s_mov_b64 exec, %[mfv]\n"
        "v_add_f64 %[q], %[q], -%[wp]\n"
        "s_and_b64 exec, %[mlive], %[mb]\n"
        "v_add_f64 %[lo], %[q], -%[wc]\n"
        "v_add_f64 %[hi], %[q], %[wc]\n"
        "v_mov_b64 %[hhi], %[wc]\n"
        "v_add_f64 %[hlo], 0, -%[wc]\n"
        // no bend: pull the pieces back inside the tube where they left it (mth / mtl hold no bending lane)
        "s_mov_b64 exec, %[mth]\n"
        "v_add_f64 %[t0], %[r], -%[h2]\n"
        "v_mul_f64 %[t0], %[t0], %[inv]\n"
        "v_mov_b32 %[khi], %[i]\n"
        "v_add_f64 %[hi], %[hi], %[t0]\n"
        "s_mov_b64 exec, %[mtl]\n"
        "v_add_f64 %[t0], -%[r], -%[h1]\n"
        "v_mul_f64 %[t0], %[t0], %[inv]\n"
        "v_mov_b32 %[klo], %[i]\n"
        "v_add_f64 %[lo], %[lo], %[t0]\n"
        // every live lane steps to its next sample
        "s_mov_b64 exec, %[mlive]\n"
        "v_add_u32 %[i], 1, %[i]\n"
        "v_add_u32 %[ai], %[pbs], %[ai]\n"
        "v_add_u32 %[awi], %[pbs], %[awi]\n"
        "v_mov_b64 %[yi], %[ynx]\n"
        "v_mov_b64 %[r], %[rnx]\n"
        "v_sub_u32 %[sp], %[i], %[k0]\n"               // the span of the next trip: its table index, and where the table ends the
        "v_cmp_gt_i32 vcc, %[limr], %[i]\n"            // lane leaves the loop (robust instantiations; never true in the plain ones)
        "v_cmp_gt_u32 %[m1], %[tsz], %[sp]\n"
        "s_and_b64 vcc, vcc, %[m1]\n"
        "s_andn2_b64 vcc, vcc, %[mdone]\n"
        "s_and_b64 exec, %[mlive], vcc\n"
        "s_mov_b64 %[mlive], exec\n"
        "s_cbranch_execnz .Lptv_walkwt_%=\n"
        "s_mov_b64 exec, %[msave]\n"
        "s_nop 1\n"
        "v_cndmask_b32_e64 %[doneflag], 0, 1, %[mdone]\n"
        : [lo] "+v"(lo), [hi] "+v"(hi), [hlo] "+v"(hlo), [hhi] "+v"(hhi), [yi] "+v"(yi), [r] "+v"(r), [i] "+v"(i), [k0] "+v"(k0),
          [klo] "+v"(klo), [khi] "+v"(khi), [ai] "+v"(ai), [awi] "+v"(awi), [ends] "+v"(ends), [types] "+v"(types),
          [mine] "+v"(mine), [next] "+v"(next), [last] "+v"(last), [doneflag] "+v"(doneflag), [sp] "+v"(sp),
          [ynx] "=&v"(ynx), [rnx] "=&v"(rnx), [wp] "=&v"(wp), [wc] "=&v"(wc), [t0] "=&v"(t0), [h1] "=&v"(h1), [h2] "=&v"(h2),
          [inv] "=&v"(inv), [q] "=&v"(q), [atab] "=&v"(atab), [brk] "=&v"(brk), [at] "=&v"(at), [aat] "=&v"(aat),
          [abw] "=&v"(abw), [code] "=&v"(code), [sh] "=&v"(sh), [bit] "=&v"(bit),
          [msave] "=&s"(msave), [mlive] "=&s"(mlive), [mcv] "=&s"(mcv), [mfv] "=&s"(mfv), [mb] "=&s"(mb), [mth] "=&s"(mth),
          [mtl] "=&s"(mtl), [mdone] "=&s"(mdone), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
        : [pbs] "s"(pbs), [limr] "s"(lim_r), [csr] "s"(cs_r), [cer] "s"(ce_r), [cem1r] "s"(cem1_r), [span] "s"(span),
          [wlo] "s"(wlo), [wd] "s"(wdelta), [rtab] "s"(rtab), [tsz] "s"(tsz), [abase] "v"(abase), [pb] "n"(PB), [pb2] "n"(2 * PB)
        : "vcc", "scc", "memory");
    w.lo = lo;
    w.hi = hi;
    w.hlo = hlo;
    w.hhi = hhi;
    w.i = i + wlo;
    w.k0 = k0 + wlo;
    w.klo = klo + wlo;
    w.khi = khi + wlo;
    rec.ends = ends;
    rec.types = types;
    rec.mine = mine;
    rec.next = next;
    rec.last = last;
    rec.done = doneflag != 0;
}

}  // namespace ptv
