// pin.hpp -- the pinning solver as a sweep: exact 1-D TV-L1 prox of every fibre, data-parallel inside the fibre
// (pincore.hpp has the algorithm; pin.hip the kernel).  The top rung of the geometry ladder for data with long pieces.
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"
#include "ops.hpp"

namespace ptv {

// Longest fibre the kernel holds in LDS (running sums of one fibre per lane group).
constexpr int kPinMaxLen = 16384;
constexpr int kPinMaxLenWeighted = 8192;   // (two planes: sums and penalties)

// Does the pinning solver take this sweep?  (lam < 0 -- tvgen lets negative penalties through -- stays with the walker,
// whose behaviour there mirrors the reference's; so does lam = 0, where the walker returns the input bit for bit and
// this solver returns it through its running sums.)
// Longer fibres are spread over a grid of workgroups that must all be resident at once (pinlong.hip, 4096 samples per
// workgroup).  How many the CURRENT device holds is asked of the runtime (compute units x resident workgroups per unit: a
// full MI355X gives 1024 / 512, a CPX / DPX partition an eighth / a half of that), not assumed.
constexpr int kPinLongBlock = 4096;
long pin_long_capacity(bool weighted);   // pinlong.hip; cached per device

// Levels a pinning kernel runs before it gives a fibre up.  Anything from white noise to one flat piece needs 12-24; data
// with exact periodic ties (a regular zigzag: every sample a bend) peel one knot per segment end and level -- n / 2 levels.
// Past the cap the fibre is handed to a walker (pin.hip: a gated sequential sweep; pinlong.hip: the caller's next rung),
// so a sweep on this rung costs at most kPinMaxLevels levels plus one sequential walk, whatever the data.
constexpr int kPinMaxLevels = 64;

inline bool pin_is_long(bool weighted, const FibreGeom &g) { return g.len > (weighted ? kPinMaxLenWeighted : kPinMaxLen); }

inline bool pin_supports(OpId op, bool weighted, const FibreGeom &g, double lam) {
    if (g.len < 2 || g.count < 1) return false;
    if (pin_is_long(weighted, g)) {
        const long wgs = ((long)g.len + kPinLongBlock - 1) / kPinLongBlock * g.count;
        if (wgs > pin_long_capacity(weighted)) return false;
        if (weighted && op != OP_PROX && op != OP_DR_COL && op != OP_DR_COL_FINAL && op != OP_DR_COL_V) return false;   // (the ops pinlong.hip builds weighted)
    }
    if (weighted)
        return op == OP_PROX || op == OP_DR_COL || op == OP_DR_COL_FINAL || op == OP_DR_ROW || op == OP_DRW_ROW_FINAL || op == OP_DR_COL_V ||
               op == OP_DR_ROW_V;
    return op != OP_DRW_ROW_FINAL && lam > 0.0;
}

// One sweep.  Strided fibres (g.inc > 1) go through transposed copies of the operands, like launch_row_along in sweep.hip.
// pieces (device, may be null): the number of pieces of the sweep's result is added to it -- the geometry policy's hint
// for whether the chunk kernels below this rung are worth a trial.
// Returns true when the sweep is done (fibres that hit the level cap included: launch_seq_gated finishes them on the
// stream), false when NOTHING was written and the caller has to run another rung: the grid-wide variant found that its
// instantiation does not fit the device after all, or hit the level cap.
// seeds: start the levels from the knots known a priori (pincore.hpp: PinLane::seed): bit 0 = jumps above 4 lambda -- finding them costs
// about a third of a level, so the caller leaves it out where the input has none --, bit 1 = the deepest knots of windows of 4 / 16 / 64
// knots (unweighted fibres of up to 4096 samples; a third of a level to a level's work -- the stages gate themselves wave by wave -- for
// five or six levels less on noise at lambda ~ 1).  The one-workgroup kernels only; the grid-wide variant starts from the fibre ends.
bool launch_pin(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *pieces = nullptr, int seeds = 3);
// (the grid-wide variant behind it, for fibres beyond kPinMaxLen: pinlong.hip)
bool launch_pin_long(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *pieces);

// sweep.hip: the sequential walker on the fibres j with flags[j] != 0 (it clears the flags it consumes); contiguous or strided
void launch_seq_gated(OpId op, bool weighted, const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *flags);

}  // namespace ptv
