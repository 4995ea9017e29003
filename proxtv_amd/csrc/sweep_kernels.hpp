// sweep_kernels.hpp -- the fibre-sweep kernels and their launch plumbing (templates on the op and on weighted / unweighted).
//
// Included by sweep.hip (policy state, dispatch, the entry points of sweep.hpp) and by sweep_unit.hip, which is compiled once per
// (op, weighted) pair: every pair instantiates its own set of kernels, so the pairs build in parallel (one translation unit took
// 4.5 minutes; seventeen take about one on eight cores) and a variant build touches only what it changes.
#pragma once

#include "sweep.hpp"

#include <cstddef>
#include <memory>

#include "chunkcore.hpp"
#include "walk_asm.hpp"

#ifndef PTV_TILE_UNROLL
#define PTV_TILE_UNROLL 4   // rows of the rebuild passes in flight together in the tile kernels (round 4, with the rebuild in branch form: DR row
                            // sweep 114.9 -> 112.8 us, plain row sweep 83.7 -> 81.7 against 1; 2 in between)
#endif
#include "pin.hpp"
#include "pointwise.hpp"
#include "policy.hpp"
#include "transposed.hpp"
#include "walker.hpp"

#include <cstdio>

// The kernels, one header each (round 6: this file used to hold all of them, 2 900 lines).  Everything lives in ptv::swp -- named, not
// anonymous: the per-op translation units of sweep_unit.hip share ChunkScratch and the declarations at the end of this file.
#include "sweep_links.hpp"           // link codes, the dirty word, the words across workgroups
#include "kernel_seq.hpp"            // kernel 1   one sequential walk per fibre
#include "sweep_window.hpp"          //            launch plan, LDS window, one chunk's walk
#include "kernel_tile.hpp"           // kernel 2   chunks over an LDS window: strided tiles
#include "kernel_along.hpp"          // kernel 2a  chunks along contiguous fibres; known runs
#include "kernel_short_global.hpp"   // kernels 1b / 2b  whole short fibres in LDS; chunks from global memory
#include "kernel_repair.hpp"         // kernels 3 / 3a  repair, sequential and one lane per failure
#include "kernel_certify.hpp"        // kernel 4   the optimality conditions behind a sweep (option certify)
#include "sweep_launch.hpp"          // host side: per-thread state, the geometry ladder, the launchers

namespace ptv {

// One translation unit per (op, weighted) pair defines these two (sweep_unit.hip); sweep.hip dispatches to them.
namespace swp {
template <int OP, bool WEIGHTED> void unit_launch(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool allow_chunked, int fam);
template <int OP, bool WEIGHTED> void unit_gated(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *flags);
template <int OP, bool WEIGHTED> void unit_warm();
template <int OP, bool WEIGHTED> long unit_certify(const SweepArgs &args, const FibreGeom &g, hipStream_t stream);
// (op, weighted) pairs that exist: X(op, weighted)
#define PTV_SWEEP_UNITS(X)                                                                                                   \
    X(OP_PROX, false) X(OP_PROX, true) X(OP_DR_COL, false) X(OP_DR_COL, true) X(OP_DR_COL_FINAL, false) X(OP_DR_COL_FINAL, true) \
    X(OP_DR_ROW, false) X(OP_DR_ROW, true) X(OP_DR_ROW_FINAL, false) X(OP_DRW_ROW_FINAL, true) X(OP_PD2_A, false)            \
    X(OP_PD2_B, false) X(OP_YANG, false) X(OP_DR_COL_V, false) X(OP_DR_COL_V, true) X(OP_DR_ROW_V, false) X(OP_DR_ROW_V, true)
#define PTV_DECLARE_UNIT(ID, W)                                                                                                         \
    template <> void unit_launch<ID, W>(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool allow_chunked, int fam); \
    template <> void unit_gated<ID, W>(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *flags);           \
    template <> void unit_warm<ID, W>();                                                                                     \
    template <> long unit_certify<ID, W>(const SweepArgs &args, const FibreGeom &g, hipStream_t stream);
PTV_SWEEP_UNITS(PTV_DECLARE_UNIT)
#undef PTV_DECLARE_UNIT
}  // namespace swp

}  // namespace ptv
