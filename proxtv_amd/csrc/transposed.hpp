// transposed.hpp -- a strided sweep run as a dimension-0 sweep on transposed copies of its operands.
//
// The kernels that work inside a fibre (sweep_along_kernel, sweep_pin_kernel) want fibres contiguous.  For a sweep along
// another dimension the operands are transposed slab by slab (pointwise.hip's slab_transpose: fibre j = slab * inc + off
// sits at j * len afterwards), the sweep runs on the copies, and its outputs are transposed back.  Inside a
// TransposeScope -- a solver whose arrays are written by sweeps only, i.e. the DR loop -- the copies are remembered by
// source pointer: the image U never changes, and the t a row sweep reads is the t the previous row sweep wrote, whose
// transposed form it still has; two of a DR row sweep's four transpositions go away.  launch_sweep() forgets the copy
// of every array a sweep is about to write.
#pragma once

#include <memory>
#include <vector>

#include "common.hpp"
#include "ops.hpp"

namespace ptv {

struct TransposeCache {
    struct Entry {
        const double *src = nullptr;
        std::unique_ptr<Scratch> copy;
    };
    bool active = false;
    std::vector<Entry> entries;
    Scratch *find(const double *src);
    void remember(const double *src, std::unique_ptr<Scratch> copy);
    void forget(const double *p);
    void clear() { entries.clear(); }
};
TransposeCache &transpose_cache();   // per host thread and per device, like the scratch pool

struct TransposeScope {
    TransposeScope() {
        transpose_cache().clear();
        transpose_cache().active = true;
    }
    ~TransposeScope() {
        transpose_cache().clear();
        transpose_cache().active = false;
    }
    TransposeScope(const TransposeScope &) = delete;
    TransposeScope &operator=(const TransposeScope &) = delete;
};

class TransposedOperands {
  public:
    // in_mask: arrays the op reads (bit 0 a, 1 b, 2 c); out_mask: arrays it writes (bit 0 o0, 1 o1); args.w (per-edge
    // penalties, len - 1 along the fibre), when set, is transposed as well
    TransposedOperands(const SweepArgs &args, unsigned in_mask, unsigned out_mask, const FibreGeom &g, hipStream_t s);
    const SweepArgs &args() const { return t_; }                       // the same sweep on the copies ...
    FibreGeom geom() const { return FibreGeom{1, g_.len, g_.count}; }   // ... whose fibres are contiguous
    void finish();                                                      // outputs back to where the caller wants them

  private:
    const double *input(const double *src, std::unique_ptr<Scratch> &own, int len);
    SweepArgs orig_, t_;
    FibreGeom g_;
    hipStream_t s_;
    unsigned out_mask_;
    long slabs_;
    size_t bytes_;
    std::unique_ptr<Scratch> ia_, ib_, ic_, iw_, o0_, o1_;
};

}  // namespace ptv
