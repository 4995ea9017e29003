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
    // A copy is the transposition of `src` under ONE slab geometry: it answers only a sweep that asks for the same
    // (stride, fibre length, slab count) -- another dimension of the same array is another copy.
    struct Shape {
        long inc = 0, slabs = 0;
        int len = 0;
        bool operator==(const Shape &o) const { return inc == o.inc && slabs == o.slabs && len == o.len; }
    };
    struct Entry {
        const double *src = nullptr;
        Shape shape;
        std::unique_ptr<Scratch> copy;
    };
    bool active = false;
    std::vector<Entry> entries;
    Scratch *find(const double *src, const Shape &shape);
    void remember(const double *src, const Shape &shape, std::unique_ptr<Scratch> copy);
    void forget(const double *p);   // every copy of p, whatever its shape: p is about to be written
    void clear() { entries.clear(); }
};
TransposeCache &transpose_cache();   // per host thread and per device, like the scratch pool

// Contract of a scope: between its construction and destruction every array a strided sweep reads is written by
// launch_sweep() only (which forgets the copies of what it writes) -- or the writer calls transpose_cache().forget().
struct TransposeScope {
    TransposeScope() : cache_(transpose_cache()) {
        cache_.clear();
        cache_.active = true;
    }
    ~TransposeScope() {   // (holds the cache it opened: nothing here can throw)
        cache_.clear();
        cache_.active = false;
    }
    TransposeScope(const TransposeScope &) = delete;
    TransposeScope &operator=(const TransposeScope &) = delete;

  private:
    TransposeCache &cache_;
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
    TransposeCache::Shape shape(int len) const { return TransposeCache::Shape{g_.inc, slabs_, len}; }
    SweepArgs orig_, t_;
    FibreGeom g_;
    hipStream_t s_;
    unsigned out_mask_;
    long slabs_;
    size_t bytes_;
    std::unique_ptr<Scratch> ia_, ib_, ic_, iw_, o0_, o1_;
};

}  // namespace ptv
