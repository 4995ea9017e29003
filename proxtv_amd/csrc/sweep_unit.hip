// sweep_unit.hip -- the sweep kernels of ONE (op, weighted) pair: compiled seventeen times by proxtv_amd/build.py with
// -DPTV_UNIT_OP=<OpId enumerator> -DPTV_UNIT_W=<true|false> (the pairs of PTV_SWEEP_UNITS in sweep_kernels.hpp).  Every pair
// instantiates its own kernels -- the op is a template parameter of all of them -- so the units share no device code and build in
// parallel; sweep.hip holds the policy state they share (chunk_state()) and dispatches to them.
#include "sweep_kernels.hpp"

#if !defined(PTV_UNIT_OP) || !defined(PTV_UNIT_W)
#error "compile with -DPTV_UNIT_OP=OP_... -DPTV_UNIT_W=true|false (see proxtv_amd/build.py)"
#endif

namespace ptv {
namespace swp {

template <>
void unit_launch<PTV_UNIT_OP, PTV_UNIT_W>(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, bool allow_chunked, int fam) {
    launch_op_w<PTV_UNIT_OP, PTV_UNIT_W>(args, g, stream, allow_chunked, fam);
    if (options().certify) launch_certify<PTV_UNIT_OP, PTV_UNIT_W>(args, g, stream);
}

// the mop-up of a kernel that gave some fibres up (pin.hip's level cap): the sequential walk of the flagged fibres only
template <>
void unit_gated<PTV_UNIT_OP, PTV_UNIT_W>(const SweepArgs &args, const FibreGeom &g, hipStream_t stream, int *flags) {
    launch_seq<PTV_UNIT_OP, PTV_UNIT_W>(args, g, stream, true, flags);
}

// the certifier alone (kernel 4): how many fibres of what a sweep of this op wrote fail the optimality conditions (-1: cannot be checked);
// the failing fibres stay flagged for nobody -- the flags are cleared by the next certified sweep's re-solve or overwritten
template <>
long unit_certify<PTV_UNIT_OP, PTV_UNIT_W>(const SweepArgs &args, const FibreGeom &g, hipStream_t stream) {
    int *flags = nullptr;
    const long failed = certify_count<PTV_UNIT_OP, PTV_UNIT_W>(args, g, stream, &flags);
    if (failed > 0) PTV_HIP(hipMemsetAsync(flags, 0, sizeof(int) * (size_t)g.count, stream));
    return failed;
}

// first use of a device: upload this unit's code object at initialisation, not in the first solve (common.hpp: warm_sweep)
template <>
void unit_warm<PTV_UNIT_OP, PTV_UNIT_W>() {
    hipFuncAttributes attr;
    PTV_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>((sweep_seq_kernel<PTV_UNIT_OP, PTV_UNIT_W, false>))));
}

}  // namespace swp
}  // namespace ptv
