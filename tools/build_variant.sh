# A/B builds: compile sweep.hip with extra flags and link it with the other (already built) objects into
# proxtv_amd/build/lib_<name>.so ; run with  PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=proxtv_amd/build/lib_<name>.so
#   bash tools/build_variant.sh <name> [extra hipcc flags...]      (-DPTV_FAST_BUILD [-DPTV_FAST_WEIGHTED]: a third of the compile time)
name=$1; shift
cd $(dirname $0)/..
B=proxtv_amd/build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-gpu-rdc -Wno-unused-function "$@" -c proxtv_amd/csrc/sweep.hip -o $B/sweep_$name.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/lib_$name.so $B/common.o $B/sweep_$name.o $B/pin.o $B/pinlong.o $B/pointwise.o $B/tv2.o $B/solvers.o $B/cabi.o && echo $B/lib_$name.so
