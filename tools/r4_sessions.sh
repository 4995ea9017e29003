#!/bin/bash
# Round-4 GPU sessions (one gpurun call each; everything lands under gpurun_out/r4/<session>/):
#     gpurun --timeout N -- 'bash tools/r4_sessions.sh <session>'
# Variant builds (tools/build_variant.sh) must exist in proxtv_amd/build/ before the call: they travel with the snapshot.
S=$1
OUT=gpurun_out/r4/$S
mkdir -p $OUT
export TMPDIR=/tmp
W=proxtv_amd/build
ab() { python tools/ab_run.py "$@"; }
alt() { PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/$W/lib_$1.so "${@:2}"; }
case $S in
s1)   # the 32-fibre tile, the table walk, non-temporal streams: parity first, then the A/B matrix, then the phase trace
  timeout 600 python -m pytest tests -m gpu -x -q -k "tile_geometries or golden_dr2 or golden_pd2 or golden_yang2 or both_forms or c4_volume or c2_full" > $OUT/parity_new_tile.log 2>&1; tail -3 $OUT/parity_new_tile.log
  PROXTV_VERBOSE=1 python tools/time_one.py c2 2>&1 | grep -i "tile kernel\|c2 lambda" > $OUT/occupancy.log; cat $OUT/occupancy.log
  ab --reps 7 --rounds 2 --cases c2,c2@0.3,c2@0.5,c3,pd2,c4,c4y,prox0,prox1,s1024,s512 base t0,tile=0 walktab=$W/lib_walktab.so walktab_t0=$W/lib_walktab.so,tile=0 form2,dr_form=2 t0_form2,tile=0,dr_form=2 walktab_form2=$W/lib_walktab.so,dr_form=2 > $OUT/ab_matrix.txt 2>&1; cat $OUT/ab_matrix.txt
  ab --reps 7 --rounds 2 --cases c2,prox1 nt=$W/lib_nt.so nt_t0=$W/lib_nt.so,tile=0 > $OUT/ab_nt.txt 2>&1; cat $OUT/ab_nt.txt
  alt walktab timeout 900 python -m pytest tests -m gpu -x -q tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py tests/test_gpu_chunk_repair.py > $OUT/parity_walktab.log 2>&1; tail -3 $OUT/parity_walktab.log
  python tools/wg_trace.py 0.1 > $OUT/wg_trace_tile32.txt 2>&1
  PROXTV_TILE=0 python tools/wg_trace.py 0.1 > $OUT/wg_trace_tile64.txt 2>&1
  grep "^##\|^# mean" $OUT/wg_trace_tile32.txt $OUT/wg_trace_tile64.txt
  ;;
*) echo "unknown session $S"; exit 2;;
esac
