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
s2)   # new defaults (32-fibre tiles on rungs 0 and 1, table walk, non-temporal streams, a-priori pins): the whole suite, then A/B
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.log 2>&1; tail -3 $OUT/pytest_default.log
  ab --reps 7 --rounds 2 --cases c2,c2@0.3,c2@0.5,c3,pd2,c4,c4y base t0,tile=0 > $OUT/ab_tiles.txt 2>&1; cat $OUT/ab_tiles.txt
  ab --reps 5 --rounds 2 --cases c2@0.7,c2@0.8,c2@1.0,c2@3.0 base noseed,pin_seed=0 > $OUT/ab_pins.txt 2>&1; cat $OUT/ab_pins.txt
  ab --reps 7 --rounds 2 --cases c2,c2@0.5,c3 base ntall=$W/lib_ntall.so nont=$W/lib_nont.so > $OUT/ab_nt.txt 2>&1; cat $OUT/ab_nt.txt
  ab --reps 7 --rounds 2 --cases c2 base keep4=$W/lib_keep4.so keep8=$W/lib_keep8.so keep16=$W/lib_keep.so > $OUT/ab_keep.txt 2>&1; cat $OUT/ab_keep.txt
  QUICK=1 bash tools/collect_profiles.sh r04q > $OUT/profiles_quick.log 2>&1; tail -4 $OUT/profiles_quick.log
  ;;
s3)   # rebuild in branch form, one-segment along-fibre instantiations, overlapped transpositions, seeds gated: suite, then A/B by knobs
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.log 2>&1; tail -3 $OUT/pytest_default.log
  ab --reps 7 --rounds 2 --cases c2,c2@0.3,c2@0.5,c3,pd2,c4,c4y,s1024,s512,prox0,prox1 base > $OUT/ab_base.txt 2>&1; cat $OUT/ab_base.txt
  ab --reps 5 --rounds 2 --cases c2@0.8,c2@1.0,c2@3.0 base nooverlap,pin_overlap=0 noseed,pin_seed=0 > $OUT/ab_pins.txt 2>&1; cat $OUT/ab_pins.txt
  ab --reps 5 --rounds 1 --cases c2@0.5,c2@0.6,c2@0.7 base rung3,chunk_mode=3 > $OUT/ab_rung3.txt 2>&1; cat $OUT/ab_rung3.txt
  ab --reps 5 --rounds 1 --cases c2@0.2,c2@0.3,c4 base rung0,chunk_mode=0 > $OUT/ab_rung0.txt 2>&1; cat $OUT/ab_rung0.txt
  ;;
s4)   # occupancy / unroll variants of the two hot kernels; rung-1 rows on the robust 32-fibre tile instead of transposed copies
  ab --reps 7 --rounds 2 --cases c2,prox0,prox1 base aw3=$W/lib_aw3.so au2=$W/lib_au2.so au8=$W/lib_au8.so tu2=$W/lib_tu2.so tu4=$W/lib_tu4.so > $OUT/ab_variants.txt 2>&1; cat $OUT/ab_variants.txt
  ab --reps 5 --rounds 2 --cases c2@0.6,c2@0.65,c2@0.7,c2@0.75 base tilerows,seed_row_along_e4=1 > $OUT/ab_tilerows.txt 2>&1; cat $OUT/ab_tilerows.txt
  ;;
s5)   # tile unroll 4, rows of rung 1 on the tile, the long-fibre TV-L2 solver: suite, then the survey of cases and the lambda sweep
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.log 2>&1; tail -3 $OUT/pytest_default.log
  timeout 300 python tools/time_cases.py > $OUT/time_cases.txt 2>&1; cat $OUT/time_cases.txt
  timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.1,0.2,0.3,0.4,0.5,0.6,0.65,0.7,0.75,0.8,1.0,3.0,10.0,30.0 > $OUT/lambda_sweep.txt 2>&1; cat $OUT/lambda_sweep.txt
  python - > $OUT/tv2_long.txt 2>&1 <<'PY'
import time, numpy as np, proxtv_amd
rng = np.random.default_rng(0)
for n in (100_000, 1_000_000, 4_000_000):
    y = np.cumsum(rng.standard_normal(n)) * 0.05 + rng.standard_normal(n)
    for lam in (1.0, 30.0):
        proxtv_amd.tv2_1d(y, lam)
        t0 = time.perf_counter(); proxtv_amd.tv2_1d(y, lam); dt = time.perf_counter() - t0
        print(f"tv2_1d n={n} lambda={lam}: {dt*1e3:.2f} ms (host arrays in and out)")
PY
  cat $OUT/tv2_long.txt
  ;;
s6)   # rung 1 near its upper end: blocks per workgroup of the robust tile (fewer links across workgroups), second-chance rounds
  timeout 300 python -m pytest tests/test_gpu_p2.py -m gpu -x -q > $OUT/pytest_p2.log 2>&1; tail -2 $OUT/pytest_p2.log
  ab --reps 5 --rounds 2 --cases c2@0.5,c2@0.6,c2@0.65,c2@0.7 base bpw2,blocks_per_wg=2 bpw4,blocks_per_wg=4 rounds8,rounds=8 rounds2,rounds=2 > $OUT/ab_rung1.txt 2>&1; cat $OUT/ab_rung1.txt
  ;;
s7)   # soak: the randomised differential test on the round-4 build (tiles, a-priori pins and the form of the DR iteration drawn at random)
  { echo "# python tools/fuzz.py <seconds> <seed> [nd|long] on one MI355X box; assertion: relative error <= 1e-9"
    python tools/fuzz.py 150 31; python tools/fuzz.py 60 32 nd; python tools/fuzz.py 90 33 long; } > $OUT/fuzz_soak.txt 2>&1; cat $OUT/fuzz_soak.txt
  ;;
s8)   # after the fix of the weighted pitch-65 tile's table: the new test, the soak again (same seeds, then fresh ones)
  timeout 600 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_2d.log 2>&1; tail -2 $OUT/pytest_2d.log
  { echo "# python tools/fuzz.py <seconds> <seed> [nd|long] on one MI355X box; assertion: relative error <= 1e-9"
    python tools/fuzz.py 150 31; python tools/fuzz.py 100 41; python tools/fuzz.py 45 42 nd; python tools/fuzz.py 60 43 long; } > $OUT/fuzz_soak.txt 2>&1; cat $OUT/fuzz_soak.txt
  ;;
s9)   # the correction of PD2 / Yang sweeps kept in registers for half a thread's rows; the robust tile searching its whole zone for known bends
  ab --reps 7 --rounds 2 --cases pd2,c4y,yang2 base keepops=$W/lib_keepops.so > $OUT/ab_keepops.txt 2>&1; cat $OUT/ab_keepops.txt
  ab --reps 5 --rounds 2 --cases c2@0.4,c2@0.5,c2@0.6,c2@0.65,c2@0.7 base look14=$W/lib_look14.so > $OUT/ab_look14.txt 2>&1; cat $OUT/ab_look14.txt
  ;;
s10)  # the last defaults (correction kept for PD2 / Yang, whole-zone search in the robust tile): suite default + rung 1, then profiles of the final build
  timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  FILES="tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py tests/test_gpu_pin.py tests/test_gpu_boundary.py"
  PROXTV_CHUNK_MODE=1 timeout 900 python -m pytest $FILES -m gpu -q > $OUT/pytest_mode1.log 2>&1; echo "pinned to rung 1: $(tail -1 $OUT/pytest_mode1.log)" | tee -a $OUT/summary.txt
  python tools/fuzz.py 60 51 > $OUT/fuzz.txt 2>&1; python tools/fuzz.py 30 52 nd >> $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" | tee -a $OUT/summary.txt
  bash tools/collect_profiles.sh r04 > $OUT/profiles.log 2>&1; tail -3 $OUT/profiles.log
  ;;
s11)  # the repair kernel jumping from failure to failure: parity where it works hardest (rungs 0 / 1 / 2 / 4 pinned, fuzz), then the upper end of rung 1
  FILES="tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py"
  timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  for m in 0 1 2 4; do PROXTV_CHUNK_MODE=$m timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_mode$m.log 2>&1; echo "pinned to rung $m: $(tail -1 $OUT/pytest_mode$m.log)" | tee -a $OUT/summary.txt; done
  python tools/fuzz.py 90 61 > $OUT/fuzz.txt 2>&1; python tools/fuzz.py 60 62 long >> $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ab --reps 5 --rounds 2 --cases c2,c2@0.4,c2@0.5,c2@0.6,c2@0.65,c2@0.7 base > $OUT/ab_repair.txt 2>&1; cat $OUT/ab_repair.txt
  ;;
s12)  # the final build (repair kernel that jumps): the suite once more, then the whole profile collection keyed to this build
  timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" | tee -a $OUT/summary.txt
  bash tools/collect_profiles.sh r04 > $OUT/profiles.log 2>&1; tail -3 $OUT/profiles.log
  ;;
s13)  # s' kept in registers for half the rows of a DR row sweep: suite (default + rung 1 + the 64-fibre tile), the seed check, profiles of this build
  timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  FILES="tests/test_gpu_parity_2d.py tests/test_gpu_fuzz.py tests/test_gpu_boundary.py"
  PROXTV_CHUNK_MODE=1 timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_mode1.log 2>&1; echo "pinned to rung 1 ($FILES): $(tail -1 $OUT/pytest_mode1.log)" | tee -a $OUT/summary.txt
  PROXTV_TILE=0 timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_tile0.log 2>&1; echo "tile=0 ($FILES): $(tail -1 $OUT/pytest_tile0.log)" | tee -a $OUT/summary.txt
  python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" | tee -a $OUT/summary.txt
  QUICK=1 bash tools/collect_profiles.sh r04k > $OUT/profiles.log 2>&1; tail -3 $OUT/profiles.log
  timeout 300 python tools/seed_check.py > $OUT/seed_check.txt 2>&1; tail -40 $OUT/seed_check.txt
  ;;
s14)  # soak of the last build
  { echo "# python tools/fuzz.py <seconds> <seed> [nd|long] on one MI355X box, build $(python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print(build.build_id())"); assertion: relative error <= 1e-9"
    python tools/fuzz.py 150 71; python tools/fuzz.py 70 72 long; python tools/fuzz.py 40 73 nd; } > $OUT/fuzz_soak.txt 2>&1; grep -v amdgpu $OUT/fuzz_soak.txt
  ;;
*) echo "unknown session $S"; exit 2;;
esac
