#!/bin/bash
# GPU sessions of the current round (one gpurun call each; everything lands under gpurun_out/r5/<session>/):
#     gpurun --timeout N -- 'bash tools/gpu_sessions.sh <session>'
# Variant builds (python -m proxtv_amd.build --variant NAME -- flags) must exist in proxtv_amd/build/ before the call: they travel
# with the snapshot.
S=$1
OUT=gpurun_out/r5/$S
mkdir -p $OUT
export TMPDIR=/tmp
W=proxtv_amd/build
ab() { python tools/ab_run.py "$@"; }
alt() { PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/$W/lib_$1.so "${@:2}"; }
case $S in
s1)   # the jobs repair merged as the default, the build split into units: the whole suite, the suite with the jobs kernel always on,
      # a short soak, the A/B of the option, then counters for EVERY hot kernel (weighted, pinning, N-D combiners included)
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  FILES="tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py tests/test_gpu_boundary.py"
  PROXTV_REPAIR_JOBS=2 timeout 600 python -m pytest $FILES -m gpu -x -q > $OUT/pytest_jobs2.log 2>&1; echo "repair_jobs=2: $(tail -1 $OUT/pytest_jobs2.log)" | tee -a $OUT/summary.txt
  PROXTV_REPAIR_JOBS=2 PROXTV_CHUNK_MODE=1 timeout 600 python -m pytest $FILES -m gpu -x -q > $OUT/pytest_jobs2_mode1.log 2>&1; echo "repair_jobs=2, rung 1: $(tail -1 $OUT/pytest_jobs2_mode1.log)" | tee -a $OUT/summary.txt
  { python tools/fuzz.py 60 61; python tools/fuzz.py 25 62 nd; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ab --reps 5 --rounds 2 --cases c2,c2@0.6,c2@0.65,c2@0.7,c2@0.75,c3 base nojobs,repair_jobs=0 jobs2,repair_jobs=2 > $OUT/ab_jobs.txt 2>&1; cat $OUT/ab_jobs.txt
  cd /tmp; R=$GRAFT_REPO_ROOT
  timeout 900 python $R/tools/kernel_counters.py collect $R/$OUT/kc > $R/$OUT/kc_collect.log 2>&1
  cd $R
  python tools/kernel_counters.py report $OUT/kc > $OUT/kernel_counters.txt 2>&1; cat $OUT/kernel_counters.txt
  python tools/kernel_counters.py traffic $OUT/kc > $OUT/pmc_traffic.json 2>&1
  find $OUT/kc -name "*.db" -delete   # (the databases are tens of MB each; the table and the json are what is kept)
  python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" | tee -a $OUT/summary.txt
  ;;
s2)   # K1: wave-uniform fast paths (interior segments / inner blocks: nothing tested per element in staging and stream-out), the
      # rebuild that keeps the chunk in registers, the slow tail skipped when every lane is done -- parity first, then A/B against the
      # build of s1 (lib_r5a.so) on one box, then the instruction counters of the new kernels
  FILES="tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_chunk_repair.py tests/test_gpu_large.py tests/test_gpu_fuzz.py"
  timeout 900 python -m pytest $FILES -m gpu -x -q > $OUT/pytest_parity.log 2>&1; echo "parity files: $(tail -1 $OUT/pytest_parity.log)" | tee $OUT/summary.txt
  PROXTV_CHUNK_MODE=1 timeout 600 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_chunk_repair.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_mode1.log 2>&1; echo "rung 1: $(tail -1 $OUT/pytest_mode1.log)" | tee -a $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c2,c2@0.3,c2@0.5,c3,pd2,c4,c4y,prox0,prox1,wprox0,wprox1,s1024,s512 base old=$W/lib_r5a.so > $OUT/ab_k1.txt 2>&1; cat $OUT/ab_k1.txt
  cd /tmp; R=$GRAFT_REPO_ROOT
  KC_SETS=insts,waves timeout 600 python $R/tools/kernel_counters.py collect $R/$OUT/kc calib+dr0.1+prox0+prox1+c3 > $R/$OUT/kc_collect.log 2>&1
  cd $R
  python tools/kernel_counters.py report $OUT/kc > $OUT/kernel_counters.txt 2>&1; cat $OUT/kernel_counters.txt
  find $OUT/kc -name "*.db" -delete
  ;;
s3)   # occupancy variants of the along-fibre kernel now that it issues 18 % fewer instructions (chunks of 15 / 13 samples: five
      # workgroups per CU; weighted chunks of 9 / 11: the two LDS planes at 16 / 12 waves per CU; 1- and 2-wave workgroups), the phase
      # trace of the new build, the lambda sweep
  timeout 300 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  for v in c15 c13 w9 aw1; do alt $v timeout 300 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_chunk_repair.py -m gpu -x -q > $OUT/pytest_$v.log 2>&1; echo "$v: $(tail -1 $OUT/pytest_$v.log)" | tee -a $OUT/summary.txt; done
  ab --reps 7 --rounds 2 --cases c2,c2@0.5,prox0,c4,s1024 base c15=$W/lib_c15.so c13=$W/lib_c13.so aw1=$W/lib_aw1.so aw2=$W/lib_aw2.so > $OUT/ab_occupancy.txt 2>&1; cat $OUT/ab_occupancy.txt
  ab --reps 7 --rounds 2 --cases c3,wprox0,c3@5 base w9=$W/lib_w9.so w11=$W/lib_w11.so > $OUT/ab_weighted.txt 2>&1; cat $OUT/ab_weighted.txt
  timeout 60 python tools/wg_trace.py > $OUT/wg_trace.txt 2>&1; grep "^##\|^# mean" $OUT/wg_trace.txt
  timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.1,0.2,0.3,0.4,0.5,0.6,0.65,0.7,0.75,0.8,1.0,3.0,10.0,30.0 > $OUT/lambda_sweep.txt 2>&1; cat $OUT/lambda_sweep.txt
  ;;
s4)   # weighted chunks of 9 samples as the default (parity + fuzz first), then: 7 samples; the row sweep's s' kept for 12 / 16 of a thread's 16
      # rows now that the staging addresses live in scalar registers; rung 1 against rung 3 at lambda 0.75 - 1 with the jobs repair;
      # the phase trace with the number of waves in flight
  timeout 600 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_large.py tests/test_gpu_fuzz.py tests/test_gpu_chunk_repair.py -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  { python tools/fuzz.py 90 71; python tools/fuzz.py 30 72 nd; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c3,wprox0,c3@5 base w7=$W/lib_w7.so > $OUT/ab_w7.txt 2>&1; cat $OUT/ab_w7.txt
  ab --reps 7 --rounds 2 --cases c2,c2@0.5,pd2,c4y,yang2 base keep12=$W/lib_keep12.so keep16=$W/lib_keep16.so > $OUT/ab_keep.txt 2>&1; cat $OUT/ab_keep.txt
  ab --reps 5 --rounds 1 --cases c2@0.7,c2@0.75,c2@0.8,c2@0.9,c2@1.0 base rung1,chunk_mode=1 rung3,chunk_mode=3 > $OUT/ab_rungs.txt 2>&1; cat $OUT/ab_rungs.txt
  timeout 60 python tools/wg_trace.py > $OUT/wg_trace.txt 2>&1; grep "^##\|^# " $OUT/wg_trace.txt
  ;;
s5)   # the along-fibre kernel taking its segments in turns (as many workgroups as the device holds: no slot waits for the dispatcher),
      # s' kept for all 16 rows: parity, A/B by option on one build, the trace; rung / form thresholds re-checked with the new kernels
  timeout 600 python -m pytest tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_large.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c2,prox0,c2@0.2,c4,c4y,pd2,s1024,s512 base turns0,along_persist=0 > $OUT/ab_persist.txt 2>&1; cat $OUT/ab_persist.txt
  ab --reps 5 --rounds 1 --cases c2@0.25,c2@0.3,c2@0.35,c2@0.4 base rung0,chunk_mode=0 noisy35,seed_noisy_e4=3500 form0,dr_form=0 > $OUT/ab_rung0.txt 2>&1; cat $OUT/ab_rung0.txt
  timeout 60 python tools/wg_trace.py > $OUT/wg_trace.txt 2>&1; grep "^##\|^# " $OUT/wg_trace.txt
  ;;
s6)   # work queues: waves (along-fibre kernel) / workgroups (tiles) draw their segments / blocks from an atomic counter instead of
      # waiting for the dispatcher (static turns lost in s5: no rebalancing).  Parity, A/B by option on one build, the trace.
  timeout 600 python -m pytest tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_large.py tests/test_gpu_fuzz.py tests/test_gpu_chunk_repair.py -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c2,prox0,prox1,c2@0.5,c3,c4,c4y,pd2,s1024,s512 base noq,along_persist=0,tile_persist=0 alongq,tile_persist=0 tileq,along_persist=0 > $OUT/ab_queue.txt 2>&1; cat $OUT/ab_queue.txt
  timeout 60 python tools/wg_trace.py > $OUT/wg_trace.txt 2>&1; grep "^##\|^# " $OUT/wg_trace.txt
  ;;
s7)   # work queues again: 64 counters per launch in separate cache lines (one counter serialised 12 288 atomics: 217 us), the draw
      # issued behind the window loads
  timeout 600 python -m pytest tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_large.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c2,prox0,prox1,c2@0.5,c3,c4y,pd2,s1024 base noq,along_persist=0,tile_persist=0 alongq,tile_persist=0 tileq,along_persist=0 > $OUT/ab_queue.txt 2>&1; cat $OUT/ab_queue.txt
  timeout 60 python tools/wg_trace.py > $OUT/wg_trace.txt 2>&1; grep "^##\|^# " $OUT/wg_trace.txt
  ;;
s8)   # replay: the along-fibre kernel verifies the structure its previous sweep recorded instead of walking (from the fourth sweep of
      # a solve on; all or nothing per wave).  Parity first (the whole suite), then A/B by option, the share of waves that replayed
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  { python tools/fuzz.py 60 81; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c2,c2@0.05,c2@0.2,pd2,c4,s1024,s2048 base walk,replay=0 > $OUT/ab_replay.txt 2>&1; cat $OUT/ab_replay.txt
  python - > $OUT/replay_share.txt 2>&1 <<'PY'
import numpy as np, torch, sys
sys.path.insert(0, '.')
from proxtv_amd import _lib, device
lib = _lib.require_device()
lib.proxtv_set_option(b"why", 1)
x = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((4096, 4096))).cuda())
buf = np.zeros(8, dtype=np.uint32)
for lam in (0.05, 0.1, 0.15, 0.2):
    lib.proxtv_debug_why(buf.ctypes.data)
    for iters in (5, 10, 20, 35):
        device.tv1_2d(x, lam, max_iters=iters)
        lib.proxtv_debug_why(buf.ctypes.data)
        waves = 16384 * iters      # column sweeps: iterations 1 .. iters - 1 and the final one
        print(f"lambda {lam} {iters:2d} iterations: {int(buf[5]):7d} wavefronts replayed of {waves} launched in {iters} column sweeps = {buf[5] / waves:.3f}")
PY
  cat $OUT/replay_share.txt
  ;;
s9)   # replay, fused: the check rides on the rebuild (a pass of its own cost a wave more than the walk it saves: s8)
  timeout 600 python -m pytest tests/test_gpu_replay.py tests/test_gpu_parity_2d.py tests/test_gpu_large.py tests/test_gpu_fuzz.py tests/test_gpu_boundary.py -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c2,c2@0.05,c2@0.2,s2048 base walk,replay=0 > $OUT/ab_replay.txt 2>&1; cat $OUT/ab_replay.txt
  python - > $OUT/replay_share.txt 2>&1 <<'PY'
import numpy as np, torch, sys
sys.path.insert(0, '.')
from proxtv_amd import _lib, device
lib = _lib.require_device()
lib.proxtv_set_option(b"why", 1)
x = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((4096, 4096))).cuda())
buf = np.zeros(8, dtype=np.uint32)
for lam in (0.05, 0.1, 0.2):
    lib.proxtv_debug_why(buf.ctypes.data)
    for iters in (5, 10, 20, 35):
        device.tv1_2d(x, lam, max_iters=iters)
        lib.proxtv_debug_why(buf.ctypes.data)
        waves = 16384 * iters
        print(f"lambda {lam} {iters:2d} iterations: {int(buf[5]):7d} wavefronts replayed, {int(buf[6]):6d} tried and walked, of {waves} launched in {iters} column sweeps = {buf[5] / waves:.3f}")
PY
  cat $OUT/replay_share.txt
  ;;
s10)  # replay, fused, the hole at the segment's end closed: parity, A/B, and the instruction counters of the column sweep with and without
  timeout 600 python -m pytest tests/test_gpu_replay.py tests/test_gpu_parity_2d.py tests/test_gpu_large.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  python tools/replay_diag.py 2>&1 | grep -v amdgpu.ids | tee $OUT/replay_diag.txt
  ab --reps 7 --rounds 2 --cases c2,c2@0.05 base walk,replay=0 > $OUT/ab_replay.txt 2>&1; cat $OUT/ab_replay.txt
  cd /tmp; R=$GRAFT_REPO_ROOT
  KC_SETS=insts,waves timeout 300 python $R/tools/kernel_counters.py collect $R/$OUT/kc_replay dr0.1 > /dev/null 2>&1
  PROXTV_REPLAY=0 KC_SETS=insts,waves timeout 300 python $R/tools/kernel_counters.py collect $R/$OUT/kc_walk dr0.1 > /dev/null 2>&1
  cd $R
  python tools/kernel_counters.py report $OUT/kc_replay 2>&1 | grep "kernel \|sweep_along_kernel<1" | tee $OUT/counters_replay.txt
  python tools/kernel_counters.py report $OUT/kc_walk 2>&1 | grep "sweep_along_kernel<1" | tee $OUT/counters_walk.txt
  find $OUT -name "*.db" -delete
  ;;
s11)  # validation of the round's final build: the whole suite under the default policy, the parity / repair / fuzz / pinning / boundary
      # files pinned to rungs 1 and 3, on the 64-fibre tile, with replay on, with the jobs repair always on; the soak
  timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_default.log 2>&1; echo "default policy, whole suite: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  FILES="tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py tests/test_gpu_pin.py tests/test_gpu_boundary.py tests/test_gpu_replay.py"
  PROXTV_CHUNK_MODE=1 timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_mode1.log 2>&1; echo "pinned to rung 1: $(tail -1 $OUT/pytest_mode1.log)" | tee -a $OUT/summary.txt
  PROXTV_CHUNK_MODE=3 timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_mode3.log 2>&1; echo "pinned to rung 3: $(tail -1 $OUT/pytest_mode3.log)" | tee -a $OUT/summary.txt
  PROXTV_TILE=0 timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_tile0.log 2>&1; echo "64-fibre tile: $(tail -1 $OUT/pytest_tile0.log)" | tee -a $OUT/summary.txt
  PROXTV_REPLAY=1 timeout 600 python -m pytest $FILES tests/test_gpu_large.py -m gpu -q > $OUT/pytest_replay.log 2>&1; echo "replay on: $(tail -1 $OUT/pytest_replay.log)" | tee -a $OUT/summary.txt
  PROXTV_REPAIR_JOBS=2 timeout 600 python -m pytest tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_fuzz.py -m gpu -q > $OUT/pytest_jobs2.log 2>&1; echo "jobs repair always on: $(tail -1 $OUT/pytest_jobs2.log)" | tee -a $OUT/summary.txt
  { echo "# python tools/fuzz.py <seconds> <seed> [nd|long] on one MI355X box; assertion: relative error <= 1e-9"
    python tools/fuzz.py 100 91; PROXTV_REPLAY=1 python tools/fuzz.py 50 92; python tools/fuzz.py 30 93 nd; python tools/fuzz.py 40 94 long; } > $OUT/fuzz_soak.txt 2>&1; grep "^fuzz\|MISMATCH" $OUT/fuzz_soak.txt | tee -a $OUT/summary.txt
  python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" | tee -a $OUT/summary.txt
  ;;
s12)  # the profiles of the round's final build
  bash tools/collect_profiles.sh r05 > $OUT/collect.log 2>&1; tail -60 $OUT/collect.log
  ;;
s13)  # knobs re-checked against the round's kernels (the second form of the DR iteration on rung 0, blocks per tile workgroup), a longer soak
  ab --reps 7 --rounds 2 --cases c2,prox1,c3,pd2 base form2,dr_form=2 bpw2,blocks_per_wg=2 bpw4,blocks_per_wg=4 > $OUT/ab_knobs.txt 2>&1; cat $OUT/ab_knobs.txt
  { echo "# python tools/fuzz.py <seconds> <seed> [nd|long] on one MI355X box; assertion: relative error <= 1e-9"
    python tools/fuzz.py 120 101; python tools/fuzz.py 40 102 nd; python tools/fuzz.py 60 103 long; } > $OUT/fuzz_soak2.txt 2>&1; grep "^fuzz\|MISMATCH" $OUT/fuzz_soak2.txt
  ;;
s14)  # work queues once more, along-fibre kernel of one-operand sweeps only: 16 counters, a wave moves to the next queue when one runs dry
  timeout 600 python -m pytest tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_large.py tests/test_gpu_fuzz.py tests/test_gpu_replay.py -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c2,prox0,c2@0.2,c4,pd2,s1024,s2048 base noq,along_queue=0 > $OUT/ab_queue.txt 2>&1; cat $OUT/ab_queue.txt
  timeout 60 python tools/wg_trace.py > $OUT/wg_trace.txt 2>&1; grep "^## col\|^# rec\|^# mean" $OUT/wg_trace.txt | sed -n 4,8p
  ;;
s15)  # the remaining settings of the earlier rounds' validation matrix on the final build: the hill-climbing policy, the in-kernel link
      # check off, pinned to the sequential rung
  FILES="tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py tests/test_gpu_pin.py tests/test_gpu_boundary.py"
  PROXTV_DETERMINISTIC=0 timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_hillclimb.log 2>&1; echo "hill-climbing policy: $(tail -1 $OUT/pytest_hillclimb.log)" | tee $OUT/summary.txt
  PROXTV_XLINK=0 timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_xlink0.log 2>&1; echo "in-kernel link check off: $(tail -1 $OUT/pytest_xlink0.log)" | tee -a $OUT/summary.txt
  PROXTV_CHUNK_MODE=5 timeout 600 python -m pytest $FILES -m gpu -q > $OUT/pytest_mode5.log 2>&1; echo "pinned to rung 5: $(tail -1 $OUT/pytest_mode5.log)" | tee -a $OUT/summary.txt
  ;;
s16)  # the table of tools/seed_check.py with the round's kernels, one more soak (which found: pinned rung 0, PD2, lambda = 6 -- s17, s18)
  timeout 300 python tools/seed_check.py > $OUT/seed_check.txt 2>&1
  { python tools/fuzz.py 80 111; python tools/fuzz.py 30 112 long; } 2>&1 | grep "^fuzz\|MISMATCH" | tee $OUT/fuzz3.txt
  ;;
s17)  # the mismatch of s16 localised: which iteration of the Dykstra loop, which fibre (the case travels as an .npy next to the call)
  timeout 150 python tools/case_diag.py $CASE_NPY $CASE_LAMBDA $OUT > $OUT/diag.txt 2>&1; cat $OUT/diag.txt
  ;;
s18)  # ... and the fibre: the operand of the 1-D prox that case_diag.py saved, one column alone, the pinned rungs, the kernel options
  timeout 120 python tools/case_diag2.py $CASE_NPZ 797 $OUT > $OUT/diag2.txt 2>&1; cat $OUT/diag2.txt
  ;;
s19)  # the hand-over of a repair walk to an unproven chunk made right (chunkcore.hpp: the first piece of an unproven lane is summed from
      # its true first row; tiles: before the barrier behind the walks): the whole suite, the soak from the case that failed on, the
      # headline's HBM counters and kernel stats and bench line for the new build
  timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  grep -q " passed" $OUT/pytest_default.log && ! grep -q "failed" $OUT/pytest_default.log || { tail -40 $OUT/pytest_default.log; exit 1; }
  { timeout 100 python tools/fuzz.py 45 111 from 1256; timeout 60 python tools/fuzz.py 25 113; } 2>&1 | grep "^fuzz\|MISMATCH" | tee $OUT/fuzz.txt
  R=$GRAFT_REPO_ROOT
  (cd /tmp; KC_SETS=fetch,write timeout 300 python $R/tools/kernel_counters.py collect $R/$OUT/kc calib+dr0.1 > $R/$OUT/kc_collect.log 2>&1
   timeout 120 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_bench -o x -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c5 > $R/$OUT/bench_under_rocprof.log 2>&1)
  python tools/rocprof_summary.py $(find $OUT/prof_bench -name "x_results.db" | head -1) > $OUT/kernel_stats.txt 2>&1; head -12 $OUT/kernel_stats.txt
  python tools/kernel_counters.py report $OUT/kc > $OUT/kernel_counters.txt 2>&1
  python tools/kernel_counters.py traffic $OUT/kc > $OUT/pmc_traffic.json 2>&1; cat $OUT/pmc_traffic.json
  rm -rf $OUT/prof_bench; find $OUT/kc -name "*.db" -delete
  cp $OUT/pmc_traffic.json profiles/r05_pmc_traffic.json   # (bench.py reads the traffic of the build it runs)
  timeout 300 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cat $OUT/bench_line.json
  ;;
s20)  # the build of s19 under the settings that steer work to the repair kernels and the tiles' robust instantiations
  FILES="tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py"
  PROXTV_CHUNK_MODE=1 timeout 100 python -m pytest $FILES -m gpu -x -q > $OUT/pytest_mode1.log 2>&1; echo "pinned to rung 1: $(tail -1 $OUT/pytest_mode1.log)" | tee $OUT/summary.txt
  PROXTV_REPAIR_JOBS=2 timeout 100 python -m pytest $FILES -m gpu -x -q > $OUT/pytest_jobs2.log 2>&1; echo "repair_jobs=2: $(tail -1 $OUT/pytest_jobs2.log)" | tee -a $OUT/summary.txt
  PROXTV_CHUNK_MODE=0 timeout 100 python -m pytest $FILES -m gpu -x -q > $OUT/pytest_mode0.log 2>&1; echo "pinned to rung 0: $(tail -1 $OUT/pytest_mode0.log)" | tee -a $OUT/summary.txt
  ;;
esac
