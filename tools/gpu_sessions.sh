#!/bin/bash
# GPU sessions of round 6 (one gpurun call each; everything lands under gpurun_out/r6/<session>/):
#     gpurun --timeout N -- 'bash tools/gpu_sessions.sh <session>'
# Variant builds (python -m proxtv_amd.build --variant NAME -- flags) must exist in proxtv_amd/build/ before the call: they travel
# with the snapshot.  (The sessions of round 5 are in the history of this file.)
S=$1
OUT=gpurun_out/r6/$S
mkdir -p $OUT
export TMPDIR=/tmp
W=proxtv_amd/build
ab() { python tools/ab_run.py "$@"; }
alt() { PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/$W/lib_$1.so "${@:2}"; }
case $S in
s1)   # the certifier, the pinned-rung matrix, the pruned kernel: new test files first (fail fast), then the whole suite, a certified soak, the bench line
  timeout 900 python -m pytest tests/test_gpu_certify.py tests/test_gpu_matrix.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "new files: $(tail -1 $OUT/pytest_new.log)" | tee $OUT/summary.txt
  timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee -a $OUT/summary.txt
  { python tools/fuzz.py 90 611; python tools/fuzz.py 30 612 nd; python tools/fuzz.py 40 613 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python - <<'P' | tee -a $OUT/summary.txt
import json,sys
d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r6/s1/bench.json').read().strip().splitlines()[-1])
print('headline ms', d['ms_per_step'], {k:round(v['avg_launch_ms']*1e3,1) for k,v in d['roofline']['by_kernel'].items()})
print({k:(round(d[k]['ms'],2), d[k]['ok']) for k in ('c3','lambda_1','hard')}, {k:(round(v['ms'],2), v['ok']) for k,v in d['c4'].items() if isinstance(v,dict)})
P
  ;;
s2)   # the certifier with the reference's EPSILON slack; A/B of the round's kernel against round 5's build and against the build without the prefix fence
  timeout 900 python -m pytest tests/test_gpu_certify.py tests/test_gpu_matrix.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "new files: $(tail -1 $OUT/pytest_new.log)" | tee $OUT/summary.txt
  timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee -a $OUT/summary.txt
  { python tools/fuzz.py 90 611; python tools/fuzz.py 30 612 nd; python tools/fuzz.py 40 613 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c2,prox0,prox1,c3,c4,c4y,pd2,s512 base r5=$W/lib_r5.so nofence=$W/lib_nofence.so > $OUT/ab_fence.txt 2>&1; cat $OUT/ab_fence.txt
  ;;
s3)   # what does the certifier still object to?  (verbose: the first failing fibres of every sweep, with the test and the margin)
  PROXTV_VERBOSE=1 timeout 600 python -m pytest tests/test_gpu_certify.py -m gpu -x -q -k "clean" > $OUT/pytest_certify.log 2>&1; grep "certify:" $OUT/pytest_certify.log | sort | uniq -c | sort -rn | head -40 | tee $OUT/summary.txt; tail -1 $OUT/pytest_certify.log | tee -a $OUT/summary.txt
  PROXTV_VERBOSE=1 python tools/fuzz.py 60 611 > $OUT/fuzz.txt 2>&1; grep "certify:" $OUT/fuzz.txt | cut -c1-260 | head -60 | tee -a $OUT/summary.txt; grep "^fuzz\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ab --reps 7 --rounds 2 --cases c2,prox0,prox1 base nofence=$W/lib_nofence.so > $OUT/ab_barrier.txt 2>&1; cat $OUT/ab_barrier.txt
  ;;
s4)   # the certifier with steps counted above the last-sample slack: new files, the whole suite, certified soaks (verbose: whatever still fails is described), the bench line
  PROXTV_VERBOSE=1 timeout 900 python -m pytest tests/test_gpu_certify.py tests/test_gpu_matrix.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "new files: $(tail -1 $OUT/pytest_new.log)" | tee $OUT/summary.txt; grep "certify:" $OUT/pytest_new.log | cut -c1-260 | sort | uniq -c | sort -rn | head -20 | tee -a $OUT/summary.txt
  timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee -a $OUT/summary.txt
  { PROXTV_VERBOSE=1 python tools/fuzz.py 120 611; python tools/fuzz.py 30 612 nd; PROXTV_VERBOSE=1 python tools/fuzz.py 60 613 long; } > $OUT/fuzz.txt 2>&1; grep "certify:" $OUT/fuzz.txt | cut -c1-260 | head -30 | tee -a $OUT/summary.txt; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
  ;;
s5)   # how often is a sweep of the headline marked dirty at all?
  python tools/dirty_rate.py 2>&1 | tee $OUT/dirty_rate.txt
  ;;
s6)   # optimistic solves + the wait for link words: tests, dirty marks again, A/B by option on one build, small images
  timeout 600 python -m pytest tests/test_gpu_optimistic.py tests/test_gpu_boundary.py tests/test_gpu_parity_2d.py tests/test_gpu_chunk_repair.py tests/test_gpu_large.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "tests: $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
  python tools/dirty_rate.py 2>&1 | tee $OUT/dirty_rate.txt
  ab --reps 9 --rounds 2 --cases c2,c2@0.2,c3,s2048,s1024,s512,s256 base plain,optimistic=0 r5=$W/lib_r5.so > $OUT/ab_optimistic.txt 2>&1; cat $OUT/ab_optimistic.txt
  ;;
s7)   # optimistic solves with the flags cleared before a second run and the back-off: the new file, then everything
  timeout 600 python -m pytest tests/test_gpu_optimistic.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "new: $(tail -1 $OUT/pytest_new.log)" | tee $OUT/summary.txt
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee -a $OUT/summary.txt
  { python tools/fuzz.py 60 621; python tools/fuzz.py 40 622 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ab --reps 9 --rounds 2 --cases c2,c2@0.2,c3,s512 base plain,optimistic=0 > $OUT/ab_optimistic.txt 2>&1; cat $OUT/ab_optimistic.txt
  ;;
s8)   # known runs: the new test file, parity files, A/B by option on one build, a certified soak on long fibres
  timeout 600 python -m pytest tests/test_gpu_runs.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "new: $(tail -1 $OUT/pytest_new.log)" | tee $OUT/summary.txt
  timeout 900 python -m pytest tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_large.py tests/test_gpu_matrix.py tests/test_gpu_certify.py -m gpu -x -q > $OUT/pytest_parity.log 2>&1; echo "parity: $(tail -1 $OUT/pytest_parity.log)" | tee -a $OUT/summary.txt
  ab --reps 9 --rounds 2 --cases c2,prox0,c2@0.13,pd2,c4,s2048,s512 base noruns,runs=0 > $OUT/ab_runs.txt 2>&1; cat $OUT/ab_runs.txt
  { python tools/fuzz.py 60 631 long; python tools/fuzz.py 60 632; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ;;
s9)   # known runs with LDS atomics and carry-chain masks: the test file, A/B by option, the phase trace of both
  timeout 600 python -m pytest tests/test_gpu_runs.py tests/test_gpu_parity_1d.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "new: $(tail -1 $OUT/pytest_new.log)" | tee $OUT/summary.txt
  ab --reps 9 --rounds 2 --cases c2,prox0,s2048,s512 base noruns,runs=0 > $OUT/ab_runs.txt 2>&1; cat $OUT/ab_runs.txt
  timeout 120 python tools/wg_trace.py > $OUT/wg_trace_runs.txt 2>&1; grep "^##\|^# mean" $OUT/wg_trace_runs.txt
  PROXTV_RUNS=0 timeout 120 python tools/wg_trace.py > $OUT/wg_trace_noruns.txt 2>&1; grep "^##\|^# mean" $OUT/wg_trace_noruns.txt
  ;;
s10)  # is the along-fibre kernel bound by how fast workgroups are dispatched?  eight waves (two fibres of 4096) per workgroup against four
  alt aw8 timeout 600 python -m pytest tests/test_gpu_runs.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_chunk_repair.py -m gpu -x -q > $OUT/pytest_aw8.log 2>&1; echo "aw8: $(tail -1 $OUT/pytest_aw8.log)" | tee $OUT/summary.txt
  ab --reps 9 --rounds 2 --cases c2,prox0,c2@0.5,c3,c4,pd2,s2048,s512 base aw8=$W/lib_aw8.so > $OUT/ab_aw8.txt 2>&1; cat $OUT/ab_aw8.txt
  ;;
s11)  # instruction counters of the column sweep with and without the known-runs path
  cd /tmp; R=$GRAFT_REPO_ROOT
  KC_SETS=insts,waves timeout 400 python $R/tools/kernel_counters.py collect $R/$OUT/kc_runs calib+dr0.1 > $R/$OUT/kc_collect.log 2>&1
  PROXTV_RUNS=0 KC_SETS=insts,waves timeout 400 python $R/tools/kernel_counters.py collect $R/$OUT/kc_noruns calib+dr0.1 >> $R/$OUT/kc_collect.log 2>&1
  cd $R
  python tools/kernel_counters.py report $OUT/kc_runs > $OUT/kernel_counters_runs.txt 2>&1; python tools/kernel_counters.py report $OUT/kc_noruns > $OUT/kernel_counters_noruns.txt 2>&1
  grep -h "along\|kernel  " $OUT/kernel_counters_runs.txt $OUT/kernel_counters_noruns.txt | cut -c1-260
  find $OUT -name "*.db" -delete
  ;;
s12)  # known runs with the waves of a workgroup on the same segment of four fibres: tests, A/B by option, counters
  timeout 600 python -m pytest tests/test_gpu_runs.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_large.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "tests: $(tail -1 $OUT/pytest_new.log)" | tee $OUT/summary.txt
  ab --reps 9 --rounds 2 --cases c2,prox0,pd2,s2048,s512 base noruns,runs=0 > $OUT/ab_runs.txt 2>&1; cat $OUT/ab_runs.txt
  timeout 120 python tools/wg_trace.py > $OUT/wg_trace_runs.txt 2>&1; grep "^##\|^# mean" $OUT/wg_trace_runs.txt | grep -A3 "column"
  ;;
s13)  # the rows two workgroups of a tile sweep stage, loaded without the streaming hint: parity, A/B against the build with the hint
  timeout 600 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_large.py tests/test_gpu_runs.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "tests: $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
  ab --reps 9 --rounds 2 --cases c2,c3,c2@0.5,pd2,c4y base nohalo=$W/lib_nohalo.so > $OUT/ab_halo.txt 2>&1; cat $OUT/ab_halo.txt
  ;;
s14)  # wave lives by segment; the many-terms PD test
  python tools/wave_life.py 2>&1 | tee $OUT/wave_life.txt
  timeout 600 python -m pytest tests/test_gpu_parity_nd.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "nd tests: $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
  ;;
s15)  # the along-fibre kernel's waves sorted (last segments in workgroups of their own): parity, wave lives, A/B against the unsorted build
  timeout 900 python -m pytest tests/test_gpu_runs.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_large.py tests/test_gpu_chunk_repair.py tests/test_gpu_optimistic.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "tests: $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
  python tools/wave_life.py 2>&1 | tee $OUT/wave_life.txt
  ab --reps 9 --rounds 2 --cases c2,prox0,c3,pd2,c4,c4y,s2048,s512 base unsorted=$W/lib_unsorted.so > $OUT/ab_sorted.txt 2>&1; cat $OUT/ab_sorted.txt
  ;;
s16)  # is the along-fibre kernel bound by the rate at which waves are launched?  two / three virtual workgroups per workgroup (sorted waves)
  for v in turns2 turns3; do alt $v timeout 600 python -m pytest tests/test_gpu_runs.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_2d.py tests/test_gpu_large.py tests/test_gpu_chunk_repair.py -m gpu -x -q > $OUT/pytest_$v.log 2>&1; echo "$v: $(tail -1 $OUT/pytest_$v.log)" | tee -a $OUT/summary.txt; done
  ab --reps 9 --rounds 2 --cases c2,prox0,pd2,s2048,s512 base unsorted=$W/lib_unsorted.so turns2=$W/lib_turns2.so turns3=$W/lib_turns3.so > $OUT/ab_turns.txt 2>&1; cat $OUT/ab_turns.txt
  ;;
s17)  # what does the column sweep cost with its phases switched off (option ablate: 1 walk + rebuild, 2 stream-out, 4 window loads)?  7 = the launch alone
  ab --reps 9 --rounds 1 --cases prox0,prox1 base a7,ablate=7 a5,ablate=5 a6,ablate=6 a3,ablate=3 a1,ablate=1 a2,ablate=2 a4,ablate=4 2> /dev/null > $OUT/ab_ablate.txt; cat $OUT/ab_ablate.txt
  ;;
final)  # the round's final build: the whole suite, smoke, certified soaks (2-D, volumes, long fibres), the pinned-rung subsets by environment
  timeout 1200 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | tee -a $OUT/summary.txt
  FILES="tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_boundary.py"
  for m in 0 1 3; do PROXTV_CHUNK_MODE=$m timeout 900 python -m pytest $FILES -m gpu -x -q > $OUT/pytest_mode$m.log 2>&1; echo "pinned rung $m: $(tail -1 $OUT/pytest_mode$m.log)" | tee -a $OUT/summary.txt; done
  PROXTV_REPAIR_JOBS=2 timeout 900 python -m pytest $FILES -m gpu -x -q > $OUT/pytest_jobs2.log 2>&1; echo "repair_jobs=2: $(tail -1 $OUT/pytest_jobs2.log)" | tee -a $OUT/summary.txt
  PROXTV_RUNS=0 PROXTV_OPTIMISTIC=0 timeout 900 python -m pytest $FILES tests/test_gpu_large.py -m gpu -x -q > $OUT/pytest_plain.log 2>&1; echo "runs=0 optimistic=0: $(tail -1 $OUT/pytest_plain.log)" | tee -a $OUT/summary.txt
  { python tools/fuzz.py 300 641; python tools/fuzz.py 300 642; python tools/fuzz.py 60 643 nd; python tools/fuzz.py 200 644 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ;;
final2)  # after the split of sweep_kernels.hpp (same code, new build id): the whole suite, smoke, a longer certified soak
  timeout 1200 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | tee -a $OUT/summary.txt
  { for sd in 651 652 653; do python tools/fuzz.py 240 $sd; done; python tools/fuzz.py 120 654 nd; python tools/fuzz.py 300 655 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ;;
final3)  # the edge masks cut at 4 lambda exactly: the runs tests, the certified full-size solves, the whole suite, certified campaigns at full size, soak
  timeout 600 python -m pytest tests/test_gpu_runs.py tests/test_gpu_large.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "runs + large: $(tail -1 $OUT/pytest_new.log)" | tee $OUT/summary.txt
  timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee -a $OUT/summary.txt
  for sd in 1 2 3; do python tools/certified_campaign.py 150 $sd > $OUT/campaign_$sd.txt 2>&1; tail -1 $OUT/campaign_$sd.txt | tee -a $OUT/summary.txt; grep "certify:" $OUT/campaign_$sd.txt | cut -c1-260 | head -10 | tee -a $OUT/summary.txt; done
  { python tools/fuzz.py 200 661; python tools/fuzz.py 200 662 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ;;
w1)   # knots known by windows (pin_seed = 2): the pinning tests, A/B by option over lambda, the rungs side by side
  timeout 900 python -m pytest tests/test_gpu_pin.py -m gpu -x -q > $OUT/pytest_pin.log 2>&1; echo "pin tests: $(tail -1 $OUT/pytest_pin.log)" | tee $OUT/summary.txt
  ab --reps 5 --rounds 2 --cases c2@0.7,c2@0.8,c2@1.0,c2@1.5,c2@3,c2@10,pd2@1.0,s2048@1.0,s1024@1.0 base jumps,pin_seed=1 > $OUT/ab_windows.txt 2>&1; cat $OUT/ab_windows.txt
  python tools/lambda_probe.py --modes 1,3 --lams 0.5,0.6,0.65,0.7 > $OUT/probe_mid.txt 2>&1; cat $OUT/probe_mid.txt
  ;;
w2)   # the window stages gated (wave ballots; the policy's half-penalty fraction): tests, A/B by option, the bench line
  timeout 900 python -m pytest tests/test_gpu_pin.py tests/test_gpu_large.py -m gpu -x -q > $OUT/pytest_pin.log 2>&1; echo "pin + large tests: $(tail -1 $OUT/pytest_pin.log)" | tee $OUT/summary.txt
  ab --reps 5 --rounds 2 --cases c2@0.8,c2@1.0,c2@1.5,c2@3,c2@5,c2@10,pd2@1.0,yang2@1.0,s2048@1.0,c4@1.0 base jumps,pin_seed=1 > $OUT/ab_windows.txt 2>&1; cat $OUT/ab_windows.txt
  python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; python - <<'P' | tee -a $OUT/summary.txt
import json
d=json.loads(open('gpurun_out/r6/w2/bench.json').read().strip().splitlines()[-1])
print('headline ms', d['ms_per_step'], {k:round(v['avg_launch_ms']*1e3,1) for k,v in d['roofline']['by_kernel'].items()})
print({k:(round(d[k]['ms'],2), d[k]['ok']) for k in ('c3','lambda_1','hard')}, {k:(round(v['ms'],2), v['ok']) for k,v in d['c4'].items() if isinstance(v,dict)})
P
  ;;
w4)   # operands of Dykstra / ADMM loops sampled mid-solve (policy_reprobe): parity files, then PD2 / Yang over lambda against the pinned rungs
  timeout 1200 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_pin.py tests/test_gpu_large.py tests/test_gpu_boundary.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "parity + pin + large: $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
  ab --reps 3 --rounds 1 --cases pd2@0.1,pd2@0.3,pd2@0.5,pd2@0.7,pd2@1.0,yang2@0.1,yang2@0.3,yang2@0.5,yang2@1.0,yang2@3,c4y,c4,c4y@1.0,c4@1.0,c2@3,c2@5 base m1,chunk_mode=1 m3,chunk_mode=3 > $OUT/ab_pd_yang.txt 2>&1; cat $OUT/ab_pd_yang.txt
  ;;
w6)   # window seeds + mid-solve samples, the whole suite; PD2 / Yang / DR over lambda under the default policy against the pinned rungs
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  ab --reps 3 --rounds 1 --cases pd2@0.4,pd2@0.5,pd2@0.6,pd2@0.7,pd2@1.0,pd2@3,yang2@0.5,yang2@1.0,yang2@3,yang2@10,c4y,c4 base m1,chunk_mode=1 m3,chunk_mode=3 > $OUT/ab_pd_yang.txt 2>&1; cat $OUT/ab_pd_yang.txt
  python tools/lambda_probe.py --modes -1 --lams 0.1,0.3,0.4,0.5,0.6,0.65,0.7,0.8,1.0,1.5,2.0,3.0,10.0,30.0 > $OUT/lambda_sweep.txt 2>&1; cat $OUT/lambda_sweep.txt
  ;;
w8)   # stages need two lanes; the policy's gate at 0.15; Yang sampled every fourth iteration and read with the general threshold
  timeout 900 python -m pytest tests/test_gpu_pin.py tests/test_gpu_parity_2d.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pin + parity 2d: $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
  ab --reps 3 --rounds 1 --cases c2@1.0,c2@2,c2@3,c2@5,hard,hard@1.0,yang2@1.0,yang2@2,yang2@3,pd2@0.6,pd2@3 base jumps,pin_seed=1 m3,chunk_mode=3 > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
  ;;
final4)  # window seeds (always on, stages gated wave by wave) + mid-solve samples: the whole suite, smoke, soaks with the options drawn at random, a certified campaign
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | tee -a $OUT/summary.txt
  PROXTV_CHUNK_MODE=3 timeout 900 python -m pytest tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_nd.py tests/test_gpu_boundary.py tests/test_gpu_large.py -m gpu -x -q > $OUT/pytest_mode3.log 2>&1; echo "pinned rung 3: $(tail -1 $OUT/pytest_mode3.log)" | tee -a $OUT/summary.txt
  { python tools/fuzz.py 240 671; python tools/fuzz.py 240 672; python tools/fuzz.py 60 673 nd; python tools/fuzz.py 120 674 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  python tools/certified_campaign.py 100 5 > $OUT/campaign_5.txt 2>&1; tail -1 $OUT/campaign_5.txt | tee -a $OUT/summary.txt
  python tools/certified_campaign.py 150 6 high > $OUT/campaign_6_high.txt 2>&1; tail -1 $OUT/campaign_6_high.txt | tee -a $OUT/summary.txt; grep "certify:" $OUT/campaign_6_high.txt | cut -c1-260 | head -10 | tee -a $OUT/summary.txt
  ;;
final5)  # the round's last build: the whole suite, smoke, the two-rank dry run of bench.py on one GPU, soaks, a certified campaign at large penalties
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | tee -a $OUT/summary.txt
  PROXTV_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench2.json 2> $OUT/bench2.err; python -c "
import json;d=json.loads(open('$OUT/bench2.json').read().strip().splitlines()[-1]);print('2 ranks on one GPU (dry run):',d['n_gpus'],'ranks',round(d['ms_per_step'],2),'ms per step, c5 gather checked',d['c5'].get('gather_checked'))" | tee -a $OUT/summary.txt
  { python tools/fuzz.py 200 681; python tools/fuzz.py 200 682; python tools/fuzz.py 50 683 nd; python tools/fuzz.py 80 684 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  python tools/certified_campaign.py 200 7 high > $OUT/campaign_7_high.txt 2>&1; tail -1 $OUT/campaign_7_high.txt | tee -a $OUT/summary.txt; grep "certify:" $OUT/campaign_7_high.txt | cut -c1-260 | head -10 | tee -a $OUT/summary.txt
  python tools/certified_campaign.py 100 8 > $OUT/campaign_8.txt 2>&1; tail -1 $OUT/campaign_8.txt | tee -a $OUT/summary.txt
  ;;
final6)  # windows on weighted fibres + the weighted threshold: the whole suite, smoke, weighted DR over its penalties, soaks, a certified campaign at large penalties
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $OUT/pytest_default.log 2>&1; echo "default: $(tail -1 $OUT/pytest_default.log)" | tee $OUT/summary.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | tee -a $OUT/summary.txt
  timeout 600 python tools/weighted_rungs.py > $OUT/weighted_rungs.txt 2>&1; grep RESULT $OUT/weighted_rungs.txt | tee -a $OUT/summary.txt
  { python tools/fuzz.py 150 691; python tools/fuzz.py 150 692; python tools/fuzz.py 40 693 nd; python tools/fuzz.py 60 694 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  python tools/certified_campaign.py 180 9 high > $OUT/campaign_9_high.txt 2>&1; tail -1 $OUT/campaign_9_high.txt | tee -a $OUT/summary.txt; grep "certify:" $OUT/campaign_9_high.txt | cut -c1-260 | head -10 | tee -a $OUT/summary.txt
  ;;
soak1)  # the round's last build (4700b7897a11) once more: certified campaigns at large and at the usual penalties, option soaks
  python tools/certified_campaign.py 360 10 high > $OUT/campaign_10_high.txt 2>&1; tail -1 $OUT/campaign_10_high.txt | tee $OUT/summary.txt; grep "certify:" $OUT/campaign_10_high.txt | cut -c1-260 | head -10 | tee -a $OUT/summary.txt
  python tools/certified_campaign.py 240 11 > $OUT/campaign_11.txt 2>&1; tail -1 $OUT/campaign_11.txt | tee -a $OUT/summary.txt
  { python tools/fuzz.py 240 701; python tools/fuzz.py 240 702; python tools/fuzz.py 100 703 long; } > $OUT/fuzz.txt 2>&1; grep "^fuzz\|MISMATCH\|Error" $OUT/fuzz.txt | tee -a $OUT/summary.txt
  ;;
*) echo "unknown session $S";;
esac
