# r5-prep GPU sessions: parity of option "repair_jobs" against the sequential repair, the A/B timing at the penalties where the
# repair kernels are a fifth of a solve, kernel traces of both.
#   /usr/local/graft/bin/gpurun --timeout 280 -- 'bash tools/r5_prep_session.sh [parity|ab|trace]...'
OUT=gpurun_out/r5p; mkdir -p $OUT
python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" | tee $OUT/summary.txt
for S in "$@"; do case $S in
parity)
  timeout 170 python tools/r5_parity.py > $OUT/parity.txt 2>&1; echo "parity rc=$? : $(tail -1 $OUT/parity.txt)" | tee -a $OUT/summary.txt
  grep -c "^ok" $OUT/parity.txt; grep -v "^ok" $OUT/parity.txt | grep -v amdgpu | head -20 ;;
ab)
  timeout 100 python tools/ab_run.py --reps 5 --rounds 2 --cases c2@0.5,c2@0.65,c2@0.7,c2@0.6 seq jobs,repair_jobs=1 > $OUT/ab.txt 2>&1
  grep -v amdgpu $OUT/ab.txt | tail -30 ;;
trace)
  cd /tmp && export TMPDIR=/tmp
  for J in 0 1; do
    PROXTV_REPAIR_JOBS=$J timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/tr$J -o x -- python $GRAFT_REPO_ROOT/tools/profile_cases.py dr0.7 > /tmp/tr$J.log 2>&1
    python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/tr$J -name "x_results.db" | head -1) > $GRAFT_REPO_ROOT/$OUT/dr0.7_kernel_stats_jobs$J.txt
    echo "== repair_jobs=$J"; head -14 $GRAFT_REPO_ROOT/$OUT/dr0.7_kernel_stats_jobs$J.txt | cut -c1-200
  done
  cd $GRAFT_REPO_ROOT ;;
soak)   # differential soak against the CPU oracle with the jobs repair drawn on in three cases of four (tools/fuzz.py)
  { echo "# python tools/fuzz.py 30 81 ; PROXTV_REPAIR_JOBS=2 python tools/fuzz.py 12 82 nd -- build $(cat $OUT/summary.txt)"
    timeout 60 python tools/fuzz.py 30 81; PROXTV_REPAIR_JOBS=2 timeout 40 python tools/fuzz.py 12 82 nd; } > $OUT/fuzz_soak.txt 2>&1; grep -v amdgpu $OUT/fuzz_soak.txt | tail -5 ;;
diag)   # what a jobs launch is made of: variants of the build (tools/build_variant.sh), each traced on DR 4096^2 at lambda 0.7
  cd /tmp && export TMPDIR=/tmp
  for V in base w64 nowalk tabdiv; do
    L=$GRAFT_REPO_ROOT/proxtv_amd/build/lib_$V.so; [ -f $L ] || continue
    PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$L PROXTV_REPAIR_JOBS=2 timeout 60 rocprofv3 --kernel-trace --stats -d /tmp/dg$V -o x -- python $GRAFT_REPO_ROOT/tools/profile_cases.py dr0.7 > /tmp/dg$V.log 2>&1
    python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/dg$V -name "x_results.db" | head -1) > $GRAFT_REPO_ROOT/$OUT/dr0.7_kernel_stats_$V.txt
    echo "== $V"; grep "sweep_repair" $GRAFT_REPO_ROOT/$OUT/dr0.7_kernel_stats_$V.txt | cut -c1-150
  done
  cd $GRAFT_REPO_ROOT ;;
esac; done
