cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 400 python -m pytest tests/test_gpu_pin.py -x -q 2>&1 | tail -5
PROXTV_VERBOSE=1 timeout 120 python $R/tools/_pol_probe.py 2>&1 | grep -v amdgpu | grep "trial\|---\|^mode" | cut -c1-150 | tail -70
timeout 120 python tools/lambda_sweep.py 0.1 0.5 0.7 1.0 3.0 10.0 2>&1 | grep -v amdgpu.ids
cd /tmp
for c in dr3.0 dr1.0; do
  rm -rf $R/gpurun_out/prof_$c
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o x -- python $R/tools/profile_cases.py $c > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_$c -name "x_results.db" | head -1) | cut -c1-150 | head -9
done
