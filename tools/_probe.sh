cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 600 python -m pytest tests/test_gpu_pin.py tests/test_gpu_boundary.py -x -q 2>&1 | tail -5
timeout 120 python tools/lambda_sweep.py 1.0 3.0 10.0 2>&1 | grep -v amdgpu.ids
cd /tmp
rm -rf $R/gpurun_out/prof_dr3.0
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dr3.0 -o x -- python $R/tools/profile_cases.py dr3.0 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_dr3.0 -name "x_results.db" | head -1) | cut -c1-150 | grep "pin_kernel\|transpose" | head -4
