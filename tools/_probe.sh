cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 600 python -m pytest tests/test_gpu_pin.py -x -q 2>&1 | tail -3
timeout 120 python tools/lambda_sweep.py 0.7 1.0 3.0 10.0 2>&1 | grep -v amdgpu.ids
cd /tmp
for c in dr3.0; do
rm -rf $R/gpurun_out/prof_$c
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o x -- python $R/tools/profile_cases.py $c > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_$c -name "x_results.db" | head -1) | cut -c1-150 | grep "pin_kernel\|transpose" | head -3
done
rm -rf $R/gpurun_out/prof_m3
PROXTV_CHUNK_MODE=3 timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_m3 -o x -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c5 > $R/gpurun_out/m3.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_m3 -name "x_results.db" | head -1) | cut -c1-150 | head -4 | tail -2
