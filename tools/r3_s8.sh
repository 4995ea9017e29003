cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s8; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity_nd.py tests/test_gpu_parity_2d.py tests/test_gpu_fuzz.py tests/test_gpu_chunk_repair.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
for c in c4 c4y; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o x -- python $R/tools/profile_cases.py $c > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find $O/prof_$c -name "x_results.db" | head -1) > $O/stats_$c.txt
done
rm -rf $O/prof_*
head -12 $O/stats_c4.txt; head -12 $O/stats_c4y.txt
