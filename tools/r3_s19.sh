cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s19; rm -rf $O; mkdir -p $O
cd $R
PROXTV_CHUNK_MODE=1 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py -m gpu -x -q > $O/pytest_mode1.log 2>&1; tail -2 $O/pytest_mode1.log
timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.3,0.5,0.6,0.7 > $O/lambda_default.txt 2>&1
cat $O/lambda_default.txt
cd /tmp
for lam in 0.5 0.7; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$lam -o x -- python $R/tools/lambda_probe.py --modes -1 --lams $lam > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find $O/prof_$lam -name "x_results.db" | head -1) > $O/stats_$lam.txt
  head -8 $O/stats_$lam.txt
done
rm -rf $O/prof_*
