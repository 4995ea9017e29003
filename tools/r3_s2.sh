cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s2; rm -rf $O; mkdir -p $O
for lam in 0.5 0.7; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$lam -o x -- python $R/tools/lambda_probe.py --modes 1 --lams $lam > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find $O/prof_$lam -name "x_results.db" | head -1) > $O/stats_mode1_$lam.txt
done
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_small -o x -- python $R/tools/small_images.py 256 1024 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $O/prof_small -name "x_results.db" | head -1) > $O/stats_small.txt
timeout 200 rocprofv3 --hip-trace --stats -d $O/prof_first -o x -- python $R/tools/first_solve.py 0.1 > $O/first.log 2>&1
find $O/prof_first -name "*stats*" | head; for f in $(find $O/prof_first -name "*hip_api_stats*csv" | head -1); do head -25 $f > $O/first_hip_api_stats.txt; done
rm -rf $O/prof_*
head -14 $O/stats_mode1_0.5.txt; head -14 $O/stats_mode1_0.7.txt; head -16 $O/stats_small.txt; cat $O/first_hip_api_stats.txt; cat $O/first.log | tail -3
