# rocprofv3 kernel stats of one case:  bash tools/r2_prof.sh <case> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_$2
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$2 -o x -- python $R/tools/profile_cases.py $1 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_$2 -name "x_results.db" | head -1) > $R/gpurun_out/stats_$2.txt
head -14 $R/gpurun_out/stats_$2.txt
