# Profile collection on the GPU box (round tag = $1, default r05): kernel stats of the headline and the other configurations, the
# per-kernel table of durations / registers / HBM traffic / SQ counters (tools/kernel_counters.py: one process per counter set), the
# bench line, the phase trace, the lambda sweep, first calls, small images, short fibres, the host-pointer path.  Everything lands
# under gpurun_out/<tag>/ ; what is to be judged is copied into profiles/ afterwards.
#   QUICK=1: kernel stats of the headline + the counters table + bench line only.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c5"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o x -- $BENCH > $O/bench_under_rocprof.log 2>&1
python $R/tools/rocprof_summary.py $(find $O/prof_bench -name "x_results.db" | head -1) > $O/${TAG}_final_kernel_stats.txt
timeout 900 python $R/tools/kernel_counters.py collect $O/kc > $O/kc_collect.log 2>&1
(cd $R && python tools/kernel_counters.py report $O/kc > $O/${TAG}_kernel_counters.txt 2> $O/kc_report.err)
(cd $R && python tools/kernel_counters.py traffic $O/kc > $O/${TAG}_pmc_traffic.json 2>> $O/kc_report.err)
if [ -z "$QUICK" ]; then
for c in c3 c4 c4y dr0.5 dr0.7 dr1.0 dr3.0; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o x -- python $R/tools/profile_cases.py $c > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find $O/prof_$c -name "x_results.db" | head -1) > $O/${TAG}_final_${c}_kernel_stats.txt
done
cd $R
timeout 60 python tools/wg_trace.py > $O/${TAG}_wg_trace.txt 2>&1
timeout 300 python tools/time_cases.py > $O/${TAG}_time_cases.txt 2>&1
timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.1,0.3,0.4,0.5,0.6,0.65,0.7,0.8,1.0,3.0,10.0,30.0 > $O/${TAG}_lambda_sweep.txt 2>&1
for w in dr fibre; do timeout 100 python tools/first_call.py $w >> $O/${TAG}_first_call.txt 2>&1; done
timeout 120 python tools/small_images.py > $O/${TAG}_small_images.txt 2>&1
timeout 120 python tools/long_fibre.py > $O/${TAG}_long_fibre.txt 2>&1
timeout 200 python tools/short_probe.py > $O/${TAG}_short_fibres.txt 2>&1

fi
cd $R
timeout 400 python bench.py > $O/${TAG}_bench_line.json 2> $O/bench.err
rm -rf $O/prof_*; find $O/kc -name "*.db" -delete
ls -la $O
head -12 $O/${TAG}_final_kernel_stats.txt
cat $O/${TAG}_kernel_counters.txt
cat $O/${TAG}_bench_line.json
