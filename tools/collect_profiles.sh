# Profile collection on the GPU box (round tag = $1, default r04): kernel stats of the headline and the other configurations, PMC
# traffic (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated), SQ instruction counters, the bench line, the phase trace, the
# lambda sweep, first calls, small images, short fibres, the host-pointer path.  Everything lands under gpurun_out/<tag>/ ; what is
# to be judged is copied into profiles/ afterwards.   QUICK=1: kernel stats of the headline + PMC traffic + bench line only.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c5"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o x -- $BENCH > $O/bench_under_rocprof.log 2>&1
python $R/tools/rocprof_summary.py $(find $O/prof_bench -name "x_results.db" | head -1) > $O/${TAG}_final_kernel_stats.txt
timeout 400 python $R/tools/pmc_traffic.py collect $O/pmc > $O/pmc_collect.log 2>&1
python $R/tools/pmc_traffic.py report $O/pmc > $O/${TAG}_pmc_traffic.json 2> $O/pmc_report.err
cp $O/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json 2>/dev/null
if [ -z "$QUICK" ]; then
for c in c3 c4 c4y dr0.5 dr0.7 dr1.0 dr3.0; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o x -- python $R/tools/profile_cases.py $c > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find $O/prof_$c -name "x_results.db" | head -1) > $O/${TAG}_final_${c}_kernel_stats.txt
done
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/sq$i -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c5 > /dev/null 2>&1
done
{
echo "# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c5   (three passes, one counter set each)"
echo "# averages per dispatch of the two hot kernels: sweep_along_kernel<1,...> = DR column sweep, sweep_chunk_kernel<3,...> = DR row sweep"
for k in 1 2 3; do python $R/tools/pmc_summary.py $(find $O/sq$k -name "p_results.db" | head -1) "sweep_along_kernel<1, false, 16, 64, false," "sweep_chunk_kernel<3, false, false, 16,"; done
} > $O/${TAG}_final_sq_counters.txt 2>&1
cd $R
timeout 60 python tools/wg_trace.py > $O/${TAG}_wg_trace.txt 2>&1
timeout 300 python tools/time_cases.py > $O/${TAG}_time_cases.txt 2>&1
timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.1,0.3,0.4,0.5,0.6,0.7,0.8,1.0,3.0,10.0,30.0 > $O/${TAG}_lambda_sweep.txt 2>&1
for w in dr fibre; do timeout 100 python tools/first_call.py $w >> $O/${TAG}_first_call.txt 2>&1; done
timeout 120 python tools/small_images.py > $O/${TAG}_small_images.txt 2>&1
timeout 120 python tools/long_fibre.py > $O/${TAG}_long_fibre.txt 2>&1
timeout 200 python tools/short_probe.py > $O/${TAG}_short_fibres.txt 2>&1
timeout 100 python tools/host_api_time.py > $O/${TAG}_host_api.txt 2>&1
fi
cd $R
timeout 400 python bench.py > $O/${TAG}_bench_line.json 2> $O/bench.err
rm -rf $O/prof_* $O/sq? $O/pmc
ls -la $O
head -12 $O/${TAG}_final_kernel_stats.txt
python -c "import json; d=json.load(open('$O/${TAG}_pmc_traffic.json')); print({k:(v['hbm_total'],v['ratio_to_algorithmic']) for k,v in d['kernels'].items()}, d.get('build_id'))"
cat $O/${TAG}_bench_line.json
