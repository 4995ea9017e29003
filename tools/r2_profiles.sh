# Round-2 profile collection on the GPU box: kernel stats of the headline and the other configurations, PMC traffic
# (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated), SQ instruction counters, the bench line, the phase trace.
# Everything lands under gpurun_out/r02/ ; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c5"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o x -- $BENCH > $O/bench_under_rocprof.log 2>&1
python $R/tools/rocprof_summary.py $(find $O/prof_bench -name "x_results.db" | head -1) > $O/r02_final_kernel_stats.txt
for c in c3 c4 c4y dr0.5 dr1.0 dr3.0; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o x -- python $R/tools/profile_cases.py $c > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find $O/prof_$c -name "x_results.db" | head -1) > $O/r02_final_${c}_kernel_stats.txt
done
# SQ counters, one pass per set (no trace domains besides --kernel-trace)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/sq$i -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c5 > /dev/null 2>&1
done
{
echo "# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c5   (three passes, one counter set each)"
echo "# averages per dispatch of the two hot kernels: sweep_along_kernel<1,...> = DR column sweep, sweep_chunk_kernel<3,...> = DR row sweep"
for k in 1 2 3; do python $R/tools/pmc_summary.py $(find $O/sq$k -name "p_results.db" | head -1) "sweep_along_kernel<1, false, 16, 64>" "sweep_chunk_kernel<3, false, false, 16, 8, 16, false>"; done
} > $O/r02_final_sq_counters.txt 2>&1
timeout 400 python $R/tools/pmc_traffic.py collect $O/pmc > $O/pmc_collect.log 2>&1
python $R/tools/pmc_traffic.py report $O/pmc > $O/r02_pmc_traffic.json 2> $O/pmc_report.err
cd $R
timeout 60 python tools/wg_trace.py > $O/r02_wg_trace.txt 2>&1
timeout 300 python tools/time_cases.py > $O/r02_time_cases.txt 2>&1
timeout 120 python tools/lambda_sweep.py > $O/r02_lambda_sweep.txt 2>&1
cp $O/r02_pmc_traffic.json $R/profiles/r02_pmc_traffic.json 2>/dev/null
timeout 300 python bench.py > $O/r02_bench_line.json 2> $O/bench.err
rm -rf $O/prof_* $O/sq? $O/pmc
ls -la $O
head -12 $O/r02_final_kernel_stats.txt
cat $O/r02_final_sq_counters.txt | head -40
python -c "import json; d=json.load(open('$O/r02_pmc_traffic.json')); print({k:(v['hbm_total'],v['ratio_to_algorithmic']) for k,v in d['kernels'].items()}, d.get('build_id'))"
cat $O/r02_bench_line.json
