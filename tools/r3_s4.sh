cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s4; rm -rf $O; mkdir -p $O
for lam in 0.1 0.5 0.7; do timeout 100 python tools/wg_trace.py $lam 1 2>&1 | grep -E "^##|^# mean" > $O/trace_mode1_$lam.txt; done
timeout 100 python tools/wg_trace.py 0.1 0 2>&1 | grep -E "^##|^# mean" > $O/trace_mode0_0.1.txt
timeout 100 python tools/wg_trace.py 0.7 2 2>&1 | grep -E "^##|^# mean" > $O/trace_mode2_0.7.txt
for f in $O/*.txt; do echo "=== $f"; cat $f; done
