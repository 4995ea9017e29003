cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s11; rm -rf $O; mkdir -p $O
WG_TRACE_WEIGHTED=1 timeout 100 python tools/wg_trace.py 0.1 2>&1 | grep -E "^##|^# mean|^ +[0-9]+ [0-9] [0-9] [0-9] +0 " > $O/trace_w.txt
PROXTV_BLOCKS_PER_WG=1 WG_TRACE_WEIGHTED=1 timeout 100 python tools/wg_trace.py 0.1 2>&1 | grep -E "^##|^# mean" > $O/trace_w_q1.txt
cat $O/trace_w.txt; echo ---- qpw=1; cat $O/trace_w_q1.txt
for q in 1 2 4 8; do PROXTV_BLOCKS_PER_WG=$q timeout 200 python tools/time_cases.py 2>&1 | grep -E "C3" ; done
