cd $GRAFT_REPO_ROOT
export PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/proxtv_amd/build/lib_wc8.so
timeout 300 python -m pytest tests/test_gpu_parity_2d.py -m gpu -x -q -k "dr2w or weighted" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "c3_weighted_dr_4096 and not global" 2>&1 | tail -3
for q in 0 1 2 4; do PROXTV_BLOCKS_PER_WG=$q timeout 200 python tools/time_one.py c3 2>&1 | tail -1; done
timeout 200 python tools/time_one.py wprox1 2>&1 | tail -1
WG_TRACE_WEIGHTED=1 timeout 100 python tools/wg_trace.py 0.1 2>&1 | grep -E "^##|^# mean" | head -6
