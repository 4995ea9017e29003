cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for L in main nw5; do
  if [ $L = main ]; then unset PROXTV_LIB PROXTV_DEBUG_ALT_LIB; else export PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/proxtv_amd/build/lib_$L.so; fi
  echo "== $L"
  timeout 100 python tools/time_one.py c2 0.1 | tail -1
  timeout 100 python tools/time_one.py prox1 0.1 | tail -1
done
done
export PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/proxtv_amd/build/lib_nw5.so
timeout 300 python -m pytest tests/test_gpu_parity_2d.py -m gpu -x -q 2>&1 | tail -2
timeout 100 python tools/wg_trace.py 0.1 2>&1 | grep -E "^##|^# mean" | head -3
