cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s20; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for m in 1 3 5; do PROXTV_CHUNK_MODE=$m timeout 900 python -m pytest tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py tests/test_gpu_pin.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_mode$m.log 2>&1; echo "mode $m: $(tail -1 $O/pytest_mode$m.log)"; done
timeout 300 python tools/lambda_probe.py --modes -1 > $O/lambda_default.txt 2>&1
cat $O/lambda_default.txt
