cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s16; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pin.py tests/test_gpu_fuzz.py tests/test_gpu_large.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PROXTV_CHUNK_MODE=3 timeout 900 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_chunk_repair.py -m gpu -x -q > $O/pytest_mode3.log 2>&1; tail -2 $O/pytest_mode3.log
timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.8,1.0,3.0,10.0,30.0 > $O/lambda_default.txt 2>&1
timeout 120 python tools/long_fibre.py > $O/long_fibre.txt 2>&1
cat $O/lambda_default.txt $O/long_fibre.txt
