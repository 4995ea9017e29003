cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s14; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PROXTV_CHUNK_MODE=1 timeout 900 python -m pytest tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_mode1.log 2>&1; tail -2 $O/pytest_mode1.log
timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.1,0.3,0.5,0.7 > $O/lambda_default.txt 2>&1
timeout 120 python tools/small_images.py > $O/small_images.txt 2>&1
timeout 300 python tools/time_cases.py > $O/time_cases.txt 2>&1
cat $O/lambda_default.txt $O/small_images.txt $O/time_cases.txt
