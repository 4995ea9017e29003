cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s3; rm -rf $O; mkdir -p $O
PROXTV_CHUNK_MODE=1 timeout 900 python -m pytest tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_mode1.log 2>&1; echo "rc $?" >> $O/pytest_mode1.log
tail -5 $O/pytest_mode1.log
timeout 900 python -m pytest tests/test_gpu_chunk_repair.py tests/test_gpu_boundary.py tests/test_gpu_fuzz.py tests/test_gpu_large.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.1,0.3,0.5,0.7,1.0 > $O/lambda_default.txt 2>&1
timeout 300 python tools/lambda_probe.py --modes 1 --lams 0.7,1.0 > $O/lambda_pinned.txt 2>&1
timeout 200 python tools/lambda_probe.py --modes 1 --lams 0.5,0.7,1.0 --opt row_along=3 > $O/lambda_rowalong.txt 2>&1
timeout 200 python tools/lambda_probe.py --modes 2 --lams 0.7,1.0 > $O/lambda_mode2.txt 2>&1
cat $O/lambda_default.txt $O/lambda_pinned.txt $O/lambda_rowalong.txt $O/lambda_mode2.txt
