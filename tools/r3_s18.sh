cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s18; rm -rf $O; mkdir -p $O
PROXTV_CHUNK_MODE=1 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_chunk_repair.py -m gpu -x -q > $O/pytest_mode1.log 2>&1; tail -2 $O/pytest_mode1.log
timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.3,0.5,0.6,0.7 > $O/lambda_default.txt 2>&1
timeout 300 python tools/lambda_probe.py --modes 1 --lams 0.5,0.7 --opt row_along=0 > $O/lambda_tile.txt 2>&1
cat $O/lambda_default.txt $O/lambda_tile.txt
