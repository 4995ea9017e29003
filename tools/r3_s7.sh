cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s7; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_nd.py tests/test_gpu_parity_2d.py tests/test_gpu_fuzz.py tests/test_gpu_chunk_repair.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/short_probe.py > $O/short.txt 2>&1
timeout 300 python tools/time_cases.py > $O/time_cases.txt 2>&1
cat $O/short.txt $O/time_cases.txt
