cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s9; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/time_cases.py 2>&1 | grep -E "C4|case" > $O/time_cases.txt
cat $O/time_cases.txt
