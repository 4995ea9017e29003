cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s10; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_fuzz.py tests/test_gpu_chunk_repair.py tests/test_gpu_pin.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "c3" > $O/pytest_c3.log 2>&1; tail -3 $O/pytest_c3.log
timeout 300 python tools/time_cases.py 2>&1 | grep -E "C3|C2|case" > $O/time_cases.txt
cat $O/time_cases.txt
