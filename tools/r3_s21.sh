cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s21; rm -rf $O; mkdir -p $O
timeout 600 python tools/seed_check.py 2048 > $O/seed_check.txt 2>&1
cat $O/seed_check.txt
timeout 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_chunk_repair.py tests/test_gpu_pin.py -m gpu -x -q 2>&1 | tail -3
