cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 600 python -m pytest tests/test_gpu_pin.py tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py -x -q 2>&1 | tail -5
timeout 120 python tools/lambda_sweep.py 0.1 0.5 0.7 1.0 3.0 10.0 2>&1 | grep -v amdgpu.ids
