mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity_nd.py tests/test_gpu_parity_1d.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
python tools/time_cases.py
PROXTV_WHOLE=0 python tools/time_cases.py 2>&1 | grep "C4"
} > gpurun_out/r2_probe5.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_probe5.log
