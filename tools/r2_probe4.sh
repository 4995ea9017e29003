mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_boundary.py tests/test_gpu_sharding.py -x -q 2>&1 | tail -15
python bench.py --steps 10 --warmup 2
PROXTV_BENCH_SHARED_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-c5 2>&1 | tail -2
} > gpurun_out/r2_probe4.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_probe4.log | tail -40
