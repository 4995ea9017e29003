# the GPU suite with every sweep pinned to geometry modes 1, 3, 5 (and adaptive), then the headline
mkdir -p gpurun_out
{
for m in -1 1 3 5; do
  echo "== PROXTV_CHUNK_MODE=$m"
  PROXTV_CHUNK_MODE=$m python -m pytest tests -x -q -m gpu 2>&1 | tail -3
done
echo "== PROXTV_ALONG=0 (transposed tile for dimension 0)"
PROXTV_ALONG=0 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 --no-cpu-baseline
} > gpurun_out/r2_modes.log 2>&1
cat gpurun_out/r2_modes.log | grep -v amdgpu.ids
