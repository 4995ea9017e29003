# round-2 probe 3: new chunk core (branch-free interior walk, ownership rebuild, s' kept in registers)
mkdir -p gpurun_out
{
python bench.py --steps 5 --warmup 2 --no-cpu-baseline
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python tools/wg_trace.py
} > gpurun_out/r2_probe3.log 2>&1
tail -120 gpurun_out/r2_probe3.log
