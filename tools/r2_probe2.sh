# round-2 probe 2: phase stagger between the workgroups of a CU
mkdir -p gpurun_out
{
for s in 0 1 2 3 4 6 8 12; do
  PROXTV_STAGGER=$s python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger', '$s', round(d['ms_per_step'],3), d['roofline']['family_ms_per_solve'])"
done
} > gpurun_out/r2_probe2.log 2>&1
cat gpurun_out/r2_probe2.log
