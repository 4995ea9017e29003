# round-2 probe 1: phase ablation of the headline solve on the round-1 build (numbers to steer the kernel diet)
mkdir -p gpurun_out
{
python bench.py --steps 5 --warmup 2 --no-cpu-baseline
bash tools/ablate.sh 0 8 9 10 12 11 13 14 15
} > gpurun_out/r2_probe1.log 2>&1
tail -20 gpurun_out/r2_probe1.log
