# phase ablation of the headline solve: PROXTV_ABLATE bits 1 = skip the walk, 2 = skip the epilogue, 4 = skip the window loads
# (8 = baseline without the repair launches); usage: bash tools/ablate.sh [list of values]
for a in ${@:-0 8 9 10 12 11 13 14 15}; do
  PROXTV_ABLATE=$a python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate', '$a', round(d['ms_per_step'],2), d['roofline']['family_ms_per_solve'])"
done
