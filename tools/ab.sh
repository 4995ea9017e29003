# A/B of the headline between the in-tree library and alternative builds:  bash tools/ab.sh lib1.so [lib2.so ...]
for k in 1 2; do
  for L in "" "$@"; do
    PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${L:-main}', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['family_ms_per_solve'].items()})"
  done
done
