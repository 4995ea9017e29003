# A/B the headline with an alternative build of the library:  bash tools/ab_headline.sh proxtv_amd/libproxtv_noov.so
ALT=$1
for k in 1 2 3; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('main', round(d['ms_per_step'],3), d['roofline']['family_ms_per_solve'])"
  cp proxtv_amd/libproxtv_amd.so /tmp/main.so; cp $ALT proxtv_amd/libproxtv_amd.so
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alt ', round(d['ms_per_step'],3), d['roofline']['family_ms_per_solve'])"
  cp /tmp/main.so proxtv_amd/libproxtv_amd.so
done
