cd $GRAFT_REPO_ROOT
O=gpurun_out/r3soak2; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" > $O/build_id.txt
timeout 100 python tools/fuzz.py 55 21 > $O/fuzz2d.txt 2>&1; tail -1 $O/fuzz2d.txt
timeout 60 python tools/fuzz.py 25 22 nd > $O/fuzznd.txt 2>&1; tail -1 $O/fuzznd.txt
timeout 80 python - > $O/fuzzlong.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import fuzz
n, w, where = fuzz.run(budget=35.0, seed=23, sizes=(96, 130, 1089, 2177, 3300, 4353))
print(f"fuzz long fibres: {n} cases, worst relative error {w:.2e} ({where})")
PY
tail -1 $O/fuzzlong.txt; cat $O/build_id.txt
