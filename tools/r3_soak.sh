cd $GRAFT_REPO_ROOT
O=gpurun_out/r3soak; rm -rf $O; mkdir -p $O
timeout 400 python tools/fuzz.py 150 7 > $O/fuzz2d.txt 2>&1; tail -2 $O/fuzz2d.txt
timeout 300 python tools/fuzz.py 90 8 nd > $O/fuzznd.txt 2>&1; tail -2 $O/fuzznd.txt
timeout 400 python - > $O/fuzzlong.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import fuzz
n, w, where = fuzz.run(budget=120.0, seed=9, sizes=(96, 130, 1089, 2177, 3300, 4353))
print(f"fuzz long fibres: {n} cases, worst relative error {w:.2e} ({where})")
PY
tail -2 $O/fuzzlong.txt
