cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s30; rm -rf $O; mkdir -p $O
for f in 1 0; do
  echo "== dr_form=$f: $(PROXTV_DR_FORM=$f timeout 100 python tools/time_one.py c3 0.4 | tail -1)"
  echo "== dr_form=$f: $(PROXTV_DR_FORM=$f timeout 100 python tools/time_one.py c3 0.6 | tail -1)"
done
timeout 300 python tools/fuzz.py 90 11 > $O/fuzz2d.txt 2>&1; tail -2 $O/fuzz2d.txt
timeout 300 python - > $O/fuzzlong.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import fuzz
n, w, where = fuzz.run(budget=70.0, seed=12, sizes=(96, 130, 1089, 2177, 3300, 4353))
print(f"fuzz long fibres: {n} cases, worst relative error {w:.2e} ({where})")
PY
tail -2 $O/fuzzlong.txt
