cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s27; rm -rf $O; mkdir -p $O
for rep in 1 2; do for f in 1 0; do
  echo "== dr_form=$f: $(PROXTV_DR_FORM=$f timeout 100 python tools/time_one.py c2 0.1 | tail -1)"
done; done
for f in 1 0; do
  echo "== dr_form=$f: $(PROXTV_DR_FORM=$f timeout 100 python tools/time_one.py c3 | tail -1)"
  echo "== dr_form=$f: $(PROXTV_DR_FORM=$f timeout 100 python tools/time_one.py c2 0.3 | tail -1)"
  echo "== dr_form=$f: $(PROXTV_DR_FORM=$f timeout 100 python tools/time_one.py c2 0.5 | tail -1)"
done
for f in 1 0; do echo "== dr_form=$f"; PROXTV_DR_FORM=$f timeout 100 python tools/small_images.py 256 1024 2048 | tail -3; done
PROXTV_DR_FORM=1 timeout 200 python tools/lambda_probe.py --lams 0.1,0.3 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
