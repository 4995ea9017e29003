cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s29; rm -rf $O; mkdir -p $O
timeout 300 python tools/lambda_probe.py --lams 0.3,0.4,0.5,0.6 2>&1 | grep "^mode"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
