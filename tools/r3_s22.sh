cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s22; rm -rf $O; mkdir -p $O
timeout 600 python tools/seed_check.py 2048 > $O/seed_check.txt 2>&1
grep -E "checker|white|<--" $O/seed_check.txt
timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.1,0.2,0.25,0.3,0.5,0.6,0.65,0.7,0.75,0.8,1.0 > $O/lambda_default.txt 2>&1
cat $O/lambda_default.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
