cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s6; rm -rf $O; mkdir -p $O
for w in dr fibre; do timeout 100 python tools/first_call.py $w >> $O/first_call.txt 2>&1; done
timeout 100 python tools/first_call.py dr 3.0 >> $O/first_call.txt 2>&1
timeout 100 python tools/host_api_time.py > $O/host_api.txt 2>&1
timeout 300 python tools/lambda_probe.py --modes -1 --lams 0.5,0.6,0.7,0.8,0.9,1.0 > $O/lambda_default.txt 2>&1
timeout 300 python tools/lambda_probe.py --modes 1,3 --lams 0.8,0.9 > $O/lambda_pinned.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity_2d.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cat $O/first_call.txt $O/host_api.txt $O/lambda_default.txt $O/lambda_pinned.txt
