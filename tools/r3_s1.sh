# Round-3 GPU session 1: the suite on the new build, then where the time goes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s1; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python tools/lambda_probe.py --modes -1 > $O/lambda_default.txt 2>&1
timeout 300 python tools/lambda_probe.py --modes 1,3 --lams 0.3,0.5,0.7,1.0 > $O/lambda_pinned.txt 2>&1
timeout 200 python tools/lambda_probe.py --modes 1 --lams 0.5,0.7,1.0 --opt row_along=3 > $O/lambda_rowalong.txt 2>&1
timeout 200 python tools/lambda_probe.py --modes 2 --lams 0.7,1.0 > $O/lambda_mode2.txt 2>&1
timeout 200 python tools/lambda_probe.py --modes -1 --lams 0.1,0.5,1.0,3.0 --opt deterministic=0 > $O/lambda_adaptive.txt 2>&1
timeout 300 python tools/time_cases.py > $O/time_cases.txt 2>&1
timeout 120 python tools/small_images.py > $O/small_images.txt 2>&1
for lam in 0.1 3.0; do timeout 120 python tools/first_solve.py $lam >> $O/first_solve.txt 2>&1; done
timeout 120 python tools/long_fibre.py > $O/long_fibre.txt 2>&1
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/lambda_default.txt $O/lambda_pinned.txt $O/lambda_rowalong.txt $O/lambda_mode2.txt $O/lambda_adaptive.txt $O/time_cases.txt $O/small_images.txt $O/first_solve.txt $O/long_fibre.txt
python -c "import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline']['by_kernel'], d.get('c5'))"
