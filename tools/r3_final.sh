# Round-3 final validation on the GPU box: the suite under the default policy, pinned to rungs 1 / 3 / 5, and with the
# in-kernel link check switched off; then the profile collection (tools/r3_profiles.sh).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" > $O/build_id.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; echo "default: $(tail -1 $O/pytest_default.log)" | tee -a $O/summary.txt
FILES="tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py tests/test_gpu_pin.py tests/test_gpu_boundary.py"
for m in 1 3 5; do PROXTV_CHUNK_MODE=$m timeout 900 python -m pytest $FILES -m gpu -q > $O/pytest_mode$m.log 2>&1; echo "pinned to rung $m: $(tail -1 $O/pytest_mode$m.log)" | tee -a $O/summary.txt; done
PROXTV_XLINK=0 timeout 900 python -m pytest $FILES -m gpu -q > $O/pytest_xlink0.log 2>&1; echo "xlink=0: $(tail -1 $O/pytest_xlink0.log)" | tee -a $O/summary.txt
PROXTV_DETERMINISTIC=0 timeout 900 python -m pytest $FILES tests/test_gpu_large.py -m gpu -q > $O/pytest_adaptive.log 2>&1; echo "deterministic=0: $(tail -1 $O/pytest_adaptive.log)" | tee -a $O/summary.txt
bash tools/r3_profiles.sh > $O/profiles.log 2>&1
tail -3 $O/profiles.log
cat $O/summary.txt $O/build_id.txt
