# Final validation on the GPU box (round tag = $1, default r04): the suite under the default policy, pinned to rungs 1 / 3 / 5, on the
# 64-fibre tile, with the in-kernel link check switched off and with the hill-climbing policy; then -- unless NOPROFILES=1 -- the
# profile collection (tools/collect_profiles.sh).
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG}final; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" > $O/build_id.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; echo "default: $(tail -1 $O/pytest_default.log)" | tee -a $O/summary.txt
FILES="tests/test_gpu_chunk_repair.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_1d.py tests/test_gpu_parity_nd.py tests/test_gpu_fuzz.py tests/test_gpu_pin.py tests/test_gpu_boundary.py"
for m in ${MODES:-1 3 5}; do PROXTV_CHUNK_MODE=$m timeout 900 python -m pytest $FILES -m gpu -q > $O/pytest_mode$m.log 2>&1; echo "pinned to rung $m: $(tail -1 $O/pytest_mode$m.log)" | tee -a $O/summary.txt; done
PROXTV_TILE=0 timeout 900 python -m pytest $FILES tests/test_gpu_large.py -m gpu -q > $O/pytest_tile0.log 2>&1; echo "tile=0: $(tail -1 $O/pytest_tile0.log)" | tee -a $O/summary.txt
PROXTV_XLINK=0 timeout 900 python -m pytest $FILES -m gpu -q > $O/pytest_xlink0.log 2>&1; echo "xlink=0: $(tail -1 $O/pytest_xlink0.log)" | tee -a $O/summary.txt
PROXTV_DETERMINISTIC=0 timeout 900 python -m pytest $FILES tests/test_gpu_large.py -m gpu -q > $O/pytest_adaptive.log 2>&1; echo "deterministic=0: $(tail -1 $O/pytest_adaptive.log)" | tee -a $O/summary.txt
[ -z "$NOPROFILES" ] && { bash tools/collect_profiles.sh $TAG > $O/profiles.log 2>&1; tail -3 $O/profiles.log; }
cat $O/summary.txt $O/build_id.txt
