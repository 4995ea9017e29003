# after the final run: the two test-side fixes (fuzz metric relative to the input's scale; the two-forms test pins the seeded policy)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final3; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print('build id', build.build_id())" > $O/build_id.txt
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; echo "default: $(tail -1 $O/pytest_default.log)" | tee -a $O/summary.txt
FILES="tests/test_gpu_fuzz.py tests/test_gpu_parity_2d.py"
PROXTV_DETERMINISTIC=0 timeout 300 python -m pytest $FILES -m gpu -q > $O/pytest_adaptive.log 2>&1; echo "deterministic=0 ($FILES): $(tail -1 $O/pytest_adaptive.log)" | tee -a $O/summary.txt
for m in 1 3 5; do PROXTV_CHUNK_MODE=$m timeout 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/pytest_mode$m.log 2>&1; echo "pinned to rung $m (tests/test_gpu_fuzz.py): $(tail -1 $O/pytest_mode$m.log)" | tee -a $O/summary.txt; done
PROXTV_XLINK=0 timeout 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/pytest_xlink0.log 2>&1; echo "xlink=0 (tests/test_gpu_fuzz.py): $(tail -1 $O/pytest_xlink0.log)" | tee -a $O/summary.txt
cat $O/build_id.txt
