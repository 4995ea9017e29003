cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final2; rm -rf $O; mkdir -p $O
for m in 1 3 5; do PROXTV_CHUNK_MODE=$m timeout 600 python -m pytest tests/test_gpu_boundary.py -m gpu -q > $O/boundary_mode$m.log 2>&1; echo "test_gpu_boundary.py pinned to rung $m: $(tail -1 $O/boundary_mode$m.log)" | tee -a $O/summary.txt; done
PROXTV_XLINK=0 timeout 600 python -m pytest tests/test_gpu_boundary.py -m gpu -q > $O/boundary_xlink0.log 2>&1; echo "test_gpu_boundary.py xlink=0: $(tail -1 $O/boundary_xlink0.log)" | tee -a $O/summary.txt
