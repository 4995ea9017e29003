# The repair kernel's jump: tools/repair_scan_check.py (lively-then-flat fibres through the pinned rungs against the sequential walk) on this
# tree's build and, if one is there, on an alternative build (proxtv_amd/build/lib_alt.so -- the round-4 variant with the bounded scan was
# checked this way: profiles/r04_repair_scan_check.txt), then the module that holds the regression test.
#   gpurun -- 'bash tools/repair_scan_session.sh'
OUT=gpurun_out/rscan; mkdir -p $OUT
{ echo "## this tree's build ($(python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print(build.build_id())"))"
  timeout 30 python tools/repair_scan_check.py 1 2>&1 | grep -v amdgpu | tail -8
  if [ -f proxtv_amd/build/lib_alt.so ]; then echo "## proxtv_amd/build/lib_alt.so"
    PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/proxtv_amd/build/lib_alt.so timeout 30 python tools/repair_scan_check.py 1 2>&1 | grep -v amdgpu | tail -8; fi; } > $OUT/check.txt 2>&1
cat $OUT/check.txt
timeout 60 python -m pytest tests/test_gpu_chunk_repair.py -m gpu -q -x > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
