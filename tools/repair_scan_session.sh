# The repair kernel's jump (round 4, fixed at its end): the build before the fix and this one on lively-then-flat fibres, then the module
# that holds the regression test.   gpurun -- 'bash tools/repair_scan_session.sh'
OUT=gpurun_out/rscan; mkdir -p $OUT
{ echo "## build before the fix (proxtv_amd/build/lib_old.so = a32f6193052f)"
  [ -f proxtv_amd/build/lib_old.so ] && PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/proxtv_amd/build/lib_old.so timeout 30 python tools/repair_scan_check.py 1 2>&1 | grep -v amdgpu | tail -8   # (a copy of that build's library, if one was kept)
  echo "## this build ($(python -c "import sys; sys.path.insert(0,'.'); from proxtv_amd import build; print(build.build_id())"))"
  timeout 30 python tools/repair_scan_check.py 1 2>&1 | grep -v amdgpu | tail -8; } > $OUT/check.txt 2>&1
cat $OUT/check.txt
timeout 60 python -m pytest tests/test_gpu_chunk_repair.py -m gpu -q -x > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
