cd $GRAFT_REPO_ROOT
for L in main c8rows c8both; do
  if [ $L = main ]; then unset PROXTV_LIB PROXTV_DEBUG_ALT_LIB; else export PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/proxtv_amd/build/lib_$L.so; fi
  echo "== $L"; timeout 100 python tools/small_images.py 128 256 512 1024 | tail -4
done
