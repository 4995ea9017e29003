cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for L in main w8 w2 u2 u8 u17; do
  if [ $L = main ]; then unset PROXTV_LIB PROXTV_DEBUG_ALT_LIB; else export PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/proxtv_amd/build/lib_$L.so; fi
  echo "== $L: $(timeout 100 python tools/time_one.py c2 0.1 | tail -1)"
done
done
