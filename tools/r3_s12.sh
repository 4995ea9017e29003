cd $GRAFT_REPO_ROOT
for L in nopipe pipeA; do
  for q in 0 1 2; do
  echo "== $L blocks_per_wg=$q"
  for c in c3 wprox1; do PROXTV_BLOCKS_PER_WG=$q PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=$PWD/proxtv_amd/build/lib_$L.so timeout 200 python tools/time_one.py $c 2>&1 | tail -1; done
  done
done
