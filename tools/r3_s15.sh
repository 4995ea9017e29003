cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for x in 1 0; do
  echo "== xlink=$x"
  PROXTV_XLINK=$x timeout 100 python tools/time_one.py c2 0.1 | tail -1
  PROXTV_XLINK=$x timeout 100 python tools/time_one.py c2 0.5 | tail -1
  PROXTV_XLINK=$x timeout 100 python tools/time_one.py c2 0.7 | tail -1
  PROXTV_XLINK=$x timeout 100 python tools/small_images.py 256 1024 | tail -2
done
done
