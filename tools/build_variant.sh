# A/B builds: the sweep units recompiled with extra flags and linked with the default build's other objects into
# proxtv_amd/build/lib_<name>.so ; run with  PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=proxtv_amd/build/lib_<name>.so  (tools/ab_run.py)
#   bash tools/build_variant.sh <name> [extra hipcc flags...]
name=$1; shift
cd $(dirname $0)/..
python -m proxtv_amd.build --variant "$name" -- "$@"
