# stage shares of the PLAIN rounds (the diagnostic build of the decoupled kernel spills: its shares are not representative)
# usage: gpurun -- bash tools/archive/gpu_stamp.sh ["eye,sphere,m" ...]    (table resolutions / patches to compare; default: the library's)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stamps
export TRT_COMPACTION=0
for pg in "${@:-default}"; do
  [ "$pg" = default ] && unset TRT_PATHGRID || export TRT_PATHGRID=$pg
  echo "=== C3 tables $pg"; TRT_HIP_LIB=$PWD/build/stamp.so timeout -k 10 120 python3 tools/archive/stamp_config.py 1920 1080 64 8 2>&1 | grep -v "^$"
  echo "=== C5 tables $pg"; TRT_HIP_LIB=$PWD/build/stamp.so timeout -k 10 120 python3 tools/archive/stamp_config.py 1920 1080 256 12 2>&1 | grep -v "^$"
done | tee gpurun_out/stamps/stamps.txt
