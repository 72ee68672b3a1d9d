# stage shares of the PLAIN rounds (the diagnostic build of the decoupled kernel spills: its shares are not representative)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
export TRT_COMPACTION=0
echo "=== C3"; TRT_HIP_LIB=$PWD/build/stamp.so timeout -k 10 120 python3 tools/stamp_config.py 1920 1080 64 8 2>&1 | grep -v "^$"
echo "=== C5"; TRT_HIP_LIB=$PWD/build/stamp.so timeout -k 10 120 python3 tools/stamp_config.py 1920 1080 256 12 2>&1 | grep -v "^$"
