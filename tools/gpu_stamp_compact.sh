cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in 0 1; do
echo "=== C3 compaction $m"; TRT_COMPACTION=$m TRT_HIP_LIB=$PWD/build/stamp.so timeout -k 10 120 python3 tools/stamp_config.py 1920 1080 64 8 2>&1 | grep -v "^$"
done
