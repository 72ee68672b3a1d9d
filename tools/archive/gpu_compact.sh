cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 tools/archive/compact_check.py > gpurun_out/compact_check.log 2>&1
echo rc=$? >> gpurun_out/compact_check.log
tail -30 gpurun_out/compact_check.log
