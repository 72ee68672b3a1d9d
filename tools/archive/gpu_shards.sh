cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in 0 1; do echo "== TRT_COMPACTION=$m"; TRT_COMPACTION=$m timeout -k 10 400 python3 tools/shard_bench.py 2>&1 | grep shard; done
