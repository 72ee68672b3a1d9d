cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pg in 64,32 0,0 64,32 0,0; do
echo "TRT_PATHGRID=$pg"
TRT_PATHGRID=$pg GPU_MAX_HW_QUEUES=8 timeout -k 10 600 python3 tools/run_configs.py 2>&1 | grep "^| C"
done
