cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
(timeout -k 10 500 python3 tools/fuzz_campaign.py 300000 6000 2>&1 | tail -2
timeout -k 10 300 python3 tools/fuzz_campaign.py 310000 3000 wide 2>&1 | tail -2
timeout -k 10 300 python3 tools/fuzz_campaign.py 320000 3000 compact 2>&1 | tail -2) | tee gpurun_out/r2h/fuzz4.txt
