cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
(timeout -k 10 400 python3 tools/fuzz_campaign.py 200000 4000 compact 2>&1 | tail -3
timeout -k 10 300 python3 tools/fuzz_campaign.py 210000 3000 2>&1 | tail -3
timeout -k 10 250 python3 tools/fuzz_campaign.py 220000 2000 lights 2>&1 | tail -3) | tee gpurun_out/r2g/fuzz3.txt
