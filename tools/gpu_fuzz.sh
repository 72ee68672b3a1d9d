cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
(timeout -k 10 300 python3 tools/fuzz_campaign.py 100000 5000 2>&1 | tail -3
timeout -k 10 250 python3 tools/fuzz_campaign.py 120000 2500 wide 2>&1 | tail -3
timeout -k 10 250 python3 tools/fuzz_campaign.py 140000 2500 lights 2>&1 | tail -3
timeout -k 10 250 python3 tools/fuzz_campaign.py 160000 3000 refract 2>&1 | tail -3
timeout -k 10 120 python3 tools/extremes.py 2>&1 | tail -10) | tee gpurun_out/r2f/fuzz2.txt
