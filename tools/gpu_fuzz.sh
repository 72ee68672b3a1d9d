cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
(timeout -k 10 400 python3 tools/fuzz_campaign.py 20000 7000 2>&1 | tail -4
timeout -k 10 300 python3 tools/fuzz_campaign.py 40000 3500 wide 2>&1 | tail -4
timeout -k 10 300 python3 tools/fuzz_campaign.py 60000 3500 lights 2>&1 | tail -4
timeout -k 10 120 python3 tools/extremes.py 2>&1 | tail -12) | tee gpurun_out/r2f/fuzz.txt
