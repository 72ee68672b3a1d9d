cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
(timeout -k 10 900 python3 tools/fuzz_campaign.py 400000 16000 2>&1 | tail -1
timeout -k 10 600 python3 tools/fuzz_campaign.py 420000 10000 compact 2>&1 | tail -1
timeout -k 10 400 python3 tools/fuzz_campaign.py 440000 5000 wide 2>&1 | tail -1
timeout -k 10 400 python3 tools/fuzz_campaign.py 450000 5000 lights 2>&1 | tail -1
timeout -k 10 400 python3 tools/fuzz_campaign.py 460000 4000 refract 2>&1 | tail -1) | tee gpurun_out/r2i/fuzz5.txt
