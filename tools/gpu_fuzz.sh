#!/bin/bash
# Randomised parity campaigns on the GPU box (tools/fuzz_campaign.py): scenes x modes as arguments "first count mode" ...
# usage: gpurun -- bash tools/gpu_fuzz.sh "100000 5000" "120000 2500 wide" "300000 3000 patches" ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
for job in "$@"; do
  echo "== fuzz_campaign.py $job"
  # the campaign's progress lines go to a file under gpurun_out/ (a run that writes nothing for seven minutes is taken to be hung)
  timeout -k 10 1000 python3 -u tools/fuzz_campaign.py $job 2>&1 | tee -a gpurun_out/fuzz/progress.log | grep -v "^\.\.\. " | tail -6
done | tee gpurun_out/fuzz/fuzz.txt
grep -q MISMATCH gpurun_out/fuzz/fuzz.txt && exit 1
exit 0
