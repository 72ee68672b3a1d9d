#!/bin/bash
# all BASELINE configs on one GPU (tools/run_configs.py) with the shading decoupled by the library's policy (-1) and never (0)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/configs
for comp in -1 0; do
  TRT_COMPACTION=$comp timeout -k 10 500 python3 tools/run_configs.py --depth 3 --min-seconds 0.6 > gpurun_out/configs/compaction_$comp.md 2> gpurun_out/configs/err.log || { tail -5 gpurun_out/configs/err.log; exit 1; }
  echo "== TRT_COMPACTION=$comp"; cat gpurun_out/configs/compaction_$comp.md
done
