#!/bin/bash
# the whole -m gpu suite (or a -k selection), as the driver runs it; log under gpurun_out/tests/
# usage: gpurun -- bash tools/gpu_tests.sh ["-k expression"]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tests
timeout -k 10 1000 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/tests/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/tests/gpu_tests.log
exit $rc
