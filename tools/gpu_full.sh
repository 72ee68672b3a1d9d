set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2g/pytest.log 2>&1 || (tail -40 gpurun_out/r2g/pytest.log; false)
tail -3 gpurun_out/r2g/pytest.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()"
GPU_MAX_HW_QUEUES=8 timeout -k 10 600 python3 tools/run_configs.py 2>&1 | grep "^|"
