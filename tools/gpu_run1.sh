set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
rocprofv3 -L > gpurun_out/r2a/counters.txt 2>&1 || true
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1 || (tail -30 gpurun_out/r2a/pytest.log; false)
tail -3 gpurun_out/r2a/pytest.log
timeout -k 10 300 python3 bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
timeout -k 10 300 python3 bench.py --animation 60 > gpurun_out/r2a/bench_anim.json 2> gpurun_out/r2a/bench_anim.err
cat gpurun_out/r2a/bench.json | cut -c1-1500
cat gpurun_out/r2a/bench_anim.json | cut -c1-900
