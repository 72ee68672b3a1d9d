set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2h/pytest.log 2>&1 || (tail -40 gpurun_out/r2h/pytest.log; false)
tail -3 gpurun_out/r2h/pytest.log
timeout -k 10 120 python3 tools/extremes.py 2>&1 | tail -10
timeout -k 10 300 python3 bench.py --no-cpu-baseline | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench ms/step %.3f d1 render %.3f verified %s'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['verified']))"
