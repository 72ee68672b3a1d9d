cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "refraction or golden or whole" 2>&1 | tail -30
timeout -k 10 300 python3 bench.py --no-cpu-baseline | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench ms/step %.3f d1 render %.3f verified %s'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['verified']))"
