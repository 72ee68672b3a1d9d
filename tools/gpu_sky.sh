cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for dim in 256 1024 2048; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --sky-dim $dim 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('sky_dim $dim', 'ms/step %.3f d1 render %.3f value %.3e alg bytes %d'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['value'], d['roofline']['algorithmic_bytes']))"
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dist or c_host or production_stages or path_ray or whole" 2>&1 | tail -3
