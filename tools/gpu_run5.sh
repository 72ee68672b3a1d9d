cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for mode in "" "--animation 60"; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline $mode | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench $mode value %.4e ms/step %.3f d1 render %.3f verified %s'%(d['value'], d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['verified']))"
done
