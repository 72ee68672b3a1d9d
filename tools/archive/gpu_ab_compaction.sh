cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for r in 1 2; do
for comp in -1 0; do
  TRT_COMPACTION=$comp timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-configs --no-moving-camera 2> gpurun_out/ab/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('compaction $comp', '%.3f G'%(d['value']/1e9), 'ms/step %.4f'%d['ms_per_step'], 'device_ms/step', d.get('device_ms_per_step'), 'd1 %.4f'%d['one_frame_at_a_time']['render_kernel_ms'], d['verified'], d['config']['workgroup_threads'], d['roofline'].get('bound_actual'), d['roofline'].get('valu_busy'), d['roofline'].get('issue_frac'))" || { tail -5 gpurun_out/ab/err.log; exit 1; }
done
done
