set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
for g in 64,16 64,32 128,16 128,32 32,8 256,16; do
  for mode in "" "--animation 60"; do
    TRT_PATHGRID=$g timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-verify --steps 20 $mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('PATHGRID $g $mode', 'ms/step %.3f d1 render %.3f rounds/trace %.2f swept %d'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['diagnostics']['exact_test_rounds_per_trace'], d['diagnostics']['swept_traces']))"
  done
done
for lg in 128,64 256,128 256,64; do
  TRT_LIGHTGRID=$lg timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-verify --steps 20 --animation 60 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('LIGHTGRID $lg anim', 'ms/step %.3f d1 render %.3f rounds/trace %.2f'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['diagnostics']['exact_test_rounds_per_trace']))"
done
