set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
for g in 64,32 64,48 64,64 96,48 128,64 64,96; do
  for mode in "" "--animation 60"; do
    TRT_PATHGRID=$g timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-verify --steps 20 $mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('PATHGRID $g $mode', 'ms/step %.3f d1 render %.3f rounds/trace %.2f swept %d'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['diagnostics']['exact_test_rounds_per_trace'], d['diagnostics']['swept_traces']))" | tee -a gpurun_out/r2d/sweep2.txt
  done
done
