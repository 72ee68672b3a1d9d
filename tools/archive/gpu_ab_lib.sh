# usage: bash tools/archive/gpu_ab_lib.sh build/a.so build/b.so ...   (the shipped library is always the first contestant)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for lib in terminalraytracer_amd/libtrt_hip.so "$@"; do
  for mode in "" "--animation 60"; do
  TRT_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 $mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib $mode', 'ms/step %.3f d1 render %.3f verified %s vgprs %d rounds/trace %.2f'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['verified'], d['kernel_info']['vgprs'], d['diagnostics']['exact_test_rounds_per_trace']))"
  done
done
done
