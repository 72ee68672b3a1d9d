#!/bin/bash
# bench.py over one setting at a time on the GPU box, config 3 and config 5 (--animation 60), a line per run.
# usage: gpurun -- bash tools/archive/gpu_bench_sweep.sh <what> <value> [<value> ...]
#   what = depth     frames in flight (--depth)                       e.g. depth 1 2 3 4
#          pathgrid  TRT_PATHGRID="eye,sphere[,patches]"              e.g. pathgrid 64,32 64,48 64,32,2
#          lightgrid TRT_LIGHTGRID="directional,point"                e.g. lightgrid 128,64 256,64
#          skydim    cubemap face size (--sky-dim, frames unverified) e.g. skydim 256 1024 2048
#          compaction TRT_COMPACTION=-1|0|1                           e.g. compaction 0 1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
what=$1; shift
mkdir -p gpurun_out/sweep
for v in "$@"; do
  for mode in "" "--animation 60"; do
    extra=""; unset TRT_PATHGRID TRT_LIGHTGRID TRT_COMPACTION
    case $what in
      depth) extra="--depth $v";;
      pathgrid) export TRT_PATHGRID=$v;;
      lightgrid) export TRT_LIGHTGRID=$v;;
      skydim) extra="--sky-dim $v";;
      compaction) export TRT_COMPACTION=$v;;
    esac
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-configs $extra $mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$what $v $mode', 'ms/step %.3f  render kernel one at a time %.3f ms  %.2f G path rays/s  verified %s  exact rounds/trace %.2f  swept %d  %s'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['value']/1e9, d['verified'], d['diagnostics']['exact_test_rounds_per_trace'], d['diagnostics']['swept_traces'], d['roofline']['kernel']))"
  done
done | tee gpurun_out/sweep/$what.txt
