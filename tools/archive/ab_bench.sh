#!/bin/bash
# A/B the production kernel across alternative builds of libtrt_hip.so in ONE process-per-variant loop on one GPU box.
# usage: tools/archive/ab_bench.sh <rounds> <lib> [<lib> ...]     (interleaved rounds; prints ms/frame per round)
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    TRT_HIP_LIB=$PWD/$lib python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-verify --depth ${DEPTH:-2} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'round $r', 'render_kernel_ms %.4f' % d['one_frame_at_a_time']['render_kernel_ms'], 'Grays/s %.3f' % (d['value']/1e9), d['kernel_info']['vgprs'], d['kernel_info']['max_blocks_per_cu'])"
  done
done
