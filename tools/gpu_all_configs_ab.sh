#!/bin/bash
# A/B of library builds over everything the default bench.py line holds: headline, moving camera, configs 2 / 4 / 5 (frames verified).
# usage: gpurun -- bash tools/gpu_all_configs_ab.sh lib [lib ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for lib in "$@"; do
export TRT_AB_NAME=$lib
TRT_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ['TRT_AB_NAME'], 'c3 %.2f'%(d['value']/1e9), 'moving %.2f'%(d['value_moving_camera']/1e9), ' '.join('%s %.2f (d1 %.3f)'%(k, v['path_rays_per_s']/1e9, v['render_kernel_ms_one_at_a_time']) for k,v in d['configs'].items()), 'ok' if d['verified'] and all(v['verified'] for v in d['configs'].values()) else 'VERIFY FAILED')"
done; done
