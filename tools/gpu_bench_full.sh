#!/bin/bash
# the default bench.py run as the driver makes it (headline + configs 2/4/5 + CPU baseline), timed; counters the profiler offers
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench
( time timeout -k 10 900 python bench.py > gpurun_out/bench/default.json 2> gpurun_out/bench/default.err ) 2> gpurun_out/bench/default.time
rc=$?
tail -3 gpurun_out/bench/default.err; cat gpurun_out/bench/default.time
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench/default.json').read().strip().splitlines()[-1])
print('headline', d['value']/1e9, 'G', d['ms_per_step'], 'ms verified', d['verified'])
for k,v in (d.get('configs') or {}).items():
    print(k, v if not isinstance(v,dict) else {kk:v[kk] for kk in ('ms_per_frame','path_rays_per_s','verified','kernel','scene_setup_s') if kk in v})
print('cpu', d.get('cpu_baseline'))
PY
timeout 120 rocprofv3 -L > gpurun_out/bench/counters.txt 2>&1; grep -i "f64\|F32\|FLOP\|SQ_INSTS_VALU" gpurun_out/bench/counters.txt | head -40
exit $rc
