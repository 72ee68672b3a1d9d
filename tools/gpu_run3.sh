set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err || (tail -20 gpurun_out/r2d/bench.err; false)
timeout -k 10 300 python3 bench.py --no-cpu-baseline --animation 60 > gpurun_out/r2d/bench_anim.json 2> gpurun_out/r2d/bench_anim.err || (tail -20 gpurun_out/r2d/bench_anim.err; false)
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --check --no-cpu-baseline --steps 10 > gpurun_out/r2d/bench_gloo2.json 2> gpurun_out/r2d/bench_gloo2.err || (tail -20 gpurun_out/r2d/bench_gloo2.err; false)
python3 - <<'PY'
import json
for f in ("bench","bench_anim","bench_gloo2"):
    d=json.load(open(f"gpurun_out/r2d/{f}.json"))
    print(f, "value %.3e ms/step %.3f verified %s d1 %s"%(d["value"], d["ms_per_step"], d["verified"], d["one_frame_at_a_time"]))
PY
