# bench.py through its three multi-GPU code paths on one GPU: the C-ABI path (world 1), its fallback (forced), the 2-rank gloo rehearsal
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/paths
show='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
print("%.4e rays/s  ms/step %.3f  verified %s  n_gpus %d  path: %s" % (d["value"], d["ms_per_step"], d["verified"], d["n_gpus"], d["config"]["multi_gpu_path"]))'
timeout -k 10 300 python3 bench.py --no-cpu-baseline | python3 -c "$show" &&
TRT_BENCH_FAIL_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline | python3 -c "$show" &&
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --no-cpu-baseline --check --steps 10 2> gpurun_out/paths/gloo2.err | tee gpurun_out/paths/gloo2.out | python3 -c "$show"
