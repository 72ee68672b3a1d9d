cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('torchrun N=1: value %.3e ms/step %.3f verified %s path %s'%(d['value'], d['ms_per_step'], d['verified'], d['config']['multi_gpu_path']))"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('gloo rehearsal N=2 on one GPU: ms/step %.3f verified %s depth %d path %s'%(d['ms_per_step'], d['verified'], d['frames_in_flight'], d['config']['multi_gpu_path']))"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dist or c_host" 2>&1 | tail -3
