"""CPU-side cost per frame of the multi-GPU loop (HipShardRenderer.render: launches, events, one gather, one index_select),
measured on one GPU with a 1-rank RCCL group and a frame so small that the GPU is never the limit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import bench
from terminalraytracer_amd.distributed import HipShardRenderer
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
scene = bench.build_scene()
r = HipShardRenderer(scene, 1920, 16, 0, 1, 0, 1, 1, depth=2)   # 16 rows, 1 bounce, 1 ray per pixel
s = r.slots[0]["frame"]
gathered = [torch.zeros_like(s.shard)]
def frame():
    r.render(scene.camera)
    dist.gather(s.shard, gathered, dst=0)          # what a root adds per frame at world > 1
    torch.index_select(gathered[0], 0, torch.arange(s.shard.shape[0], device="cuda:0"), out=gathered[0].clone())
for _ in range(20):
    frame()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    frame()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"CPU enqueue time per frame {1e6 * (t1 - t0) / n:.0f} us; with the GPU drained {1e6 * (t2 - t0) / n:.0f} us")
r.close(); dist.destroy_process_group()
