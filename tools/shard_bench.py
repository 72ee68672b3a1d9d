"""Per-rank render time of a 1/world shard of the bench frame on ONE GPU, serial and with frames pipelined
over `depth` contexts/streams (predicts multi-GPU strong scaling without a multi-GPU node)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from terminalraytracer_amd import hip

scene = bench.build_scene()
for (w, h, world) in ((1920, 1080, 1), (1920, 1080, 2), (1920, 1080, 4), (1920, 1080, 8), (3840, 2160, 8)):
    cam = scene.camera.copy()
    cam[13] = 5 * float(w) / float(h)
    rs = hip.RowSet.shard(w, h, 0, world, 8)
    rows = hip.lib().trt_rowset_rows(C.byref(rs))
    line = f"{w}x{h} shard 1/{world}:"
    for depth in (1, 2, 3):
        ctxs, streams, fbs = [], [], []
        for i in range(depth):
            c = hip.Context(0)
            c.set_scene(scene)
            st = torch.cuda.Stream()
            c.set_stream(st.cuda_stream)
            ctxs.append(c), streams.append(st)
            fbs.append(torch.zeros(rows * w * 3, dtype=torch.float64, device="cuda:0"))
        n = 60
        for it in range(6 + n):
            if it == 6:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            i = it % depth
            ctxs[i].render_device(cam, rs, 8, 10, fbs[i].data_ptr(), fbs[i].numel() * 8)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        line += f"  depth {depth}: {dt * 1e3:.3f} ms/frame"
        for c in ctxs:
            c.set_stream(None)
            c.close()
    print(line)
