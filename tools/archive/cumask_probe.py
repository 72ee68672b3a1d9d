"""Throughput of the bench frame against the number of compute units a CU-masked stream leaves out (trt_reserve_cus).
Run on the GPU box: python tools/archive/cumask_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    import bench
    from terminalraytracer_amd import hip
    scene = bench.build_scene()
    w, h = bench.W, bench.H
    rs = hip.RowSet.whole(w, h)
    fb = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda:0")
    for reserve in (0, 1, 8, 32, 64, 128):
        c = hip.Context(0)
        c.set_scene(scene)
        if reserve:
            c.reserve_cus(reserve)
        for i in range(8):
            if i == 2:
                c.synchronize()
                t0 = time.perf_counter()
            c.render_device(scene.camera, rs, bench.BOUNCES, bench.SPP, fb.data_ptr(), fb.numel() * 8)
        c.synchronize()
        print(f"reserve {reserve}: {(time.perf_counter() - t0) / 6 * 1e3:.3f} ms/frame  (expected x{256 / (256 - reserve):.3f})", flush=True)
        c.close()


if __name__ == "__main__":
    main()
