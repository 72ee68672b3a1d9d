"""Does the gather of frame f overlap the rendering of later frames?  One GPU renders a 1/8 shard of the bench frame through
the same slot/event structure as HipShardRenderer; the gather is stood in for by a busy-wait kernel of fixed length on the
main stream (a gather occupies the stream, not the CUs).  Prints ms per frame for several numbers of frames in flight."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from terminalraytracer_amd import hip

# calibrate torch.cuda._sleep
torch.cuda._sleep(1000); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize()
CLOCK = 20_000_000 / (time.perf_counter() - t0)
print(f"_sleep clock {CLOCK / 1e6:.0f} MHz")
scene = bench.build_scene()
w, h = bench.W, bench.H
for world in (8, 2):
    rs = hip.RowSet.shard(w, h, 0, world, 8)
    rows = hip.lib().trt_rowset_rows(C.byref(rs))
    for gather_ms in (0.0, 0.2, 0.4):
        line = f"1/{world} shard, stand-in gather {gather_ms:.1f} ms:"
        for depth, reserve in ((1, 0), (2, 0), (3, 0), (4, 0), (2, 8), (3, 8), (3, 16)):
            slots = []
            for i in range(depth):
                c = hip.Context(0); c.set_scene(scene)
                if reserve:
                    c.reserve_cus(reserve)
                    st = torch.cuda.ExternalStream(c.stream_ptr())
                else:
                    st = torch.cuda.Stream(); c.set_stream(st.cuda_stream)
                slots.append({"ctx": c, "stream": st, "fb": torch.zeros(rows * w * 3, dtype=torch.float64, device="cuda:0"),
                              "rendered": torch.cuda.Event(), "consumed": torch.cuda.Event()})
            main = torch.cuda.Stream()   # not the null stream: CU-masked streams are 'blocking' streams and would serialise with it
            spin = int(gather_ms * 1e-3 * CLOCK)   # torch.cuda._sleep counts ticks of the clock calibrated below
            def frame(k):
                s = slots[k % depth]
                if k >= depth:
                    s["stream"].wait_event(s["consumed"])
                s["ctx"].render_device(scene.camera, rs, bench.BOUNCES, bench.SPP, s["fb"].data_ptr(), s["fb"].numel() * 8)
                s["rendered"].record(s["stream"])
                main.wait_event(s["rendered"])
                if spin:
                    with torch.cuda.stream(main):
                        torch.cuda._sleep(spin)
                s["consumed"].record(main)
            for k in range(2 * depth + 4):
                frame(k)
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for k in range(2 * depth + 4, 2 * depth + 4 + n):
                frame(k)
            torch.cuda.synchronize()
            line += f"  d{depth}" + (f"/r{reserve}" if reserve else "") + f": {(time.perf_counter() - t0) / n * 1e3:.3f}"
            for s in slots:
                if not reserve:
                    s["ctx"].set_stream(None)
                s["ctx"].close()
        print(line, flush=True)
