#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X frame producer.

Workload (BASELINE.json metric / configs[2], SURVEY.md 8d "C3"): one 1920x1080 frame of the
SYNTH-v0 scene (64 random spheres, seed 1234, mixed reflectivity) + checker floor + 2 lights,
8 bounces, 10 rays per pixel, 256^2 procedural cubemap.  A step = one frame.  Scene, cubemap
and camera are resident on the GPU before the timed region; frames are rendered into device
memory ("off-screen framebuffer").

N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME frame is sharded by
interleaved 8-row tiles across the ranks and assembled on rank 0 with one RCCL gather per
frame ("scaling": "strong").

metric: path rays/s = trace calls issued by the bounce loop (primary + secondary rays,
TRT.c:1024) per second.  The per-frame ray count is deterministic; it is taken once from the
kernel's counting variant in an UNTIMED pass (the timed kernel carries no counters).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# every HIP stream of a process is dealt onto one of a few hardware queues (4 by default), and streams that share a queue
# run one after the other: the render streams of the frames in flight, the assembly stream and RCCL's must not collide
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H, SPHERES, BOUNCES, SPP, SKY_DIM, SEED = 1920, 1080, 64, 8, 10, 256, 1234
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # FP64 vector with FMA; 39.3 without (contraction must stay off here)


def build_scene():
    from terminalraytracer_amd import scenes as S
    cam = S.orbit_camera(1.0, W, H)
    return S.synth_scene(SPHERES, S.synth_sky(SKY_DIM), cam, seed=SEED)


def algorithmic_bytes(width, height, n, dim, ld, lp):
    # framebuffer store + cubemap read once + scene records
    return width * height * 24 + 6 * dim * dim * 3 + n * 72 + 352 + 48 * ld + 56 * lp


def cpu_baseline(scene):
    """Reference CPU path on this box's host cores: the genuine reference (oracle/_ref, compiled from the
    reference's own file) renders ONE frame of the same workload, single-threaded like the original
    (about 10-15 s of CPU work).  Falls back to the repo's restatement ("port") where the reference
    library is absent.  The all-core OpenMP port is reported beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support as T
    from terminalraytracer_amd import layout as L
    from terminalraytracer_amd import scenes as S
    threads = min(os.cpu_count() or 1, 64)
    t0 = time.perf_counter()
    _, st = T.oracle_render(scene, W, H, BOUNCES, SPP, threads=threads)  # also yields the ray count of the frame
    dt_all = time.perf_counter() - t0
    ref = os.path.join(ROOT, "oracle", "_ref", f"libtrtref_b{BOUNCES}_s{SPP}_w480_h280.so")
    out = {"unit": "path rays/s", "cores": 1, "sample": f"1 full {W}x{H} frame of the same scene/camera (the whole step)"}
    if os.path.exists(ref):
        lib = C.CDLL(ref)
        lib.project_scene.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Screen)]
        sc = scene.as_scene()
        screen, px = S.new_screen(W, H)
        t0 = time.perf_counter()
        lib.project_scene(C.byref(sc), C.byref(screen))
        dt = time.perf_counter() - t0
        out.update(kind="reference", value=st.path_rays / dt, seconds=dt)
    else:
        t0 = time.perf_counter()
        T.oracle_render(scene, W, H, BOUNCES, SPP, threads=1)
        dt = time.perf_counter() - t0
        out.update(kind="port", value=st.path_rays / dt, seconds=dt)
    out["port_all_cores"] = {"value": st.path_rays / dt_all, "cores": threads, "seconds": dt_all}
    out["path_rays_in_sample"] = st.path_rays
    return out


def measured_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic.json; FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 corrections applied)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        t = json.load(fh)
    return t.get("hbm_bytes_per_launch"), t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel", type=int, default=0, help="0 production, 1 reference-order (debug)")
    ap.add_argument("--units", type=int, default=0, help="work units of the production kernel: 0 auto, 1 pixels, 2 samples")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--check", action="store_true", help="rank 0 compares the assembled frame with a single-renderer frame (untimed)")
    ap.add_argument("--depth", type=int, default=0, help="frames in flight per GPU (1 = strictly one frame at a time; 0 = 2 on one GPU, 3 on several)")
    ap.add_argument("--reserve-cus", type=int, default=-1, help="compute units kept free of render workgroups so that the gather's kernels can "
                                                               "run beside them (-1 = 16 from 8 GPUs on, else 0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from terminalraytracer_amd import hip
    from terminalraytracer_amd.distributed import HipShardRenderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if args.depth <= 0:
        args.depth = 2 if world == 1 else 3
    if args.reserve_cus < 0:  # pays only where a rank's frame is short against the gather (tools/gather_sim.py)
        args.reserve_cus = 16 if world >= 8 else 0
    local = local % max(1, torch.cuda.device_count())  # rehearsals may put several ranks on one GPU
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local)

    scene = build_scene()
    r = HipShardRenderer(scene, W, H, rank, world, local, BOUNCES, SPP, depth=args.depth, reserve_cus=args.reserve_cus)
    r.for_each_context(lambda c: (c.set_kernel(args.kernel), c.set_work_units(args.units)))

    # untimed: per-frame ray counts of this rank's rows (counting variant of the kernel)
    r.ctx.enable_counters(True)
    r.render(scene.camera)
    torch.cuda.synchronize()
    path_rays, shadow_rays = r.ctx.read_counters()
    diag = r.ctx.read_diagnostics()
    r.ctx.enable_counters(False)
    for _ in range(args.depth - 1):  # bring every slot to the same state before warm-up
        r.render(scene.camera)
    torch.cuda.synchronize()
    counts = torch.tensor([path_rays, shadow_rays], dtype=torch.float64, device=f"cuda:{local}")
    if world > 1:
        dist.all_reduce(counts)
    path_total, shadow_total = float(counts[0].item()), float(counts[1].item())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.check:  # the sharded, gathered frame must equal the frame of one renderer, bit for bit
        frame = r.render(scene.camera)
        torch.cuda.synchronize()
        if rank == 0:
            with hip.Context(local) as single:
                single.set_scene(scene)
                whole = single.render_host(scene.camera, hip.RowSet.whole(W, H), BOUNCES, SPP)
            same = np.array_equal(frame.cpu().numpy().view(np.uint64), whole.view(np.uint64))
            print(f"CHECK sharded({world}) == single: {same}", file=sys.stderr)
            assert same
        for _ in range(args.depth - 1):
            r.render(scene.camera)
        barrier()

    for _ in range(args.warmup):
        r.render(scene.camera)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.render(scene.camera)
    barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=f"cuda:{local}")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    seconds = float(elapsed.item())

    kernel_ms = r.kernel_times(args.steps)  # HIP events on each launch stream, the timed launches only
    if rank == 0:
        ms_step = seconds / args.steps * 1e3
        kavg = float(np.mean(kernel_ms))
        rows = hip.lib().trt_rowset_rows(C.byref(r.sharded.rowset))
        alg = algorithmic_bytes(W, rows, SPHERES, SKY_DIM, scene.dir_lights.shape[0], scene.point_lights.shape[0])
        achieved = alg / (kavg * 1e-3) / 1e9
        flops = (path_total + shadow_total) / world * (25 * SPHERES + 17)  # SURVEY 8(d) reference op count, per rank
        out = {
            "metric": "path rays/s (primary+secondary) at 1920x1080, 64 spheres, 8 bounces",
            "value": path_total * args.steps / seconds,
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 1920x1080, SYNTH-v0 64 spheres seed 1234 + checker floor, "
                                   "1 directional + 1 point light, 8 bounces, 10 rays/pixel, 256^2 procedural cubemap, "
                                   "off-screen f64 framebuffer", "sharding": f"{world} x interleaved 8-row tiles, 1 gather/frame",
                       "kernel": {0: "persistent waves, synchronous rounds", 1: "reference-order", 2: "persistent state machine"}[args.kernel],
                       "frames_in_flight": args.depth, "reserved_cus": args.reserve_cus},
            "rays_per_frame": {"path": path_total, "shadow": shadow_total},
            "all_rays_per_s": (path_total + shadow_total) * args.steps / seconds,
            "kernel_ms_avg": kavg,
            "frames_in_flight": args.depth,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic()[0] if world == 1 else None,
                         "algorithmic_bytes": alg,
                         "note": "VALU-issue-bound path: algorithmic HBM bytes are ~1 B/ray (SURVEY 8d); see fp64_valu and DESIGN.md 5. "
                                 "achieved = algorithmic bytes per launch / average launch duration (HIP events); with "
                                 "frames_in_flight > 1 launches overlap, so a launch lasts longer than ms_per_step"},
            "fp64_valu": {"achieved_tflops_reference_opcount": flops / (kavg * 1e-3) / 1e12, "peak_tflops_fma": FP64_PEAK_TFLOPS,
                          "peak_tflops_no_fma": FP64_PEAK_TFLOPS / 2,
                          "note": "reference op count 25*N+17 per trace; the kernel does fewer ops than that (FP32 culling)"},
            "valu_issue": (measured_traffic()[1] or {}).get("valu"),  # committed PMC summary of this same command
            "kernel_info": r.ctx.kernel_info(),
            # rounds kernel (id 0): one round = one path ray per lane (+ one shadow ray per light for the lanes that hit something)
            "diagnostics": dict(diag, path_lane_utilisation=path_rays / max(1, 64 * diag["wave_loop_trips"]),
                                shadow_lane_utilisation=shadow_rays / max(1, 64 * diag["wave_loop_trips"] * max(1, scene.dir_lights.shape[0] + scene.point_lights.shape[0])),
                                exact_test_rounds_per_trace=diag["phase2_rounds"] / max(1, path_rays + shadow_rays) * 64)
            if args.kernel == 0 and args.units != 1 else
            dict(diag, lane_utilisation=(path_rays + shadow_rays) / max(1, 64 * diag["wave_loop_trips"])),
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(scene)
            except Exception as e:  # the checker libraries are built by __graft_entry__.build(); never lose the GPU line over them
                out["cpu_baseline"] = {"value": None, "unit": "path rays/s", "cores": 1, "kind": "port", "sample": "not measured",
                                       "error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
