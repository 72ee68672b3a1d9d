#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X frame producer.

Default workload (BASELINE.json metric / configs[2], SURVEY.md 8d "C3"): one 1920x1080 frame of the
SYNTH-v0 scene (64 random spheres, seed 1234, mixed reflectivity) + checker floor + 2 lights,
8 bounces, 10 rays per pixel, 256^2 procedural cubemap.  A step = one frame.  Scene, cubemap
and camera are resident on the GPU before the timed region; frames are rendered into device
memory ("off-screen framebuffer").

--animation F (BASELINE configs[4], "C5"): 256 spheres, 12 bounces, the reference's orbiting camera
(TerminalRayTracer.c:1327-1336) at t = f/60, f = 0..F-1, a NEW camera every step; reports sustained
frames/s next to path rays/s.

N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME frame is sharded by
interleaved 8-row tiles across the ranks and assembled on rank 0 with one RCCL gather per
frame ("scaling": "strong").  Row tiles, the RCCL communicator, the gather and the frame pipeline
live in the library behind the C-ABI (trt_dist_*, include/trt_hip.h section 3): the same calls a C
host makes; torch.distributed only carries the communicator id to the ranks and times the run.
--backend gloo selects the PyTorch-level rehearsal path (several ranks sharing one GPU).

metric: path rays/s = trace calls issued by the bounce loop (primary + secondary rays,
TRT.c:1024) per second.  The per-frame ray count is deterministic; it is taken once from the
kernel's counting variant in an UNTIMED pass (the timed kernel carries no counters).

The frame that is timed is CHECKED: after the timed loop rank 0 hashes the last frame (FNV-1a-64 of the
double framebuffer, untimed) and compares it with the hash the GENUINE reference produced for the same
scene and camera (tests/golden/golden_full.json, made by tests/golden/make_golden_full.py); a wrong
pixel anywhere fails the run (exit code 3, "verified": false in the line).
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
# this pool's host driver offers dmabuf IPC only: without it RCCL's peer buffers fail with hipIpcGetMemHandle: invalid argument
# (exported by the image already; kept here for a launcher that builds its own environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SPP, SKY_DIM, SEED = 10, 256, 1234
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GOLDEN = os.path.join(ROOT, "tests", "golden")

WORKLOADS = {
    # name: (width, height, spheres, bounce limit, golden case of the frame at orbit time 1.0 / of animation frames)
    "c2": dict(width=1920, height=1080, spheres=8, bounces=4, golden={1.0: "c2_1080p_8sph_b4"},
               text="BASELINE configs[1]: 1920x1080, SYNTH-v0 8 spheres + checker floor, 2 lights, 4 bounces, 10 rays/pixel"),
    "c4": dict(width=3840, height=2160, spheres=64, bounces=8, golden={1.0: "c4_2160p_64sph_b8"},
               text="BASELINE configs[3] on ONE GPU: 3840x2160, SYNTH-v0 64 spheres + checker floor, 2 lights, 8 bounces, 10 rays/pixel"),
    "c3": dict(width=1920, height=1080, spheres=64, bounces=8, golden={1.0: "c3_1080p_64sph_b8"},
               orbit_golden={0: "c3_1080p_64sph_b8_f0", 19: "c3_1080p_64sph_b8_f19", 59: "c3_1080p_64sph_b8_f59"},
               text="BASELINE configs[2]: 1920x1080, SYNTH-v0 64 spheres seed 1234 + checker floor, 1 directional + 1 point light, "
                    "8 bounces, 10 rays/pixel, 256^2 procedural cubemap, off-screen f64 framebuffer"),
    "c5": dict(width=1920, height=1080, spheres=256, bounces=12, golden={0: "c5_1080p_256sph_b12_f0", 59: "c5_1080p_256sph_b12_f59"},
               text="BASELINE configs[4]: 1920x1080, SYNTH-v0 256 spheres seed 1234 + checker floor, 1 directional + 1 point light, "
                    "12 bounces, 10 rays/pixel, 256^2 procedural cubemap, orbiting camera t = f/60 (a new camera every frame), "
                    "off-screen f64 framebuffer"),
}


def stored_camera(width, height, t=1.0):
    """The reference's own orbit camera at time t (tests/golden/cameras.npz, dumped from the genuine reference: libm's
    sin/cos may differ in the last place between boxes) with screen_width = 5*W/H."""
    d = np.load(os.path.join(GOLDEN, "cameras.npz"))
    i = int(np.argmin(np.abs(d["t"] - t)))
    assert abs(d["t"][i] - t) < 1e-12, t
    cam = d["camera"][i].copy()
    cam[13] = 5 * float(width) / float(height)
    return cam


def animation_cameras(width, height, frames):
    d = np.load(os.path.join(GOLDEN, "cameras_anim.npz"))
    cams = d["camera"][np.arange(frames) % len(d["t"])].copy()
    cams[:, 13] = 5 * float(width) / float(height)
    return [c for c in cams]


def build_scene(workload="c3", sky_dim=SKY_DIM):
    from terminalraytracer_amd import scenes as S
    w = WORKLOADS[workload]
    return S.synth_scene(w["spheres"], S.synth_sky(sky_dim), stored_camera(w["width"], w["height"], 1.0), seed=SEED)


# names kept for the tools/ scripts
W, H, SPHERES, BOUNCES = 1920, 1080, 64, 8


def golden_hash(name):
    with open(os.path.join(GOLDEN, "golden_full.json")) as fh:
        for c in json.load(fh)["cases"]:
            if c["name"] == name:
                return c
    raise KeyError(name)


def algorithmic_bytes(width, height, n, dim, ld, lp):
    # framebuffer store + cubemap read once + scene records
    return width * height * 24 + 6 * dim * dim * 3 + n * 72 + 352 + 48 * ld + 56 * lp


def cpu_baseline(scene, width, height, bounces, full_frame):
    """Reference CPU path on this box's host cores: the genuine reference (oracle/_ref, compiled from the
    reference's own file) renders a bounded sample of the same workload, single-threaded like the original:
    the whole 1080p frame for the 64-sphere workload (10-15 s), the same scene and camera at 480x270 for the
    256-sphere one (a full frame would take minutes).  Falls back to the repo's restatement ("port") where the
    reference library is absent.  The all-core OpenMP port is reported beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support as T
    from terminalraytracer_amd import layout as L
    from terminalraytracer_amd import scenes as S
    w, h = (width, height) if full_frame else (480, 270)
    threads = min(os.cpu_count() or 1, 64)
    t0 = time.perf_counter()
    _, st = T.oracle_render(scene, w, h, bounces, SPP, threads=threads)  # also yields the ray count of the sample
    dt_all = time.perf_counter() - t0
    ref = os.path.join(ROOT, "oracle", "_ref", f"libtrtref_b{bounces}_s{SPP}_w480_h280.so")
    out = {"unit": "path rays/s", "cores": 1, "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
           "sample": f"1 {w}x{h} frame of the same scene/camera" + (" (the whole step)" if full_frame else " (1/16 of the step's pixels)")}
    if os.path.exists(ref):
        lib = C.CDLL(ref)
        lib.project_scene.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Screen)]
        sc = scene.as_scene()
        screen, px = S.new_screen(w, h)
        t0 = time.perf_counter()
        lib.project_scene(C.byref(sc), C.byref(screen))
        dt = time.perf_counter() - t0
        out.update(kind="reference", value=st.path_rays / dt, seconds=dt)
    else:
        t0 = time.perf_counter()
        T.oracle_render(scene, w, h, bounces, SPP, threads=1)
        dt = time.perf_counter() - t0
        out.update(kind="port", value=st.path_rays / dt, seconds=dt)
    out["port_all_cores"] = {"value": st.path_rays / dt_all, "cores": threads, "seconds": dt_all}
    out["path_rays_in_sample"] = st.path_rays
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


FP64_PEAK_TFLOPS, FP64_PEAK_WITHOUT_FMA = 78.6, 39.3  # MI355X FP64 vector: half the 157.3 TFLOP/s FP32 vector figure of MI355X_MICROARCH.md


def compute_executed(profile, kernel_ms):
    """SURVEY 8d's bounding roofline, on EXECUTED work: FP64 wave instructions of one launch by class (COMMITTED rocprofv3 PMC
    pass: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes, an FMA = 2 flops) over the render kernel's own duration measured in
    THIS run.  frac is against the FMA peak; issue_frac is the share of the FP64 pipe's issue slots in use (an add or a multiply
    takes the slot an FMA would: with -ffp-contract=off, which bit-exactness demands, most slots carry one flop)."""
    if not profile or "flops_per_launch" not in profile:
        return None
    seconds = kernel_ms * 1e-3
    achieved = profile["flops_per_launch"] / seconds / 1e12
    return {"flops": profile["flops_per_launch"], "achieved_tflops": achieved, "peak": FP64_PEAK_TFLOPS, "peak_without_fma": FP64_PEAK_WITHOUT_FMA,
            "frac": achieved / FP64_PEAK_TFLOPS, "frac_of_peak_without_fma": achieved / FP64_PEAK_WITHOUT_FMA,
            "issue_frac": profile["fp64_lane_slots_per_launch"] / seconds / (FP64_PEAK_WITHOUT_FMA * 1e12),
            "fp64_share_of_valu_instructions": profile["fp64_share_of_valu_instructions"], "kernel_ms": kernel_ms,
            "basis": profile["basis"], "source": profile["source"]}


def default_depth(workload, world=1):
    """Frames in flight per GPU when --depth is not given: 3, and 2 for config 5 on one GPU -- its 256-thread workgroups (four per
    CU, the plain rounds with patches) of three frames get in each other's way: 20.69 against 20.27 G path rays/s over 300 frames,
    twice (profiles/r04/f_ab_log.txt); configs 2 / 3 / 4 are best with 3 (config 3: 34.74 against 33.84 with 2).  On several GPUs
    the gather needs the third slot (see --depth)."""
    return 2 if workload == "c5" and world == 1 else 3


def other_configs(hip, host, torch, local, depth_given, tile_rows):
    """BASELINE configs 2, 4 (on one GPU) and 5 through the same calls and the same frames-in-flight loop as the headline, each
    frame checked against the hash the GENUINE reference produced for it (tests/golden/golden_full.json): what the driver sees
    of them.  Ray counts: the reference's own (golden_full.json) for the stills, the kernel's counting variant (untimed) for the
    60 cameras of the orbit."""
    out = {}
    for name, steps, frames in (("c2", 30, 0), ("c4", 12, 0), ("c5", 60, 60)):
        wl = WORKLOADS[name]
        w, h, b = wl["width"], wl["height"], wl["bounces"]
        scene = build_scene(name)
        cams = animation_cameras(w, h, frames) if frames else [scene.camera]
        t_setup = time.perf_counter()
        depth = depth_given or default_depth(name)
        d = hip.Dist(local, scene, None, 0, 1, w, h, tile_rows=tile_rows, frames_in_flight=depth)
        setup_s = time.perf_counter() - t_setup
        try:
            if frames:
                ctx0, rows = d.context(0), hip.RowSet.whole(w, h)
                scratch = torch.zeros(h * w * 3, dtype=torch.float64, device=f"cuda:{local}")
                ctx0.enable_counters(True)
                path = []
                for cam in cams:
                    ctx0.render_device(cam, rows, b, SPP, scratch.data_ptr(), scratch.numel() * 8)
                    path.append(ctx0.read_counters()[0])
                ctx0.enable_counters(False)
                del scratch
            else:
                path = [golden_hash(wl["golden"][1.0])["path_rays"]]
            for i in range(3):
                d.render(cams[i % len(cams)], b, SPP)
            d.synchronize()
            t0 = time.perf_counter()
            last = None
            for i in range(steps):
                last = d.render(cams[i % len(cams)], b, SPP)
            d.synchronize()
            seconds = time.perf_counter() - t0
            checks = []
            probes = [((steps - 1) % len(cams), last)]
            if frames and (steps - 1) % len(cams) != 0:
                probes.append((0, None))
            for index, frame in probes:
                key = index if frames else 1.0
                if key not in wl["golden"]:
                    continue
                if frame is None:
                    frame = d.render(cams[index], b, SPP)
                want = golden_hash(wl["golden"][key])
                got = host.fnv1a64(d.fetch(frame))
                checks.append({"frame": wl["golden"][key], "fnv": got, "reference_fnv": want["fb_fnv"], "ok": got == want["fb_fnv"]})
            rays = float(sum(path[i % len(path)] for i in range(steps)))
            variant = d.context(0).render_variant()
            # strictly one frame at a time: the render kernel's own duration (HIP events on its stream)
            ctx0, rows = d.context(0), hip.RowSet.whole(w, h)
            scratch = torch.zeros(h * w * 3, dtype=torch.float64, device=f"cuda:{local}")
            for i in range(6):
                ctx0.render_device(cams[i % len(cams)], rows, b, SPP, scratch.data_ptr(), scratch.numel() * 8)
                ctx0.synchronize()
            render_ms = float(np.mean(ctx0.render_kernel_times(5)[0]))
            del scratch
            out[name] = {"workload": wl["text"], "steps": steps, "frames_in_flight": depth, "ms_per_frame": seconds / steps * 1e3,
                         "frames_per_s": steps / seconds, "path_rays_per_s": rays / seconds, "path_rays_per_frame": rays / steps,
                         "verified": bool(checks) and all(c["ok"] for c in checks), "verification": checks,
                         "kernel": "render_rounds_kernel<false, false, %s%s>" % ("true" if variant["decoupled"] else "false",
                                                                                  ", true" if d.context(0).path_patches()[0] else ""),
                         "render_kernel_ms_one_at_a_time": render_ms, "scene_setup_s": setup_s,
                         # one copy of the scene's tables per device, whatever the number of frame slots (trt_share_scene)
                         "scene_tables": d.context(0).scene_info()}
            whole = committed_profile() or {}
            prof = whole.get("config5") if name == "c5" else None
            if prof:
                out[name]["compute_executed"] = compute_executed(prof.get("compute_executed"), render_ms)
                out[name]["valu"] = prof.get("valu")
                # config 5's memory side (its tables leave every cache): COMMITTED PMC passes of this very command
                alg5 = algorithmic_bytes(w, h, wl["spheres"], SKY_DIM, 1, 1)
                out[name]["roofline"] = {"bound": "hbm", "achieved": alg5 / (render_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": alg5 / (render_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg5,
                                         "traffic": prof.get("hbm_bytes_per_launch"), "traffic_over_algorithmic": prof.get("traffic_over_algorithmic"),
                                         "traffic_breakdown": prof.get("breakdown_bytes"), "traffic_source": prof.get("source"),
                                         "profile": profile_identity(whole), "kernel_ms": render_ms}
        finally:
            d.close()
    return out


def kernel_source_hash():
    """What the committed profile constants are tied to: a hash of everything the device code is made of (the kernel sources and
    the compile flags of the Makefile).  tools/make_traffic_json.py stores it when a PMC pass is turned into profiles/traffic.json;
    a kernel that has changed since shows as profile_matches_build = false in the line."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "terminalraytracer_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hpp", ".h", ".hip")):
            with open(os.path.join(csrc, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    with open(os.path.join(ROOT, "Makefile")) as fh:
        h.update("".join(l for l in fh if l.startswith("HIPFLAGS")).encode())
    return h.hexdigest()[:16]


def profile_identity(prof):
    """{head, source hash of the kernel the constants were measured on, whether that is the kernel of this tree}"""
    if not prof:
        return None
    here = kernel_source_hash()
    return {"profile_head": prof.get("head"), "profile_kernel_source_hash": prof.get("kernel_source_hash"), "kernel_source_hash": here,
            "profile_matches_build": prof.get("kernel_source_hash") == here}


def committed_profile():
    """HBM bytes per launch and VALU counters from the COMMITTED rocprofv3 PMC passes of this same command
    (profiles/traffic.json: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 corrections applied).  They are not
    measured in this run: the record says which commit's kernel they were taken with."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed frames (default 200: 0.28 s of rendering, so that the last frames' tails do not weigh on the mean; 60 with --animation)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--animation", type=int, default=0, metavar="F",
                    help="config 5: 256 spheres, 12 bounces, orbit cameras t = f/60 for f < F, a new camera every step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 2, 4 and 5 (rendered and verified after the headline on one GPU)")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of the timed frame against the reference's hash")
    ap.add_argument("--no-moving-camera", action="store_true", help="skip the headline's second loop (the reference's orbiting camera)")
    ap.add_argument("--kernel", type=int, default=0, help="0 production, 1 reference-order (debug)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--rccl-stand-in", action="store_true", help="TEST HOOK: bind the library TRT_RCCL_LIB names in RCCL's place "
                    "(tests/rccl_stub.cpp: several ranks on one GPU); without this flag the variable is ignored")
    ap.add_argument("--allow-fallback", action="store_true", help="if the product path (trt_dist_* behind the C-ABI) fails on any rank, time the "
                    "PyTorch-level gather instead and say so in the line; WITHOUT this flag (the default) such a failure ends every rank with "
                    "exit code 5 and the reason on stderr: a number for a path that is not the product is not a number")
    ap.add_argument("--check", action="store_true", help="rank 0 compares the assembled frame with a single-renderer frame (untimed)")
    ap.add_argument("--depth", type=int, default=0, help="frames in flight per GPU (1 = strictly one frame at a time; 0 = the default: 3, and 2 for config 5 on one GPU: default_depth(); the "
                                                         "next frames' workgroups fill the CUs that a frame's tail leaves idle -- on one GPU "
                                                         "1.86 / 1.67 / 1.65 / 1.66 ms per frame with 1 / 2 / 3 / 4 -- and on several GPUs the "
                                                         "gather of frame f can only get on the machine when the persistent workgroups of frame "
                                                         "f+1 retire, so with two slots frame f+2 would wait for it)")
    ap.add_argument("--tile-rows", type=int, default=8, help="rows per interleaved tile of the row sharding")
    ap.add_argument("--sky-dim", type=int, default=SKY_DIM, help="cubemap face size (the frame is only verified at the default, 256)")
    ap.add_argument("--reserve-cus", type=int, default=0, help="compute units kept free of render workgroups so that the gather's kernels can "
                                                              "run beside them (a guess until an 8-GPU run has been made: default 0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from terminalraytracer_amd import hip, host
    from terminalraytracer_amd.distributed import HipShardRenderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    auto_depth = args.depth <= 0
    workload = "c5" if args.animation > 0 else "c3"
    if auto_depth:
        args.depth = default_depth(workload, world)
    wl = WORKLOADS[workload]
    width, height, bounces = wl["width"], wl["height"], wl["bounces"]
    if args.steps <= 0:
        args.steps = args.animation if args.animation > 0 else 200
    local = local % max(1, torch.cuda.device_count())  # rehearsals may put several ranks on one GPU
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local)

    scene = build_scene(workload, args.sky_dim)
    if args.sky_dim != SKY_DIM:
        args.no_verify = True  # the reference's hashes are for the 256^2 cubemap
    cameras = animation_cameras(width, height, args.animation) if args.animation > 0 else [scene.camera]
    camera_of = lambda step: cameras[step % len(cameras)]  # noqa: E731
    # --backend gloo: several ranks may share one GPU, which RCCL cannot do: the PyTorch-level rehearsal path -- unless
    # TRT_RCCL_LIB names a stand-in for RCCL (tests/rccl_stub.cpp), with which the product path itself runs on one GPU
    stand_in = os.environ.get("TRT_RCCL_LIB") if args.rccl_stand_in else None
    if stand_in:  # test hook: the library honours TRT_RCCL_LIB only when told to, before its first use of RCCL
        hip.dist_allow_rccl_override(True)
    rehearsal = world > 1 and args.backend != "nccl" and not stand_in
    carrier = f"cuda:{local}" if args.backend == "nccl" else "cpu"  # where torch.distributed's own tensors live
    fallback_reason = None
    made = {}

    def torch_level_renderer():  # PyTorch-level sharding: trt_render_device per rank, torch.distributed gather
        r = HipShardRenderer(scene, width, height, rank, world, local, bounces, SPP, tile_rows=args.tile_rows, depth=args.depth,
                             reserve_cus=args.reserve_cus)
        return r, [slot["ctx"] for slot in r.slots], r.sharded.rowset, r.render, (lambda frame: frame.cpu().numpy())

    if rehearsal:  # over gloo several ranks may share one GPU, which RCCL cannot do
        r, contexts, rowset, render, fetch = torch_level_renderer()
    else:          # the product path: trt_dist_* behind the C-ABI
        uid = None
        if world > 1:
            box = torch.zeros(128, dtype=torch.uint8, device=carrier)
            if rank == 0:
                box.copy_(torch.frombuffer(bytearray(hip.dist_unique_id()), dtype=torch.uint8))
            dist.broadcast(box, 0)
            uid = bytes(box.cpu().numpy().tobytes())
        # The library's communicator meets its peers here for the first time in the life of the process: create it and push one
        # frame through it on a side thread with a deadline, and let the ranks agree on the outcome.  If any rank failed (an
        # error, or no frame within the deadline), EVERY rank exits non-zero with the reason on stderr -- unless --allow-fallback
        # was given: then every rank falls back to the PyTorch-level gather and the line says so.
        import threading
        problem = []

        def first_contact():
            try:
                if os.environ.get("TRT_BENCH_FAIL_DIST"):  # test hook for the fallback
                    raise RuntimeError("TRT_BENCH_FAIL_DIST is set")
                t_setup = time.perf_counter()
                d = hip.Dist(local, scene, uid, rank, world, width, height, tile_rows=args.tile_rows, frames_in_flight=args.depth,
                             reserved_cus=args.reserve_cus)
                made["setup_s"] = time.perf_counter() - t_setup
                made["dist"] = d
                frame = d.render(camera_of(0), bounces, SPP)
                d.synchronize()
                if rank == 0:
                    d.fetch(frame)
                made["rccl_ranks"] = d.comm_ranks()  # what RCCL itself says of the communicator (ncclCommCount)
                if world > 1 and made["rccl_ranks"] != world:
                    raise RuntimeError(f"the communicator has {made['rccl_ranks']} ranks, WORLD_SIZE is {world}")
            except Exception as e:  # noqa: BLE001
                problem.append(f"{type(e).__name__}: {e}")

        worker = threading.Thread(target=first_contact, daemon=True)
        worker.start()
        worker.join(timeout=180.0 if world > 1 else None)
        if worker.is_alive():
            # The thread is stuck inside the library (ncclCommInitRank or a GPU wait) and keeps its communicator and its streams on
            # this GPU: nothing timed beside it would mean anything, and interpreter exit could block in its teardown.  Say so and
            # leave -- the peers' next collective fails and they leave as well.  (An EXCEPTION, below, is a clean failure: then
            # every rank falls back.)
            print(f"bench: rank {rank}: no frame through trt_dist_* within 180 s; giving up", file=sys.stderr, flush=True)
            os._exit(4)
        ok = torch.tensor([0 if problem else 1], dtype=torch.int32, device=carrier)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            r = made["dist"]
            contexts = [r.context(i) for i in range(args.depth)]
            rowset = hip.RowSet.shard(width, height, rank, world, args.tile_rows)
            render = lambda cam: r.render(cam, bounces, SPP)     # noqa: E731  -> device address of the frame on rank 0
            fetch = r.fetch
        else:
            fallback_reason = problem[0] if problem else "another rank failed"
            if "dist" in made:
                made["dist"].close()
            if not args.allow_fallback:
                # the product path failed: no line, no number.  Every rank has taken part in the agreement above, so every rank
                # leaves here, in step, through the interpreter's normal exit (nothing is re-executed).
                print(f"bench: rank {rank}: the C-ABI multi-GPU path (trt_dist_*) failed: {fallback_reason}; no fallback was allowed "
                      "(--allow-fallback), giving up", file=sys.stderr, flush=True)
                if world > 1:
                    dist.destroy_process_group()
                sys.exit(5)
            print(f"bench: rank {rank}: C-ABI multi-GPU path unavailable ({fallback_reason}); falling back to the PyTorch-level gather", file=sys.stderr)
            r, contexts, rowset, render, fetch = torch_level_renderer()
    for c in contexts:
        c.set_kernel(args.kernel)
    ctx0 = contexts[0]
    local_rows = hip.lib().trt_rowset_rows(C.byref(rowset))
    scratch = torch.zeros(max(1, local_rows) * width * 3, dtype=torch.float64, device=f"cuda:{local}")  # for the untimed passes

    # untimed: ray counts of this rank's rows for every distinct camera (counting variant of the kernel)
    ctx0.enable_counters(True)
    path_per_cam, shadow_per_cam, diag = [], [], None
    for cam in cameras:
        ctx0.render_device(cam, rowset, bounces, SPP, scratch.data_ptr(), scratch.numel() * 8)
        p, s = ctx0.read_counters()
        path_per_cam.append(p)
        shadow_per_cam.append(s)
        diag = diag or ctx0.read_diagnostics()
    ctx0.enable_counters(False)
    torch.cuda.synchronize()
    counts = torch.tensor([path_per_cam, shadow_per_cam], dtype=torch.float64, device=carrier)
    if world > 1:
        dist.all_reduce(counts)
    path_cam, shadow_cam = counts[0].cpu().numpy(), counts[1].cpu().numpy()
    path_timed = float(sum(path_cam[i % len(cameras)] for i in range(args.steps)))
    shadow_timed = float(sum(shadow_cam[i % len(cameras)] for i in range(args.steps)))

    def barrier():
        # the library's own work first: its communicator still has grouped send/recv queued on its stream, and torch's barrier is
        # a collective on ANOTHER communicator of the same GPU -- two communicators with work in flight, issued in an order that
        # differs from rank to rank, is the documented NCCL/RCCL deadlock
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: strictly one frame at a time on ONE context, every launch bracketed by HIP events on its stream; the render
    # kernel's own duration (the roofline's denominator) and the ordered-mean kernel's behind it
    d1 = min(args.steps, 20)
    for i in range(2 + d1):
        ctx0.render_device(camera_of(i), rowset, bounces, SPP, scratch.data_ptr(), scratch.numel() * 8)
        ctx0.synchronize()
    render_ms, reduce_ms = ctx0.render_kernel_times(d1)
    render_ms_avg, reduce_ms_avg = float(np.mean(render_ms)), float(np.mean(reduce_ms))
    barrier()

    if args.check:  # the sharded, gathered frame must equal the frame of one renderer, bit for bit
        frame = render(scene.camera)
        torch.cuda.synchronize()
        if rank == 0:
            with hip.Context(local) as single:
                single.set_scene(scene)
                whole = single.render_host(scene.camera, hip.RowSet.whole(width, height), bounces, SPP)
            same = np.array_equal(fetch(frame).view(np.uint64), whole.view(np.uint64))
            print(f"CHECK sharded({world}) == single: {same}", file=sys.stderr)
            assert same
        barrier()

    for i in range(args.warmup):
        render(camera_of(i))
    barrier()
    launched_before = [c.launch_count() for c in contexts]
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = render(camera_of(i))
    barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=carrier)
    # the same loop by the DEVICE's clock: from the start of the first timed launch to the end of the last one (HIP events on the
    # contexts' streams; the frames take turns over the contexts, so the earliest start and the latest end are found pairwise)
    device_span_ms = None
    try:
        took = [(c, a, c.launch_count()) for c, a in zip(contexts, launched_before) if c.launch_count() > a]
        if took and sum(b - a for _, a, b in took) == args.steps:
            device_span_ms = max(c0.launch_span_ms(a0, c1, b1 - 1) for c0, a0, _ in took for c1, _, b1 in took)
    except hip.TrtError:
        device_span_ms = None  # more launches per context than the event ring holds (--steps in the thousands)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    seconds = float(elapsed.item())
    # per rank: what its shard took to render and what its part of the gather took (the slots' most recent frames; HIP events)
    per_rank = None
    if hasattr(r, "frame_times"):
        mine = torch.tensor(list(r.frame_times()), dtype=torch.float64, device=carrier)
        every = [torch.zeros_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(every, mine)
        else:
            every = [mine]
        per_rank = [{"rank": i, "render_ms": round(float(t[0]), 4), "gather_ms": round(float(t[1]), 4)} for i, t in enumerate(every)]

    # untimed: the frame the timed loop produced last (and, for the animation, frame 59) against the GENUINE reference's hash
    verified, checks = None, []
    if not args.no_verify:
        probes = [((args.steps - 1) % len(cameras), False)]
        if args.animation > 59 and (args.steps - 1) % len(cameras) != 59:
            probes.append((59, True))
        for index, again in probes:
            key = index if args.animation > 0 else 1.0
            if key not in wl["golden"]:
                continue
            frame = render(cameras[index]) if again else last  # a collective: every rank renders, rank 0 gets the frame
            torch.cuda.synchronize()
            if rank == 0:
                want = golden_hash(wl["golden"][key])
                got = host.fnv1a64(fetch(frame))
                checks.append({"frame": wl["golden"][key], "fnv": got, "reference_fnv": want["fb_fnv"], "ok": got == want["fb_fnv"]})
            barrier()
        if rank == 0:
            verified = bool(checks) and all(c["ok"] for c in checks)

    # The headline as the reference USES the path: main() moves the camera every frame (TRT.c:1327-1339), so the eye's two
    # candidate tables are rebuilt inside every frame.  The same loop -- same scene, steps, frames in flight -- fed the reference's 60
    # orbit cameras, the last frame checked against the genuine reference's hash of that very frame.  (The orbit sees other
    # frames than the still, so the ray counts are those of its own cameras.)
    moving = None
    if world == 1 and workload == "c3" and args.animation == 0 and args.kernel == 0 and args.sky_dim == SKY_DIM and not args.no_moving_camera:
        orbit = animation_cameras(width, height, 60)
        ctx0.enable_counters(True)
        orbit_path = []
        for cam in orbit[:max(1, min(args.steps, 60))]:
            ctx0.render_device(cam, rowset, bounces, SPP, scratch.data_ptr(), scratch.numel() * 8)
            orbit_path.append(ctx0.read_counters()[0])
        ctx0.enable_counters(False)
        for i in range(args.warmup):
            render(orbit[i % 60])
        barrier()
        t1 = time.perf_counter()
        last_moving = None
        for i in range(args.steps):
            last_moving = render(orbit[i % 60])
        barrier()
        moving_seconds = time.perf_counter() - t1
        rays = float(sum(orbit_path[i % len(orbit_path)] for i in range(args.steps)))
        moving = {"value": rays / moving_seconds, "unit": "rays/s", "ms_per_step": moving_seconds / args.steps * 1e3, "steps": args.steps,
                  "camera": "the reference's orbit, t = f/60: a new camera and new eye tables every frame (TRT.c:1327-1339)",
                  "verification": [], "verified": None}
        if not args.no_verify:
            index = (args.steps - 1) % 60
            probes = [(index, last_moving)] if index in wl["orbit_golden"] else [(59, None), (0, None)]
            for index, frame in probes:
                frame = frame if frame is not None else render(orbit[index])
                torch.cuda.synchronize()
                want = golden_hash(wl["orbit_golden"][index])
                got = host.fnv1a64(fetch(frame))
                moving["verification"].append({"frame": wl["orbit_golden"][index], "fnv": got, "reference_fnv": want["fb_fnv"], "ok": got == want["fb_fnv"]})
            moving["verified"] = all(c["ok"] for c in moving["verification"])

    if rank == 0:
        ms_step = seconds / args.steps * 1e3
        rows = local_rows
        nd, npt = scene.dir_lights.shape[0], scene.point_lights.shape[0]
        alg = algorithmic_bytes(width, rows, wl["spheres"], args.sky_dim, nd, npt)
        achieved = alg / (render_ms_avg * 1e-3) / 1e9
        path_mean = float(np.mean(path_cam))
        prof = committed_profile() if (world == 1 and workload == "c3") else None
        variant = ctx0.render_variant()  # the shading decoupled from the owning lane or the plain rounds
        variant_name = "render_rounds_kernel<false, false, %s%s>" % ("true" if variant["decoupled"] else "false",
                                                                    ", true" if ctx0.path_patches()[0] else "")
        out = {
            "metric": "path rays/s (primary+secondary) at 1920x1080, 64 spheres, 8 bounces" if workload == "c3" else
                      "path rays/s (primary+secondary) at 1920x1080, 256 spheres, 12 bounces, orbiting camera",
            "value": path_timed / seconds,
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["text"] if args.sky_dim == SKY_DIM else wl["text"].replace("256^2", f"{args.sky_dim}^2"), "sharding": f"{world} x interleaved {args.tile_rows}-row tiles, 1 gather/frame",
                       "multi_gpu_path": "PyTorch-level rehearsal (gloo)" if rehearsal else
                                         ("FALLBACK (--allow-fallback): PyTorch-level gather (torch.distributed " + args.backend + "); trt_dist_* failed: " + fallback_reason
                                          if fallback_reason is not None else
                                          "C-ABI trt_dist_* (send/recv gather inside the library) over a STAND-IN for RCCL: " + stand_in if stand_in and world > 1 else
                                          "C-ABI trt_dist_* (RCCL send/recv gather inside the library)"),
                       "kernel": {0: "persistent waves, synchronous rounds" + (", shading decoupled from the owning lane" if variant["decoupled"] else ""),
                                  1: "reference-order"}[args.kernel],
                       "workgroup_threads": variant["workgroup_threads"],
                       "frames_in_flight": args.depth, "reserved_cus": args.reserve_cus,
                       "camera": "static: the eye's two candidate tables are built once, before the timed region (an orbit rebuilds them every "
                                 "frame, ~0.06 ms: see configs.c5)" if args.animation == 0 else "a new camera every frame (the eye's tables are rebuilt per frame, inside the timed region)"},
            "frames_per_s": args.steps / seconds,
            # the same loop under the reference's real usage: a new camera every frame (its own ray counts; see `moving_camera`)
            "value_moving_camera": moving["value"] if moving else None,
            "moving_camera": moving,
            "scene_setup_s": made.get("setup_s") if not rehearsal else None,
            "scene_tables": ctx0.scene_info(),
            "per_rank": per_rank,
            "rccl_library": hip.dist_rccl_library() if world > 1 and not rehearsal else None,
            # the ranks of the communicator the product path created, as the library bound above reports them (ncclCommCount)
            "rccl_ranks": made.get("rccl_ranks") if world > 1 and not rehearsal and fallback_reason is None else None,
            "rays_per_frame": {"path": path_mean, "shadow": float(np.mean(shadow_cam))},
            "all_rays_per_s": (path_timed + shadow_timed) / seconds,
            "frames_in_flight": args.depth,
            "verified": verified, "verification": checks,
            # one frame at a time (no overlap between consecutive frames): render kernel + ordered mean, HIP events per launch
            "one_frame_at_a_time": {"render_kernel_ms": render_ms_avg, "reduce_kernel_ms": reduce_ms_avg, "launches": d1,
                                    "path_rays_per_s": path_mean / ((render_ms_avg + reduce_ms_avg) * 1e-3)},
            # the timed loop by the device's own clock (HIP events, first launch's start to last launch's end): frames overlap, so this is
            # what a frame costs the GPU in the pipelined loop; it must not exceed ms_per_step (which adds the host's part of the loop)
            "device_ms_per_step": device_span_ms / args.steps if device_span_ms is not None else None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # what actually bounds the kernel: the issue of vector instructions, most of them per-ray FP64
                         # geometry.  COMMITTED counters of this kernel (profiles/traffic.json), not measured in this run.
                         "bound_actual": "valu_issue",
                         "valu_busy": ((prof or {}).get("valu") or {}).get("valu_busy_measured"),
                         "fp64_share": ((prof or {}).get("compute_executed") or {}).get("fp64_share_of_valu_instructions"),
                         "issue_frac": (compute_executed((prof or {}).get("compute_executed"), render_ms_avg) or {}).get("issue_frac"),
                         "traffic": (prof or {}).get("hbm_bytes_per_launch"),
                         "traffic_source": (prof or {}).get("source"),
                         "profile": profile_identity(prof),
                         "algorithmic_bytes": alg, "kernel": variant_name, "kernel_ms": render_ms_avg,
                         "achieved_by_step": alg / (ms_step * 1e-3) / 1e9,
                         "traffic_over_algorithmic": ((prof or {}).get("hbm_bytes_per_launch") or 0) / alg if prof else None,
                         "compute_executed": compute_executed((prof or {}).get("compute_executed"), render_ms_avg),
                         "note": "achieved = algorithmic bytes of a frame (SURVEY 8d) / the render kernel's own average duration, HIP events "
                                 "on its stream, frames launched strictly one at a time (one_frame_at_a_time); the timed region keeps "
                                 "frames_in_flight frames in flight, so consecutive launches overlap there.  The path is VALU-issue-bound: "
                                 "~1 algorithmic byte per ray, frac << 1 by nature (SURVEY 0, DESIGN.md 5); `traffic` and `valu` are "
                                 "COMMITTED rocprofv3 PMC results (profiles/traffic.json), not measured in this run"},
            # SURVEY 8d, compute basis: what the REFERENCE's algorithm spends on these rays (every trace_ray call tests all N spheres
            # and the ground: 25 N + 17 FP64 flops), over the render kernel's own duration.  The candidate tables skip most of those
            # tests, so this is work avoided as much as work done and may exceed the FP64 vector peak; what the hardware executed is
            # in `valu` (instructions, measured VALU busy).
            "compute": {"basis": "reference-equivalent FP64 flops = trace_ray calls x (25 N + 17)",
                        "flops_per_frame": (path_mean + float(np.mean(shadow_cam))) * (25 * wl["spheres"] + 17),
                        # NOT a rate of executed work and not to be set against a peak: the reference's flops for these rays over the
                        # kernel's duration.  The tables avoid most of them (roofline.compute_executed is what the hardware did).
                        "reference_equivalent_tflops": (path_mean + float(np.mean(shadow_cam))) * (25 * wl["spheres"] + 17) / (render_ms_avg * 1e-3) / 1e12,
                        "unit": "TFLOP/s of the REFERENCE's algorithm, mostly avoided rather than executed",
                        "fp64_vector_peak_for_orientation": FP64_PEAK_TFLOPS,
                        "peak_source": "MI355X FP64 vector: half the 157.3 TFLOP/s FP32 vector figure of MI355X_MICROARCH.md; -ffp-contract=off halves it again"},
            "valu": (prof or {}).get("valu"),
            "kernel_info": ctx0.kernel_info(),
            # one round = one path ray per lane (+ one shadow ray per light for the lanes that hit something)
            "diagnostics": dict(diag, path_lane_utilisation=path_per_cam[0] / max(1, 64 * diag["wave_loop_trips"]),
                                shadow_lane_utilisation=shadow_per_cam[0] / max(1, 64 * diag["shading_passes"] * max(1, nd + npt)),
                                exact_test_rounds_per_trace=diag["phase2_rounds"] / max(1, path_per_cam[0] + shadow_per_cam[0]) * 64)
            if args.kernel == 0 else diag,
        }
        if world == 1 and workload == "c3" and not args.no_configs and args.kernel == 0 and args.sky_dim == SKY_DIM:
            r.close()  # the headline's buffers make room
            r = None
            try:
                out["configs"] = other_configs(hip, host, torch, local, 0 if auto_depth else args.depth, args.tile_rows)
            except Exception as e:  # never lose the headline over the rest
                out["configs"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(scene, width, height, bounces, full_frame=(workload == "c3"))
            except Exception as e:  # the checker libraries are built by __graft_entry__.build(); never lose the GPU line over them
                out["cpu_baseline"] = {"value": None, "unit": "path rays/s", "cores": 1, "kind": "port", "sample": "not measured",
                                       "error": f"{type(e).__name__}: {e}"}
        # the line's "verified" covers everything in it: the headline's frame, the moving-camera frame and the other configs; a
        # config that could not be measured at all (an exception) counts as unverified too
        if moving and moving["verified"] is False:
            verified = False
            checks = checks + [v for v in moving["verification"] if not v["ok"]]
        configs = out.get("configs")
        if configs is not None and not args.no_verify:
            if "error" in configs:
                verified = False
                checks = checks + [{"frame": "configs c2/c4/c5", "ok": False, "error": configs["error"]}]
            for c in configs.values():
                if isinstance(c, dict) and c.get("verified") is False:
                    verified = False
                    checks = checks + [v for v in c.get("verification", []) if not v["ok"]]
        out["verified"], out["verification"] = verified, checks
        print(json.dumps(out))
    if r is not None:
        r.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and verified is False:
        print("bench.py: the timed frame differs from the reference's frame: " + json.dumps(checks), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
