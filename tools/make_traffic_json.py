"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py --json) of `bench.py --depth 1`:
HBM bytes per launch (FETCH_SIZE / WRITE_SIZE from separate passes, gfx950 correction of MI355X_MICROARCH.md applied) and the
VALU counters.  The FP64 instruction counters (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64) give the EXECUTED FP64 work of a launch, for config 3 and,
from a second summary, for config 5.
The record names the kernel it was measured on: `head` (git HEAD of the tree the profile was taken from -- run this script in that
tree) and `kernel_source_hash` (bench.kernel_source_hash(): the kernel sources + compile flags); bench.py compares the latter with
its own tree and says profile_matches_build = false when the constants are stale.
usage: python tools/make_traffic_json.py <pmc_summary.json> <tag> [<c5_pmc_summary.json>] [<round dir, default r04>] > profiles/traffic.json"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

s = json.load(open(sys.argv[1]))
tag = sys.argv[2]
c5 = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
rnd = sys.argv[4] if len(sys.argv) > 4 else "r04"


def executed_fp64(counters, kernel_name, where):
    """FP64 work the hardware EXECUTED in one launch: wave-level instructions by class x 64 lanes (a wave instruction occupies the
    FP64 pipe for all 64 lanes whatever its exec mask), an FMA counted as two flops."""
    add, mul, fma, trans = (counters[k]["mean"] for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    valu = counters["SQ_INSTS_VALU"]["mean"]
    f32 = sum(counters[k]["mean"] for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32") if k in counters)
    return {"source": where + ", " + kernel_name.replace("trt::", "") + ", means over the dispatches",
            "wave_instructions": {"add_f64": add, "mul_f64": mul, "fma_f64": fma, "trans_f64": trans, "all_valu": valu, "add_mul_fma_f32": f32},
            "fp64_share_of_valu_instructions": (add + mul + fma + trans) / valu,
            "flops_per_launch": 64.0 * (add + mul + trans) + 128.0 * fma,
            "fp64_lane_slots_per_launch": 64.0 * (add + mul + fma + trans),
            "basis": "issued: 64 lanes per wave instruction, FMA = 2 flops; x lane_activity for the flops of active lanes only"}

rk = next(k for k in s if "render_rounds_kernel<false" in k)
dk = next(k for k in s if "reduce_samples_kernel" in k)
R, D = s[rk], s[dk]
m = lambda d, k: d[k]["mean"]  # noqa: E731
KB = 1000.0  # the counters are in KB
render_w, render_f = m(R, "WRITE_SIZE") * KB, m(R, "FETCH_SIZE") * KB
reduce_w, reduce_f = m(D, "WRITE_SIZE") * KB, m(D, "FETCH_SIZE") * KB
total = render_w + 2 * render_f + reduce_w + 2 * reduce_f
gui = m(R, "GRBM_GUI_ACTIVE") / 8.0
try:
    head = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=bench.ROOT, text=True).strip()
    dirty = bool(subprocess.check_output(["git", "status", "--porcelain", "--", "terminalraytracer_amd/csrc", "Makefile"], cwd=bench.ROOT, text=True).strip())
except Exception:  # noqa: BLE001
    head, dirty = None, None
out = {
    "head": head, "head_dirty_kernel_sources": dirty, "kernel_source_hash": bench.kernel_source_hash(),
    "workload": "bench.py default (1920x1080, 64 spheres, 8 bounces, 10 rays/pixel), production kernels " + rk.replace("trt::", "") + " + reduce_samples_kernel, --depth 1",
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace; profiles/{rnd}/{tag}_pmc_summary.txt (tools/profile_round.sh); means over the dispatches",
    "per_launch_KB": {"render_rounds_kernel": {"FETCH_SIZE": m(R, "FETCH_SIZE"), "WRITE_SIZE": m(R, "WRITE_SIZE")},
                      "reduce_samples_kernel": {"FETCH_SIZE": m(D, "FETCH_SIZE"), "WRITE_SIZE": m(D, "WRITE_SIZE")}},
    "correction": "MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read: the reduce kernel streams the 497.7 MB scratch "
                  "and the counter reads 243 MB, i.e. the x2 correction is confirmed on this very kernel; it is applied to both kernels' reads (for the render kernel's scattered "
                  "4-8 B table and texel reads that is an upper bound). WRITE_SIZE is taken as is: the reduce kernel's 48.6 MB equals the framebuffer.",
    "hbm_bytes_per_launch": round(total),
    "breakdown_bytes": {"scratch_write_render": round(render_w), "scratch_read_reduce": round(2 * reduce_f), "framebuffer_write_reduce": round(reduce_w),
                        "cubemap_and_table_reads_render_upper_bound": round(2 * render_f)},
    "algorithmic_bytes_per_launch": 50951112,
    "note": "traffic is many times the algorithmic bytes BY DESIGN: work units are single samples and each sample's colour (24 B) passes through a sample-major scratch so that "
            "the per-pixel mean is formed in the reference's order. It costs 0.10 ms (reduce kernel) of a 2.0 ms frame. The candidate tables (7 MB of list cells at 64 spheres) are "
            "read once per trace with 8-byte loads. The frame is bound by VALU issue and dependent-load latency, not by HBM.",
    "valu": {
        "source": f"profiles/{rnd}/{tag}_pmc_summary.txt (rocprofv3 --pmc SQ_* / GRBM_GUI_ACTIVE), " + rk.replace("trt::", "") + ", means over the dispatches",
        "SQ_INSTS_VALU_per_launch": m(R, "SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": m(R, "SQ_ACTIVE_INST_VALU"), "SQ_WAVES": m(R, "SQ_WAVES"), "simds": 1024,
        "GRBM_GUI_ACTIVE_per_xcd": gui,
        "valu_busy_measured": 4.0 * m(R, "SQ_ACTIVE_INST_VALU") / (1024 * gui),
        "valu_busy_formula": "4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE per XCD): the gfx94x VALUBusy expression of rocprofv3 -L (SQ_ACTIVE_INST_VALU counts quad-cycles)",
        "lane_activity": m(R, "SQ_THREAD_CYCLES_VALU") / (64.0 * m(R, "SQ_ACTIVE_INST_VALU")),
        "wave_cycles_waiting_any_frac": m(R, "SQ_WAIT_ANY") / m(R, "SQ_WAVE_CYCLES"),
        "wave_cycles_waiting_inst_frac": m(R, "SQ_WAIT_INST_ANY") / m(R, "SQ_WAVE_CYCLES"),
    },
}
if "SQ_INSTS_VALU_FMA_F64" in R:
    out["compute_executed"] = executed_fp64(R, rk, f"profiles/{rnd}/{tag}_pmc_summary.txt (rocprofv3 --pmc SQ_INSTS_VALU_*_F64)")
if c5:
    ck = next(k for k in c5 if "render_rounds_kernel<false" in k)
    C = c5[ck]
    gui5 = m(C, "GRBM_GUI_ACTIVE") / 8.0
    alg5 = bench.algorithmic_bytes(1920, 1080, 256, 256, 1, 1)
    mem5 = {}
    if "FETCH_SIZE" in C and "WRITE_SIZE" in C:
        dk5 = next(k for k in c5 if "reduce_samples_kernel" in k)
        D5 = c5[dk5]
        rw, rf, dw, df = m(C, "WRITE_SIZE") * KB, m(C, "FETCH_SIZE") * KB, m(D5, "WRITE_SIZE") * KB, m(D5, "FETCH_SIZE") * KB
        total5 = rw + 2 * rf + dw + 2 * df
        mem5 = {"per_launch_KB": {"render_rounds_kernel": {"FETCH_SIZE": m(C, "FETCH_SIZE"), "WRITE_SIZE": m(C, "WRITE_SIZE")},
                                  "reduce_samples_kernel": {"FETCH_SIZE": m(D5, "FETCH_SIZE"), "WRITE_SIZE": m(D5, "WRITE_SIZE")}},
                "hbm_bytes_per_launch": round(total5), "algorithmic_bytes_per_launch": alg5, "traffic_over_algorithmic": total5 / alg5,
                "breakdown_bytes": {"scratch_write_render": round(rw), "scratch_read_reduce": round(2 * df), "framebuffer_write_reduce": round(dw),
                                    "cubemap_and_table_reads_render_upper_bound": round(2 * rf)},
                "source": f"profiles/{rnd}/{tag}_c5_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; x2 on reads as for config 3)"}
    out["config5"] = {
        "workload": "bench.py --animation 60 --depth 1 (1920x1080, 256 spheres, 12 bounces, orbit), " + ck.replace("trt::", ""),
        **mem5,
        "valu": {"SQ_INSTS_VALU_per_launch": m(C, "SQ_INSTS_VALU"), "valu_busy_measured": 4.0 * m(C, "SQ_ACTIVE_INST_VALU") / (1024 * gui5),
                 "lane_activity": m(C, "SQ_THREAD_CYCLES_VALU") / (64.0 * m(C, "SQ_ACTIVE_INST_VALU")),
                 "source": f"profiles/{rnd}/{tag}_c5_pmc_summary.txt"},
        "compute_executed": executed_fp64(C, ck, f"profiles/{rnd}/{tag}_c5_pmc_summary.txt"),
    }
print(json.dumps(out, indent=1))
