"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py --json) of `bench.py --depth 1`:
HBM bytes per launch (FETCH_SIZE / WRITE_SIZE from separate passes, gfx950 correction of MI355X_MICROARCH.md applied) and the
VALU counters.  usage: python tools/make_traffic_json.py <pmc_summary.json> <tag> > profiles/traffic.json"""
import json
import sys

s = json.load(open(sys.argv[1]))
tag = sys.argv[2]
rk = next(k for k in s if "render_rounds_kernel<false" in k)
dk = next(k for k in s if "reduce_samples_kernel" in k)
R, D = s[rk], s[dk]
m = lambda d, k: d[k]["mean"]  # noqa: E731
KB = 1000.0  # the counters are in KB
render_w, render_f = m(R, "WRITE_SIZE") * KB, m(R, "FETCH_SIZE") * KB
reduce_w, reduce_f = m(D, "WRITE_SIZE") * KB, m(D, "FETCH_SIZE") * KB
total = render_w + 2 * render_f + reduce_w + 2 * reduce_f
gui = m(R, "GRBM_GUI_ACTIVE") / 8.0
out = {
    "workload": "bench.py default (1920x1080, 64 spheres, 8 bounces, 10 rays/pixel), production kernels " + rk.replace("trt::", "") + " + reduce_samples_kernel, --depth 1",
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace; profiles/r02/{tag}_pmc_summary.txt (tools/profile_round.sh); means over the dispatches",
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
        "source": f"profiles/r02/{tag}_pmc_summary.txt (rocprofv3 --pmc SQ_* / GRBM_GUI_ACTIVE), " + rk.replace("trt::", "") + ", means over the dispatches",
        "SQ_INSTS_VALU_per_launch": m(R, "SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": m(R, "SQ_ACTIVE_INST_VALU"), "SQ_WAVES": m(R, "SQ_WAVES"), "simds": 1024,
        "GRBM_GUI_ACTIVE_per_xcd": gui,
        "valu_busy_measured": 4.0 * m(R, "SQ_ACTIVE_INST_VALU") / (1024 * gui),
        "valu_busy_formula": "4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE per XCD): the gfx94x VALUBusy expression of rocprofv3 -L (SQ_ACTIVE_INST_VALU counts quad-cycles)",
        "lane_activity": m(R, "SQ_THREAD_CYCLES_VALU") / (64.0 * m(R, "SQ_ACTIVE_INST_VALU")),
        "wave_cycles_waiting_any_frac": m(R, "SQ_WAIT_ANY") / m(R, "SQ_WAVE_CYCLES"),
        "wave_cycles_waiting_inst_frac": m(R, "SQ_WAIT_INST_ANY") / m(R, "SQ_WAVE_CYCLES"),
    },
}
print(json.dumps(out, indent=1))
