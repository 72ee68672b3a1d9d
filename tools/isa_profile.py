"""Executed instructions per stage, round and kind of the SHIPPING render kernels (build/isa_profile.so from
tools/build_isa_profile.sh; post-pass: tools/isa_profile_pass.py).  Renders one frame of a config and prints the table.

usage: TRT_HIP_LIB=$PWD/build/isa_profile.so python tools/isa_profile.py W H SPHERES BOUNCES [compaction 0|1|-1]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KINDS = ["valu", "fp64", "trans", "cmp", "cndmask", "mov", "salu", "wait", "lds", "vmem", "visits"]
# an interval STARTS at its boundary (csrc/trt_rounds.hpp: TRT_STAMP_AT / TRT_MARK_AT) and runs to the next one passed
NAMES = {63: "prologue (image, first hand-out)", 22: "units+primary", 0: "unit(next_dir)", 1: "P set-up (family, member, cell)",
         2: "P sweep / list thinning", 3: "P exact tests", 4: "P plane", 5: "P post: hit (back, normal)", 6: "P post: sky texel, reflect",
         7: "S entry", 24: "Sd look-up", 8: "Sd trace set-up", 9: "Sd sweep / thinning", 10: "Sd exact tests", 11: "Sd plane", 12: "Sd tail",
         13: "Sd lit accumulate, loop", 25: "Sp unit(to_light), look-up", 14: "Sp any-hit search", 17: "Sp tail (after any-hit)",
         26: "Sp full: set-up", 27: "Sp full: sweep", 28: "Sp full: exact", 29: "Sp full: plane", 19: "Sp lit accumulate, loop",
         20: "END (colour, store, bookkeeping)", 21: "loop edge"}
ORDER = [22, 0, 1, 2, 3, 4, 5, 6, 7, 24, 8, 9, 10, 11, 12, 13, 25, 14, 17, 26, 27, 28, 29, 19, 20, 21, 63]


def child(w, h, n, b, compaction):
    os.environ["TRT_PRINT_PROFILE"] = "1"
    from terminalraytracer_amd import hip, scenes as S
    scene = S.synth_scene(n, S.synth_sky(256), S.orbit_camera(1.0, w, h))
    with hip.Context(0) as ctx:
        ctx.set_scene(scene)
        ctx.set_compaction(compaction)
        ctx.render_host(scene.camera, hip.RowSet.whole(w, h), b, 10)
        print("variant", ctx.render_variant(), "kernel_info", ctx.kernel_info(), file=sys.stderr)
        ctx.read_counters()


def main():
    if sys.argv[1] == "--child":
        child(*(int(x) for x in sys.argv[2:7]))
        return
    w, h, n, b = (int(x) for x in sys.argv[1:5])
    compaction = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(w), str(h), str(n), str(b), str(compaction)], capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr[-3000:])
        raise SystemExit(r.returncode)
    table = {}
    for l in r.stderr.split("\n"):
        m = re.match(r"profile (\d+) (\d+) (\d+)", l)
        if m:
            table.setdefault(int(m.group(2)), {})[KINDS[int(m.group(1))]] = int(m.group(3))
        elif l.startswith("variant"):
            print("#", l)
    rounds = table.get(0, {}).get("visits", 0)
    print("=== %dx%d, %d spheres, %d bounces, compaction %d: instructions per wave and round (%d rounds)" % (w, h, n, b, compaction, rounds))
    print("%-36s" % "interval" + "".join("%8s" % k for k in KINDS[:-1]) + "  passes/round")
    tot = dict.fromkeys(KINDS, 0)
    for slot in ORDER + sorted(set(table) - set(ORDER)):
        row = table.get(slot)
        if not row:
            continue
        print("%-36s" % NAMES.get(slot, "slot %d" % slot) + "".join("%8.1f" % (row.get(k, 0) / rounds) for k in KINDS[:-1]) + "%10.3f" % (row.get("visits", 0) / rounds))
        if slot != 63:
            for k in KINDS:
                tot[k] += row.get(k, 0)
    print("%-36s" % "total of a round (no prologue)" + "".join("%8.1f" % (tot[k] / rounds) for k in KINDS[:-1]))


if __name__ == "__main__":
    main()
