"""Post-pass over the compiler's gfx950 assembly of a -DTRT_MARKS=2 build: the ISA PROFILE of the shipping render kernels.

The stage boundaries of a -DTRT_MARKS=2 build are `s_mov_b32 m0, <slot> ; MARK` (TRT_STAMP_AT / TRT_MARK_AT in
csrc/trt_common.hpp); nothing else in these kernels touches m0 (checked here).  At the head of every basic block of the
NON-counting instantiations (`render_rounds_kernel<false, ...>`: the ones that ship) this pass inserts, for every KIND of
instruction the block holds,
    v_readlane_b32 s101, v<240 + kind>, m0 ; s_add_u32 s101, s101, N ; v_writelane_b32 v<240 + kind>, s101, m0
between a save and a restore of SCC (s100), so that lane `slot` of v<240 + kind> ends up holding the number of instructions of
that kind the wave EXECUTED in the intervals that START at boundary `slot` (lane 63: before the first boundary).  Kind 10
counts the boundaries themselves (visits: boundary 0 is passed once per round).  The kernel's epilogue adds the eleven
registers to counters[40 ...] (csrc/trt_rounds.hpp), `TRT_PRINT_PROFILE=1` prints them (trt_read_counters).

Unlike tools/archive/count_isa.py (stamp sums in 48 SGPRs, which the diagnostic build spills to VGPR lanes: ~140 extra moves per
round, and the COUNTING instantiation), the code profiled here is the shipping instantiation's own register allocation and
instruction selection; only the scheduling barriers at the boundaries differ.  One library, one run, every kind.

usage: python tools/isa_profile_pass.py in.s out.s"""
import re
import sys

KINDS = [
    ("valu", lambda op: op.startswith("v_")),
    ("fp64", lambda op: op.startswith("v_") and ("_f64" in op) and not op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_cmp", "v_cvt", "v_frexp", "v_ldexp", "v_div_s", "v_div_fi", "v_div_fm"))),
    ("trans", lambda op: op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos"))),
    ("cmp", lambda op: op.startswith("v_cmp")),
    ("cndmask", lambda op: op.startswith("v_cndmask")),
    ("mov", lambda op: op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane"))),
    ("salu", lambda op: op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop"))),
    ("wait", lambda op: op.startswith(("s_waitcnt", "s_nop"))),
    ("lds", lambda op: op.startswith("ds_")),
    ("vmem", lambda op: op.startswith(("global_", "flat_", "buffer_", "scratch_"))),
    ("visits", None),
]
BRANCH = ("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")
FIRST_VGPR = 240
MARK = re.compile(r"^s_mov_b32 m0, (\d+)\s*;\s*MARK")
DUMP = re.compile(r"^v_mov_b32(_e32)? v\d+, v(24\d|250)\b")


def classify(line):
    """'label', 'insn' or '' (blank, comment, directive) for a line of the assembly"""
    code = line.split(";")[0].split("//")[0].strip()
    if not code:
        return ""
    if code.endswith(":"):
        return "label"
    if code.startswith("."):
        return ""
    return "insn"


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).read().split("\n")
    out, i, kernels = [], 0, 0
    in_instrumented_desc = set()
    while i < len(lines):
        l = lines[i]
        if l.startswith("_ZN3trt20render_rounds_kernelILb0E") and ":" in l and l.split(":")[0].endswith("E"):
            name = l.split(":")[0]
            end = next(j for j in range(i + 1, len(lines)) if lines[j].strip().startswith(".Lfunc_end"))
            body = lines[i + 1:end]
            out.append(l)
            out.append("\ts_mov_b32 m0, 63")
            for k in range(len(KINDS)):
                out.append("\tv_mov_b32_e32 v%d, 0" % (FIRST_VGPR + k))
            blocks, cur = [], []
            for b in body:
                t = b.strip()
                what = classify(b)
                if what == "label" and cur:
                    blocks.append(cur)
                    cur = []
                cur.append(b)
                op = t.split()[0] if what == "insn" else ""
                if op.startswith(BRANCH) or MARK.match(t):
                    blocks.append(cur)  # a boundary ends a block too: what follows belongs to the new interval
                    cur = []
            if cur:
                blocks.append(cur)
            for blk in blocks:
                n = [0] * len(KINDS)
                for b in blk:
                    t = b.strip()
                    if classify(b) != "insn":
                        continue
                    m = MARK.match(t)
                    if m:
                        continue  # counted as a visit of the NEW interval, below
                    if DUMP.match(t):
                        continue  # the epilogue's reads of the profile registers
                    if re.search(r"\bm0\b", t.split(";")[0]):
                        raise SystemExit("the kernel uses m0: " + t)
                    if re.search(r"\bs10[01]\b", t.split(";")[0]):
                        raise SystemExit("the kernel uses s100 / s101: " + t)
                    op = t.split()[0]
                    for k, (_, want) in enumerate(KINDS):
                        if want and want(op):
                            n[k] += 1
                k = 0
                while k < len(blk) and classify(blk[k]) != "insn":
                    out.append(blk[k])
                    k += 1
                if any(n) and k < len(blk):
                    out.append("\ts_cselect_b32 s100, 1, 0")
                    for kk, c in enumerate(n):
                        if c:
                            out.append("\tv_readlane_b32 s101, v%d, m0" % (FIRST_VGPR + kk))
                            out.append("\ts_add_u32 s101, s101, %d" % c)
                            out.append("\tv_writelane_b32 v%d, s101, m0" % (FIRST_VGPR + kk))
                    out.append("\ts_cmp_lg_u32 s100, 0")
                for b in blk[k:]:
                    out.append(b)
                    if MARK.match(b.strip()):  # a visit of the interval that starts here
                        vk = FIRST_VGPR + len(KINDS) - 1
                        out.append("\ts_nop 0")
                        out.append("\ts_cselect_b32 s100, 1, 0")
                        out.append("\tv_readlane_b32 s101, v%d, m0" % vk)
                        out.append("\ts_add_u32 s101, s101, 1")
                        out.append("\tv_writelane_b32 v%d, s101, m0" % vk)
                        out.append("\ts_cmp_lg_u32 s100, 0")
            kernels += 1
            in_instrumented_desc.add(name)
            i = end
            continue
        out.append(l)
        i += 1
    # kernel descriptors of the instrumented kernels: registers up to v250 / s101
    text = "\n".join(out).split("\n")
    cur = None
    for j, l in enumerate(text):
        m = re.match(r"^\s*\.amdhsa_kernel (\S+)", l)
        if m:
            cur = m.group(1)
        if l.strip().startswith(".end_amdhsa_kernel"):
            cur = None
        if cur in in_instrumented_desc:
            m = re.match(r"^(\s*)\.amdhsa_next_free_vgpr (\d+)", l)
            if m:
                if int(m.group(2)) > FIRST_VGPR:
                    raise SystemExit("%s allocates v%s" % (cur, m.group(2)))
                text[j] = "%s.amdhsa_next_free_vgpr %d" % (m.group(1), FIRST_VGPR + len(KINDS))
            m = re.match(r"^(\s*)\.amdhsa_next_free_sgpr (\d+)", l)
            if m:
                if int(m.group(2)) > 100:
                    raise SystemExit("%s allocates s%s" % (cur, m.group(2)))
                text[j] = "%s.amdhsa_next_free_sgpr 102" % m.group(1)
            m = re.match(r"^(\s*)\.amdhsa_accum_offset (\d+)", l)
            if m:
                text[j] = "%s.amdhsa_accum_offset %d" % (m.group(1), (FIRST_VGPR + len(KINDS) + 3) // 4 * 4)
    # the code object's metadata (what the runtime's occupancy queries read): the same register counts
    entry = {}
    def close(entry):
        if entry.get("name") in in_instrumented_desc:
            j = entry["vgpr"]
            text[j] = re.sub(r"\d+\s*$", str(FIRST_VGPR + len(KINDS)), text[j])
            j = entry["sgpr"]
            text[j] = re.sub(r"\d+\s*$", "108", text[j])
    for j, l in enumerate(text):
        if re.match(r"^  - \.agpr_count:", l) or l.strip().startswith(".end_amdgpu_metadata") or l.startswith("amdhsa.target"):
            close(entry)
            entry = {}
        m = re.match(r"^    \.name:\s+(\S+)", l)
        if m:
            entry["name"] = m.group(1)
        if re.match(r"^    \.vgpr_count:", l):
            entry["vgpr"] = j
        if re.match(r"^    \.sgpr_count:", l):
            entry["sgpr"] = j
    open(dst, "w").write("\n".join(text))
    print("isa_profile_pass: %d kernels instrumented" % kernels, file=sys.stderr)


if __name__ == "__main__":
    main()
