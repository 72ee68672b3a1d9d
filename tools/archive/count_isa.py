"""Post-pass over the compiler's gfx950 assembly of a -DTRT_STAMP=2 build: makes s101 a count of executed instructions.

At the head of every basic block of the counting render kernels (`render_rounds_kernel<true, ...>`) it inserts
    s_cselect_b32 s100, 1, 0 ; s_add_u32 s101, s101, N ; s_cmp_lg_u32 s100, 0        (SCC saved and restored)
with N = the number of instructions of the chosen KIND in that block, and `s_mov_b32 s101, 0` at the kernel's entry.  The
stamps of the diagnostic build (TRT_STAMP_AT) then read s101 instead of the clock, so the per-stage sums that
TRT_PRINT_STAMPS prints are wave-level instruction counts per stage.  The kernels must leave s100 / s101 alone
(.amdhsa_next_free_sgpr <= 100: checked), and the pass raises it to 102.

usage: python tools/archive/count_isa.py in.s out.s <kind>     kind: valu | fp64 | trans | salu | lds | vmem | cndmask | all"""
import re
import sys

KINDS = {
    "valu": lambda op: op.startswith("v_"),
    "fp64": lambda op: op.startswith("v_") and ("_f64" in op) and not op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_cmp", "v_cvt", "v_frexp", "v_ldexp", "v_div_s", "v_div_fi", "v_div_fm")),
    "trans": lambda op: op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")),
    "cmp": lambda op: op.startswith("v_cmp"),
    "cndmask": lambda op: op.startswith("v_cndmask"),
    "mov": lambda op: op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane")),
    "salu": lambda op: op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")),
    "wait": lambda op: op.startswith(("s_waitcnt", "s_nop")),
    "lds": lambda op: op.startswith("ds_"),
    "vmem": lambda op: op.startswith(("global_", "flat_", "buffer_", "scratch_")),
    "all": lambda op: True,
}
BRANCH = ("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")


def main():
    src, dst, kind = sys.argv[1], sys.argv[2], sys.argv[3]
    want = KINDS[kind]
    lines = open(src).read().split("\n")
    out, i, kernels = [], 0, 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_ZN3trt20render_rounds_kernelILb1E") and l.split(":")[0].endswith("E") and ":" in l:
            # the body runs to s_endpgm ... .amdhsa_kernel follows after it
            end = next(j for j in range(i + 1, len(lines)) if lines[j].strip().startswith(".Lfunc_end"))
            body = lines[i + 1:end]
            out.append(l)
            out.append("\ts_mov_b32 s101, 0")
            # split into blocks
            blocks, cur = [], []
            for b in body:
                t = b.strip()
                is_label = t.endswith(":") and not t.startswith((";", "//"))
                if is_label and cur:
                    blocks.append(cur)
                    cur = []
                cur.append(b)
                op = t.split()[0] if t and not t.startswith((";", ".", "//")) and not is_label else ""
                if op.startswith(BRANCH):
                    blocks.append(cur)
                    cur = []
            if cur:
                blocks.append(cur)
            for blk in blocks:
                n = 0
                for b in blk:
                    t = b.strip()
                    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                        continue
                    if "s101" in t or "s100" in t:
                        if not t.startswith("s_mov_b32") or "s101" not in t.split(",")[-1]:
                            raise SystemExit("the kernel uses s100 / s101: " + t)
                        continue
                    n += 1 if want(t.split()[0]) else 0
                # labels first, then the count, then the instructions
                k = 0
                while k < len(blk) and (blk[k].strip().endswith(":") or not blk[k].strip() or blk[k].strip().startswith((";", ".", "//"))):
                    out.append(blk[k])
                    k += 1
                if n and k < len(blk):
                    out.append("\ts_cselect_b32 s100, 1, 0")
                    out.append("\ts_add_u32 s101, s101, %d" % n)
                    out.append("\ts_cmp_lg_u32 s100, 0")
                out.extend(blk[k:])
            kernels += 1
            i = end
            continue
        m = re.match(r"^(\s*)\.amdhsa_next_free_sgpr (\d+)", l)
        if m and int(m.group(2)) < 102:
            l = "%s.amdhsa_next_free_sgpr 102" % m.group(1)
        out.append(l)
        i += 1
    open(dst, "w").write("\n".join(out))
    print("count_isa: %d kernels instrumented for '%s'" % (kernels, kind), file=sys.stderr)


if __name__ == "__main__":
    main()
