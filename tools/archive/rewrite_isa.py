"""Post-pass over the compiler's gfx950 assembly of the device code (Makefile: hipcc -S -> this -> assembler).

v_cndmask_b32 in its short VOP2 encoding (`_e32`, mask implicitly in VCC) is a SLOW instruction on gfx950: 15.5 cycles per wave
instruction back to back at four waves per SIMD, 16 behind a compare at one wave, against 4.0 / 5.7 for the VOP3 encoding (`_e64`)
of the very same operation with the very same mask register (tools/archive/ubench_cndmask.hip, tools/archive/ubench_valu.hip).  The compiler's
SIShrinkInstructions pass picks the short form whenever the mask is VCC and has no switch to stop it, and the frame producer's
closest-hit bookkeeping is runs of such selects.  This pass re-encodes them: same opcode, same operands, same result, 4 bytes longer.
Only selects whose first source is a VGPR or an inline constant are touched (VOP3 on gfx9 takes no literal, and an SGPR source
beside the VCC mask would be a second constant-bus read); the compiler emits no other kind here, and the pass counts what it leaves.

usage: python tools/archive/rewrite_isa.py in.s out.s"""
import re
import sys

PAT = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+)(v\d+), (v\d+|-?\d+|-?\d+\.\d+), (v\d+), vcc(\s*(;.*)?)$")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    changed = kept = 0
    out = []
    with open(src) as fh:
        for line in fh:
            m = PAT.match(line.rstrip("\n"))
            if m:
                out.append(f"{m.group(1)}v_cndmask_b32_e64{m.group(2)}{m.group(3)}, {m.group(4)}, {m.group(5)}, vcc{m.group(6)}\n")
                changed += 1
            else:
                if "v_cndmask_b32_e32" in line:
                    kept += 1
                out.append(line)
    with open(dst, "w") as fh:
        fh.writelines(out)
    print(f"rewrite_isa: {changed} v_cndmask_b32_e32 -> _e64, {kept} left as they are", file=sys.stderr)


if __name__ == "__main__":
    main()
