"""Static instruction counts per stage of a render kernel, from the compiler's assembly of a -DTRT_MARKS=1 build
(hipcc --save-temps): the stage boundaries are `; MARK <slot>` comments.  Instructions are attributed to the mark that
precedes them in layout order; loop bodies count once.  usage: python tools/archive/isa_stage_counts.py <file.s> <mangled-kernel-substring>"""
import collections, re, sys
path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3trt") and want in l and l.rstrip().endswith(("E:", ")")) or (l.startswith("_ZN3trt") and want in l and ":" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith("s_endpgm"))
names = {"22": "loop edge", "0": "units+primary", "1": "unit(next_dir)", "trace_0": "trace: set-up/table load", "trace_1": "trace: sweep / list filter",
         "trace_2": "trace: exact tests", "trace_3": "trace: plane", "6": "P post: hit", "7": "P post: sky, reflect, enqueue", "8": "Sd look-up",
         "13": "Sd tail", "14": "Sp unit/look-up", "19": "Sp tail", "20": "S pass edge / lit accumulate", "21": "collect + END"}
cur = outer = "prologue"
counts = collections.OrderedDict()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r";\s*MARK (\S+)", t)
    if m:
        k = m.group(1)
        if k.startswith("trace_"):  # trace() is inlined at three sites: name its phases after the stage mark that precedes them
            cur = outer + " > " + names[k]
        else:
            outer = cur = "after " + names.get(k, k)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "flat_", "scratch_", "buffer_")) else "other")
    counts.setdefault(cur, collections.Counter())[kind] += 1
print("%-62s %6s %6s %5s %5s" % ("region (instructions AFTER the mark)", "valu", "salu", "lds", "vmem"))
tot = collections.Counter()
for k, c in counts.items():
    print("%-62s %6d %6d %5d %5d" % (k, c["valu"], c["salu"], c["lds"], c["vmem"]))
    tot.update(c)
print("%-62s %6d %6d %5d %5d" % ("total", tot["valu"], tot["salu"], tot["lds"], tot["vmem"]))
