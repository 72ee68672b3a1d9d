#!/bin/bash
# Executed instructions per stage and round of the plain rounds (build/count_<kind>.so from tools/archive/build_isa_count.sh), config 3 and 5
# usage: gpurun -- bash tools/archive/gpu_count.sh [kind ...]     -> gpurun_out/counts/<config>_<kind>.txt and a table
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/counts; mkdir -p $out
export TRT_COMPACTION=0
kinds="${@:-valu fp64 trans cmp cndmask mov salu wait lds vmem}"
for k in $kinds; do
  TRT_HIP_LIB=$PWD/build/count_$k.so timeout -k 10 120 python3 tools/archive/stamp_config.py 1920 1080 64 8 > $out/c3_$k.txt 2>&1 || { tail -5 $out/c3_$k.txt; exit 1; }
  TRT_HIP_LIB=$PWD/build/count_$k.so timeout -k 10 120 python3 tools/archive/stamp_config.py 1920 1080 256 12 > $out/c5_$k.txt 2>&1 || { tail -5 $out/c5_$k.txt; exit 1; }
done
python3 - $out $kinds <<'PY'
import re, sys
out, kinds = sys.argv[1], sys.argv[2:]
for cfg in ("c3", "c5"):
    table, trips = {}, None
    for k in kinds:
        for l in open("%s/%s_%s.txt" % (out, cfg, k)):
            m = re.match(r"stamp (.{16})\s+[\d.]+ %\s+(\d+)", l)
            if m:
                table.setdefault(m.group(1).strip(), {})[k] = int(m.group(2))
            m = re.search(r"'wave_loop_trips': (\d+)", l)
            if m:
                trips = int(m.group(1))
    print("=== %s: instructions per wave and round (%d rounds)" % (cfg, trips))
    print("%-18s" % "stage" + "".join("%9s" % k for k in kinds))
    tot = dict.fromkeys(kinds, 0)
    for st, row in table.items():
        print("%-18s" % st + "".join("%9.1f" % (row.get(k, 0) / trips) for k in kinds))
        for k in kinds:
            tot[k] += row.get(k, 0)
    print("%-18s" % "total" + "".join("%9.1f" % (tot[k] / trips) for k in kinds))
PY
