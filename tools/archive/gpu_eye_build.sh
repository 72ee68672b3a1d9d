#!/bin/bash
# kernel time of the eye-table builder (and of the scene's table builders) at 64 and 256 spheres, for the shipped library and
# any alternative builds given.   usage: gpurun -- bash tools/archive/gpu_eye_build.sh [lib ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in terminalraytracer_amd/libtrt_hip.so "$@"; do
  for n in 64 256; do
    O=gpurun_out/eyebuild/$(basename $lib .so)_$n; rm -rf $O; mkdir -p $O
    TRT_HIP_LIB=$PWD/$lib timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O -o p -- python3 tools/archive/experiments/eye_build_time.py $n > $O/out.txt 2>&1 || { tail -5 $O/out.txt; exit 1; }
    echo "== $lib n=$n"; python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "build" in r["Name"] or "pack" in r["Name"]:
        print("  %-50s calls %4s avg %9.1f us min %9.1f" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  done
done
