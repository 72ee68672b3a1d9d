#!/bin/bash
# PCIe-inclusive cost of the drop-in project_scene() from plain C: examples/trt_demo at 1920x1080 without drawing
set -e
tmp=$(mktemp -d)
python - "$tmp" <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import support as T
d = os.path.join(sys.argv[1], "colors"); os.makedirs(d)
for f in T.FACES:
    open(os.path.join(d, f + ".ppm"), "wb").write(T.golden_ppm_raw("colors", f))
PY
if [ -n "$TRT_PRINT_HOST_TIMES" ]; then ./examples/trt_demo "$tmp/colors" 6 1920 1080 --no-draw 2>&1 | tail -9; ./examples/trt_demo "$tmp/colors" 6 480 280 --no-draw 2>&1 | tail -6; rm -rf "$tmp"; exit 0; fi
./examples/trt_demo "$tmp/colors" 30 1920 1080 --no-draw
./examples/trt_demo "$tmp/colors" 200 480 280 --no-draw
./examples/trt_demo "$tmp/colors" 30 1920 1080 --no-draw --rgb8
./examples/trt_demo "$tmp/colors" 200 480 280 --no-draw --rgb8
rm -rf "$tmp"
