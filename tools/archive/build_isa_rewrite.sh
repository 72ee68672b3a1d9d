#!/bin/bash
# libtrt_hip.so with the render unit's device assembly passed through tools/archive/rewrite_isa.py: hipcc -S (device) -> rewrite -> assemble ->
# link -> bundle -> host compile with the bundle embedded -> shared library with the library's other units.
# usage: tools/archive/build_isa_rewrite.sh <out.so> [extra -D flags]
set -e
OUT=${1:-build/e64.so}; shift || true
LLVM=/opt/rocm/lib/llvm/bin
CSRC=terminalraytracer_amd/csrc
TUNE="$*"
TAG=$(echo "$TUNE" | sed -e 's/-D/_/g' -e 's/=/_/g' -e 's/ //g')
FLAGS="$TUNE --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -I$CSRC -Wno-unused-function"
T=build/isa_$(basename $OUT .so); mkdir -p $T
make -s -j6 lib LIB=$T/plain.so TUNE="$TUNE" > /dev/null
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o $T/dev.s $CSRC/trt_render.hip
python3 tools/archive/rewrite_isa.py $T/dev.s $T/dev2.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev2.s -o $T/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c -o $T/trt_render.o $CSRC/trt_render.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT $T/trt_render.o build/trt_capi$TAG.o build/trt_tables$TAG.o build/trt_diag$TAG.o build/trt_dropin$TAG.o \
    build/trt_dist.o build/host_trt_*.o -ldl
echo built $OUT
