#!/bin/bash
# Diagnostic libraries whose stage "stamps" are INSTRUCTION COUNTS: the render unit's -DTRT_STAMP=2 device assembly ->
# tools/archive/count_isa.py (one kind of instruction per library) -> assemble -> link with the library's other units.
# (Superseded by tools/build_isa_profile.sh, which profiles the SHIPPING instantiations; kept for the stamp build's counting kernel.)
# usage: tools/archive/build_isa_count.sh <kind> [...]   -> build/count_<kind>.so
set -e
LLVM=/opt/rocm/lib/llvm/bin
CSRC=terminalraytracer_amd/csrc
TUNE="-DTRT_STAMP=2"
TAG=$(echo "$TUNE" | sed -e 's/-D/_/g' -e 's/=/_/g' -e 's/ //g')
FLAGS="$TUNE --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -I$CSRC -Wno-unused-function"
T=build/isa_count; mkdir -p $T
make -s -j6 lib LIB=$T/plain.so TUNE="$TUNE" > /dev/null
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o $T/dev.s $CSRC/trt_render.hip
for kind in "$@"; do
  python3 tools/archive/count_isa.py $T/dev.s $T/dev_$kind.s $kind
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev_$kind.s -o $T/dev_$kind.o
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev_$kind.out $T/dev_$kind.o
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev_$kind.out -output=$T/dev_$kind.hipfb
  /opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev_$kind.hipfb -c -o $T/trt_render_$kind.o $CSRC/trt_render.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build/count_$kind.so $T/trt_render_$kind.o build/trt_capi$TAG.o build/trt_tables$TAG.o build/trt_diag$TAG.o \
      build/trt_dropin$TAG.o build/trt_dist.o build/host_trt_*.o -ldl
  echo built build/count_$kind.so
done
