#!/bin/bash
# The ISA-profile library: the render unit's -DTRT_MARKS=2 device assembly -> tools/isa_profile_pass.py -> assemble -> link with the
# library's other units (built with the same switches).
# usage: tools/build_isa_profile.sh [extra -D switches]   -> build/isa_profile.so   (run it with tools/isa_profile.py)
set -e
LLVM=/opt/rocm/lib/llvm/bin
CSRC=terminalraytracer_amd/csrc
TUNE="-DTRT_MARKS=2 $*"
TAG=$(echo "$TUNE" | sed -e 's/-D/_/g' -e 's/=/_/g' -e 's/ //g')
FLAGS="$TUNE --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -I$CSRC -Wno-unused-function"
T=build/isa_profile; mkdir -p $T
make -s -j6 lib LIB=$T/plain.so TUNE="$TUNE" > /dev/null   # every unit with the same switches (the counters' size depends on them)
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o $T/dev.s $CSRC/trt_render.hip
python3 tools/isa_profile_pass.py $T/dev.s $T/dev_p.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev_p.s -o $T/dev_p.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev_p.out $T/dev_p.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev_p.out -output=$T/dev_p.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev_p.hipfb -c -o $T/trt_render_p.o $CSRC/trt_render.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ${OUT:-build/isa_profile.so} $T/trt_render_p.o build/trt_capi$TAG.o build/trt_tables$TAG.o build/trt_diag$TAG.o \
    build/trt_dropin$TAG.o build/trt_dist.o build/host_trt_*.o -ldl
echo built ${OUT:-build/isa_profile.so}
