#!/bin/bash
# libtrt_hip.so with the device assembly passed through tools/rewrite_isa.py: hipcc -S (device) -> rewrite -> assemble -> link ->
# bundle -> host compile with the bundle embedded -> shared library.  usage: tools/build_isa_rewrite.sh <out.so> [extra -D flags]
set -e
OUT=${1:-build/e64.so}; shift || true
LLVM=/opt/rocm/lib/llvm/bin
CSRC=terminalraytracer_amd/csrc
FLAGS="$* --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -I$CSRC -Wno-unused-function"
T=build/isa_$(basename $OUT .so); mkdir -p $T
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o $T/dev.s $CSRC/trt_capi.hip
python3 tools/rewrite_isa.py $T/dev.s $T/dev2.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev2.s -o $T/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c -o $T/trt_capi.o $CSRC/trt_capi.hip
make -s build/trt_dist.o build/host_trt_camera.o build/host_trt_emit.o build/host_trt_hash.o build/host_trt_ppm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT $T/trt_capi.o build/trt_dist.o build/host_trt_*.o -ldl
echo built $OUT
