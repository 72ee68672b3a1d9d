# Top-level build: libtrt_hip.so (HIP kernels + C-ABI, gfx950 only) and the CPU checkers under oracle/.
HIPCC ?= /opt/rocm/bin/hipcc
CSRC  := terminalraytracer_amd/csrc
LIB   := terminalraytracer_amd/libtrt_hip.so
# -ffp-contract=off: results must be bit-identical to the reference's non-FMA x86-64 build
HIPFLAGS := --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -std=c++17 \
            -Iinclude -I$(CSRC) -Wall -Wno-unused-function

.PHONY: all lib oracle clean resource-usage
all: lib oracle

lib: $(LIB)

$(LIB): $(CSRC)/trt_capi.hip $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/host/*.c) include/trt.h include/trt_hip.h
	$(HIPCC) $(HIPFLAGS) -o $@ $(CSRC)/trt_capi.hip $(wildcard $(CSRC)/host/*.c)

oracle:
	$(MAKE) -C oracle all

# compiler's view of registers / LDS / occupancy per kernel
resource-usage:
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -o /tmp/trt_ru.so $(CSRC)/trt_capi.hip 2>&1 | grep -E "remark" || true

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean
