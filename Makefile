# Top-level build: libtrt_hip.so (HIP kernels + C-ABI + host-side C, gfx950 only), the C demo driver and
# the CPU checkers under oracle/.
HIPCC ?= /opt/rocm/bin/hipcc
CC    ?= gcc
CSRC  := terminalraytracer_amd/csrc
LIB   ?= terminalraytracer_amd/libtrt_hip.so
BUILD := build
# -ffp-contract=off: results must be bit-identical to the reference's non-FMA x86-64 build.
# -fno-slp-vectorize: packed FP32 (v_pk_fma_f32) buys nothing on gfx950 and costs registers.
# extra -D switches for kernel-tuning A/B builds, e.g. make lib LIB=build/w4.so TUNE=-DTRT_PERSISTENT_WAVES=4
TUNE ?=
empty :=
space := $(empty) $(empty)
TAG := $(subst $(space),,$(subst =,_,$(subst -D,_,$(TUNE))))
HIPFLAGS := $(TUNE) --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 \
            -Iinclude -I$(CSRC) -Wall -Wno-unused-function $(EXTRA)
HOSTFLAGS := -O2 -ffp-contract=off -fno-fast-math -fPIC -std=c11 -Iinclude -Wall -Wextra
HOST_SRC := $(wildcard $(CSRC)/host/*.c)
HOST_OBJ := $(patsubst $(CSRC)/host/%.c,$(BUILD)/host_%.o,$(HOST_SRC))

.PHONY: all lib demo oracle stub clean resource-usage
all: lib demo oracle stub

lib: $(LIB)

$(BUILD)/host_%.o: $(CSRC)/host/%.c include/trt.h include/trt_host.h
	@mkdir -p $(BUILD)
	$(CC) $(HOSTFLAGS) -c -o $@ $<

# the library's translation units (csrc/trt_context.hpp says what each holds); trt_render is the one that instantiates the production kernel
UNITS := trt_capi trt_tables trt_render trt_diag trt_dropin
UNIT_OBJ := $(patsubst %,$(BUILD)/%$(TAG).o,$(UNITS))
HIP_HEADERS := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.h) include/trt.h include/trt_hip.h include/trt_hip_diag.h

$(BUILD)/trt_render$(TAG).o: $(CSRC)/trt_render.hip $(HIP_HEADERS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c -o $@ $(CSRC)/trt_render.hip 2> $(BUILD)/resource_usage$(TAG).txt \
		|| (cat $(BUILD)/resource_usage$(TAG).txt; false)
	@grep -E "error|warning:" $(BUILD)/resource_usage$(TAG).txt || true
	@awk '/Function Name: .*render_rounds_kernelILb0/ {f=1} f && /VGPRs:/ {split($$0,a,"VGPRs: "); v=a[2]+0; print "render_rounds_kernel<false>: " v " VGPRs" (v>128 ? "  ** WARNING: more than 128 -> 3 waves/SIMD **" : " (4 waves/SIMD)"); exit}' $(BUILD)/resource_usage$(TAG).txt

$(BUILD)/%$(TAG).o: $(CSRC)/%.hip $(HIP_HEADERS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(BUILD)/trt_dist.o: $(CSRC)/trt_dist.hip include/trt.h include/trt_hip.h include/trt_hip_diag.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(CSRC)/trt_dist.hip

# RCCL is bound at run time by trt_dist.hip (dlopen): the library does not link against it
$(LIB): $(UNIT_OBJ) $(BUILD)/trt_dist.o $(HOST_OBJ)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $@ $(UNIT_OBJ) $(BUILD)/trt_dist.o $(HOST_OBJ) -ldl

demo: examples/trt_demo examples/trt_dist_demo
examples/%: examples/%.c $(LIB) include/trt_hip.h include/trt_host.h
	$(CC) -O2 -std=gnu11 -Iinclude -o $@ $< -Lterminalraytracer_amd -ltrt_hip -lm -Wl,-rpath,'$$ORIGIN/../terminalraytracer_amd'

oracle:
	$(MAKE) -C oracle all

# TEST INFRASTRUCTURE: a stand-in for the eight RCCL entry points trt_dist.hip binds, so that several ranks can share the one GPU of a
# test box (selected by TRT_RCCL_LIB, tests only)
stub: tests/_build/librccl_stub.so tests/_build/dist_ranks
# TEST HOST: several ranks of trt_dist_* as threads of one process (the 8-way split of BASELINE configs 4 and 5 on one GPU)
tests/_build/dist_ranks: tests/dist_ranks.c $(LIB) include/trt_hip.h include/trt_host.h
	@mkdir -p tests/_build
	$(CC) -O2 -std=gnu11 -Iinclude -o $@ $< -Lterminalraytracer_amd -ltrt_hip -lpthread -lm -Wl,-rpath,'$$ORIGIN/../../terminalraytracer_amd'
tests/_build/librccl_stub.so: tests/rccl_stub.cpp
	@mkdir -p tests/_build
	g++ -O2 -fPIC -shared -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o $@ $< -L/opt/rocm/lib -lamdhip64 -lrt -lpthread

# compiler's view of registers / LDS / occupancy per kernel
resource-usage:
	$(HIPCC) $(HIPFLAGS) -c -Rpass-analysis=kernel-resource-usage -o /tmp/trt_ru.o $(CSRC)/trt_render.hip 2>&1 | grep -E "remark" || true

clean:
	rm -rf $(LIB) $(BUILD) examples/trt_demo examples/trt_dist_demo
	$(MAKE) -C oracle clean
