# Top-level build: the gfx950 engine library and the CPU checker.
#   make            -> kangaroo_amd/libkangaroo_amd.so (HIP, gfx950) + oracle/libkgoracle.so
#   make ref        -> oracle/_ref (reference CPU sources compiled in place; needs /root/reference)
HIPCC    ?= hipcc
CXX      ?= g++
HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument
CSRC     := kangaroo_amd/csrc
BUILD    := build

all: kangaroo_amd/libkangaroo_amd.so oracle dropin_test

$(BUILD)/engine_capi.o: $(CSRC)/engine_capi.hip $(CSRC)/walk_common.h $(CSRC)/walk_kernel_rot.h $(CSRC)/walk_kernel_stream.h $(CSRC)/herd_kernel.h $(CSRC)/kat_kernel.h $(CSRC)/fe29.h $(CSRC)/modinv30.h include/kangaroo_amd.h include/kangaroo_amd_host.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(CSRC)/engine_capi.hip

$(BUILD)/host_secp.o: $(CSRC)/host_secp.cpp $(CSRC)/fe29.h $(CSRC)/modinv30.h include/kangaroo_amd.h include/kangaroo_amd_host.h
	@mkdir -p $(BUILD)
	$(CXX) -O2 -std=c++17 -fPIC -c -o $@ $(CSRC)/host_secp.cpp

kangaroo_amd/libkangaroo_amd.so: $(BUILD)/engine_capi.o $(BUILD)/host_secp.o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^ -lpthread

oracle:
	$(MAKE) -C oracle all

# test-only: the C++ drop-in class compiled against stand-in headers and linked with a small driver, so that
# tests/test_gpu_dropin.py can RUN it on the GPU box (the reference tree does not travel)
dropin_test: tests/_build/dropin_host
tests/_build/dropin_host: tests/dropin_host.cpp kangaroo_amd/dropin/GPUEngine_amd.cpp tests/dropin_stub/engine.h tests/dropin_stub/gpuengine_standin.h tests/dropin_stub/int.h tests/dropin_stub/constants.h include/kangaroo_amd.h kangaroo_amd/libkangaroo_amd.so
	@mkdir -p tests/_build
	$(CXX) -O2 -std=c++17 -Itests/dropin_stub -Iinclude -o $@ tests/dropin_host.cpp kangaroo_amd/dropin/GPUEngine_amd.cpp -Lkangaroo_amd -lkangaroo_amd -Wl,-rpath,'$$ORIGIN/../../kangaroo_amd'

ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf $(BUILD) kangaroo_amd/libkangaroo_amd.so
	$(MAKE) -C oracle clean

.PHONY: all oracle ref clean dropin_test
