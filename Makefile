# Top-level build: the gfx950 engine library and the CPU checker.
#   make            -> kangaroo_amd/libkangaroo_amd.so (HIP, gfx950) + oracle/libkgoracle.so
#   make ref        -> oracle/_ref (reference CPU sources compiled in place; needs /root/reference)
HIPCC    ?= hipcc
HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-command-line-argument
CSRC     := kangaroo_amd/csrc

all: kangaroo_amd/libkangaroo_amd.so oracle

kangaroo_amd/libkangaroo_amd.so: $(CSRC)/engine_capi.hip $(CSRC)/walk_kernel.h $(CSRC)/fe29.h $(CSRC)/modinv30.h include/kangaroo_amd.h
	$(HIPCC) $(HIPFLAGS) -o $@ $(CSRC)/engine_capi.hip

oracle:
	$(MAKE) -C oracle all

ref:
	$(MAKE) -C oracle ref

clean:
	rm -f kangaroo_amd/libkangaroo_amd.so
	$(MAKE) -C oracle clean

.PHONY: all oracle ref clean
