# Top-level build for C/C++ users (the Python entry point __graft_entry__.build() does the same).
#   make            libmdct_hip.so + the C++ CLI + the CPU checker
#   make lib | cli | oracle | ref | clean
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := simd_dct_amd/csrc
LIB     := simd_dct_amd/libmdct_hip.so
# -ffp-contract=off: bit-exactness contract (no FMA).  -fno-slp-vectorize: the SLP vectoriser packs adjacent scalar ops
# with v_mov shuffles; where packing pays (the q32 butterflies) it is written by hand.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -Wall -Iinclude -I$(CSRC)

all: lib cli oracle

lib: $(LIB)
$(LIB): $(CSRC)/mdct_kernels.hip $(CSRC)/mdct_api.hip $(CSRC)/shim.hip $(CSRC)/comm.hip $(CSRC)/stages.hip $(CSRC)/mdct_kernels.h include/mdct.h include/simd_dct_shim.h
	$(HIPCC) $(HIPFLAGS) -shared $(CSRC)/mdct_kernels.hip $(CSRC)/mdct_api.hip $(CSRC)/shim.hip $(CSRC)/comm.hip $(CSRC)/stages.hip -ldl -o $@

cli: tools/simd_dct_cli
tools/simd_dct_cli: tools/simd_dct_cli.cpp $(LIB)
	$(HIPCC) -O2 -std=c++17 -x hip --offload-arch=$(ARCH) -Iinclude $< -Lsimd_dct_amd -lmdct_hip -Wl,-rpath,'$$ORIGIN/../simd_dct_amd' -o $@

oracle:
	$(MAKE) -C oracle
ref:
	$(MAKE) -C oracle ref

clean:
	rm -f $(LIB) tools/simd_dct_cli
	$(MAKE) -C oracle clean

.PHONY: all lib cli oracle ref clean
