# Top-level build for C/C++ users (the Python entry point __graft_entry__.build() does the same).
#   make            libmdct_hip.so + the C++ CLI + the CPU checker
#   make lib | cli | jpeg_example | oracle | ref | clean
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := simd_dct_amd/csrc
LIB     := simd_dct_amd/libmdct_hip.so
# -ffp-contract=off: bit-exactness contract (no FMA).  -fno-slp-vectorize: the SLP vectoriser packs adjacent scalar ops
# with v_mov shuffles; where packing pays (the q32 butterflies) it is written by hand.
# -mllvm -disable-vector-combine: LLVM's VectorCombine rewrites a scalar float op on elements of a register pair as a PACKED op with one
# useless half plus a v_mov (twice the issue cycles of the scalar op): the AAN row passes keep 6 / 12 unpaired scalar operations per line
# on purpose (fused int16 round trip 1024 -> 952 vector instructions per wave, 752 -> 632 of them packed).
HIPFLAGS := --offload-arch=$(ARCH) -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -disable-vector-combine -std=c++17 -fPIC -Wall -Iinclude -I$(CSRC)

all: lib cli oracle

lib: $(LIB)
$(LIB): $(CSRC)/shim_host.h $(CSRC)/mdct_kernels.hip $(CSRC)/mdct_api.hip $(CSRC)/shim.hip $(CSRC)/comm.hip $(CSRC)/stages.hip $(CSRC)/mdct_kernels.h $(CSRC)/scan_records.h $(CSRC)/huffman_rows.h $(CSRC)/pack_rows.h $(CSRC)/wg_sync.h $(CSRC)/batch_plan.h include/mdct.h include/simd_dct_shim.h
	$(HIPCC) $(HIPFLAGS) -shared $(CSRC)/mdct_kernels.hip $(CSRC)/mdct_api.hip $(CSRC)/shim.hip $(CSRC)/comm.hip $(CSRC)/stages.hip -ldl -o $@

cli: tools/simd_dct_cli
tools/simd_dct_cli: tools/simd_dct_cli.cpp tools/node_pipeline.h $(LIB)
	$(HIPCC) -O2 -std=c++17 -x hip --offload-arch=$(ARCH) -Iinclude -Itools $< -Lsimd_dct_amd -lmdct_hip -Wl,-rpath,'$$ORIGIN/../simd_dct_amd' -o $@

# plain C on the C-ABI: pixels -> baseline JPEG (tools/mdct_jpeg.c)
jpeg_example: tools/mdct_jpeg
# (no dependency on $(LIB): a GPU test builds this while the library is loaded, so it must never relink it as a side effect; run `make lib` first)
tools/mdct_jpeg: tools/mdct_jpeg.c include/mdct.h
	gcc -O2 -std=c99 -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude $< -Lsimd_dct_amd -lmdct_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../simd_dct_amd' -Wl,-rpath,/opt/rocm/lib -o $@

oracle:
	$(MAKE) -C oracle
ref:
	$(MAKE) -C oracle ref

clean:
	rm -f $(LIB) tools/simd_dct_cli tools/mdct_jpeg
	$(MAKE) -C oracle clean

.PHONY: all lib cli jpeg_example oracle ref clean
