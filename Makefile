# Builds the product library (HIP kernels + C ABI) for gfx950 and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
CSRC := bio_amd/csrc
LIB := bio_amd/lib/libbioalign.so
OBJS := build/kernels.o build/kernels_affine.o build/kernels_long.o build/kernels_fasta.o build/kernels_wide.o build/kernels_bp.o build/api.o

all: $(LIB) oracle tests/cpp/bio_test

$(LIB): $(OBJS)
	@mkdir -p bio_amd/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS)

build/kernels.o: $(CSRC)/kernels.hip $(CSRC)/common.h $(CSRC)/device_helpers.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

build/kernels_affine.o: $(CSRC)/kernels_affine.hip $(CSRC)/common.h $(CSRC)/device_helpers.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

build/kernels_long.o: $(CSRC)/kernels_long.hip $(CSRC)/common.h $(CSRC)/device_helpers.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

build/kernels_wide.o: $(CSRC)/kernels_wide.hip $(CSRC)/common.h $(CSRC)/device_helpers.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

build/kernels_bp.o: $(CSRC)/kernels_bp.hip $(CSRC)/common.h $(CSRC)/device_helpers.h $(CSRC)/bp_column.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

build/kernels_fasta.o: $(CSRC)/kernels_fasta.hip $(CSRC)/common.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

build/api.o: $(CSRC)/api.cpp $(CSRC)/common.h include/bioalign.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c -o $@ $<

# C++ host mirror (include/bio.hpp) and its tests; plain g++, links the C ABI only
tests/cpp/bio_test: tests/cpp/bio_test.cpp tests/cpp/golden_vectors.inc include/bio.hpp include/bio_matrices.inc include/bioalign.h $(LIB)
	g++ -O2 -std=c++17 -Wall -o $@ tests/cpp/bio_test.cpp -Lbio_amd/lib -lbioalign -Wl,-rpath,'$$ORIGIN/../../bio_amd/lib'

oracle:
	$(MAKE) -C oracle libbio_oracle.so

clean:
	rm -rf build $(LIB) tests/cpp/bio_test
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
