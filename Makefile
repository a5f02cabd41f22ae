# Builds the HIP library (gfx950), the C++ host layer and the CPU oracle, all in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
CSRC := raytracer_amd/csrc
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wl,-Bsymbolic -ffp-contract=off -fno-fast-math \
            -Wall -Wno-unused-function -Iinclude $(if $(DEV),-DRT_DEV_KNOBS)
HOSTFLAGS := -O2 -std=c++17 -fPIC -shared -Wl,-Bsymbolic -ffp-contract=off -fno-fast-math -Wall -Wextra -Iinclude

all: $(CSRC)/librt_amd.so $(CSRC)/librt_host.so oracle

$(CSRC)/librt_amd.so: $(CSRC)/rt_amd.hip $(wildcard $(CSRC)/*.h) include/rt_amd.h
	$(HIPCC) $(HIPFLAGS) -o $@ $(CSRC)/rt_amd.hip

$(CSRC)/librt_host.so: $(wildcard $(CSRC)/host/*.cpp) $(wildcard $(CSRC)/host/*.h) $(wildcard $(CSRC)/*.h) include/rt_host.hpp include/rt_amd.h include/rt_host.h $(CSRC)/librt_amd.so
	$(CXX) $(HOSTFLAGS) -o $@ $(wildcard $(CSRC)/host/*.cpp) -L$(CSRC) -lrt_amd -lz -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(CSRC)/*.so oracle/*.so

.PHONY: all oracle clean
