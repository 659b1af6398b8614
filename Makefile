# Top-level build: HIP library (gfx950), host mirror library, CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := shenqi_amd/csrc
LIBDIR := shenqi_amd/lib
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Iinclude
HIPSRC := $(CSRC)/capi.hip $(CSRC)/grav_walk.hip $(CSRC)/grav_group.hip $(CSRC)/pm.hip $(CSRC)/sph.hip $(CSRC)/sph_capi.hip $(CSRC)/sph_resident.hip $(CSRC)/fft3d.hip $(CSRC)/tree_build.hip $(CSRC)/dynamics.hip $(CSRC)/timestep.hip $(CSRC)/fof.hip $(CSRC)/exchange.hip $(CSRC)/toptree.hip
HIPOBJ := $(patsubst $(CSRC)/%.hip,$(LIBDIR)/%.o,$(HIPSRC))

all: $(LIBDIR)/libshenqi_hip.so host oracle

$(LIBDIR)/%.o: $(CSRC)/%.hip $(CSRC)/common.hpp include/shenqi_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libshenqi_hip.so: $(HIPOBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIPOBJ) -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib

host: $(LIBDIR)/libshenqi_hip.so
	$(MAKE) -C shenqi_amd/host

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR); $(MAKE) -C oracle clean; $(MAKE) -C shenqi_amd/host clean
.PHONY: all host oracle clean
