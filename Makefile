# Builds everything in-tree for gfx950:
#   hinge_amd/lib/libhinge_hip.so   HIP kernels + C ABI (include/hinge_hip.h)
#   oracle/libhinge_oracle.so       CPU oracle (test infrastructure)
#   oracle/_ref/libhinge_ref.so     reference library code, when /root/reference is present
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude

CSRC = hinge_amd/csrc
LIB  = hinge_amd/lib/libhinge_hip.so

BIN  = hinge_amd/bin
HOST = hinge_amd/host
HOSTDEPS = $(wildcard $(HOST)/*.h) include/hinge_hip.h $(LIB)
PROGS = $(BIN)/Reads_filter $(BIN)/get_maximal_reads $(BIN)/hinging $(BIN)/consensus $(BIN)/draft_assembly $(BIN)/hinge_pipeline $(BIN)/hinge

SYNTHIO = hinge_amd/lib/libhinge_synthio.so

all: $(LIB) $(PROGS) $(SYNTHIO) oracle

$(BIN)/Reads_filter: $(HOST)/filter_main.cpp $(HOSTDEPS)
	mkdir -p $(BIN)
	$(HIPCC) -O2 -std=c++17 -w -pthread -o $@ $< -Lhinge_amd/lib -lhinge_hip -lz -Wl,-rpath,'$$ORIGIN/../lib'

$(BIN)/get_maximal_reads: $(HOST)/maximal_main.cpp $(HOSTDEPS)
	mkdir -p $(BIN)
	$(HIPCC) -O2 -std=c++17 -w -pthread -o $@ $< -Lhinge_amd/lib -lhinge_hip -lz -Wl,-rpath,'$$ORIGIN/../lib'

$(BIN)/hinging: $(HOST)/layout_main.cpp $(HOSTDEPS)
	mkdir -p $(BIN)
	$(HIPCC) -O2 -std=c++17 -w -pthread -o $@ $< -Lhinge_amd/lib -lhinge_hip -lz -Wl,-rpath,'$$ORIGIN/../lib'

$(BIN)/hinge_pipeline: $(HOST)/pipeline_main.cpp $(wildcard $(HOST)/*.cpp) $(HOSTDEPS)
	mkdir -p $(BIN)
	$(HIPCC) -O2 -std=c++17 -w -pthread -o $@ $< -Lhinge_amd/lib -lhinge_hip -lz -Wl,-rpath,'$$ORIGIN/../lib'

$(BIN)/consensus: $(HOST)/consensus_main.cpp $(HOSTDEPS)
	mkdir -p $(BIN)
	$(HIPCC) -O2 -std=c++17 -w -pthread -o $@ $< -Lhinge_amd/lib -lhinge_hip -lz -Wl,-rpath,'$$ORIGIN/../lib'

$(BIN)/draft_assembly: $(HOST)/draft_main.cpp $(HOSTDEPS)
	mkdir -p $(BIN)
	$(HIPCC) -O2 -std=c++17 -w -pthread -o $@ $< -Lhinge_amd/lib -lhinge_hip -lz -Wl,-rpath,'$$ORIGIN/../lib'

$(BIN)/hinge: $(HOST)/hinge
	mkdir -p $(BIN)
	cp $< $@
	chmod +x $@

$(LIB): $(wildcard $(CSRC)/*.hip $(CSRC)/*.h $(CSRC)/*.inc) include/hinge_hip.h
	mkdir -p hinge_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/hinge_capi.hip -ldl

# test / bench tooling: fast writer of synthetic .las files (no GPU code)
$(SYNTHIO): hinge_amd/tools_c/synth_io.c
	mkdir -p hinge_amd/lib
	gcc -O2 -fPIC -shared -o $@ $<

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf hinge_amd/lib hinge_amd/bin
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
