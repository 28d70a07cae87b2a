# Builds the two native libraries without Python (same commands as `python -m csgn_amd.build`):
#   csgn_amd/lib/libcsgn_hip.so   the C ABI (include/csgn_hip.h): hand-written gfx950 kernels
#   csgn_amd/lib/libcertFHE.so    the drop-in certFHE:: classes (include/certfhe/) over that ABI
# hipcc cross-compiles for gfx950 without a GPU.  `make check` also builds the test-only oracle.
HIPCC   ?= $(or $(shell command -v hipcc 2>/dev/null),/opt/rocm/bin/hipcc)
CXX     ?= g++
CSRC    := csgn_amd/csrc
LIBDIR  := csgn_amd/lib
HIP_SRC := $(addprefix $(CSRC)/,csgn_capi.hip csgn_mul.hip csgn_add.hip csgn_decrypt.hip csgn_encrypt.hip \
                                csgn_permute.hip csgn_compact.hip csgn_harness.hip csgn_bitlen.hip csgn_tuning.cpp)
HIP_HDR := $(wildcard $(CSRC)/*.h) include/csgn_hip.h
CLS_SRC := $(sort $(wildcard $(CSRC)/certfhe/*.cpp))
CLS_HDR := $(wildcard include/certfhe/*.h) $(wildcard $(CSRC)/certfhe/*.h)

.PHONY: all check clean
all: $(LIBDIR)/libcsgn_hip.so $(LIBDIR)/libcertFHE.so

$(LIBDIR)/libcsgn_hip.so: $(HIP_SRC) $(HIP_HDR)
	mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -Iinclude -I$(CSRC) -o $@ $(HIP_SRC)

$(LIBDIR)/libcertFHE.so: $(CLS_SRC) $(CLS_HDR) $(LIBDIR)/libcsgn_hip.so
	$(CXX) -std=c++11 -O2 -fPIC -shared -Iinclude -Iinclude/certfhe -o $@ $(CLS_SRC) \
	    -L$(LIBDIR) -lcsgn_hip '-Wl,-rpath,$$ORIGIN'

check: all
	$(MAKE) -C oracle all
	python -m pytest tests -q -m "not gpu"

clean:
	rm -f $(LIBDIR)/libcsgn_hip.so $(LIBDIR)/libcertFHE.so
