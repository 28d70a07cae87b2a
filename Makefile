# Builds the native libraries without Python (same commands as `python -m csgn_amd.build`):
#   csgn_amd/lib/libcsgn_hip.so    the C ABI (include/csgn_hip.h): hand-written gfx950 kernels
#   csgn_amd/lib/libcsgn_shard.so  batch sharding + RCCL all-gather of term counts (include/csgn_shard.h)
#   csgn_amd/lib/libcertFHE.so     the drop-in certFHE:: classes (include/certfhe/) over that ABI
#   csgn_amd/lib/libcertFHE_shard.so  certFHE::ShardGroup / ShardedBatch: one batch over the GPUs of a node
# `make tools` adds tools/bin/shard_mul (thread-per-GPU driver) and tools/bin/bench_mul.
# hipcc cross-compiles for gfx950 without a GPU.  `make check` also builds the test-only oracle.
HIPCC   ?= $(or $(shell command -v hipcc 2>/dev/null),/opt/rocm/bin/hipcc)
CXX     ?= g++
CSRC    := csgn_amd/csrc
LIBDIR  := csgn_amd/lib
HIP_SRC := $(addprefix $(CSRC)/,csgn_capi.hip csgn_circuit.hip csgn_mul.hip csgn_add.hip csgn_smallops.hip csgn_decrypt.hip csgn_encrypt.hip \
                                csgn_permute.hip csgn_compact.hip csgn_harness.hip csgn_bitlen.hip csgn_tuning.cpp)
OBJDIR  := $(LIBDIR)/obj
HIP_OBJ := $(patsubst $(CSRC)/%,$(OBJDIR)/%.o,$(basename $(HIP_SRC)))
HIP_FLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-inline-asm -Iinclude -I$(CSRC)
HIP_HDR := $(wildcard $(CSRC)/*.h) include/csgn_hip.h
CLS_SRC := $(sort $(wildcard $(CSRC)/certfhe/*.cpp))
CLS_HDR := $(wildcard include/certfhe/*.h) $(wildcard $(CSRC)/certfhe/*.h)

ROCM_LIB ?= $(dir $(HIPCC))../lib

.PHONY: all tools check clean
all: $(LIBDIR)/libcsgn_hip.so $(LIBDIR)/libcsgn_shard.so $(LIBDIR)/libcertFHE.so $(LIBDIR)/libcertFHE_shard.so

$(LIBDIR)/libcsgn_shard.so: $(CSRC)/csgn_shard.hip include/csgn_shard.h include/csgn_hip.h
	mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -Iinclude -o $@ $(CSRC)/csgn_shard.hip -L$(ROCM_LIB) -lrccl

tools: tools/bin/shard_mul tools/bin/bench_mul tools/bin/bench_native

tools/bin/shard_mul: tools/shard_mul.cpp $(LIBDIR)/libcertFHE_shard.so
	mkdir -p tools/bin
	$(CXX) -std=c++11 -O2 -Iinclude -Iinclude/certfhe -o $@ $< -L$(LIBDIR) -lcertFHE_shard -lcertFHE -lcsgn_shard -lcsgn_hip \
	    -lpthread '-Wl,-rpath,$(abspath $(LIBDIR))' '-Wl,-rpath,$(abspath $(ROCM_LIB))'

# certFHE::ShardGroup / ShardedBatch over libcsgn_shard.so (RCCL); apart from libcertFHE.so so that
# single-GPU users never map RCCL
$(LIBDIR)/libcertFHE_shard.so: $(wildcard $(CSRC)/certfhe_shard/*.cpp) $(CLS_HDR) include/csgn_shard.h \
        $(LIBDIR)/libcertFHE.so $(LIBDIR)/libcsgn_shard.so
	$(CXX) -std=c++11 -O2 -fPIC -shared -pthread -Iinclude -Iinclude/certfhe -o $@ $(wildcard $(CSRC)/certfhe_shard/*.cpp) \
	    -L$(LIBDIR) -lcertFHE -lcsgn_shard -lcsgn_hip '-Wl,-rpath,$$ORIGIN'

# bench.py's measurement without torch: thread per GPU over the C ABI, strict RCCL (bench.py --native-ranks)
tools/bin/bench_native: tools/bench_native.cpp $(LIBDIR)/libcsgn_hip.so $(LIBDIR)/libcsgn_shard.so
	mkdir -p tools/bin
	$(CXX) -std=c++11 -O2 -Wall -Iinclude -o $@ $< -L$(LIBDIR) -lcsgn_shard -lcsgn_hip -lpthread \
	    '-Wl,-rpath,$(abspath $(LIBDIR))' '-Wl,-rpath,$(abspath $(ROCM_LIB))'

# dev probes: per-wave cycle stamps of the wave-cooperative ragged multiply; scalar-path touch rate
tools/bin/coop_probe: tools/coop_probe.hip $(HIP_SRC) $(HIP_HDR)
	mkdir -p tools/bin
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -Wno-inline-asm -DCSGN_COOP_STAMPS $(COOP_PROBE_FLAGS) -Iinclude -I$(CSRC) -o $@ tools/coop_probe.hip $(CSRC)/csgn_tuning.cpp

tools/bin/sprefetch_bench: tools/sprefetch_bench.hip
	mkdir -p tools/bin
	$(HIPCC) --offload-arch=gfx950 -O3 -o $@ $<

tools/bin/graph_memset_probe: tools/graph_memset_probe.hip
	mkdir -p tools/bin
	$(HIPCC) --offload-arch=gfx950 -O2 -o $@ $<

tools/bin/wpattern_bench: tools/wpattern_bench.hip
	mkdir -p tools/bin
	$(HIPCC) --offload-arch=gfx950 -O3 -o $@ $<

tools/bin/bench_mul: tools/bench_mul.cpp $(LIBDIR)/libcsgn_hip.so
	mkdir -p tools/bin
	$(CXX) -std=c++11 -O2 -Iinclude -o $@ $< -L$(LIBDIR) -lcsgn_hip '-Wl,-rpath,$(abspath $(LIBDIR))'

# one object per translation unit (make -j compiles them side by side)
$(OBJDIR)/%.o: $(CSRC)/%.hip $(HIP_HDR)
	mkdir -p $(OBJDIR)
	$(HIPCC) $(HIP_FLAGS) -c -o $@ $<
$(OBJDIR)/%.o: $(CSRC)/%.cpp $(HIP_HDR)
	mkdir -p $(OBJDIR)
	$(HIPCC) $(HIP_FLAGS) -x hip -c -o $@ $<

# The wave-cooperative ragged multiply keeps loads outside the compiler's books (inline-assembly loads into a reserved
# v127, hand-counted waits): the generated code is checked wherever the library is compiled, and a finding fails the build.
$(OBJDIR)/csgn_mul.s: $(CSRC)/csgn_mul.hip $(HIP_HDR)
	mkdir -p $(OBJDIR)
	$(HIPCC) $(HIP_FLAGS) -S --cuda-device-only -o $@ $< 2>/dev/null
$(OBJDIR)/coop_isa.ok: $(OBJDIR)/csgn_mul.s tools/check_coop_isa.py
	python3 tools/check_coop_isa.py $<
	echo ok > $@

$(LIBDIR)/libcsgn_hip.so: $(HIP_OBJ) $(OBJDIR)/coop_isa.ok
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIP_OBJ)

$(LIBDIR)/libcertFHE.so: $(CLS_SRC) $(CLS_HDR) $(LIBDIR)/libcsgn_hip.so
	$(CXX) -std=c++11 -O2 -fPIC -shared -Iinclude -Iinclude/certfhe -o $@ $(CLS_SRC) \
	    -L$(LIBDIR) -lcsgn_hip '-Wl,-rpath,$$ORIGIN'

check: all
	$(MAKE) -C oracle all
	python -m pytest tests -q -m "not gpu"

clean:
	rm -rf $(OBJDIR)
	rm -f $(LIBDIR)/libcsgn_hip.so $(LIBDIR)/libcsgn_shard.so $(LIBDIR)/libcertFHE.so $(LIBDIR)/libcertFHE_shard.so tools/bin/shard_mul tools/bin/bench_mul tools/bin/bench_native
