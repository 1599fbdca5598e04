# Builds the three native pieces in-tree (the .so files travel to the GPU box, they are not in git):
#   raytracer_amd/lib/librtcore.so   HIP render core for gfx950 (the product)
#   raytracer_amd/lib/librthost.so   host-side mirror of the reference's scene set-up (the product's host)
#   raytracer_amd/bin/rtcli          C++ mirror of `cabal run raytracer <config.json>` (app/Main.hs)
#   oracle/_build/liboracle.so       CPU FP64 restatement of the reference (TEST INFRASTRUCTURE ONLY)
HIPCC   ?= hipcc
CXX     ?= g++
ARCH    ?= gfx950
RT_MIN_WAVES ?= 4

CORE_SRC := raytracer_amd/csrc/rtcore.hip
CORE_HDR := $(wildcard raytracer_amd/csrc/*.h) include/rtcore.h
KERNEL_DEFS ?=
HOST_SRC := $(wildcard raytracer_amd/csrc/host/*.cpp)
HOST_HDR := include/rthost.h include/rtcore.h

all: core host cli oracle

core: raytracer_amd/lib/librtcore.so
host: raytracer_amd/lib/librthost.so
cli: raytracer_amd/bin/rtcli
oracle: oracle/_build/liboracle.so

# -ffp-contract=off: the reference (GHC) forms no FMAs, and hit decisions at shared edges and vertices depend on the last bit
# of t, u and v.  With contraction off the kernel's FP64 arithmetic rounds like the reference's (measured: radiance equal to
# ~1e-16, no primitive-id mismatches); the FP32 box tests use explicit fmaf and are not affected.
# RT_BUILD_ID: digest of the kernel sources + compile-time knobs, returned by rt_build_id(); counter records under profiles/
# carry it so that bench.py can tell when they were taken on another kernel
# CORE_OPT: -O2, measured against -O3 / -Os / -O1 / -fno-slp-vectorize / -fno-unroll-loops on one box (tools/ab.sh): C4 40.86 ms at -O2,
# 41.17 at -O3, 40.91 at -Os, 41.72 at -O1; C5 508.5 against 514.7, C2 14.11 against 14.18, C3 15.58 against 15.68.  No fast-math at
# any level: the results are the same bits.
CORE_OPT ?= -O2
raytracer_amd/lib/librtcore.so: $(CORE_SRC) $(CORE_HDR) Makefile
	@mkdir -p raytracer_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) $(CORE_OPT) -std=c++17 -fPIC -shared -ffp-contract=off -DRT_MIN_WAVES=$(RT_MIN_WAVES) $(KERNEL_DEFS) \
	    -DRT_BUILD_ID=\"$$( (cat $(CORE_SRC) $(sort $(CORE_HDR)); echo "$(RT_MIN_WAVES) $(KERNEL_DEFS) $(CORE_OPT)") | sha256sum | cut -c1-16)\" $(CORE_SRC) -o $@

# -ffp-contract=off: the BVH builder's sort keys must round exactly like the reference's doubles
raytracer_amd/lib/librthost.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p raytracer_amd/lib
	$(CXX) -O2 -std=c++17 -fPIC -shared -ffp-contract=off -pthread $(HOST_SRC) -lz -o $@

raytracer_amd/bin/rtcli: raytracer_amd/csrc/rtcli.cpp raytracer_amd/lib/librtcore.so raytracer_amd/lib/librthost.so
	@mkdir -p raytracer_amd/bin
	$(CXX) -O2 -std=c++17 raytracer_amd/csrc/rtcli.cpp -Iinclude -Lraytracer_amd/lib -lrtcore -lrthost \
	    -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath-link,/opt/rocm/lib -o $@

# -ffp-contract=off: GHC forms no FMAs; neither may the restatement
oracle/_build/liboracle.so: oracle/oracle.cpp include/rtcore.h
	@mkdir -p oracle/_build
	$(CXX) -O2 -std=c++17 -fPIC -shared -ffp-contract=off -pthread oracle/oracle.cpp -o $@

clean:
	rm -rf raytracer_amd/lib raytracer_amd/bin oracle/_build

.PHONY: all core host cli oracle clean
