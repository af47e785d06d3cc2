# Builds the product library (HIP kernels + C ABI, gfx950 only) without Python:
#
#   make                      decaf377_amd/lib/libdecaf377_amd.so, libdecaf377_amd_check.so and the CPU oracle
#   make lib                  the product library only (what rust/build.rs and a C consumer link)
#   make lib FB_BITS=14       options below; a non-default build goes to build/variants/<VARIANT>.so when VARIANT is set
#   make check                the same sources with -DD377_CHECK_INVARIANTS (the reference's debug assertions)
#   make oracle               test infrastructure (oracle/Makefile), never linked into the product
#   make tools                the native pieces the kept developer tools load or run (tools/README.md): build/row_proto.so,
#                             tools/clock_vs_traffic, tools/valu_mix, tools/valu_mix2
#
# Options ("config / flags"):
#   ARCH              offload architecture (gfx950; the kernels are written for nothing else)
#   FB_BITS           comb width of the fixed-base table: 23 (default: 5.9 GB), 21 (1.6 GB), 18 (235 MB), 16, 14, 12 or 8   -> -DD377_FB_BITS
#   DCB_K             elements per lane per batched inversion: 8 (default), 4, 16      -> -DD377_DCB_K
#   WAVES_PER_SIMD    occupancy the chunked kernels are built for: 2 (default)         -> -DD377_WAVES_PER_SIMD
#   CHECK_INVARIANTS  1: curve-equation checks on the device in the product library too
#   EXTRA             further -D flags (A/B builds: tools/build_variant.sh)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
FB_BITS ?= 23
DCB_K ?= 8
WAVES_PER_SIMD ?= 2
CHECK_INVARIANTS ?= 0
EXTRA ?=
VARIANT ?=

CSRC := decaf377_amd/csrc
LIBDIR := decaf377_amd/lib
HDRS := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.inc) include/decaf377_amd.h
UNITS := d377 msm codec_chunked batch_msm
DEFS := -DD377_FB_BITS=$(FB_BITS) -DD377_DCB_K=$(DCB_K) -DD377_WAVES_PER_SIMD=$(WAVES_PER_SIMD) $(EXTRA)
ifeq ($(CHECK_INVARIANTS),1)
DEFS += -DD377_CHECK_INVARIANTS
endif
HIPFLAGS := -O3 --offload-arch=$(ARCH) -std=c++17 -fPIC

ifeq ($(VARIANT),)
OBJDIR := build/obj
LIB := $(LIBDIR)/libdecaf377_amd.so
else
OBJDIR := build/obj_$(VARIANT)
LIB := build/variants/$(VARIANT).so
endif
CHECK_OBJDIR := build/obj_check
CHECK_LIB := $(LIBDIR)/libdecaf377_amd_check.so

all: lib check oracle
lib: $(LIB)
check: $(CHECK_LIB)

# the options are part of the build's identity: a change of FB_BITS etc. rebuilds the objects
$(OBJDIR)/.flags: FORCE
	@mkdir -p $(OBJDIR)
	@echo '$(HIPCC) $(HIPFLAGS) $(DEFS)' | cmp -s - $@ || echo '$(HIPCC) $(HIPFLAGS) $(DEFS)' > $@
$(CHECK_OBJDIR)/.flags: FORCE
	@mkdir -p $(CHECK_OBJDIR)
	@echo '$(HIPCC) $(HIPFLAGS) $(DEFS) -DD377_CHECK_INVARIANTS' | cmp -s - $@ || echo '$(HIPCC) $(HIPFLAGS) $(DEFS) -DD377_CHECK_INVARIANTS' > $@

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS) $(OBJDIR)/.flags
	$(HIPCC) $(HIPFLAGS) $(DEFS) -c $< -o $@
$(CHECK_OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS) $(CHECK_OBJDIR)/.flags
	$(HIPCC) $(HIPFLAGS) $(DEFS) -DD377_CHECK_INVARIANTS -c $< -o $@

$(LIB): $(UNITS:%=$(OBJDIR)/%.o)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $^
$(CHECK_LIB): $(UNITS:%=$(CHECK_OBJDIR)/%.o)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $^

oracle:
	$(MAKE) -C oracle

# developer tools (tools/README.md): what tools/row_proto.py, row_point_check.py, row_invert_check.py dlopen and what
# clock_vs_traffic.sh / bench.py's MAC-ceiling note run
tools: build/row_proto.so tools/clock_vs_traffic tools/valu_mix tools/valu_mix2
build/row_proto.so: tools/row_proto.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -shared $< -o $@
tools/clock_vs_traffic: tools/clock_vs_traffic.hip $(HDRS)
	$(HIPCC) -O3 --offload-arch=$(ARCH) -std=c++17 -I$(CSRC) $< -o $@
tools/valu_mix: tools/valu_mix.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) -std=c++17 $< -o $@
tools/valu_mix2: tools/valu_mix2.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) -std=c++17 $< -o $@

clean:
	rm -rf build/obj build/obj_check build/obj_* build/variants $(LIBDIR)/*.so
	$(MAKE) -C oracle clean

FORCE:
.PHONY: all lib check oracle tools clean FORCE
