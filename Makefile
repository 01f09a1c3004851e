# Builds libsow_amd.so (HIP, gfx950 only) in-tree.  `make` or `python -c "import __graft_entry__ as g; g.build()"`.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := sow_amd/csrc
ifdef STAMPS
OBJDIR  := build/obj_stamps
LIB     := sow_amd/lib/libsow_amd_stamps.so
EXTRA   := -DSOW_STAMPS
else ifdef VARIANT
# A/B builds: make VARIANT=name DEFS="-DSOMETHING" -> sow_amd/lib/libsow_amd_name.so (select with SOW_AMD_LIB=...)
OBJDIR  := build/obj_$(VARIANT)
LIB     := sow_amd/lib/libsow_amd_$(VARIANT).so
EXTRA   := $(DEFS)
else
OBJDIR  := build/obj
LIB     := sow_amd/lib/libsow_amd.so
EXTRA   :=
endif
SRCS    := $(wildcard $(CSRC)/*.hip)
OBJS    := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HDRS    := $(wildcard $(CSRC)/*.hpp) include/sow_amd.h
CXXFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) $(EXTRA) -Wall -Wno-unused-function -Wno-unused-variable \
            -Wno-unused-but-set-variable

all: $(LIB)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p sow_amd/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

clean:
	rm -rf build sow_amd/lib/*.so

.PHONY: all clean
