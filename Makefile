# Builds libspectral.so (gfx950 only) in-tree so it travels to the GPU box with the snapshot.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := pyfft_amd/csrc
LIBDIR  := pyfft_amd/lib
OBJDIR  := build/obj
FLAGS   := -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -Wall -Wno-unused-function

SRCS    := $(wildcard $(CSRC)/*.hip)
OBJS    := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HDRS    := $(wildcard $(CSRC)/*.h) include/spectral.h

all: $(LIBDIR)/libspectral.so $(LIBDIR)/libspectral_packed.so

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS) Makefile
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIBDIR)/libspectral.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

# the same library with the radix-16 butterflies in packed fp32 (fft_core.h, SP_PACKED): an option that measured no
# faster; built so that tests/test_gpu_variants.py keeps it correct
POBJDIR := build/obj_packed
POBJS   := $(patsubst $(CSRC)/%.hip,$(POBJDIR)/%.o,$(SRCS))
$(POBJDIR)/%.o: $(CSRC)/%.hip $(HDRS) Makefile
	@mkdir -p $(POBJDIR)
	$(HIPCC) $(FLAGS) -DSP_PACKED=1 -c $< -o $@
$(LIBDIR)/libspectral_packed.so: $(POBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(POBJS) -o $@

clean:
	rm -rf build $(LIBDIR)/libspectral.so $(LIBDIR)/libspectral_packed.so

.PHONY: all clean
