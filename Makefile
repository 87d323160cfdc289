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

all: $(LIBDIR)/libspectral.so

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS) Makefile
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIBDIR)/libspectral.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

clean:
	rm -rf build $(LIBDIR)/libspectral.so

.PHONY: all clean
