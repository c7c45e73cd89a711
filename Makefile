# Builds the gfx950 kernel library in-tree.  `make` (or __graft_entry__.build()) produces
# mdn-vae_amd/lib/libmdnvae_hip.so; the GPU box only ever loads that prebuilt file.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := mdn-vae_amd
SRC := $(wildcard $(PKG)/csrc/*.hip) $(wildcard $(PKG)/csrc/*.cpp)
OBJ := $(patsubst $(PKG)/csrc/%,build/%.o,$(SRC))
LIB := $(PKG)/lib/libmdnvae_hip.so
CXXFLAGS := $(EXTRA) -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Iinclude

all: $(LIB)

build/%.hip.o: $(PKG)/csrc/%.hip $(PKG)/csrc/common.h include/mdnvae_hip.h
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

build/%.cpp.o: $(PKG)/csrc/%.cpp $(PKG)/csrc/common.h include/mdnvae_hip.h
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p $(PKG)/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJ)

clean:
	rm -rf build $(LIB)

.PHONY: all clean
