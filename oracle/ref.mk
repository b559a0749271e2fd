# TEST INFRASTRUCTURE — oracle/_ref: the reference's OWN native op (the pybind11 module `D3D`,
# 3D/dcn/src/vision.cpp + deform_conv.h + cuda/deform_conv_cuda.cu + cuda/deform_im2col_cuda.cuh + cpu/deform_cpu.cpp),
# compiled UNMODIFIED from where the sources lie under /root/reference by hipcc for gfx950.  The reference's build
# (3D/dcn/setup.py) refuses without CUDA and its .cuh includes THC/THCAtomics.cuh (gone from modern torch); the four shim
# headers in oracle/ref_shim/ (cuda.h, cuda_runtime.h, THC/THCAtomics.cuh, ATen/cuda/CUDAContext.h: CUDA runtime names ->
# HIP, current-stream accessor, old AT_DISPATCH spelling) are all it takes.  No reference source is copied into this repo;
# outputs go to oracle/_ref/ only (git-ignored, travels to the GPU box).  Used by tests/ to pin the C oracle and the HIP
# kernels to the reference's own arithmetic; never linked or imported by the product.
#
#   make -f ref.mk            (from oracle/; needs /root/reference and torch's headers)
REF    ?= /root/reference/3D/dcn/src
HIPCC  ?= /opt/rocm/bin/hipcc
PY     ?= python3
TORCH  := $(shell $(PY) -c 'import os,torch;print(os.path.dirname(torch.__file__))')
PYINC  := $(shell $(PY) -c 'import sysconfig;print(sysconfig.get_paths()["include"])')
PBINC  := $(shell $(PY) -c 'import pybind11;print(pybind11.get_include())')
OUT    := _ref
FLAGS  := --offload-arch=gfx950 -O2 -fPIC -std=c++17 -w -x hip -DWITH_CUDA -DUSE_ROCM -D__HIP_PLATFORM_AMD__ \
          -DTORCH_EXTENSION_NAME=D3D -DTORCH_API_INCLUDE_EXTENSION_H \
          -Iref_shim -I$(REF) -I$(TORCH)/include -I$(TORCH)/include/torch/csrc/api/include -I$(PYINC) -I$(PBINC)
OBJS   := $(OUT)/deform_conv_cuda.o $(OUT)/vision.o $(OUT)/deform_cpu.o

all: $(OUT)/D3D.so

$(OUT)/deform_conv_cuda.o: $(REF)/cuda/deform_conv_cuda.cu $(REF)/cuda/deform_im2col_cuda.cuh $(wildcard ref_shim/*.h ref_shim/*/*.cuh ref_shim/*/*/*.h)
	@mkdir -p $(OUT)
	$(HIPCC) $(FLAGS) -c $< -o $@
$(OUT)/vision.o: $(REF)/vision.cpp $(REF)/deform_conv.h
	@mkdir -p $(OUT)
	$(HIPCC) $(FLAGS) -c $< -o $@
$(OUT)/deform_cpu.o: $(REF)/cpu/deform_cpu.cpp
	@mkdir -p $(OUT)
	$(HIPCC) $(FLAGS) -c $< -o $@
$(OUT)/D3D.so: $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OBJS) -L$(TORCH)/lib -lc10 -lc10_hip -ltorch -ltorch_cpu -ltorch_hip -ltorch_python \
	    -Wl,-rpath,$(TORCH)/lib -o $@
clean:
	rm -rf $(OUT)
.PHONY: all clean
