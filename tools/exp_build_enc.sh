#!/bin/bash
# Timing experiments on the GPU box for the encoder kernels: rebuild encoder_kernels with -DPCV_ENC_EXP=<n> (results are wrong
# in some of these builds) and link over the library in the box's scratch copy of the repository.  0 restores the product build;
# `base` builds perceive_amd/csrc/enc_base.hip.txt (a copy of an older encoder_kernels.hip put there for an A/B) instead.
set -e
cd "$(dirname "$0")/../perceive_amd/csrc"
SRC=encoder_kernels.hip; DEF=-DPCV_ENC_EXP=$1
if [ "$1" = base ]; then cp enc_base.hip.txt /tmp/enc_base.hip; SRC=/tmp/enc_base.hip; DEF="-I."; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $DEF -I../../include -Wno-unused-function -Wno-unused-result -Wno-unused-value -c $SRC -o /tmp/enc_exp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libperceive_hip.so /tmp/enc_exp.o scan_kernels.o context.o model.o searcher.o sqlite_build.o text_model.o tokenizer.o torch_archive.o
