#!/bin/bash
# Timing experiments on the GPU box: rebuild scan_kernels with -DPCV_EXP=<n> (scan_kernels.hip lists them; results are wrong in
# some of these builds) and link over the library in the box's scratch copy of the repository.
#   bash tools/exp_build.sh 1   (0 restores the product build)
set -e
cd "$(dirname "$0")/../perceive_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPCV_EXP=$1 -Wno-unused-function -Wno-unused-result -Wno-unused-value -c scan_kernels.hip -o /tmp/scan_exp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libperceive_hip.so encoder_kernels.o /tmp/scan_exp.o context.o model.o searcher.o sqlite_build.o text_model.o tokenizer.o torch_archive.o
