#!/bin/bash
# Diagnostic build of the library with per-wave time stamps in scan_mfma8_kernel (-DPCV_STAMPS): perceive_amd/libperceive_hip_stamps.so.
# Never loaded by the product (perceive_amd/_ffi.py binds libperceive_hip.so).  To use it, copy it over libperceive_hip.so in the
# scratch copy of the repository on the GPU box:
#   gpurun -- 'cp perceive_amd/libperceive_hip_stamps.so perceive_amd/libperceive_hip.so &&
#              PCV_STAMPS_FILE=gpurun_out/s.bin python tools/ab_scan.py --rows 12500000 && python tools/read_stamps.py gpurun_out/s.bin'
# PCV_STAMPS_FILE names the file the stamps of every pass are appended to; tools/read_stamps.py prints where the waves' time went.
# PCV_STAMPS_EXTRA=-DPCV_STAMPS_TIMELINE: slots 5..7 of a streaming wave = time at the end of its 8th, 32nd, 96th block.
set -e
cd "$(dirname "$0")/../perceive_amd/csrc"
B=/tmp/pcv_stamps_build; mkdir -p $B
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPCV_STAMPS $PCV_STAMPS_EXTRA -Wno-unused-function -Wno-unused-result -Wno-unused-value"
for f in *.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o $B/${f%.hip}.o & done
for f in *.cpp; do /opt/rocm/bin/hipcc $FLAGS -x hip -c $f -o $B/${f%.cpp}.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libperceive_hip_stamps.so $B/*.o
ls -la ../libperceive_hip_stamps.so
