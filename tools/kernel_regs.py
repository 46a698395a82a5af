#!/usr/bin/env python3
"""Registers and scratch of the kernels in a -save-temps gfx950 assembly file:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -c csrc/scan_kernels.hip -o /tmp/x.o && python tools/kernel_regs.py *gfx950*.s [name filter]"""
import re
import sys

name = None
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for line in open(sys.argv[1]):
    m = re.search(r"\.amdhsa_kernel (\S+)", line)
    if m:
        name, rec = m.group(1), {}
    for key in ("next_free_vgpr", "private_segment_fixed_size", "group_segment_fixed_size", "accum_offset"):
        m = re.search(r"\.amdhsa_" + key + r" (\d+)", line)
        if m and name:
            rec[key] = int(m.group(1))
    if ".end_amdhsa_kernel" in line and name:
        if flt in name:
            print(f"{name[:110]:110s} vgpr {rec.get('next_free_vgpr')} scratch {rec.get('private_segment_fixed_size')} lds {rec.get('group_segment_fixed_size')}")
        name = None
