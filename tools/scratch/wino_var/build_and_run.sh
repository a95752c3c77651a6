#!/bin/bash
# build experiment variants of winograd.hip (macro WINO_ABL) into standalone libs and time layer D with rocprofv3
R=/root/repo
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -fno-slp-vectorize -DWINO_ABL=$v -I $R/include -I $R/meta-interpolation_amd/csrc $R/meta-interpolation_amd/csrc/winograd.hip $R/meta-interpolation_amd/csrc/loss.hip -o /tmp/libwino_$v.so || exit 1
done
