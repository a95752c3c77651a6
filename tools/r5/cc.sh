#!/bin/bash
# tools/r5/cc.sh UNIT.hip [flags]: compile one translation unit with the library's flags (warnings shown), object to /tmp
R=/root/repo; CS=$R/meta-interpolation_amd/csrc
U=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -Wall -Wno-unused-function -I $R/include -I $CS "$@" -c $CS/$U -o /tmp/cc_$(basename $U .hip).o 2>&1 | grep -v "^clang++: warning: argument unused" | grep -B1 -A4 "warning:\|error:" | head -60
