#!/bin/bash
# Build libgnr.so (gfx950) in-tree.  Usage: graspnerf_amd/csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -o libgnr.so gnr_kernels.hip gnr_head.hip gnr_pack.cpp "$@"
