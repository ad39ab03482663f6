#!/bin/bash
# Build an A/B variant of the library in-tree:  tools/build_variant.sh NAME [extra hipcc flags]  ->  graspnerf_amd/csrc/libgnr_NAME.so
# (selected at load time with GNR_LIB=libgnr_NAME.so; tools/ab_chain.py times a list of them in one GPU call)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../graspnerf_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -fno-slp-vectorize -o libgnr_$NAME.so \
  gnr_kernels.hip gnr_head.hip gnr_post.hip gnr_img.hip gnr_pack.cpp gnr_pack_dev.hip gnr_host_rng.cpp "$@"
