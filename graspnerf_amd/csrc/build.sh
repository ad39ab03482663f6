#!/bin/bash
# Build libgnr.so (gfx950) in-tree.  Usage: graspnerf_amd/csrc/build.sh [extra hipcc flags]
# -fno-slp-vectorize: packed f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 from SLP-packed scalar code) costs more than the
# two scalar ops it replaces when it sits between MFMAs (MI355X_MICROARCH.md); measured -1 % on k_chain, -1.2 % per step.
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -fno-slp-vectorize \
      -o libgnr.so gnr_kernels.hip gnr_head.hip gnr_post.hip gnr_pack.cpp gnr_host_rng.cpp "$@"
