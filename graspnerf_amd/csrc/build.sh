#!/bin/bash
# Build libgnr.so (gfx950) in-tree.  Usage: graspnerf_amd/csrc/build.sh [extra hipcc flags]
# -fno-slp-vectorize: packed f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 from SLP-packed scalar code) costs more than the
# two scalar ops it replaces when it sits between MFMAs (MI355X_MICROARCH.md); measured -1 % on k_chain, -1.2 % per step.
set -e
cd "$(dirname "$0")"
SRC="gnr_kernels.hip gnr_head.hip gnr_post.hip gnr_img.hip gnr_pack.cpp gnr_host_rng.cpp"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -fno-slp-vectorize"
# libgnr.so: the product (k_chain multiplies on the f16 matrix cores with fp32 operands as fp16 pairs).
# libgnr_f32mfma.so: the same sources with the chain on the fp32-input MFMA (-DGNR_SPLIT16=0); measurement companion only:
# bench.py times it next to the product (GNR_LIB selects the library), nothing else loads it.
hipcc $FLAGS -o libgnr.so $SRC "$@" &
P1=$!
hipcc $FLAGS -DGNR_SPLIT16=0 -o libgnr_f32mfma.so $SRC "$@" &
P2=$!
wait $P1
wait $P2
