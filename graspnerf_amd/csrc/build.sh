#!/bin/bash
# Build libgnr.so (gfx950) in-tree.  Usage: graspnerf_amd/csrc/build.sh [extra hipcc flags]
# -fno-slp-vectorize: packed f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 from SLP-packed scalar code) costs more than the
# two scalar ops it replaces when it sits between MFMAs (MI355X_MICROARCH.md); measured -1 % on k_chain, -1.2 % per step.
set -e
cd "$(dirname "$0")"
SRC="gnr_kernels.hip gnr_head.hip gnr_post.hip gnr_img.hip gnr_pack.cpp gnr_pack_dev.hip gnr_host_rng.cpp"
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
# libgnr_torch.so: torch.ops.graspnerf.* (gnr_torch_ops.cpp: TORCH_LIBRARY registration of the C ABI; host C++ only, links libgnr.so).
# Optional: nothing on the product path uses the registered operators, so an environment without the pieces it needs (g++, the
# torch headers, -lc10_hip / -ltorch_hip) gets a warning, not a failed package build (tests/test_torch_ops.py skips without it).
PY=${PYTHON:-$(command -v python3 || command -v python)}
ROCM=${ROCM_PATH:-$( (hipconfig --rocmpath 2>/dev/null) || echo /opt/rocm)}
(
  set -e
  TORCH_DIR=$($PY -c 'import os, torch; print(os.path.dirname(torch.__file__))')
  ABI=$($PY -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')
  g++ -O2 -std=c++17 -shared -fPIC -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI=$ABI -Wno-deprecated-declarations \
      -I$TORCH_DIR/include -I$TORCH_DIR/include/torch/csrc/api/include -I$ROCM/include \
      gnr_torch_ops.cpp -o libgnr_torch.so -L$TORCH_DIR/lib -ltorch -ltorch_cpu -lc10 -lc10_hip -ltorch_hip -L. -lgnr \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,$TORCH_DIR/lib
) || { rm -f libgnr_torch.so; echo "WARNING: libgnr_torch.so (torch.ops.graspnerf.*) was not built; the C-ABI library libgnr.so is complete without it" >&2; }
