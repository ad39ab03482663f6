#!/bin/bash
# k_ray dense layers as pair MFMAs (increment 1: q/k/v + fc): parity of the product build, A/B against the fp32 FMA build
cd /root/repo; mkdir -p gpurun_out/m
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/m/parity.txt
GNR_LIB=libgnr_rayf.so timeout 300 python tools/ab_ray_mfma.py --save gpurun_out/m/ray_f.npz > gpurun_out/m/ab_f.txt 2>&1
timeout 300 python tools/ab_ray_mfma.py --compare gpurun_out/m/ray_f.npz > gpurun_out/m/ab_m.txt 2>&1
rm -f gpurun_out/m/ray_f.npz
tail -3 gpurun_out/m/parity.txt; grep -h "run" gpurun_out/m/ab_f.txt gpurun_out/m/ab_m.txt
