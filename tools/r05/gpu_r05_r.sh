#!/bin/bash
# per-point half of gnr_geo_dual_bwd on the matrix cores: tests, A/B + float64 check
cd /root/repo; mkdir -p gpurun_out/r
timeout 900 python -m pytest tests/test_ray_tail.py tests/test_determinism.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r/tests.txt
tail -3 gpurun_out/r/tests.txt
timeout 600 python tools/ab_geo_dual.py > gpurun_out/r/ab.txt 2>&1
grep -v AB_JSON gpurun_out/r/ab.txt | tail -8 | cut -c1-1500
