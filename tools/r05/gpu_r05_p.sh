#!/bin/bash
# final tree: whole GPU suite, then tools/collect_profiles.sh r05_p (rocprofv3 stats of the bench and of train steps, PMC passes, default bench line)
cd /root/repo; mkdir -p gpurun_out/r05_p
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r05_p/gpu_suite.log
tail -3 gpurun_out/r05_p/gpu_suite.log
timeout 1500 bash tools/collect_profiles.sh r05_p all > gpurun_out/r05_p/collect.log 2>&1
tail -50 gpurun_out/r05_p/collect.log | cut -c1-200
