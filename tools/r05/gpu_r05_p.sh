#!/bin/bash
# final tree: whole GPU suite, then tools/collect_profiles.sh ${TAG} (rocprofv3 stats of the bench and of train steps, PMC passes, default bench line)
TAG=${1:-r05_p}
cd /root/repo; mkdir -p gpurun_out/${TAG}
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/${TAG}/gpu_suite.log
tail -3 gpurun_out/${TAG}/gpu_suite.log
timeout 1500 bash tools/collect_profiles.sh ${TAG} all > gpurun_out/${TAG}/collect.log 2>&1
tail -50 gpurun_out/${TAG}/collect.log | cut -c1-200
