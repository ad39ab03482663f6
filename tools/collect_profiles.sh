#!/bin/bash
# Collect the round's profile artefacts on the GPU box:  bash tools/collect_profiles.sh <tag>   (e.g. r01_d)
# 1. rocprofv3 --kernel-trace --stats of the default bench.py run  -> profiles/<tag>_bench_kernel_stats.txt + bench JSON
# 2. PMC passes (one run per counter set, no tracing)              -> profiles/<tag>_pmc_counters.json
TAG=${1:-r01_x}
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$TAG/bench_under_rocprof.json 2> $R/gpurun_out/$TAG/trace.log
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/$TAG/pmc/$N -o p -- python $R/tools/run_hot.py --iters 1 > $R/gpurun_out/$TAG/pmc_$N.log 2>&1
done
cd $R
DB=$(find gpurun_out/$TAG/trace -name "*.db" | head -1)
python tools/prof_summary.py $DB "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (B=32 scenes/step, 1x MI355X)" > gpurun_out/$TAG/bench_kernel_stats.txt
python tools/pmc_summary.py gpurun_out/$TAG/pmc gpurun_out/$TAG/pmc_counters.json
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
head -12 gpurun_out/$TAG/bench_kernel_stats.txt
tail -1 gpurun_out/$TAG/bench.json | cut -c1-300
rm -rf gpurun_out/$TAG/trace gpurun_out/$TAG/pmc
