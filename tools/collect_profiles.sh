#!/bin/bash
# Collect the round's profile artefacts on the GPU box:  bash tools/collect_profiles.sh <tag>   (e.g. r02_a)
# 1. rocprofv3 --kernel-trace --stats of the forward leg of bench.py  -> <tag>_bench_kernel_stats.txt
# 2. PMC passes (one run per counter set, no tracing): forward (tools/run_hot.py) and sample_volume backward
#    (tools/time_volume_bwd.py)                                        -> <tag>_pmc_counters.json
# 3. rocprofv3 --kernel-trace --stats of steady-state train steps      -> <tag>_train_step_kernel_stats.txt
# 4. the default bench.py line (forward + train_step + with_backbones + cpu_baseline) -> <tag>_bench.json
# Everything lands in gpurun_out/<tag>/; copy what should be judged into profiles/.
TAG=${1:-r03_x}
MODE=${2:-all}      # all | pmc (only the counter passes + the stamped json: enough to re-tie the counters to an edited kernel source)
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
[ $MODE = all ] && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-backbones --no-f32-build --no-live-pmc > $R/gpurun_out/$TAG/bench_under_rocprof.json 2> $R/gpurun_out/$TAG/trace.log
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  N=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/$TAG/pmc/$N -o p -- python $R/tools/run_hot.py --iters 1 > $R/gpurun_out/$TAG/pmc_$N.log 2>&1
done
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/$TAG/pmc/bwd_$N -o p -- python $R/tools/time_volume_bwd.py --scenes 8 > $R/gpurun_out/$TAG/pmc_bwd_$N.log 2>&1
done
[ $MODE = all ] && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace_train -o t -- python $R/tools/train_step_bench.py --steps 3 --warmup 3 > $R/gpurun_out/$TAG/train_under_rocprof.json 2> $R/gpurun_out/$TAG/trace_train.log
cd $R
if [ $MODE = pmc ]; then
  python tools/pmc_summary.py gpurun_out/$TAG/pmc gpurun_out/$TAG/pmc_counters.json > gpurun_out/$TAG/pmc_summary.log
  cp gpurun_out/$TAG/pmc_counters.json profiles/${TAG}_pmc_counters.json
  rm -rf gpurun_out/$TAG/pmc
  exit 0
fi
DB=$(find gpurun_out/$TAG/trace -name "*.db" | head -1)
python tools/prof_summary.py $DB "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-backbones (B=32 scenes/step, 1x MI355X; includes the parity gate's launches and the 5 fully-bracketed steps)" > gpurun_out/$TAG/bench_kernel_stats.txt
DB=$(find gpurun_out/$TAG/trace_train -name "*.db" | head -1)
python tools/prof_summary.py $DB "$TAG: rocprofv3 --kernel-trace --stats -- python tools/train_step_bench.py --steps 3 --warmup 3 (8 scenes per step, 1x MI355X); kernels that started in the last 330 ms of the trace = the steady-state steps" --last-ms 330 > gpurun_out/$TAG/train_step_kernel_stats.txt
python tools/pmc_summary.py gpurun_out/$TAG/pmc gpurun_out/$TAG/pmc_counters.json > gpurun_out/$TAG/pmc_summary.log
# the counters the bench line replays must be the ones just collected: put them where bench.py looks (profiles/, newest by name)
cp gpurun_out/$TAG/pmc_counters.json profiles/${TAG}_pmc_counters.json
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
[ -x tools/ubench/split_mfma ] && timeout 60 tools/ubench/split_mfma > gpurun_out/$TAG/split_mfma_ubench.txt 2>&1
[ -x tools/ubench/mov_rates ] && timeout 60 tools/ubench/mov_rates > gpurun_out/$TAG/mov_rates_ubench.txt 2>&1
head -14 gpurun_out/$TAG/bench_kernel_stats.txt
head -30 gpurun_out/$TAG/train_step_kernel_stats.txt | cut -c1-130
tail -1 gpurun_out/$TAG/bench.json | cut -c1-300
rm -rf gpurun_out/$TAG/trace gpurun_out/$TAG/pmc gpurun_out/$TAG/trace_train
