#!/bin/bash
# round 5, GPU call K: what the N > 1 gradient exchange costs a step, at the box's CPU budget and at 2 CPUs, with and without c10d's monitoring threads
R=$PWD; T=r05_k; O=$R/gpurun_out/$T; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 400 python tools/train_step_bench.py --steps 12 --warmup 20 --flat-exchange-steps 12 $EXTRA > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/$name.json').read().splitlines() if l.startswith('{"metric"')][-1])
print('$name', round(d['ms_per_step'],2), d['with_flat_gradient_exchange'])
PY
}
EXTRA="" run cpus16_default A=1
EXTRA="--cpus 2" run cpus2_default A=1
EXTRA="--cpus 2" run cpus2_nomonitor TORCH_NCCL_ENABLE_MONITORING=0
EXTRA="--cpus 2" run cpus2_nomonitor_nowatch TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0
EXTRA="--cpus 2" run cpus2_avoidrecord TORCH_NCCL_AVOID_RECORD_STREAMS=1
