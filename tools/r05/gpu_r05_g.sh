#!/bin/bash
# round 5, GPU call G: train step with the partner-wavefront k_view1_bwd (per-kernel table), whole suite
R=$PWD; T=r05_g; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; tail -3 $O/suite.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backbones --no-f32-build --no-train-2cpu > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); t=d['train_step']
print('fwd', d['value'], 'train', t['value'], t['ms_per_step'], 'readback', t['value_with_per_step_readback'], t['split_ms_per_step'])
print('   ', {k:v for k,v in t['hip_kernels_ms_per_step'].items() if v>0.3})
print('   roofline', t['roofline']['frac'], t['roofline']['ms_per_launch'])
PY
