#!/bin/bash
# round 5, GPU call D: per-kernel table of the train step, pair build against the fp32-MFMA backward build, alternating
R=$PWD; T=r05_d; O=$R/gpurun_out/$T; mkdir -p $O
timeout 300 python -m pytest tests/test_model_mirror.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2; do
for L in libgnr.so libgnr_bwdf32.so; do
  GNR_LIB=$L timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backbones --no-f32-build --no-train-2cpu > $O/bench_${L}_$rep.json 2> $O/bench_${L}_$rep.err
  python - <<PY
import json
d=json.loads(open('$O/bench_${L}_$rep.json').read().strip().splitlines()[-1]); t=d['train_step']
print('$L rep $rep', 'fwd', d['value'], 'train', t['value'], t['ms_per_step'], 'readback', t['value_with_per_step_readback'], t['split_ms_per_step'])
print('   ', {k:v for k,v in t['hip_kernels_ms_per_step'].items() if v>0.3})
PY
done
done
