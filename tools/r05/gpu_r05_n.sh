#!/bin/bash
# bit-reproducible feature-map gradients (gnr_feature_grad_mode(1)): tests in both modes, cost per launch and per train step
cd /root/repo; mkdir -p gpurun_out/n
timeout 1200 python -m pytest tests/test_determinism.py tests/test_bwd_twins.py tests/test_train_step.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/n/tests.txt
tail -4 gpurun_out/n/tests.txt
timeout 300 python tools/time_volume_bwd.py --scenes 8 > gpurun_out/n/vol_bwd_float.txt 2>&1
timeout 300 python tools/time_volume_bwd.py --scenes 8 --fixed-point-feature-grads > gpurun_out/n/vol_bwd_fixed.txt 2>&1
tail -12 gpurun_out/n/vol_bwd_float.txt; tail -12 gpurun_out/n/vol_bwd_fixed.txt
timeout 400 python tools/train_step_bench.py --steps 16 --warmup 24 2>/dev/null | grep '"metric"' > gpurun_out/n/train_float.json
timeout 400 python tools/train_step_bench.py --steps 16 --warmup 24 --reproducible-feature-grads 2>/dev/null | grep '"metric"' > gpurun_out/n/train_fixed.json
python - <<'PY'
import json
for n in ('float', 'fixed'):
    d = json.loads(open('gpurun_out/n/train_%s.json' % n).read())
    print(n, round(d['ms_per_step'], 3), 'ms/step', round(d['value'], 2), 'scenes/s', d['loss'])
PY
