#!/bin/bash
# round 5, GPU call H: partner-wavefront k_view2_bwd -- gradient tests (bounded), A/B per kernel, train step
R=$PWD; T=r05_h; O=$R/gpurun_out/$T; mkdir -p $O
timeout 300 python -m pytest tests/test_bwd_twins.py tests/test_determinism.py -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 300 python -m pytest tests/test_train_step.py -m gpu -x -q > $O/tests_train.log 2>&1; tail -3 $O/tests_train.log
cat > /tmp/ab_pw2.py <<'PY'
import sys, time, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
from graspnerf_amd import weights, _lib
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
wnp = dict(np.load('/root/repo/tests/golden/weights_seed0.npz'))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
can = weights.canonical_blob(wnp, 'coarse')
hp.set_bwd_weights(weights.pack_bwd(can))
can_dev = torch.from_numpy(can).cuda()
bref, _ = batch_scenes([make_scene(i, 'cfg2', with_query_image=False) for i in range(8)])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
dvol = torch.randn(8, 1, 40, 40, 40, device='cuda')
prep = hp.prepare(bref, 40); hp.sample_volume_train(bref, 40, prepared=prep)
L = _lib.lib()
res, outs = {}, {}
for rep in range(3):
    for on in (3, 1 + 0, 0):
        name = {3: 'both partner', 1: 'both partner', 0: 'single'}[on]
        if on == 1: continue
        L.gnr_debug_view1_partner(on)
        for _ in range(2): o = hp.sample_volume_bwd(dvol, can_dev)
        torch.cuda.synchronize(); _lib.timing_begin(only='k_view')
        for _ in range(5): o = hp.sample_volume_bwd(dvol, can_dev)
        torch.cuda.synchronize(); t = _lib.timing_end()
        res.setdefault(name, []).append({k.split('@')[0]: round(v[1] / 5, 4) for k, v in t.items()})
        outs[on] = o
L.gnr_debug_view1_partner(1)
d = {k: float((a - b).abs().max() / b.abs().max()) for k, a, b in zip(('dcan', 'dray', 'dimg'), outs[3], outs[0])}
print(json.dumps({'ms_volume_8_scenes': res, 'partner_vs_single_rel_diff': d}))
PY
timeout 300 python /tmp/ab_pw2.py > $O/ab_pw2.json 2> $O/ab_pw2.err; cat $O/ab_pw2.json; tail -3 $O/ab_pw2.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backbones --no-f32-build --no-train-2cpu > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); t=d['train_step']
print('fwd', d['value'], 'train', t['value'], t['ms_per_step'], 'readback', t['value_with_per_step_readback'], t['split_ms_per_step'])
print('   ', {k:v for k,v in t['hip_kernels_ms_per_step'].items() if v>0.3})
PY
