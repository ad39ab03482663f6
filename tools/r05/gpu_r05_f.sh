#!/bin/bash
# round 5, GPU call F: the partner-wavefront k_view1_bwd -- gradients first (bounded: every test under timeout), then A/B against k_view1_bwd
R=$PWD; T=r05_f; O=$R/gpurun_out/$T; mkdir -p $O
timeout 300 python -m pytest tests/test_bwd_twins.py tests/test_determinism.py -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python -m pytest tests/test_train_step.py -m gpu -x -q > $O/tests_train.log 2>&1; tail -3 $O/tests_train.log
cat > /tmp/ab_pw.py <<'PY'
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
res = {}
outs = {}
for rep in range(3):
    for on in (1, 0):
        L.gnr_debug_view1_partner(on)
        for _ in range(2): o = hp.sample_volume_bwd(dvol, can_dev)
        torch.cuda.synchronize(); _lib.timing_begin(only='k_view1_bwd')
        for _ in range(5): o = hp.sample_volume_bwd(dvol, can_dev)
        torch.cuda.synchronize(); t = _lib.timing_end()
        res.setdefault('partner' if on else 'single', []).append(round(sum(v[1] for v in t.values()) / 5, 4))
        outs[on] = o
L.gnr_debug_view1_partner(1)
d = {k: float((a - b).abs().max() / b.abs().max()) for k, a, b in zip(('dcan', 'dray', 'dimg'), outs[1], outs[0])}
print(json.dumps({'k_view1_bwd_ms_volume_8_scenes': res, 'partner_vs_single_rel_diff': d}))
PY
timeout 300 python /tmp/ab_pw.py > $O/ab_pw.json 2> $O/ab_pw.err; cat $O/ab_pw.json; tail -3 $O/ab_pw.err
