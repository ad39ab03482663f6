"""A/B of the Morton traversal order of the inference render passes' rays (csrc/gnr_kernels.hip k_ray_order) in ONE process:
GNR_OPT_RAY_ORDER_MORTON off / on around the same B = 32 forward step; per-kernel ms (HIP events on the launch stream) and a
bit-equality check of every render output between the two orders.   python tools/ab_ray_order.py [--batch 32] [--steps 20]"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights, _lib
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--steps', type=int, default=20)
a = ap.parse_args()
L = _lib.lib()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
bref, bque = batch_scenes([make_scene(i, 'cfg2') for i in range(a.batch)])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}


def step():
    prep = hp.prepare(bref, 40, 512, 40)
    hp.sample_volume(bref, 40, prepared=prep)
    return hp.render(bref, bque, prepared=prep, debug=True)


res, outs = {}, {}
for rep in range(2):
    for on in (0, 1):
        hp.set_option('ray_order_morton', bool(on))
        for _ in range(3):
            o = step()
        torch.cuda.synchronize()
        outs[on] = o
        _lib.timing_begin()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        t = _lib.timing_end()
        res[(rep, on)] = {k: round(v[1] / a.steps, 4) for k, v in t.items() if k.startswith(('k_chain.render', 'k_ray.render', 'k_ray_order', 'k_points_rays'))}
        print('sorted' if on else 'caller order', 'run', rep, res[(rep, on)], 'render-side total', round(sum(res[(rep, on)].values()), 4), flush=True)
hp.set_option('ray_order_morton', False)                   # the default
same = True
for lvl in (0, 1):
    for k in outs[0][lvl]:
        if not torch.equal(outs[0][lvl][k], outs[1][lvl][k]):
            same = False
            print('DIFFERS', lvl, k, float((outs[0][lvl][k].float() - outs[1][lvl][k].float()).abs().max()))
same = same and torch.equal(outs[0][2], outs[1][2])
print('AB_JSON ' + json.dumps({'bit_identical_outputs': bool(same), 'runs': {f'{r}_{"sorted" if o else "caller"}': v for (r, o), v in res.items()}}))
