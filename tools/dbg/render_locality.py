"""Is the render instantiation's per-point gap to the volume instantiation memory latency?  Times k_chain.render with the benchmark's 512
random pixels per scene against 512 pixels of one compact 16 x 32 block (taps of neighbouring rays share cache lines, as the volume's do)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights, _lib
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
scenes = [make_scene(i, 'cfg2', with_query_image=False) for i in range(32)]
bref, bque = batch_scenes(scenes)
out = {}
for name in ('random pixels', 'compact 16 x 32 block', 'compact block, morton'):
    q = {k: v.copy() for k, v in bque.items()}
    if name != 'random pixels':
        ys, xs = np.meshgrid(np.arange(16), np.arange(32), indexing='ij')
        blk = np.stack([xs.reshape(-1) + 240, ys.reshape(-1) + 136], -1).astype(q['coords'].dtype)
        q['coords'][:] = blk[None]
    hp.set_option('ray_order_morton', name.endswith('morton'))
    r = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
    qq = {k: torch.from_numpy(v).cuda() for k, v in q.items()}
    for _ in range(3):
        prep = hp.prepare(r, 40, 512, 40); hp.sample_volume(r, 40, prepared=prep); hp.render(r, qq, prepared=prep)
    torch.cuda.synchronize()
    _lib.timing_begin()
    for _ in range(5):
        prep = hp.prepare(r, 40, 512, 40); hp.sample_volume(r, 40, prepared=prep); hp.render(r, qq, prepared=prep)
    torch.cuda.synchronize()
    t = _lib.timing_end()
    out[name] = {k: round(v[1] / 5, 4) for k, v in t.items() if k.startswith('k_chain') or k.startswith('k_ray')}
print(json.dumps(out, indent=1))
