"""Run the hot path a few times (for profilers).  python tools/run_hot.py [--batch B] [--iters N] [--volume-only]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--volume-only', action='store_true')
ap.add_argument('--sort-rays', action='store_true', help='experiment: visit the rays of a scene in pixel-Morton order')
ap.add_argument('--distinct', action='store_true', help='32 distinct scenes instead of one scene repeated')
ap.add_argument('--ray-order', action='store_true', help='GNR_OPT_RAY_ORDER_MORTON: the library lays the rays out in pixel-Morton order internally (bit-identical outputs)')
a = ap.parse_args()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
if a.ray_order:
    hp.set_option('ray_order_morton', True)
one = make_scene(0, 'cfg2', with_query_image=False)
scenes = [one] * a.batch if (a.batch > 4 and not a.distinct) else [make_scene(i, 'cfg2', with_query_image=False) for i in range(a.batch)]
bref, bque = batch_scenes(scenes)
if a.sort_rays:
    def morton(x, y):
        k = np.zeros_like(x, dtype=np.int64)
        for b in range(10):
            k |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
        return k
    c = bque['coords'].astype(np.int64)
    order = np.argsort(morton(c[..., 0], c[..., 1]), axis=1)
    bque['coords'] = np.take_along_axis(bque['coords'], order[..., None], 1)
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
for it in range(a.iters):
    torch.cuda.synchronize(); t = time.perf_counter()
    prep = hp.prepare(bref, 40, 512, 40)
    hp.sample_volume(bref, 40, prepared=prep)
    if not a.volume_only:
        hp.render(bref, bque, prepared=prep)
    torch.cuda.synchronize()
    print(f'iter {it}: {(time.perf_counter() - t) * 1e3:.3f} ms for {a.batch} scenes', flush=True)
from graspnerf_amd import _lib
_lib.timing_begin()
for it in range(5):
    prep = hp.prepare(bref, 40, 512, 40)
    hp.sample_volume(bref, 40, prepared=prep)
    hp.render(bref, bque, prepared=prep)
torch.cuda.synchronize()
print({k: round(v[1] / v[0], 4) for k, v in _lib.timing_end().items()})
print('chain kernel ms/launch:', hp.time_chain_kernel(bref, 40, iters=5))
