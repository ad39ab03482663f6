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
a = ap.parse_args()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
one = make_scene(0, 'cfg2', with_query_image=False)
scenes = [one] * a.batch if a.batch > 4 else [make_scene(i, 'cfg2', with_query_image=False) for i in range(a.batch)]
bref, bque = batch_scenes(scenes)
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
print('chain kernel ms/launch:', hp.time_chain_kernel(bref, 40, iters=5))
