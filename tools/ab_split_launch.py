"""GNR_OPT_SPLIT_LAUNCH against the single-stream step in one process: B = 32 forward step, outputs compared bitwise, wall time per step
(whole step: prepare + sample_volume + render) alternating between the two settings.  python tools/ab_split_launch.py [--steps 100]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=100)
ap.add_argument('--batch', type=int, default=32)
a = ap.parse_args()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
bref, bque = batch_scenes([make_scene(i, 'cfg2', with_query_image=False) for i in range(a.batch)])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
def step():
    prep = hp.prepare(bref, 40, 512, 40)
    vol = hp.sample_volume(bref, 40, prepared=prep)
    out = hp.render(bref, bque, prepared=prep)
    return vol, out
def timed(n):
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3 / n
hp.set_option('split_launch', False); v0, _ = step(); v0 = v0.clone()
hp.set_option('split_launch', True); v1, _ = step(); v1 = v1.clone()
torch.cuda.synchronize()
res = {'volume_bitwise_equal': bool(torch.equal(v0, v1)), 'single_stream_ms': [], 'split_launch_ms': []}
for rep in range(3):
    hp.set_option('split_launch', False); res['single_stream_ms'].append(round(timed(a.steps), 4))
    hp.set_option('split_launch', True); res['split_launch_ms'].append(round(timed(a.steps), 4))
print(json.dumps(res))
