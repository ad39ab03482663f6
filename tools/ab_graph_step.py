"""The B = 32 forward step launched eagerly (three ABI calls, ~17 kernel launches) against one hipGraph replay of the same calls.
python tools/ab_graph_step.py [--steps 200]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=200)
a = ap.parse_args()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
bref, bque = batch_scenes([make_scene(i, 'cfg2', with_query_image=False) for i in range(32)])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
def step():
    prep = hp.prepare(bref, 40, 512, 40)
    vol = hp.sample_volume(bref, 40, prepared=prep)
    out = hp.render(bref, bque, prepared=prep)
    return vol, out
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
v_eager, o_eager = step()
v_eager = v_eager.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    v_g, o_g = step()
g.replay(); torch.cuda.synchronize()
def timed(f, n):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3 / n
res = {'volume_bitwise_equal': bool(torch.equal(v_eager, v_g)), 'eager_ms': [], 'graph_ms': []}
for _ in range(3):
    res['eager_ms'].append(round(timed(step, a.steps), 4))
    res['graph_ms'].append(round(timed(g.replay, a.steps), 4))
print(json.dumps(res))
