"""Scene-group pipelining of the forward step (round-4 review item 7), measured through the product library:
the B = 32 step as three sweeps over all scenes (prepare, volume, coarse + fine render: what bench.py times) against the same
step issued group by group (G scenes: prepare -> volume -> render of the group, then the next group), so that a scene's inputs
are consumed by all three chain launches while they are still in the Infinity Cache.  Outputs must be bitwise identical.
    python tools/ab_scene_groups.py [--groups 8 16] [--steps 30] [--out gpurun_out/x.json] [--once G]
--once G: one step in groups of G (0 = ungrouped) and exit -- the command rocprofv3 --pmc wraps."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--groups', type=int, nargs='+', default=[8, 16])
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--repeat', type=int, default=3)
ap.add_argument('--once', type=int, default=-1)
ap.add_argument('--out', default='')
a = ap.parse_args()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
B = a.batch
bref, bque = batch_scenes([make_scene(i, 'cfg2') for i in range(B)])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}


def step(G):
    if G <= 0 or G >= B:
        prep = hp.prepare(bref, 40, 512, 40)
        vol = hp.sample_volume(bref, 40, prepared=prep)
        co, fi = hp.render(bref, bque, prepared=prep)
        return [vol], [co], [fi]
    vols, cos, fis = [], [], []
    for g0 in range(0, B, G):
        r = {k: v[g0:g0 + G] for k, v in bref.items()}
        q = {k: v[g0:g0 + G] for k, v in bque.items()}
        prep = hp.prepare(r, 40, 512, 40)
        vols.append(hp.sample_volume(r, 40, prepared=prep))
        co, fi = hp.render(r, q, prepared=prep)
        cos.append(co), fis.append(fi)
    return vols, cos, fis


if a.once >= 0:
    step(a.once); torch.cuda.synchronize()
    step(a.once); torch.cuda.synchronize()
    sys.exit(0)
ref_v, ref_c, ref_f = step(0)
res = {'batch': B, 'rows': []}
for G in a.groups:
    v, c, f = step(G)
    same = torch.equal(torch.cat(v), ref_v[0])
    for k in ref_c[0]:
        same = same and torch.equal(torch.cat([x[k] for x in c]), ref_c[0][k]) and torch.equal(torch.cat([x[k] for x in f]), ref_f[0][k])
    res[f'groups_of_{G}_bitwise_equal'] = bool(same)
for rep in range(a.repeat):
    for G in [0] + a.groups:
        for _ in range(3):
            step(G)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(G)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        res['rows'].append({'scenes_per_group': G or B, 'ms_per_step': round(ms, 4), 'scenes_per_s': round(B / ms * 1e3, 1)})
print(json.dumps(res, indent=1))
if a.out:
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, 'w'), indent=1)
