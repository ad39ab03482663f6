"""A/B of k_ray's per-sample dense layers on the f16 matrix cores against their fp32 FMA form.  One library per process (GNR_LIB);
the same B = 32 forward step; per-kernel ms (HIP events on the launch stream); --save writes the outputs, --compare reads the other
build's and prints the largest differences.
  * adopted (the product): the tail of k_ray<true>'s in-forward VJP as a register-resident MFMA chain (GNR_RAY_GEO_MFMA=1);
    the fp32 form:  tools/build_variant.sh rayf -DGNR_RAY_GEO_MFMA=0
  * not adopted: the forward dense layers (q / k / v projections, fc): docs/experiments/r05_ray_mfma.patch on commit 0243025
    GNR_LIB=libgnr_rayf.so python tools/ab_ray_mfma.py --save gpurun_out/ray_f.npz
    python tools/ab_ray_mfma.py --compare gpurun_out/ray_f.npz
Results: profiles/r05_m_ray_mfma_ab.json."""
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
ap.add_argument('--save', default=None)
ap.add_argument('--compare', default=None)
a = ap.parse_args()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
bref, bque = batch_scenes([make_scene(i, 'cfg2') for i in range(a.batch)])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}


def step():
    prep = hp.prepare(bref, 40, 512, 40)
    vol = hp.sample_volume(bref, 40, prepared=prep)
    return vol, hp.render(bref, bque, prepared=prep, debug=True)


for _ in range(3):
    vol, o = step()
torch.cuda.synchronize()
res = []
for rep in range(2):
    _lib.timing_begin()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    t = _lib.timing_end()
    res.append({k: round(v[1] / a.steps, 4) for k, v in t.items() if k.startswith('k_ray')})
    res[-1]['path_total'] = round(sum(v[1] for v in t.values()) / a.steps, 4)
    print(os.environ.get('GNR_LIB', 'libgnr.so'), 'run', rep, res[-1], flush=True)
flat = {'volume': (vol[0] if isinstance(vol, (tuple, list)) else vol).float().cpu().numpy()}
for lvl in (0, 1):
    for k, v in o[lvl].items():
        flat[f'l{lvl}_{k}'] = v.float().cpu().numpy()
if a.save:
    np.savez(a.save, **flat)
cmp = None
if a.compare:
    other = np.load(a.compare)
    cmp = {}
    for k, v in flat.items():
        d = np.abs(v - other[k])
        cmp[k] = {'max_abs': float(d.max()), 'max_abs_over_scale': float(d.max() / max(1e-30, np.abs(other[k]).max())), 'n_differ': int((d > 0).sum()), 'n': int(d.size)}
    print(json.dumps(cmp, indent=1))
print('AB_JSON ' + json.dumps({'lib': os.environ.get('GNR_LIB', 'libgnr.so'), 'runs': res, 'compare': cmp}))
