"""Time a list of library builds against each other in ONE GPU call (each in its own process: the library is chosen at load
time by GNR_LIB).  python tools/ab_chain.py libgnr.so libgnr_x.so ... [--steps N] [--check]
Prints, per build, the per-kernel ms of a B = 32 forward step (HIP events on the launch stream, include/gnr.h gnr_timing_*),
the standalone chain-kernel time, and with --check the volume / coarse-render error of scene 0 against the reference golden
(ablation builds are expected to fail it)."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, %(root)r)
from graspnerf_amd import weights, _lib
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
wnp = dict(np.load(os.path.join(%(root)r, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
scenes = [make_scene(i, 'cfg2', with_query_image=False) for i in range(%(batch)d)]
bref, bque = batch_scenes(scenes)
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
def step():
    prep = hp.prepare(bref, 40, 512, 40)
    vol = hp.sample_volume(bref, 40, prepared=prep)
    out = hp.render(bref, bque, prepared=prep)
    return vol, out
for _ in range(3): vol, out = step()
torch.cuda.synchronize()
res = {}
if %(check)d:
    g = np.load(os.path.join(%(root)r, 'tests/golden/golden_cfg2.npz'))
    v = vol[0] if isinstance(vol, (tuple, list)) else vol
    res['volume_err'] = float(np.abs(v[0].cpu().numpy().reshape(-1) - g['volume'].reshape(-1)).max())
    c = out[0] if isinstance(out, (tuple, list)) else out
    for k in ('sdf_values', 'alpha_values', 'colors_nr', 'render_depth'):
        if k in c: res[k + '_err'] = float(np.abs(c[k][0].cpu().numpy().reshape(-1) - g['render.' + k].reshape(-1)).max())
t0 = time.perf_counter()
for _ in range(%(steps)d): step()
torch.cuda.synchronize()
res['ms_per_step'] = (time.perf_counter() - t0) * 1e3 / %(steps)d
_lib.timing_begin()
for _ in range(5): step()
torch.cuda.synchronize()
res['kernels'] = {k: round(v[1] / 5, 4) for k, v in _lib.timing_end().items()}
res['chain_alone'] = hp.time_chain_kernel(bref, 40, iters=5)
print('AB_RESULT ' + json.dumps(res))
'''
ap = argparse.ArgumentParser()
ap.add_argument('libs', nargs='+')
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--check', action='store_true')
ap.add_argument('--repeat', type=int, default=1)
a = ap.parse_args()
rows = []
for rep in range(a.repeat):
    for lib in a.libs:
        env = dict(os.environ, GNR_LIB=lib)
        p = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, batch=a.batch, steps=a.steps, check=int(a.check))],
                           env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith('AB_RESULT ')]
        if not line:
            print(f'{lib}: FAILED rc={p.returncode}\n{p.stderr[-1500:]}', flush=True)
            continue
        r = json.loads(line[0][10:])
        k = r['kernels']
        print(f"{lib:28s} step {r['ms_per_step']:.3f} ms | chain.vol {k.get('k_chain.volume', 0):.4f} chain.render {k.get('k_chain.render', 0):.4f} "
              f"ray.render {k.get('k_ray.render', 0):.4f} ray.vol {k.get('k_ray.volume', 0):.4f} repack {k.get('k_repack_feats@gnr_prepare', 0):.4f} twins {k.get('k_chain.volume.fp32_twin', 0) + k.get('k_chain.render.fp32_twin', 0):.4f} | alone {r['chain_alone']:.4f}"
              + (' | ' + ' '.join(f'{kk}={vv:.2e}' for kk, vv in r.items() if kk.endswith('_err')) if a.check else ''), flush=True)
        rows.append(dict(lib=lib, **r))
print('AB_JSON ' + json.dumps(rows))
