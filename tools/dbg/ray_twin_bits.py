"""Do k_ray<true, MM> and its fp32 twin give the same forward bits?  The scene of tests/test_range_guard.py with a 1e5 weight in
mean_decoder.0 (packer bit 2: every chain launch is the fp32 twin's), run twice as is and once with GNR_OPT_FP32_CHAIN."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
w = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
if not os.environ.get('PLAIN'):
    for lvl in ('dist_decoder.', 'fine_dist_decoder.'):
        k = lvl + 'mean_decoder.0.weight'; w[k] = w[k].copy(); w[k][3, 5] = 1e5
ref, que = make_scene(0, 'cfg1')
if os.environ.get('PRE'):      # what tests/test_range_guard.py runs before: feature maps x 30 / x 3000 on HotPaths of their own
    w0 = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
    for f in (1.0, 30.0, 3000.0):
        hp0 = HotPath(weights.pack_state_dict(w0, 'coarse'), weights.pack_state_dict(w0, 'fine'))
        r0 = dict(ref, ray_feats=ref['ray_feats'] * np.float32(f), img_feats=ref['img_feats'] * np.float32(f))
        b0, q0 = batch_scenes([(r0, que)])
        p0 = hp0.prepare(b0, 16, que['coords'].shape[0], 16)
        hp0.sample_volume(b0, 16, prepared=p0); hp0.render(b0, q0, {'depth_sample_num': 16, 'fine_depth_sample_num': 16}, prepared=p0)
        print('pre', f, hp0.range_status(p0))
        if os.environ.get('PRE') == '2':
            prev = hp0.force_fp32_chain(True)
            hp0.sample_volume(b0, 16, prepared=hp0.prepare(b0, 16, que['coords'].shape[0], 16)); hp0.force_fp32_chain(prev)
        del hp0, p0
hp = HotPath(weights.pack_state_dict(w, 'coarse'), weights.pack_state_dict(w, 'fine'))
CFG = {'depth_sample_num': 16, 'fine_depth_sample_num': 16}
def run():
    bref, bque = batch_scenes([(ref, que)])
    prep = hp.prepare(bref, 16, que['coords'].shape[0], 16)
    vol = hp.sample_volume(bref, 16, prepared=prep)
    co, fi = hp.render(bref, bque, CFG, prepared=prep)
    return {**{'c_' + k: v.cpu().numpy() for k, v in co.items()}, **{'f_' + k: v.cpu().numpy() for k, v in fi.items()}, 'vol': vol.cpu().numpy()}, hp.range_status(prep)
a, fa = run(); b, fb = run()
prev = hp.force_fp32_chain(True); c, fc = run(); hp.force_fp32_chain(prev)
print('flags', fa, fb, fc)
for k in a:
    d1 = int((a[k] != b[k]).sum()); d2 = int((a[k] != c[k]).sum())
    if d1 or d2:
        idx = np.argwhere(a[k] != c[k])[:4]
        print(f'{k:24s} run-vs-run {d1:7d}  run-vs-forced {d2:7d} of {a[k].size}  first {idx.tolist()}  {[ (float(a[k][tuple(i)]), float(c[k][tuple(i)])) for i in idx[:2]]}')
print('done')
