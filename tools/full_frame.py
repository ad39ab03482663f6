"""Full-frame rendering of one scene (SURVEY §8d "optional full-frame figure rn = 147 456"): every pixel of the
288x512 query view, coarse + fine, in ONE call (the reference loops over 36 chunks of ray_batch_num = 4096 rays,
renderer.py:203-215).  python tools/full_frame.py [--iters N] [--check]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--check', action='store_true', help='compare a strided subset of rays with a separate small call')
a = ap.parse_args()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
ref, que = make_scene(0, 'cfg2', with_query_image=True)
H, W = ref['imgs'].shape[-2:]
ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
que = dict(que, coords=np.stack([xs, ys], -1).reshape(-1, 2).astype(np.float32))
bref, bque = batch_scenes([(ref, que)])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
cfg = {'ray_batch_num': 4096}
rn = H * W
for it in range(a.iters):
    torch.cuda.synchronize(); t = time.perf_counter()
    prep = hp.prepare(bref, 1, rn, 40)
    co, fi = hp.render(bref, bque, cfg, prepared=prep)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f'iter {it}: {dt * 1e3:.2f} ms for {rn} rays x (40+40) samples  ({rn / dt / 1e6:.2f} Mrays/s)', flush=True)
print('workspace MB:', prep[2].numel() / 2 ** 20, ' sdf_gradient_error shape', tuple(co['sdf_gradient_error'].shape))
print('max alloc MB:', torch.cuda.max_memory_allocated() / 2 ** 20)
if a.check:
    sel = np.arange(0, rn, 577)
    sque = {k: (v[:, sel] if k == 'coords' else v) for k, v in bque.items()}
    fd = fi['depth'][:, sel]
    co2, fi2 = hp.render(bref, sque, {}, fine_depth_in=fd)
    torch.cuda.synchronize()
    for k in ('sdf_values', 'alpha_values', 'hit_prob_nr', 'pixel_colors_nr', 'render_depth'):
        d1 = (co[k][:, sel] - co2[k]).abs().max().item()
        d2 = (fi[k][:, sel] - fi2[k]).abs().max().item()
        print(f'  {k}: coarse max|d| {d1:.2e}  fine max|d| {d2:.2e}')
        assert d1 == 0.0 and d2 < 1e-5, k
    assert torch.equal(co['ray_mask'][:, sel], co2['ray_mask'])
    print('subset check OK (coarse bit-identical)')
