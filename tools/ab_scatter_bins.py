"""A/B of the feature-map gradient scatter of the first view loop's backward (csrc/gnr_bwd_scatter.inc): binned (rows parked in HBM,
summed per pixel by k_scatter_gather) against direct float atomics out of k_view1_bwd_pw, on the benched training shapes:
the 8-scene 40^3 volume launch and an 8-scene 512 x 40 render pass.  Prints per-kernel times (the library's own HIP-event
brackets), the whole stage, and the distance between the two modes' d_ray_feats / d_img_feats (both are float sums in an
arrival order: they differ from each other like two runs of either do).
python tools/ab_scatter_bins.py [--scenes 8] [--json out.json]"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights, _lib
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene

ap = argparse.ArgumentParser()
ap.add_argument('--scenes', type=int, default=8)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--json', default=None)
a = ap.parse_args()
L = _lib.lib()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
can = weights.canonical_blob(wnp, 'coarse')
hp.set_bwd_weights(weights.pack_bwd(can), weights.pack_bwd(weights.canonical_blob(wnp, 'fine')))
can_dev = torch.from_numpy(can).cuda()
scenes = [make_scene(i, 'cfg2', with_query_image=False) for i in range(a.scenes)]
bref, bque = batch_scenes(scenes)
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
B, rn = bque['coords'].shape[:2]
torch.manual_seed(0)
dvol = torch.randn(a.scenes, 1, 40, 40, 40, device='cuda')
cfg = {'depth_sample_num': 40, 'fine_depth_sample_num': 40}


def kernel_table(f, n):
    """per-label (count, total ms) over n calls of f, from the library's event brackets"""
    f(); torch.cuda.synchronize()
    L.gnr_timing_begin()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 16)
    L.gnr_timing_end(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        lab, cnt, ms = line.rsplit(' ', 2)
        out[lab] = float(ms) / n
    return out


def wall(f, n):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


res = {}
prep = hp.prepare(bref, 40, rn, 40)
# ---- volume
hp.sample_volume_train(bref, 40, prepared=prep)
outs = {}
for mode in (0, 1):
    hp.set_option('direct_scatter', not mode)
    f = lambda: hp.sample_volume_bwd(dvol, can_dev, stages=1)       # stage 1 alone: dS1 stays what the full backward below left
    hp.sample_volume_bwd(dvol, can_dev, stages=0x1f)
    outs[mode] = [x.clone() for x in hp.sample_volume_bwd(dvol, can_dev, stages=1)]
    tab = kernel_table(f, a.iters)
    keep = {k: round(v, 4) for k, v in tab.items() if 'view1' in k or 'scatter' in k or 'unpack' in k}
    res[f'volume_bins{mode}'] = {'kernels_ms': keep, 'sum_ms': round(sum(keep.values()), 4), 'stage_wall_ms': round(wall(f, a.iters), 4)}
    print('volume', 'binned' if mode else 'direct', res[f'volume_bins{mode}'])
for name, i in (('d_ray_feats', 1), ('d_img_feats', 2)):
    d = (outs[0][i] - outs[1][i]).abs().max().item()
    s = outs[0][i].abs().max().item()
    res[f'volume_{name}_max_abs_diff_over_max'] = d / s
    print(f'volume {name}: max|direct - binned| / max|direct| = {d / s:.3e}   (max {s:.3e})')
d = (outs[0][0] - outs[1][0]).abs().max().item()
print('volume d_canonical (stage-1 ranges) max diff', d)
res['volume_d_canonical_max_abs_diff'] = d

# ---- render pass (coarse level, coarse depths)
stats, colors, geo, ctx = hp.render_chain_train(bque, None, 'coarse', cfg, prep)
dstats = torch.randn(B, rn * 40, 65, device='cuda') * 1e-3
dcolors = torch.randn(B, rn * 40, 3, device='cuda') * 1e-3
outs = {}
for mode in (0, 1):
    hp.set_option('direct_scatter', not mode)
    f = lambda: hp.render_chain_bwd(ctx, dstats, dcolors)
    outs[mode] = [x.clone() for x in f()]
    tab = kernel_table(f, a.iters)
    keep = {k: round(v, 4) for k, v in tab.items() if 'view1' in k or 'scatter' in k or 'unpack' in k}
    res[f'render_bins{mode}'] = {'kernels_ms': keep, 'sum_ms': round(sum(keep.values()), 4), 'all_kernels_ms': round(sum(tab.values()), 4)}
    print('render', 'binned' if mode else 'direct', res[f'render_bins{mode}'])
for name, i in (('d_ray_feats', 1), ('d_img_feats', 2)):
    d = (outs[0][i] - outs[1][i]).abs().max().item()
    s = outs[0][i].abs().max().item()
    res[f'render_{name}_max_abs_diff_over_max'] = d / s
    print(f'render {name}: max|direct - binned| / max|direct| = {d / s:.3e}   (max {s:.3e})')
res['render_d_canonical_equal'] = bool(torch.equal(outs[0][0], outs[1][0]))
print('render d_canonical equal:', res['render_d_canonical_equal'])
hp.set_option('direct_scatter', False)
if a.json:
    os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
    json.dump(res, open(a.json, 'w'), indent=1)
