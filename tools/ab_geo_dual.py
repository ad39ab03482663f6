"""A/B of the per-point half of gnr_geo_dual_bwd: k_geo_dual_bwd_pts_mm (16 points as MFMA columns, fp16-pair chain: the default) against
k_geo_dual_bwd_pts (fp32 FMAs, one lane per point: GNR_OPT_GEO_DUAL_FP32) in one process, on P synthetic points with
inputs of the magnitudes of a training step (statistics O(1), adjoints 1e-6) and of a scaled-up one; per-kernel ms (HIP events on the
launch stream), agreement of d stats / geometry_fc's gradients, and both against a float64 evaluation on the host (a sample of points).
    python tools/ab_geo_dual.py [--points 163840]"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights, _lib
from graspnerf_amd.hotpath import HotPath

ap = argparse.ArgumentParser()
ap.add_argument('--points', type=int, default=8 * 512 * 40)
ap.add_argument('--iters', type=int, default=10)
a = ap.parse_args()
L = _lib.lib()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
can_np = weights.canonical_blob(wnp, 'fine')
canon = torch.from_numpy(can_np).cuda()
rng = np.random.default_rng(3)
P = a.points


def f64_reference(stats, pts, gamma, gbar, gdbar, idx):
    """d stats of the points `idx` in float64 (the algebra of graspnerf_amd/ray_tail.py::tail_backward, geometry_fc part)."""
    W1 = wnp['fine_agg_net.agg_impl.geometry_fc.0.weight'].astype(np.float64); b1 = wnp['fine_agg_net.agg_impl.geometry_fc.0.bias'].astype(np.float64)
    W2 = wnp['fine_agg_net.agg_impl.geometry_fc.2.weight'].astype(np.float64); b2 = wnp['fine_agg_net.agg_impl.geometry_fc.2.bias'].astype(np.float64)
    s, p, gm, gb, gdb = (x[idx].astype(np.float64) for x in (stats, pts, gamma, gbar, gdbar))
    emb = np.concatenate([p, np.sin(p), np.cos(p), np.sin(2 * p), np.cos(2 * p), np.sin(4 * p), np.cos(4 * p)], 1)
    embd = np.concatenate([gm, np.cos(p) * gm, -np.sin(p) * gm, 2 * np.cos(2 * p) * gm, -2 * np.sin(2 * p) * gm, 4 * np.cos(4 * p) * gm, -4 * np.sin(4 * p) * gm], 1)
    x = np.concatenate([s[:, :65], emb], 1)
    h1p = x @ W1.T + b1; h1pd = embd @ W1[:, 65:].T
    e = np.exp(np.minimum(h1p, 0)); h1 = np.where(h1p > 0, h1p, e - 1); e1 = np.where(h1p > 0, 1.0, e); d1 = np.where(h1p > 0, 0.0, e)
    h1d = e1 * h1pd
    gp = h1 @ W2.T + b2; gpd = h1d @ W2.T
    eg = np.exp(np.minimum(gp, 0)); e2 = np.where(gp > 0, 1.0, eg); d2 = np.where(gp > 0, 0.0, eg)
    gpbar = e2 * gb + d2 * gpd * gdb; gpdbar = e2 * gdb
    hb = gpbar @ W2; hdb = gpdbar @ W2
    h1pbar = e1 * hb + d1 * h1pd * hdb
    return (h1pbar @ W1[:, :65])


out = {}
for label, sx, sa in (('training-step magnitudes', 1.0, 1e-6), ('statistics x 3e4, adjoints x 1e3', 3e4, 1e3)):
    stats = (rng.standard_normal((P, 66)) * sx).astype(np.float32); stats[:, 32:64] = np.abs(stats[:, 32:64]); stats[:, 64] = rng.uniform(0, 1, P); stats[:, 65] = 6
    pts = rng.uniform(-0.5, 0.5, (P, 3)).astype(np.float32)
    gamma = (rng.standard_normal((P, 3)) * sa).astype(np.float32)
    gbar = (rng.standard_normal((P, 16)) * sa).astype(np.float32); gdbar = (rng.standard_normal((P, 16)) * sa * 0.1).astype(np.float32)
    dev = [torch.from_numpy(x).cuda() for x in (stats, pts, gamma, gbar, gdbar)]
    res = {}
    for on in (0, 1, 0, 1):
        hp.set_option('geo_dual_fp32', not on)
        for _ in range(2):
            ds, dc = hp.geo_dual_bwd(canon, *dev)
        torch.cuda.synchronize()
        _lib.timing_begin()
        for _ in range(a.iters):
            hp.geo_dual_bwd(canon, *dev)
        torch.cuda.synchronize()
        t = _lib.timing_end()
        res.setdefault(on, {'ms': []})['ms'].append({k.split('@')[0]: round(v[1] / a.iters, 4) for k, v in t.items()})
        res[on]['ds'], res[on]['dc'] = ds.double().cpu().numpy(), dc.double().cpu().numpy()
    hp.set_option('geo_dual_fp32', False)
    idx = rng.choice(P, 4096, replace=False)
    ref = f64_reference(stats, pts, gamma, gbar, gdbar, idx)
    sc = np.abs(ref).max()
    rec = {'ms_fp32_fma': res[0]['ms'], 'ms_matrix_cores': res[1]['ms'],
           'dstats_max_abs_diff_over_scale': float(np.abs(res[0]['ds'] - res[1]['ds']).max() / np.abs(res[0]['ds']).max()),
           'dcanonical_max_abs_diff_over_scale': float(np.abs(res[0]['dc'] - res[1]['dc']).max() / np.abs(res[0]['dc']).max()),
           'vs_float64_on_4096_points': {'fp32_fma_max': float(np.abs(res[0]['ds'][idx, :65] - ref).max() / sc), 'matrix_cores_max': float(np.abs(res[1]['ds'][idx, :65] - ref).max() / sc),
                                         'fp32_fma_rms': float(np.sqrt(((res[0]['ds'][idx, :65] - ref) ** 2).mean()) / sc), 'matrix_cores_rms': float(np.sqrt(((res[1]['ds'][idx, :65] - ref) ** 2).mean()) / sc)},
           'finite': bool(np.isfinite(res[1]['ds']).all() and np.isfinite(res[1]['dc']).all())}
    out[label] = rec
    print(label, json.dumps(rec), flush=True)
print('AB_JSON ' + json.dumps({'points': P, 'cases': out}))
