"""Per-(view, sample) feature gradients of the coarse render pass: the rows k_view1_bwd_pw parks for the binned scatter (read out of the
training workspace) against autograd's d loss / d (gathered ray features) in float64 and float32.  Scene 0 of the benched shape."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
import reference_autograd as ag
from test_bwd_arbiter import CFG, RN, DN
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
if os.environ.get('SMOOTH'):
    for k in ('agg_net.prob_embed.0.bias', 'fine_agg_net.prob_embed.0.bias'):
        wnp[k] = wnp[k] + np.float32(8.0)
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
hp.set_bwd_weights(weights.pack_bwd(weights.canonical_blob(wnp, 'coarse')), weights.pack_bwd(weights.canonical_blob(wnp, 'fine')))
SC = int(os.environ.get('SCENE', 0))
scene = make_scene(SC, 'cfg2', with_query_image=False)
bref, bque = batch_scenes([scene])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
prep = hp.prepare(bref, 40, RN, DN)
stats, colors, geo, ctx = hp.render_chain_train(bq, None, 'coarse', CFG, prep)
g = torch.Generator().manual_seed(1)
ds = torch.randn(1, RN * DN, 65, generator=g).cuda() / (RN * DN * 65)
dc = torch.randn(1, RN * DN, 3, generator=g).cuda() / (RN * DN * 3)
dcan, dray, dimg = hp.render_chain_bwd(ctx, ds, dc)
torch.cuda.synchronize()
scene_s, keep, ws, tws = ctx[:4]
V, P = 6, RN * DN
T = (P + 15) // 16
N = T * V * 16
al = lambda x: (x + 255) & ~255
bins = V * scene_s.fh * scene_s.fw
scatter_total = al(N * 256) + al(N * 16) + al(N * 4) + al(bins * 4) + al(bins * 4) + al(N * 4) + al(V * 4)
need = hp.L.gnr_render_chain_train_workspace_bytes(C.byref(scene_s), RN, DN)
off = need - scatter_total
rows = tws[off:off + N * 256].view(torch.float32).reshape(T, V, 16, 64)
keys = tws[off + al(N * 256) + al(N * 16):off + al(N * 256) + al(N * 16) + N * 4].view(torch.int32).reshape(T, V, 16)
pos = torch.tensor([16 * (c // 8) + c % 8 for c in range(32)], device='cuda')
hip_rows = rows[..., pos].permute(1, 0, 2, 3).reshape(V, T * 16, 32)[:, :P].double()        # [V, P, 32] ray channels
live = (keys.permute(1, 0, 2).reshape(V, T * 16)[:, :P] >= 0)
res = {}
for dt in (torch.float64, torch.float32):
    Pm = {k: torch.from_numpy(v).cuda().to(dt).requires_grad_(True) for k, v in wnp.items()}
    tref = {k: (torch.from_numpy(v).cuda().to(dt) if v.dtype.kind == 'f' else torch.from_numpy(v).cuda()) for k, v in scene[0].items()}
    q1 = {'coords': bq['coords'][0].to(dt), 'pose': bq['pose'][0].to(dt), 'K': bq['K'][0].to(dt), 'depth_range': bq['depth_range'][0].to(dt)}
    depth = geo['depth'][0].to(dt)
    pts, qdir = ag.ray_points(q1, depth)
    uv, z, mask, dirv = ag.project(pts, tref['poses'], tref['Ks'], *scene[0]['imgs'].shape[-2:])
    f_ray, rgb, f_img = ag._gather(tref, uv, mask)
    f_ray = f_ray.detach().requires_grad_(True)
    near, far = -1 / q1['depth_range'][0], -1 / q1['depth_range'][1]
    di = (-1 / depth - near) / (far - near)
    half = torch.cat([di[:, 1:] - di[:, :-1], torch.full_like(di[:, :1], 1e6)], -1) / 2
    ext = torch.cat([half[:, :1], half], -1)
    hit, vis = ag.decode_hit_vis(Pm, 'dist_decoder.', f_ray, z, mask, tref['depth_range'], ext[:, :-1].reshape(-1), ext[:, 1:].reshape(-1))
    taps = {}
    qd = qdir[:, None].expand(RN, DN, 3).reshape(-1, 3)
    _, _, col = ag.aggregate(Pm, 'agg_net.', f_ray, rgb, f_img, hit, vis, mask, dirv, qd, pts, RN, DN, False, True, taps)
    v2 = taps['v2']
    wbar = (v2 / (v2.sum(0, keepdim=True) + 1e-8)).mean(0)
    st = torch.cat([taps['mean'], taps['var'], wbar], -1)
    ((st * ds[0].to(dt)).sum() + (col.reshape(-1, 3) * dc[0].to(dt)).sum()).backward()
    res[dt] = f_ray.grad.double().reshape(V, P, 32)
    if dt == torch.float64:
        mask64, hit64, vis64 = mask, hit.detach(), vis.detach()
g64, g32 = res[torch.float64], res[torch.float32]
print('live rows (hip)', int(live.sum()), 'mask (f64)', int(mask64.sum()), 'disagree', int((live != mask64.bool()).sum()))
eh = (hip_rows - g64).abs().amax(-1)
e3 = (g32 - g64).abs().amax(-1)
sc = g64.abs().amax(-1)
print('per-row max err: hip rms', float(eh.pow(2).mean().sqrt()), 'torch32 rms', float(e3.pow(2).mean().sqrt()), ' row scale rms', float(sc.pow(2).mean().sqrt()))
print('SUMMARY scene', SC, 'sum sq err hip', float(eh.pow(2).sum()), 'torch32', float(e3.pow(2).sum()), ' rows only hip is off', int(((eh > 1e-9) & (eh > 10 * e3)).sum()), ' rows only torch32 is off', int(((e3 > 1e-9) & (e3 > 10 * eh)).sum()), ' rows both off', int(((e3 > 1e-9) & (eh > 1e-9) & (e3 <= 10 * eh) & (eh <= 10 * e3)).sum()))
idx = torch.argsort(eh.reshape(-1), descending=True)[:15]
for f in idx.tolist():
    v, i = divmod(f, P)
    print(f'  view {v} sample {i} ray {i // DN} k {i % DN}: err hip {float(eh[v, i]):.3e} torch32 {float(e3[v, i]):.3e} scale {float(sc[v, i]):.3e} live {bool(live[v, i])} mask {int(mask64[v, i])} '
          f'nvalid {float(stats[0, i, 65])} hit {float(hit64.reshape(V, P)[v, i]):.3e} vis {float(vis64.reshape(V, P)[v, i]):.3e} depth {float(geo["depth"][0].reshape(-1)[i]):.4f}')

# ---- the worst row in detail
f = int(torch.argsort(eh.reshape(-1), descending=True)[0])
v, i = divmod(f, P)
print('worst row: view', v, 'sample', i)
print(' hip ', [f'{x:+.3e}' for x in hip_rows[v, i, :8].tolist()])
print(' f64 ', [f'{x:+.3e}' for x in g64[v, i, :8].tolist()])
print(' f32 ', [f'{x:+.3e}' for x in g32[v, i, :8].tolist()])
dt = torch.float64
Pm = {k: torch.from_numpy(w).cuda().to(dt) for k, w in wnp.items()}
tref = {k: (torch.from_numpy(w).cuda().to(dt) if w.dtype.kind == 'f' else torch.from_numpy(w).cuda()) for k, w in scene[0].items()}
with torch.no_grad():
    fr, _, _ = ag._gather(tref, uv, mask)
    pe_in = torch.cat([fr, ((hit64.reshape(V, P) - 0.5) * 2)[..., None], ((vis64.reshape(V, P) - 0.5) * 2)[..., None]], -1)
    pre = ag._lin(pe_in, Pm, 'agg_net.prob_embed.0')[v, i]
    print(' prob_embed.0 pre-activations closest to the ReLU kink (f64):', sorted(pre.abs().tolist())[:4])
    mean = ag._mlp3(fr, Pm, 'dist_decoder.mean_decoder', torch.nn.functional.softplus)[v, i]
    var = ag._mlp3(fr, Pm, 'dist_decoder.var_decoder', torch.nn.functional.softplus)[v, i] + 0.05
    print(' mean', mean.tolist(), 'var', var.tolist(), 'z', float(z[v, i]), 'depth_range', tref['depth_range'][v].tolist())
    near_r, far_r = -1 / tref['depth_range'][v, 0], -1 / tref['depth_range'][v, 1]
    dhat = (-1 / torch.clamp(z[v, i], min=1e-5) - near_r) / (far_r - near_r)
    lo, hi = ext[:, :-1].reshape(-1)[i], ext[:, 1:].reshape(-1)[i]
    print(' dhat', float(dhat), 'lo', float(lo), 'hi', float(hi), ' tanh args near', ((dhat - lo - mean) * var).tolist(), 'far', ((dhat + hi - mean) * var).tolist())

# ---- parameter gradients of this one-scene pass
got = weights.split_canonical(dcan, 'coarse')
print('parameter gradients (one scene, coarse pass): relative error of the HIP path / of torch fp32 against float64')
gr = {}
for dt in (torch.float64, torch.float32):
    Pm = {k: torch.from_numpy(w).cuda().to(dt).requires_grad_(True) for k, w in wnp.items()}
    tref = {k: (torch.from_numpy(w).cuda().to(dt) if w.dtype.kind == 'f' else torch.from_numpy(w).cuda()) for k, w in scene[0].items()}
    q1 = {'coords': bq['coords'][0].to(dt), 'pose': bq['pose'][0].to(dt), 'K': bq['K'][0].to(dt), 'depth_range': bq['depth_range'][0].to(dt)}
    from test_bwd_arbiter import _chain
    st, col = _chain(ag, Pm, tref, q1, geo['depth'][0].to(dt), 'dist_decoder.', 'agg_net.', RN, DN, scene[0]['imgs'].shape[-2:])
    ((st * ds[0].to(dt)).sum() + (col * dc[0].to(dt)).sum()).backward()
    gr[dt] = {k: p.grad.double() for k, p in Pm.items() if p.grad is not None}
for k in ('agg_net.agg_impl.neuray_fc.2.bias', 'agg_net.agg_impl.neuray_fc.0.bias', 'agg_net.agg_impl.neuray_fc.0.weight', 'dist_decoder.var_decoder.4.bias', 'dist_decoder.mean_decoder.4.bias',
          'agg_net.prob_embed.0.bias', 'agg_net.agg_impl.base_fc.2.weight', 'agg_net.agg_impl.vis_fc.0.weight', 'agg_net.agg_impl.rgb_fc.0.weight'):
    a, b, c = got[k].double(), gr[torch.float64][k], gr[torch.float32][k]
    n = float(b.norm())
    print(f'  {k:45s} hip {float((a - b).norm()) / n:.3e}  torch32 {float((c - b).norm()) / n:.3e}   |g| {n:.3e}  hip {a.reshape(-1)[:3].tolist()} f64 {b.reshape(-1)[:3].tolist()}')
