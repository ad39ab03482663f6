"""Where does the coarse render pass's chain differ from the float64 statement?  Scene 0 of the benched shape: HIP statistics / colours
against tests/reference_autograd.py in double, worst samples first."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
import reference_autograd as ag
from test_bwd_arbiter import _chain, CFG, RN, DN
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
scene = make_scene(0, 'cfg2', with_query_image=False)
bref, bque = batch_scenes([scene])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
prep = hp.prepare(bref, 40, RN, DN)
stats, colors, geo, ctx = hp.render_chain_train(bq, None, 'coarse', CFG, prep)
torch.cuda.synchronize()
for dt in (torch.float64, torch.float32):
    P = {k: torch.from_numpy(v).cuda().to(dt) for k, v in wnp.items()}
    tref = {k: (torch.from_numpy(v).cuda().to(dt) if v.dtype.kind == 'f' else torch.from_numpy(v).cuda()) for k, v in scene[0].items()}
    q1 = {'coords': bq['coords'][0].to(dt), 'pose': bq['pose'][0].to(dt), 'K': bq['K'][0].to(dt), 'depth_range': bq['depth_range'][0].to(dt)}
    with torch.no_grad():
        st, col = _chain(ag, P, tref, q1, geo['depth'][0].to(dt), 'dist_decoder.', 'agg_net.', RN, DN, scene[0]['imgs'].shape[-2:])
    if dt == torch.float64:
        st64, col64 = st, col
    e = (stats[0, :, :65].to(torch.float64) - st64).abs() if dt == torch.float64 else (st.double() - st64).abs()
    per = e.max(1)[0]
    print(dt, 'stats: rms', float(e.pow(2).mean().sqrt()), 'p99', float(torch.quantile(per, 0.99)), 'max', float(per.max()))
    idx = torch.argsort(per, descending=True)[:12]
    for i in idx.tolist():
        print('   sample', i, 'ray', i // DN, 'k', i % DN, 'err', float(per[i]), 'nvalid', float(stats[0, i, 65]), 'depth', float(geo['depth'][0].reshape(-1)[i]), 'col', e[i].argmax().item())

# ---- backward: d_img_feats / d_ray_feats of the coarse pass, scene 0, against fp64 / fp32 autograd
hp.set_bwd_weights(weights.pack_bwd(weights.canonical_blob(wnp, 'coarse')), weights.pack_bwd(weights.canonical_blob(wnp, 'fine')))
g = torch.Generator().manual_seed(1)
ds = torch.randn(1, RN * DN, 65, generator=g).cuda() / (RN * DN * 65)
dc = torch.randn(1, RN * DN, 3, generator=g).cuda() / (RN * DN * 3)
dcan, dray, dimg = hp.render_chain_bwd(ctx, ds, dc)
torch.cuda.synchronize()
res = {}
for dt in (torch.float64, torch.float32):
    P = {k: torch.from_numpy(v).cuda().to(dt).requires_grad_(True) for k, v in wnp.items()}
    tref = {k: (torch.from_numpy(v).cuda().to(dt) if v.dtype.kind == 'f' else torch.from_numpy(v).cuda()) for k, v in scene[0].items()}
    tref['ray_feats'].requires_grad_(True); tref['img_feats'].requires_grad_(True)
    q1 = {'coords': bq['coords'][0].to(dt), 'pose': bq['pose'][0].to(dt), 'K': bq['K'][0].to(dt), 'depth_range': bq['depth_range'][0].to(dt)}
    st, col = _chain(ag, P, tref, q1, geo['depth'][0].to(dt), 'dist_decoder.', 'agg_net.', RN, DN, scene[0]['imgs'].shape[-2:])
    ((st * ds[0].to(dt)).sum() + (col * dc[0].to(dt)).sum()).backward()
    res[dt] = (tref['img_feats'].grad.double(), tref['ray_feats'].grad.double())
for name, hipg, i in (('d_img_feats', dimg[0].double(), 0), ('d_ray_feats', dray[0].double(), 1)):
    g64, g32 = res[torch.float64][i], res[torch.float32][i]
    eh, e3 = (hipg - g64).abs(), (g32 - g64).abs()
    print(name, 'rms hip', float(eh.pow(2).mean().sqrt()), 'rms torch32', float(e3.pow(2).mean().sqrt()), 'scale', float(g64.pow(2).mean().sqrt()))
    flat = torch.argsort(eh.reshape(-1), descending=True)[:10]
    V, C, fh, fw = g64.shape
    for f in flat.tolist():
        v, c, y, x = np.unravel_index(f, (V, C, fh, fw))
        print(f'   view {v} ch {c} y {y} x {x}: hip {float(hipg[v, c, y, x]):+.6e} f64 {float(g64[v, c, y, x]):+.6e} f32 {float(g32[v, c, y, x]):+.6e}')
    # per-pixel error energy: how concentrated?
    pe = eh.pow(2).sum(1).reshape(V, -1)
    tot = float(pe.sum())
    top = torch.sort(pe.reshape(-1), descending=True)[0]
    print('   share of the squared error in the worst 10 / 100 / 1000 pixels:', float(top[:10].sum()) / tot, float(top[:100].sum()) / tot, float(top[:1000].sum()) / tot)

# ---- which samples tap view 5 around feature pixel (123..124, 30)?
dt = torch.float64
tref = {k: (torch.from_numpy(v).cuda().to(dt) if v.dtype.kind == 'f' else torch.from_numpy(v).cuda()) for k, v in scene[0].items()}
q1 = {'coords': bq['coords'][0].to(dt), 'pose': bq['pose'][0].to(dt), 'K': bq['K'][0].to(dt), 'depth_range': bq['depth_range'][0].to(dt)}
pts, qdir = ag.ray_points(q1, geo['depth'][0].to(dt))
H, W = scene[0]['imgs'].shape[-2:]
uv, z, mask, dirv = ag.project(pts, tref['poses'], tref['Ks'], H, W)
fh, fw = scene[0]['img_feats'].shape[-2:]
px = uv[5, :, 0] * fw / (W - 1) - 0.5
py = uv[5, :, 1] * fh / (H - 1) - 0.5
sel = ((px > 122) & (px < 125.5) & (py > 28.5) & (py < 31.5)).nonzero().reshape(-1)
print('samples near the outlier pixel:', sel.tolist())
for i in sel.tolist():
    print(f'  sample {i} ray {i // DN} k {i % DN} px {float(px[i]):.6f} py {float(py[i]):.6f} z5 {float(z[5, i]):.5f} mask (all views) {mask[:, i].int().tolist()} nvalid(hip) {float(stats[0, i, 65])}'
          f' stats err {float((stats[0, i, :65].double() - st64[i]).abs().max()):.3e} ds {float(ds[0, i].abs().max()):.2e}')
