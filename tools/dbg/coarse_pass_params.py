"""Parameter gradients of ONE coarse render pass over B scenes (nothing else runs on the HotPath): HIP against float64 / float32 autograd."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
import reference_autograd as ag
from test_bwd_arbiter import _chain, CFG, RN, DN
B = int(os.environ.get('B', 2))
FIRST = int(os.environ.get('FIRST', 0))
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
if os.environ.get('SMOOTH'):
    for k in ('agg_net.prob_embed.0.bias', 'fine_agg_net.prob_embed.0.bias'):
        wnp[k] = wnp[k] + np.float32(8.0)
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
hp.set_bwd_weights(weights.pack_bwd(weights.canonical_blob(wnp, 'coarse')), weights.pack_bwd(weights.canonical_blob(wnp, 'fine')))
scenes = [make_scene(FIRST + i, 'cfg2', with_query_image=False) for i in range(B)]
bref, bque = batch_scenes(scenes)
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
prep = hp.prepare(bref, 40, RN, DN)
stats, colors, geo, ctx = hp.render_chain_train(bq, None, 'coarse', CFG, prep)
g = torch.Generator().manual_seed(1)
ds = torch.randn(B, RN * DN, 65, generator=g).cuda() / (B * RN * DN * 65)
dc = torch.randn(B, RN * DN, 3, generator=g).cuda() / (B * RN * DN * 3)
hw = scenes[0][0]['imgs'].shape[-2:]
drop = 0
for dt in (torch.float64, torch.float32):
    for b in range(B):
        tref = {k: torch.from_numpy(scenes[b][0][k]).cuda().to(dt) for k in ('poses', 'Ks')}
        q1 = {'coords': bq['coords'][b].to(dt), 'pose': bq['pose'][b].to(dt), 'K': bq['K'][b].to(dt), 'depth_range': bq['depth_range'][b].to(dt)}
        pts, _ = ag.ray_points(q1, geo['depth'][b].to(dt))
        differ = ag.project(pts, tref['poses'], tref['Ks'], *hw)[2].sum(0).float() != stats[b, :, 65]
        drop += int(differ.sum()); ds[b, differ] = 0; dc[b, differ] = 0
print('samples dropped on image borders:', drop)
dcan, dray, dimg = hp.render_chain_bwd(ctx, ds, dc)
torch.cuda.synchronize()
got = weights.split_canonical(dcan, 'coarse')
gr, per_scene = {}, {}
for dt in (torch.float64, torch.float32):
    Pm = {k: torch.from_numpy(w).cuda().to(dt).requires_grad_(True) for k, w in wnp.items()}
    for b in range(B):
        tref = {k: (torch.from_numpy(w).cuda().to(dt) if w.dtype.kind == 'f' else torch.from_numpy(w).cuda()) for k, w in scenes[b][0].items()}
        q1 = {'coords': bq['coords'][b].to(dt), 'pose': bq['pose'][b].to(dt), 'K': bq['K'][b].to(dt), 'depth_range': bq['depth_range'][b].to(dt)}
        st, col = _chain(ag, Pm, tref, q1, geo['depth'][b].to(dt), 'dist_decoder.', 'agg_net.', RN, DN, scenes[b][0]['imgs'].shape[-2:])
        if dt == torch.float64:
            e = (stats[b, :, :65].double() - st.detach()).abs()
            print(f'scene {b}: forward statistics vs f64: rms {float(e.pow(2).mean().sqrt()):.3e} max {float(e.max()):.3e}; colours max {float((colors[b].double() - col.detach()).abs().max()):.3e}')
        ((st * ds[b].to(dt)).sum() + (col * dc[b].to(dt)).sum()).backward()
    gr[dt] = {k: p.grad.double() for k, p in Pm.items() if p.grad is not None}
for k in ('agg_net.agg_impl.neuray_fc.2.bias', 'agg_net.agg_impl.neuray_fc.0.weight', 'dist_decoder.var_decoder.4.bias', 'dist_decoder.var_decoder.2.bias', 'dist_decoder.var_decoder.0.bias', 'dist_decoder.var_decoder.4.weight', 'dist_decoder.aw_decoder.4.bias', 'dist_decoder.aw_decoder.2.bias', 'dist_decoder.mean_decoder.4.bias',
          'agg_net.prob_embed.0.bias', 'agg_net.agg_impl.base_fc.2.weight', 'agg_net.agg_impl.vis_fc.0.weight', 'agg_net.agg_impl.rgb_fc.0.weight'):
    a, b_, c = got[k].double(), gr[torch.float64][k], gr[torch.float32][k]
    n = float(b_.norm())
    print(f'  {k:45s} hip {float((a - b_).norm()) / n:.3e}  torch32 {float((c - b_).norm()) / n:.3e}   |g| {n:.3e}')
