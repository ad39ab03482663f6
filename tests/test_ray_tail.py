"""Per-ray tail of the render path in training: the dual-number backward (tensor algebra: tests/reference_autograd.py; device:
graspnerf_amd/ray_tail.py over k_geo_dual_fwd / k_ray_dual_bwd / k_geo_dual_bwd) against autograd's double backward of
reference_autograd.sdf_tail (second order through the in-forward SDF gradient, ibrnet.py:497-504)."""
import numpy as np
import pytest
import torch

from graspnerf_amd import ray_tail as prt, weights
import reference_autograd as ag
rt = ag            # attn_core / tail_backward / tail_weights in tensor algebra live next to the statement
from graspnerf_amd.synth import make_scene

AGG = 'agg_net.'


def _tail_params(weights_np, dtype, device='cpu', agg=AGG):
    return {agg + 'agg_impl.' + k: torch.from_numpy(weights_np[agg + 'agg_impl.' + k]).to(device=device, dtype=dtype).requires_grad_(True)
            for k in prt.TAIL_KEYS}


def _case(seed, rn, dn, dtype, device='cpu'):
    g = torch.Generator().manual_seed(seed)
    N = rn * dn
    stats = torch.randn(N, 65, generator=g).to(device=device, dtype=dtype)
    nvalid = torch.randint(0, 4, (N,), generator=g).to(device=device, dtype=dtype)
    pts = (0.3 * torch.randn(N, 3, generator=g)).to(device=device, dtype=dtype)
    a = torch.randn(rn, dn, generator=g).to(device=device, dtype=dtype)
    gamma = torch.randn(rn, dn, 3, generator=g).to(device=device, dtype=dtype)
    return stats, nvalid, pts, a, gamma


def _autograd_reference(P, stats, nvalid, pts, rn, dn, a, gamma, agg=AGG):
    stats = stats.clone().requires_grad_(True)
    sdf, grad = ag.sdf_tail(P, agg, stats[:, :32], stats[:, 32:64], stats[:, 64:65], nvalid, pts, rn, dn, True)
    names = list(P)
    gs = torch.autograd.grad((a * sdf).sum() + (gamma * grad).sum(), [stats] + [P[k] for k in names])
    return sdf.detach(), grad.detach(), gs[0], dict(zip(names, gs[1:]))


def _rel(x, y):
    return float((x - y).abs().max() / (y.abs().max() + 1e-30))


@pytest.mark.parametrize('rn,dn', [(5, 12), (2, 40)])
def test_dual_backward_equals_double_backward_fp64(rn, dn, weights_np):
    """Exact algebra: every gradient of Phi = <a, sdf> + <gamma, grad> to 1e-12 in float64."""
    P = _tail_params(weights_np, torch.float64)
    stats, nvalid, pts, a, gamma = _case(rn, rn, dn, torch.float64)
    sdf, _, dstats_ref, G_ref = _autograd_reference(P, stats, nvalid, pts, rn, dn, a, gamma)
    assert 0 < float((sdf.abs() < 1).double().mean()) < 1 or dn < 40      # both clipped and live samples in the case
    with torch.no_grad():
        dstats, G = rt.tail_backward(P, AGG, stats, nvalid, pts, rn, dn, a, gamma)
    assert _rel(dstats, dstats_ref) < 1e-12
    for k, g in G_ref.items():
        assert _rel(G[k], g) < 1e-12, k


def _hip_core(hp, level):
    """k_ray_dual_bwd through the C ABI with attn_core's signature."""
    def core(W, g, gd, a, nvalid):
        gbar, gdbar, dt = hp.ray_tail_dual_bwd(level, g, gd, a, nvalid)
        m = lambda k: dt[256 * k:256 * (k + 1)].reshape(16, 16)
        return gbar, gdbar, {'wq': m(0), 'wk': m(1), 'wv': m(2), 'wfc': m(3), 'lnw': dt[1024:1040], 'lnb': dt[1040:1056],
                             'weff': dt[1056:1072], 'beff': dt[1072]}
    return core


@pytest.mark.gpu
@pytest.mark.parametrize('R,dn,level', [(33, 40, 'coarse'), (7, 16, 'fine'), (300, 40, 'fine'), (3, 64, 'coarse'), (11, 5, 'coarse'), (64, 64, 'fine'), (5, 80, 'fine'), (70, 128, 'fine')])
def test_dual_core_kernel_matches_tensor_algebra(R, dn, level, weights_np):
    """k_ray_dual_bwd through the C ABI against ray_tail.attn_core on the same inputs (masked rows, clipped samples,
    rays straddling wavefronts, more rays than one workgroup holds)."""
    from graspnerf_amd.hotpath import HotPath
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    agg = 'agg_net.' if level == 'coarse' else 'fine_agg_net.'
    P = {k: torch.from_numpy(v).cuda() for k, v in weights_np.items()}
    W = rt.tail_weights(P, agg)
    g = torch.Generator().manual_seed(R * dn)
    gg = torch.randn(R, dn, 16, generator=g).cuda()
    gd = torch.randn(R, dn, 16, generator=g).cuda()
    a = torch.randn(R, dn, generator=g).cuda()
    nvalid = torch.randint(0, 4, (R, dn), generator=g).float().cuda()
    tb, tdb, G = rt.attn_core(W, gg, gd, a, nvalid)
    hb, hdb, H = _hip_core(hp, level)(W, gg, gd, a, nvalid)
    torch.cuda.synchronize()
    assert _rel(hb, tb) < 5e-4 and _rel(hdb, tdb) < 5e-4
    # the parameter gradients are sums over all R * dn samples.  The kernel carries the cancellation-prone scalar sums
    # (beff = sum of d sdf; lnw, lnb, weff follow from sum of d sdf * LayerNorm output) in double from the lane on, and the four
    # matrices through fp32 MFMA partials that a fixed-order reduction adds in double: every entry within 2e-5 of the largest entry
    # of the float64 evaluation (measured: <= 7e-7, profiles/r04_*_parity_errors.json) -- also where 6 700 O(1) terms cancel to
    # 9e-3 (case 70-128-fine, beff: 4e-8 of the result; the atomics of round 3 left it at 1.7e-3)
    W64 = rt.tail_weights({k: v.double() for k, v in P.items()}, agg)
    _, _, G64 = rt.attn_core(W64, gg.double(), gd.double(), a.double(), nvalid.double())
    from conftest import PARITY_LOG
    for k in G:
        scale = float(G64[k].abs().max())
        e_hip, e_t32 = float((H[k].double() - G64[k]).abs().max()), float((G[k].double() - G64[k]).abs().max())
        PARITY_LOG.append({'what': f'k_ray_dual_bwd d {k} vs float64 [{R}x{dn} {level}]', 'max_abs': e_hip, 'tol': 2e-5 * scale,
                           'max_over_tol': e_hip / (2e-5 * scale), 'torch_fp32_abs': e_t32})
        assert e_hip <= 2e-5 * scale, (k, e_hip, e_t32, scale)


@pytest.mark.gpu
@pytest.mark.parametrize('V,rn,dn', [(4, 5, 7), (6, 33, 40), (3, 64, 64), (3, 20, 128)])
def test_tail_forward_and_backward_on_a_render_pass(V, rn, dn, weights_np):
    """After the HIP chain of a pass: k_ray<true>'s sdf / gradient against sdf_tail on the chain's statistics, then the
    whole tail backward (HIP core) against autograd's double backward for random upstream (a, gamma)."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    from graspnerf_amd.synth import CONFIGS
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    ref, que = make_scene(70 + V, dict(CONFIGS['cfg1'], V=V, rn=rn))
    bref, bque = batch_scenes([(ref, que)])
    prep = hp.prepare(bref, 1, rn, dn)
    rng = np.random.default_rng(rn)
    depth = torch.sort(torch.from_numpy(rng.uniform(0.25, 0.75, (rn, dn)).astype(np.float32)), -1)[0].cuda()
    # (the pass runs on caller-given depths: up to 128 of them, fine_depth_use_all; the coarse sampler's own counts stay <= 64)
    cfg = {'depth_sample_num': min(dn, 64), 'fine_depth_sample_num': min(dn, 64), 'ray_mask_view_num': 2, 'ray_mask_point_num': 8}
    bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
    stats, colors, geo, ctx = hp.render_chain_train(bq, depth[None], 'fine', cfg, prep)
    fw = hp.render_tail_train(ctx, bq, depth[None], colors)
    sdf, grad = fw['sdf_values'], fw['sdf_gradient']
    agg = 'fine_agg_net.'
    P = _tail_params(weights_np, torch.float32, 'cuda', agg)
    q1 = {'coords': bq['coords'][0], 'pose': bq['pose'][0], 'K': bq['K'][0]}
    pts, qdir = ag.ray_points(q1, depth)
    # the ray geometry the kernels export (S2, render_ops.py:4-39) against the statement's
    assert _rel(geo['pts'], pts) < 1e-6 and _rel(geo['qdir'], qdir) < 1e-6 and torch.equal(geo['depth'][0], depth)
    a = torch.from_numpy(rng.standard_normal((rn, dn)).astype(np.float32)).cuda()
    gamma = torch.from_numpy(rng.standard_normal((rn, dn, 3)).astype(np.float32)).cuda()
    st = stats[0]
    sdf_ref, grad_ref, dstats_ref, G_ref = _autograd_reference(P, st[:, :65], st[:, 65], pts, rn, dn, a, gamma, agg)
    torch.cuda.synchronize()
    assert float((sdf[0] - sdf_ref).abs().max()) < 1e-3 * max(1.0, float(sdf_ref.abs().max()))
    assert _rel(grad[0], grad_ref) < 1e-3
    hp.can_dev = {'fine': torch.from_numpy(weights.canonical_blob(weights_np, 'fine')).cuda()}
    with torch.no_grad():                                                  # the product's device orchestration
        dstats, G = prt.tail_backward(hp, 'fine', P, agg, st, geo['pts'], rn, dn, a, gamma)
    torch.cuda.synchronize()
    assert _rel(dstats[:, :65], dstats_ref) < 1e-3 and float(dstats[:, 65].abs().max()) == 0
    for k, g in G_ref.items():
        assert _rel(G[k], g) < 2e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize('R,dn,level', [(70, 40, 'coarse'), (5, 7, 'fine'), (130, 64, 'coarse'), (33, 80, 'fine'), (70, 128, 'fine')])
def test_composite_backward_kernel(R, dn, level, weights_np):
    """k_composite_bwd against autograd over reference_autograd.composite (NeuS alpha, cumprod compositing, eikonal term) for a
    random upstream on every output, including the gradient of deviation_network.variance."""
    from graspnerf_amd.hotpath import HotPath
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    agg = 'agg_net.' if level == 'coarse' else 'fine_agg_net.'
    key = agg + 'deviation_network.variance'
    P = {key: torch.from_numpy(np.asarray(weights_np[key])).cuda().requires_grad_(True)}
    g = torch.Generator().manual_seed(R + dn)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda()
    sdf = (0.05 * rnd(R, dn)).requires_grad_(True)
    grad = rnd(R, dn, 3).requires_grad_(True)
    col = torch.rand(R, dn, 3, generator=g).cuda().requires_grad_(True)
    qdir = torch.nn.functional.normalize(rnd(R, 3), dim=1)
    depth = torch.sort(0.25 + 0.5 * torch.rand(R, dn, generator=g), -1)[0].cuda()
    nvalid = torch.full((R, dn), 3.0).cuda()
    cfg = {'ray_mask_view_num': 2, 'ray_mask_point_num': 8}
    out = ag.composite(P, agg, sdf, grad, col, nvalid, qdir, depth, {}, (96, 128), cfg)
    up = {'pixel_colors_nr': rnd(1, R, 3), 'render_depth': rnd(1, R), 'sdf_gradient_error': rnd(1, 1), 'alpha_values': rnd(1, R, dn),
          'hit_prob_nr': rnd(1, R, dn)}
    ref = torch.autograd.grad(sum((out[k] * v).sum() for k, v in up.items()), [sdf, grad, col, P[key]])
    wg = (up['sdf_gradient_error'][0, 0] / (R * dn)).expand(R).contiguous()
    a, gamma, dcol, dvar = hp.composite_bwd(level, sdf.detach(), grad.detach(), col.detach(), depth, qdir, up['pixel_colors_nr'][0],
                                            up['render_depth'][0], wg, up['alpha_values'][0], up['hit_prob_nr'][0])
    torch.cuda.synchronize()
    assert 0.05 < float(((out['alpha_values'] > 0) & (out['alpha_values'] < 1)).float().mean())      # unsaturated samples exist
    for got, want, name in ((a, ref[0], 'd sdf'), (gamma, ref[1], 'd grad'), (dcol, ref[2], 'd colours'), (dvar.reshape(()), ref[3], 'd variance')):
        assert _rel(got, want) < 1e-3, name


@pytest.mark.gpu
@pytest.mark.parametrize('P_,level', [(700, 'coarse'), (129, 'fine'), (5, 'coarse')])
def test_geometry_dual_kernels(P_, level, weights_np):
    """k_geo_dual_fwd / k_geo_dual_bwd (the two ELU layers of geometry_fc on dual numbers) inside tail_backward against the
    tensor-algebra statement: same d stats and the same gradients for all 14 tail parameters."""
    from graspnerf_amd.hotpath import HotPath
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    agg = 'agg_net.' if level == 'coarse' else 'fine_agg_net.'
    canon = torch.from_numpy(weights.canonical_blob(weights_np, level)).cuda()
    P = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in weights_np.items()}
    rn, dn = 1, P_
    if P_ > 64:
        dn = 7 if P_ % 7 == 0 else 43 if P_ % 43 == 0 else 1
        rn = P_ // dn
    if dn < 3:
        rn, dn = 1, P_
    stats, nvalid, pts, a, gamma = _case(P_, rn, dn, torch.float32, 'cuda')
    hp.can_dev = {level: canon}
    with torch.no_grad():
        want_ds, want = rt.tail_backward(P, agg, stats, nvalid, pts, rn, dn, a, gamma, _hip_core(hp, level))
        got_ds, got = prt.tail_backward(hp, level, P, agg, torch.cat([stats, nvalid[:, None]], 1), pts, rn, dn, a, gamma)
    torch.cuda.synchronize()
    assert _rel(got_ds[:, :65], want_ds) < 1e-3
    for k in want:
        assert _rel(got[k], want[k]) < 1e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize('scale_x,scale_adj', [(1.0, 1e-6), (3e4, 1e3), (1.0, 1e-20)])
def test_geo_dual_bwd_on_the_matrix_cores_agrees_with_the_fp32_kernel(scale_x, scale_adj, weights_np):
    """gnr_geo_dual_bwd's per-point half as a chained fp16-pair MFMA (k_geo_dual_bwd_pts_mm, the default) against the fp32 FMA kernel
    (GNR_OPT_GEO_DUAL_FP32): d stats and geometry_fc's gradients equal to 2e-5 of their scale for training-step magnitudes
    and for adjoints of 1e-20 (every operand block is normalised before it is split); with statistics near the fp16 limit (pre-activations
    of 1e5: a hidden unit whose pre-activation rounds to the other side of the ELU's kink changes its derivative from 1 to ~0, in either
    kernel -- tools/ab_geo_dual.py: both are then 1.5e-3 off a float64 evaluation at the worst point, 1.4e-5 rms) they agree point by
    point except at those crossings, stay finite and do not overflow the fp16 halves; a ragged point count."""
    from graspnerf_amd.hotpath import HotPath
    from graspnerf_amd import _lib
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    canon = torch.from_numpy(weights.canonical_blob(weights_np, 'fine')).cuda()
    rng = np.random.default_rng(11)
    Pn = 4 * 512 * 40 + 13
    stats = (rng.standard_normal((Pn, 66)) * scale_x).astype(np.float32)
    stats[:, 32:64] = np.abs(stats[:, 32:64]); stats[:, 64] = rng.uniform(0, 1, Pn); stats[:, 65] = 6
    pts = rng.uniform(-0.5, 0.5, (Pn, 3)).astype(np.float32)
    gamma = (rng.standard_normal((Pn, 3)) * scale_adj).astype(np.float32)
    gbar = (rng.standard_normal((Pn, 16)) * scale_adj).astype(np.float32)
    gdbar = (rng.standard_normal((Pn, 16)) * scale_adj * 0.1).astype(np.float32)
    L = _lib.lib()
    prev = hp.set_option('geo_dual_fp32', False)
    try:
        got = [x.double() for x in hp.geo_dual_bwd(canon, stats, pts, gamma, gbar, gdbar)]
        hp.set_option('geo_dual_fp32', True)
        want = [x.double() for x in hp.geo_dual_bwd(canon, stats, pts, gamma, gbar, gdbar)]
    finally:
        hp.set_option('geo_dual_fp32', prev)
    torch.cuda.synchronize()
    for g, w, name in zip(got, want, ('d stats', 'd geometry_fc')):
        assert bool(torch.isfinite(g).all()), name
        assert float(w.abs().max()) > 0, name
        if scale_x == 1.0:
            assert float((g - w).abs().max()) <= 2e-5 * float(w.abs().max()), name
        elif name == 'd stats':                                     # per point: all but the kink crossings agree (tools/dbg/geo_dual_regime.py: 0.1 % of the points)
            row = (g - w).abs().max(1)[0] / w.abs().max(1)[0].clamp(min=1e-30)
            assert float(row.median()) <= 2e-5 and float((row > 1e-3).double().mean()) <= 5e-3, (float(row.median()), float((row > 1e-3).double().mean()))
        else:                                                       # a sum over all points, the crossings included
            assert float((g - w).pow(2).mean().sqrt()) <= 0.1 * float(w.pow(2).mean().sqrt()), name
    assert float(got[0][:, 65].abs().max()) == 0.0                     # n_valid has no gradient


@pytest.mark.gpu
def test_geo_dual_bwd_weight_without_an_fp16_pair_falls_back_to_the_fp32_kernel(weights_np):
    """A geometry_fc weight of 1e5 has no fp16 pair: k_pack_geo_dual stores inf for it, the matrix-core kernel's outputs turn non-finite, its
    range word makes the fp32 kernel launched behind it recompute the call: the outputs are that kernel's, bit for bit."""
    from graspnerf_amd.hotpath import HotPath
    from graspnerf_amd import _lib
    w = dict(weights_np)
    k = 'fine_agg_net.agg_impl.geometry_fc.2.weight'
    w[k] = w[k].copy(); w[k][5, 11] = 1e5
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    canon = torch.from_numpy(weights.canonical_blob(w, 'fine')).cuda()
    rng = np.random.default_rng(2)
    Pn = 5000
    stats = rng.standard_normal((Pn, 66)).astype(np.float32); stats[:, 32:64] = np.abs(stats[:, 32:64]); stats[:, 65] = 6
    pts = rng.uniform(-0.5, 0.5, (Pn, 3)).astype(np.float32)
    gamma, gbar, gdbar = (rng.standard_normal(sh).astype(np.float32) * 1e-3 for sh in ((Pn, 3), (Pn, 16), (Pn, 16)))
    L = _lib.lib()
    prev = hp.set_option('geo_dual_fp32', False)
    try:
        got = [x.clone() for x in hp.geo_dual_bwd(canon, stats, pts, gamma, gbar, gdbar)]
        hp.set_option('geo_dual_fp32', True)
        want = [x.clone() for x in hp.geo_dual_bwd(canon, stats, pts, gamma, gbar, gdbar)]
    finally:
        hp.set_option('geo_dual_fp32', prev)
    torch.cuda.synchronize()
    for g, x in zip(got, want):
        assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
        assert torch.equal(g, x)


def _geo_dual_fwd_f64(weights_np, level, stats, pts, gamma):
    """geometry_fc (ibrnet.py:488-489) on dual numbers in float64: x = [mean 32, var 32, wbar, embed(p) 21] (neus.py:37-45), tangent along gamma."""
    agg = ('agg_net.' if level == 'coarse' else 'fine_agg_net.') + 'agg_impl.geometry_fc.'
    W1, b1, W2, b2 = (torch.from_numpy(np.asarray(weights_np[agg + k])).double() for k in ('0.weight', '0.bias', '2.weight', '2.bias'))
    st, p, gm = (torch.from_numpy(x).double() for x in (stats, pts, gamma))
    emb = [p] + [f(p * m) for m in (1.0, 2.0, 4.0) for f in (torch.sin, torch.cos)]
    demb = [gm] + [d * gm for m in (1.0, 2.0, 4.0) for d in (m * torch.cos(p * m), -m * torch.sin(p * m))]
    x = torch.cat([st[:, :65]] + emb, 1)
    xd = torch.cat(demb, 1)
    h = x @ W1.T + b1
    hd = xd @ W1[:, 65:].T
    elu = lambda u: torch.where(u > 0, u, torch.expm1(u))
    delu = lambda u: torch.where(u > 0, torch.ones_like(u), torch.exp(u))
    gp = elu(h) @ W2.T + b2
    gpd = (delu(h) * hd) @ W2.T
    return elu(gp), delu(gp) * gpd


@pytest.mark.gpu
@pytest.mark.parametrize('scale_x,scale_g,Pn', [(1.0, 1.0, 4 * 512 * 40 + 13), (1.0, 1e-20, 777), (3e4, 1e3, 20000), (1.0, 1.0, 5)])
def test_geo_dual_fwd_on_the_matrix_cores_agrees_with_the_fp32_kernel_and_float64(scale_x, scale_g, Pn, weights_np):
    """gnr_geo_dual_fwd as a chained fp16-pair MFMA (k_geo_dual_fwd_mm, the default since round 6) against the fp32 FMA kernel
    (GNR_OPT_GEO_DUAL_FP32) and a float64 evaluation: value and tangent of geometry_fc's output; the matrix-core kernel is no farther from
    float64 than the fp32 kernel (x 1.5); tangents of 1e-20 (operand blocks are normalised before they are split), statistics near the
    fp16 limit, ragged point counts."""
    from graspnerf_amd.hotpath import HotPath
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    canon = torch.from_numpy(weights.canonical_blob(weights_np, 'fine')).cuda()
    rng = np.random.default_rng(12)
    stats = (rng.standard_normal((Pn, 66)) * scale_x).astype(np.float32)
    stats[:, 32:64] = np.abs(stats[:, 32:64]); stats[:, 64] = rng.uniform(0, 1, Pn); stats[:, 65] = 6
    pts = rng.uniform(-0.5, 0.5, (Pn, 3)).astype(np.float32)
    gamma = (rng.standard_normal((Pn, 3)) * scale_g).astype(np.float32)
    prev = hp.set_option('geo_dual_fp32', False)
    try:
        got = [x.double().cpu() for x in hp.geo_dual_fwd(canon, stats, pts, gamma)]
        hp.set_option('geo_dual_fp32', True)
        f32 = [x.double().cpu() for x in hp.geo_dual_fwd(canon, stats, pts, gamma)]
    finally:
        hp.set_option('geo_dual_fp32', prev)
    want = _geo_dual_fwd_f64(weights_np, 'fine', stats, pts, gamma)
    for g, f, w, name in zip(got, f32, want, ('g', 'gd')):
        assert bool(torch.isfinite(g).all()), name
        sc = float(w.abs().max())
        assert sc > 0
        e_mm, e_32 = float((g - w).pow(2).mean().sqrt()) / sc, float((f - w).pow(2).mean().sqrt()) / sc
        assert e_mm <= 1.5 * e_32 + 1e-9, (name, e_mm, e_32)
        if scale_x == 1.0:
            assert float((g - f).abs().max()) <= 2e-5 * sc, name
        else:                               # pre-activations of 1e5: a hidden unit on the other side of the ELU's kink in one of the two kernels
            row = (g - f).abs().max(1)[0] / f.abs().max(1)[0].clamp(min=1e-30)
            assert float(row.median()) <= 2e-5 and float((row > 1e-3).double().mean()) <= 5e-3, name


@pytest.mark.gpu
def test_geo_dual_fwd_weight_without_an_fp16_pair_falls_back_to_the_fp32_kernel(weights_np):
    """A geometry_fc weight of 1e5 has no fp16 pair (k_pack_geo_dual stores inf): k_geo_dual_fwd_mm's outputs turn non-finite, its range word
    makes the fp32 kernel launched behind it recompute the call -- the outputs are that kernel's, bit for bit."""
    from graspnerf_amd.hotpath import HotPath
    w = dict(weights_np)
    k = 'fine_agg_net.agg_impl.geometry_fc.2.weight'
    w[k] = w[k].copy(); w[k][5, 11] = 1e5
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    canon = torch.from_numpy(weights.canonical_blob(w, 'fine')).cuda()
    rng = np.random.default_rng(3)
    Pn = 5000
    stats = rng.standard_normal((Pn, 66)).astype(np.float32); stats[:, 32:64] = np.abs(stats[:, 32:64]); stats[:, 65] = 6
    pts = rng.uniform(-0.5, 0.5, (Pn, 3)).astype(np.float32)
    gamma = rng.standard_normal((Pn, 3)).astype(np.float32) * 1e-3
    prev = hp.set_option('geo_dual_fp32', False)
    try:
        got = [x.clone() for x in hp.geo_dual_fwd(canon, stats, pts, gamma)]
        hp.set_option('geo_dual_fp32', True)
        want = [x.clone() for x in hp.geo_dual_fwd(canon, stats, pts, gamma)]
    finally:
        hp.set_option('geo_dual_fp32', prev)
    torch.cuda.synchronize()
    for g, x in zip(got, want):
        assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
        assert torch.equal(g, x)


def test_positive_cumprod_backward_equals_autograd():
    """reference_autograd._CumprodPositive: torch.cumprod's values, and its gradient for strictly positive factors, without the
    `(x == 0).any()` host read of the stock backward."""
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(7, 41, generator=g, dtype=torch.float64) + 1e-10).requires_grad_(True)
    up = torch.randn(7, 41, generator=g, dtype=torch.float64)
    want, = torch.autograd.grad((torch.cumprod(x, -1) * up).sum(), x)
    y = ag._CumprodPositive.apply(x)
    got, = torch.autograd.grad((y * up).sum(), x)
    assert torch.equal(y, torch.cumprod(x, -1)) and _rel(got, want) < 1e-12
