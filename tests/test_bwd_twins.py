"""Backward twins of the HIP path against PyTorch autograd over tests/reference_autograd.py (which is itself pinned
to the reference's backward by tests/test_train_step.py).  Round 1: gnr_depth_mean_bwd."""
import numpy as np
import pytest
import torch

from graspnerf_amd import weights
from graspnerf_amd.synth import make_scene


def test_bwd_blob_layout(weights_np):
    """CPU: the transposed fragments hold W^T where the chained-MFMA convention expects it."""
    from graspnerf_amd import _lib
    can = weights.canonical_blob(weights_np, 'coarse')
    pb = weights.pack_bwd(can)
    assert pb.size == _lib.lib().gnr_packed_bwd_floats() >= 2048
    W2 = weights_np['dist_decoder.mean_decoder.2.weight']
    W1 = weights_np['dist_decoder.mean_decoder.0.weight']
    nat = lambda j, g: 16 * (j // 4) + 4 * g + j % 4
    for j in range(8):
        for nb in range(2):
            for lane in range(64):
                r16, g = lane & 15, lane >> 4
                # frag[(j, nb, lane)] = W^T[out(nb, lane&15)][in = nat(j, g)] = W[nat(j, g)][out]
                assert pb[(j * 64 + lane) * 2 + nb] == W2[nat(j, g), 16 * nb + r16]
                ch = 8 * (r16 // 4) + 4 * nb + r16 % 4                  # gather layout of the output rows
                assert pb[1024 + (j * 64 + lane) * 2 + nb] == W1[nat(j, g), ch]
    g = np.arange(36958, dtype=np.float32)
    parts = weights.split_canonical(g, 'fine')
    assert parts['fine_dist_decoder.mean_decoder.2.weight'][0, 0] == 1056 and len(parts) == 63


@pytest.mark.gpu
@pytest.mark.parametrize('level,pn', [('coarse', 333), ('fine', 64)])
def test_depth_mean_bwd_matches_autograd(level, pn, weights_np):
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    import reference_autograd as ag
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    hp.set_bwd_weights(weights.pack_bwd(weights.canonical_blob(weights_np, 'coarse')),
                       weights.pack_bwd(weights.canonical_blob(weights_np, 'fine')))
    scenes = [make_scene(s, 'cfg1') for s in (0, 1)]
    bref, _ = batch_scenes(scenes)
    rng = np.random.default_rng(2)
    H, Wd = bref['imgs'].shape[-2:]
    coords = np.stack([rng.uniform(-1, Wd, (2, pn)), rng.uniform(-1, H, (2, pn))], -1).astype(np.float32)
    dmean = rng.standard_normal((2, 3, pn, 2)).astype(np.float32)
    prep = hp.prepare(bref, 1)
    dcan, dray = hp.depth_mean_bwd(bref, coords, dmean, level, prepared=prep)
    torch.cuda.synchronize()
    got = weights.split_canonical(dcan.cpu().numpy(), level)
    dec = 'dist_decoder.' if level == 'coarse' else 'fine_dist_decoder.'
    # autograd reference, scene by scene
    P = {k: torch.from_numpy(v).cuda().requires_grad_(k.startswith(dec + 'mean_decoder')) for k, v in weights_np.items()}
    ray = torch.from_numpy(bref['ray_feats']).cuda().requires_grad_(True)
    tot = 0
    for b in range(2):
        ref = {'imgs': torch.from_numpy(bref['imgs'][b]).cuda(), 'ray_feats': ray[b]}
        m = ag.depth_mean(P, ref, torch.from_numpy(coords[b]).cuda(), dec)           # coords are used as (x, y)
        tot = tot + (m * torch.from_numpy(dmean[b]).cuda()).sum()
    tot.backward()
    for name in ('0.weight', '0.bias', '2.weight', '2.bias', '4.weight', '4.bias'):
        k = dec + 'mean_decoder.' + name
        r = P[k].grad.cpu().numpy()
        assert np.abs(got[k] - r).max() <= 2e-4 * np.abs(r).max() + 1e-6, k
    others = [k for k in got if 'mean_decoder' not in k]
    assert all(np.all(got[k] == 0) for k in others)
    r = ray.grad.cpu().numpy()
    assert np.abs(dray.cpu().numpy() - r).max() <= 2e-4 * np.abs(r).max() + 1e-6
    # forward values of the twin pair agree as well
    fwd = hp.depth_mean(bref, coords, level, prepared=prep).cpu().numpy()
    with torch.no_grad():
        m0 = ag.depth_mean(P, {'imgs': torch.from_numpy(bref['imgs'][0]).cuda(), 'ray_feats': ray[0]},
                           torch.from_numpy(coords[0]).cuda(), dec).cpu().numpy()
    assert np.abs(fwd[0] - m0).max() < 1e-4


# ---- sample_volume backward, stage by stage against autograd taps ------------------------------------------------
def _regs_to_feats(buf, tiles, V, nregs, P):
    """[tiles][V][nregs][64 lanes] register dump -> [V, P, nregs, 4 groups] (lane = 16*g + r, point = 16*tile + r)."""
    a = buf[:tiles * V * nregs * 64].reshape(tiles, V, nregs, 4, 16)
    return a.permute(1, 0, 4, 2, 3).reshape(V, tiles * 16, nregs, 4)[:, :P]


def _nat(x8):
    """[..., 8 regs, 4 groups] natural layout -> [..., 32] features (feature = 16*(j//4) + 4*g + j%4)."""
    out = torch.empty(*x8.shape[:-2], 32, dtype=x8.dtype, device=x8.device)
    for j in range(8):
        for g in range(4):
            out[..., 16 * (j // 4) + 4 * g + j % 4] = x8[..., j, g]
    return out


@pytest.fixture(scope='module')
def vol_bwd_case(weights_np):
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    import reference_autograd as ag
    res = 16
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    can = weights.canonical_blob(weights_np, 'coarse')
    hp.set_bwd_weights(weights.pack_bwd(can))
    ref, _ = make_scene(3, 'cfg1')
    bref, _ = batch_scenes([(ref, make_scene(3, 'cfg1')[1])])
    vol = hp.sample_volume_train(bref, res)
    rng = np.random.default_rng(5)
    dvol = torch.from_numpy(rng.standard_normal((1, 1, res, res, res)).astype(np.float32)).cuda()
    # autograd reference with taps
    P = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in weights_np.items()}
    tref = {k: torch.from_numpy(v).cuda() for k, v in ref.items()}
    tref['ray_feats'].requires_grad_(True)
    tref['img_feats'].requires_grad_(True)
    taps = {}
    vol_ag = ag.sample_volume(P, tref, res, taps=taps)
    (vol_ag * dvol).sum().backward()
    return dict(hp=hp, can=torch.from_numpy(can).cuda(), vol=vol, vol_ag=vol_ag.detach(), dvol=dvol, P=P, taps=taps, tref=tref, res=res)


def _close(got, ref, what, rel=3e-4, atol=2e-6):
    """atol: some parameters have an exactly-zero gradient (a bias in front of the softmax over views); both sides then
    hold accumulated rounding noise."""
    got, ref = got.detach().float().cpu().numpy(), ref.detach().float().cpu().numpy()
    scale = max(np.abs(ref).max(), 1e-8)
    assert np.abs(got - ref).max() <= rel * scale + atol, f'{what}: max|d| {np.abs(got - ref).max():.3e} vs scale {scale:.3e}'


@pytest.mark.gpu
def test_volume_bwd_tail_and_geometry(vol_bwd_case):
    c = vol_bwd_case
    hp, res, taps, P = c['hp'], c['res'], c['taps'], c['P']
    _close(c['vol'], c['vol_ag'], 'training forward volume', rel=1e-4)
    dcan, _, _ = hp.sample_volume_bwd(c['dvol'], c['can'], stages=16 | 8)
    torch.cuda.synchronize()
    scene = hp._train_ctx[0]
    npts = res ** 3
    _close(hp.train_ws_section('dg16', scene, res)[:npts * 16].reshape(npts, 16), taps['g16'].grad, 'd g16')
    got = weights.split_canonical(dcan, 'coarse')
    for k in ('ray_attention.w_qs.weight', 'ray_attention.w_ks.weight', 'ray_attention.w_vs.weight', 'ray_attention.fc.weight',
              'ray_attention.layer_norm.weight', 'ray_attention.layer_norm.bias', 'out_geometry_fc.0.weight',
              'out_geometry_fc.0.bias', 'out_geometry_fc.1.weight', 'out_geometry_fc.1.bias',
              'geometry_fc.2.weight', 'geometry_fc.2.bias', 'geometry_fc.0.weight', 'geometry_fc.0.bias'):
        kk = 'agg_net.agg_impl.' + k
        _close(got[kk], P[kk].grad, kk)
    V, tiles = scene.V, npts // 16
    d = _regs_to_feats(hp.train_ws_section('dS2', scene, res), tiles, V, 9, npts)        # [V,P,9,4]
    # h also feeds vis_fc2 (stage 3 adds that path): compare the direct part through the weighted mean / variance
    with torch.no_grad():
        w2 = taps['v2'] / (taps['v2'].sum(0, keepdim=True) + 1e-8)
        direct = w2 * (taps['mean'].grad[None] + 2 * (taps['h'] - taps['mean'][None]) * taps['var'].grad[None])
    _close(_nat(d[:, :, :8]), direct, 'd h_v (direct part)')
    _close(d[:, :, 8, 0], taps['v2'].grad[..., 0], 'd v2_v')


@pytest.mark.gpu
def test_volume_bwd_second_view_loop(vol_bwd_case):
    c = vol_bwd_case
    hp, res, taps, P = c['hp'], c['res'], c['taps'], c['P']
    dcan, _, _ = hp.sample_volume_bwd(c['dvol'], c['can'], stages=16 | 8 | 4)
    torch.cuda.synchronize()
    scene = hp._train_ctx[0]
    npts, tiles = res ** 3, res ** 3 // 16
    got = weights.split_canonical(dcan, 'coarse')
    for k in ('vis_fc2.2.weight', 'vis_fc2.2.bias', 'vis_fc2.0.weight', 'vis_fc2.0.bias', 'vis_fc.2.weight', 'vis_fc.2.bias',
              'vis_fc.0.weight', 'vis_fc.0.bias', 'base_fc.2.weight', 'base_fc.2.bias', 'base_fc.0.bias'):
        kk = 'agg_net.agg_impl.' + k
        _close(got[kk], P[kk].grad, kk)
    kk = 'agg_net.agg_impl.base_fc.0.weight'
    _close(got[kk][:, 140:], P[kk].grad[:, 140:], kk + '[:,140:]')
    dG = _regs_to_feats(hp.train_ws_section('dG', scene, res), tiles, 1, 16, npts)[0]          # [P,16,4]
    dGn = torch.empty(npts, 64, device=dG.device)
    for j in range(16):
        for g in range(4):
            dGn[:, 16 * (j // 4) + 4 * g + j % 4] = dG[:, j, g]
    _close(dGn, taps['G'].grad, 'd G')


def _xslots_to_channels(d9):
    """[..., 9 slots, 4 groups] x-slot layout -> [..., 35] channels [r g b | img feats 32] (slot j<8: 3+8g+j; slot 8: g<3)."""
    out = torch.zeros(*d9.shape[:-2], 35, dtype=d9.dtype, device=d9.device)
    for g in range(4):
        for j in range(8):
            out[..., 3 + 8 * g + j] = d9[..., j, g]
        if g < 3:
            out[..., g] = d9[..., 8, g]
    return out


@pytest.mark.gpu
def test_volume_bwd_hoist_and_first_reduction(vol_bwd_case):
    c = vol_bwd_case
    hp, res, taps, P = c['hp'], c['res'], c['taps'], c['P']
    dcan, _, _ = hp.sample_volume_bwd(c['dvol'], c['can'], stages=16 | 8 | 4 | 2)
    torch.cuda.synchronize()
    scene = hp._train_ctx[0]
    npts, tiles, V = res ** 3, res ** 3 // 16, scene.V
    got = weights.split_canonical(dcan, 'coarse')
    kk = 'agg_net.agg_impl.base_fc.0.weight'
    _close(got[kk], P[kk].grad, kk)
    d = _regs_to_feats(hp.train_ws_section('dS1', scene, res), tiles, V, 18, npts)           # [V,P,18,4]
    _close(_xslots_to_channels(d[:, :, :9]), taps['x'].grad, 'd x_v (per-view + statistics paths)')
    _close(d[:, :, 17, 0], taps['gate'].grad[..., 0], 'd gate_v')


@pytest.mark.gpu
def test_volume_bwd_complete(vol_bwd_case):
    """All five stages: every coarse-level parameter gradient and both feature-map gradients of sample_volume."""
    c = vol_bwd_case
    hp, P, tref = c['hp'], c['P'], c['tref']
    dcan, dray, dimg = hp.sample_volume_bwd(c['dvol'], c['can'])
    torch.cuda.synchronize()
    got = weights.split_canonical(dcan, 'coarse')
    checked = 0
    for k, gv in got.items():
        ref = P[k].grad
        if ref is None:                                  # not on the volume path (rgb_fc, variance)
            assert float(gv.abs().max()) == 0.0, k
            continue
        _close(gv, ref, k, rel=5e-4)
        checked += 1
    assert checked >= 50
    _close(dray[0], tref['ray_feats'].grad, 'd ray_feats', rel=5e-4)
    _close(dimg[0], tref['img_feats'].grad, 'd img_feats', rel=5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('V,res,B', [(2, 5, 2), (6, 8, 1), (8, 6, 1), (5, 7, 3)])
def test_volume_bwd_other_shapes(V, res, B, weights_np):
    """View counts 2..8, grids whose point count is not a multiple of the 16-point tile, several scenes per call."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    import reference_autograd as ag
    from graspnerf_amd.synth import CONFIGS
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    can = weights.canonical_blob(weights_np, 'coarse')
    hp.set_bwd_weights(weights.pack_bwd(can))
    scenes = [make_scene(40 + i, dict(CONFIGS['cfg1'], V=V, rn=4)) for i in range(B)]
    bref, _ = batch_scenes(scenes)
    vol = hp.sample_volume_train(bref, res)
    rng = np.random.default_rng(V * 100 + res)
    dvol = torch.from_numpy(rng.standard_normal((B, 1, res, res, res)).astype(np.float32)).cuda()
    dcan, dray, dimg = hp.sample_volume_bwd(dvol, torch.from_numpy(can).cuda())
    torch.cuda.synchronize()
    P = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in weights_np.items()}
    rays, imgs = [], []
    for b in range(B):
        tref = {k: torch.from_numpy(v).cuda() for k, v in scenes[b][0].items()}
        tref['ray_feats'].requires_grad_(True); tref['img_feats'].requires_grad_(True)
        v_ag = ag.sample_volume(P, tref, res)
        _close(vol[b:b + 1], v_ag, f'volume scene {b}', rel=2e-4)
        (v_ag * dvol[b:b + 1]).sum().backward()
        rays.append(tref['ray_feats'].grad); imgs.append(tref['img_feats'].grad)
    got = weights.split_canonical(dcan, 'coarse')
    for k, gv in got.items():
        if P[k].grad is not None:
            _close(gv, P[k].grad, k, rel=1e-3)
    _close(dray, torch.stack(rays), 'd ray_feats', rel=1e-3)
    _close(dimg, torch.stack(imgs), 'd img_feats', rel=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('V,rn,dn', [(4, 5, 7), (6, 33, 40), (2, 3, 16)])
def test_render_chain_twin_matches_autograd(V, rn, dn, weights_np):
    """One render pass: statistics / colours of the HIP chain against the autograd chain, and the gradients a random
    upstream (d stats, d colours) produces for the level's parameters and both feature maps (fine level)."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    import reference_autograd as ag
    from graspnerf_amd.synth import CONFIGS
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    hp.set_bwd_weights(weights.pack_bwd(weights.canonical_blob(weights_np, 'coarse')), weights.pack_bwd(weights.canonical_blob(weights_np, 'fine')))
    ref, que = make_scene(60 + V, dict(CONFIGS['cfg1'], V=V, rn=rn))
    bref, bque = batch_scenes([(ref, que)])
    prep = hp.prepare(bref, 1, rn, dn)
    rng = np.random.default_rng(rn)
    depth = torch.sort(torch.from_numpy(rng.uniform(0.25, 0.75, (rn, dn)).astype(np.float32)), -1)[0].cuda()
    cfg = {'depth_sample_num': dn, 'fine_depth_sample_num': dn, 'ray_mask_view_num': 2, 'ray_mask_point_num': 8}
    bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
    stats, colors, geo, ctx = hp.render_chain_train(bq, depth[None], 'fine', cfg, prep)
    # autograd chain on the same points
    P = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in weights_np.items()}
    tref = {k: torch.from_numpy(v).cuda() for k, v in ref.items()}
    tref['ray_feats'].requires_grad_(True); tref['img_feats'].requires_grad_(True)
    q1 = {'coords': bq['coords'][0], 'pose': bq['pose'][0], 'K': bq['K'][0], 'depth_range': bq['depth_range'][0]}
    pts, qdir = ag.ray_points(q1, depth)
    h, w = ref['imgs'].shape[-2:]
    uv, z, mask, dirv = ag.project(pts, tref['poses'], tref['Ks'], h, w)
    f_ray, rgb, f_img = ag._gather(tref, uv, mask)
    near, far = -1 / q1['depth_range'][0], -1 / q1['depth_range'][1]
    di = (-1 / depth - near) / (far - near)
    half = torch.cat([di[:, 1:] - di[:, :-1], torch.full_like(di[:, :1], 1e6)], -1) / 2
    ext = torch.cat([half[:, :1], half], -1)
    hit, vis = ag.decode_hit_vis(P, 'fine_dist_decoder.', f_ray, z, mask, tref['depth_range'], ext[:, :-1].reshape(-1), ext[:, 1:].reshape(-1))
    taps = {}
    qd = qdir[:, None].expand(rn, dn, 3).reshape(-1, 3)
    _, _, col = ag.aggregate(P, 'fine_agg_net.', f_ray, rgb, f_img, hit, vis, mask, dirv, qd, pts, rn, dn, False, True, taps)
    v2 = taps['v2']
    wbar = (v2 / (v2.sum(0, keepdim=True) + 1e-8)).mean(0)
    ref_stats = torch.cat([taps['mean'], taps['var'], wbar], -1)
    _close(stats[0, :, :65], ref_stats, 'chain statistics', rel=2e-4)
    _close(colors[0], col.reshape(-1, 3), 'chain colours', rel=2e-4)
    assert torch.equal(stats[0, :, 65], mask.sum(0).float())
    ds = torch.from_numpy(rng.standard_normal((rn * dn, 65)).astype(np.float32)).cuda()
    dc = torch.from_numpy(rng.standard_normal((rn * dn, 3)).astype(np.float32)).cuda()
    ((ref_stats * ds).sum() + (col.reshape(-1, 3) * dc).sum()).backward()
    dcan, dray, dimg = hp.render_chain_bwd(ctx, ds[None], dc[None])
    torch.cuda.synchronize()
    got = weights.split_canonical(dcan, 'fine')
    n = 0
    for k, gv in got.items():
        if k.endswith('rgb_fc.4.bias'):                  # bias in front of the softmax over views: the gradient is exactly
            assert float(gv.abs().max()) < 1e-4          # zero, both sides hold rounding noise that grows with the point count
            continue
        if P[k].grad is not None:
            _close(gv, P[k].grad, k, rel=1e-3)
            n += 1
    assert n >= 40
    _close(dray[0], tref['ray_feats'].grad, 'd ray_feats', rel=1e-3)
    _close(dimg[0], tref['img_feats'].grad, 'd img_feats', rel=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('V,res,B', [(6, 8, 2), (3, 5, 1)])
def test_partner_wavefront_kernels_agree_with_the_single_wavefront_kernels(V, res, B, weights_np):
    """Round 5: the two view loops' backward runs as k_view{1,2}_bwd_pw (a compute wavefront and its partner per tile, two per SIMD);
    the single-wavefront kernels of rounds 1-4 stay in the library (use_vis levels take k_view1_bwd<true>) behind
    GNR_OPT_VIEW{1,2}_ONE_WAVEFRONT.  Same sums in another order: parameter gradients (per tensor) and feature-map gradients equal to 2e-5
    of their scale -- on the volume points and on a render pass, with small (1e-4) and large (1e3) upstream gradients."""
    from graspnerf_amd import _lib
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    from graspnerf_amd.synth import CONFIGS
    L = _lib.lib()
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    can = {lvl: weights.canonical_blob(weights_np, lvl) for lvl in ('coarse', 'fine')}
    hp.set_bwd_weights(weights.pack_bwd(can['coarse']), weights.pack_bwd(can['fine']))
    rn, dn = 21, 16
    scenes = [make_scene(70 + i, dict(CONFIGS['cfg1'], V=V, rn=rn)) for i in range(B)]
    bref, bque = batch_scenes(scenes)
    rng = np.random.default_rng(V + res)
    dvol = torch.from_numpy(rng.standard_normal((B, 1, res, res, res)).astype(np.float32)).cuda() * 1e-4     # small gradients: the power-of-two normalisation at work
    depth = torch.sort(torch.from_numpy(rng.uniform(0.25, 0.75, (B, rn, dn)).astype(np.float32)), -1)[0].cuda()
    ds = torch.from_numpy(rng.standard_normal((B, rn * dn, 65)).astype(np.float32)).cuda() * 1e3            # and large ones
    dc = torch.from_numpy(rng.standard_normal((B, rn * dn, 3)).astype(np.float32)).cuda() * 1e3
    cfg = {'depth_sample_num': dn, 'fine_depth_sample_num': dn, 'ray_mask_view_num': 2, 'ray_mask_point_num': 8}
    bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
    out = {}
    try:
        for mode in (1, 0):
            hp.set_option('view1_one_wavefront', not mode); hp.set_option('view2_one_wavefront', not mode)
            hp.sample_volume_train(bref, res)
            vol_g = hp.sample_volume_bwd(dvol, torch.from_numpy(can['coarse']).cuda())
            prep = hp.prepare(bref, 1, rn, dn)
            _, _, _, ctx = hp.render_chain_train(bq, depth, 'fine', cfg, prep)
            ren_g = hp.render_chain_bwd(ctx, ds, dc)
            torch.cuda.synchronize()
            out[mode] = [t.clone() for t in vol_g] + [t.clone() for t in ren_g]
    finally:
        hp.set_option('view1_one_wavefront', False); hp.set_option('view2_one_wavefront', False)
    names = ['volume d_canonical', 'volume d_ray_feats', 'volume d_img_feats', 'render d_canonical', 'render d_ray_feats', 'render d_img_feats']
    for name, a, b in zip(names, out[1], out[0]):
        assert torch.isfinite(a).all() and float(b.abs().max()) > 0, name
        tol = 2e-5           # another summation order of fp32 partial sums (sums with cancellation: a bias of 7e-7 moved by 2e-12)
        # per parameter tensor for the blobs (their scales differ by orders of magnitude), whole tensor for the feature maps
        if 'canonical' in name:
            lvl = 'coarse' if name.startswith('volume') else 'fine'
            ga, gb = weights.split_canonical(a[:36958], lvl), weights.split_canonical(b[:36958], lvl)
            for k in ga:
                sc = float(gb[k].abs().max())
                if sc > 0 and not k.endswith('rgb_fc.4.bias'):         # (a bias in front of a softmax over views: exactly zero, both sides hold rounding noise)
                    assert float((ga[k] - gb[k]).abs().max()) <= tol * sc + 1e-30, (name, k, float((ga[k] - gb[k]).abs().max()), sc)
        else:
            assert float((a - b).abs().max()) <= tol * float(b.abs().max()), name
