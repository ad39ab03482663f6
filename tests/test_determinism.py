"""Parameter gradients of the backward twins are deterministic (csrc/gnr_bwd.inc: per-wavefront partial slots + a
fixed-order reduction in double, no float atomics): two calls on the same inputs return the same BITS, like the
reference's CPU backward (ibrnet.py:497-504 + autograd, trainer.py:146-158), and every entry of every partial slot is
stored by exactly one wavefront (the poisoned run would show a missing store as NaN in the reduced gradient).
The feature-map gradients are a bilinear scatter: with float atomics (the default, like ATen's grid_sampler backward on a GPU) they
may differ between runs within rounding; with GNR_OPT_FEATURE_GRAD_FIXED (64-bit fixed-point adds, include/gnr.h) they are equal
bit for bit as well, and equal to the float path's within its own run-to-run spread.  Every test runs in both modes."""
import numpy as np
import pytest
import torch

from graspnerf_amd import weights, _lib
from graspnerf_amd.synth import make_scene, CONFIGS

pytestmark = pytest.mark.gpu


def _hot(weights_np):
    from graspnerf_amd.hotpath import HotPath
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    hp.set_bwd_weights(weights.pack_bwd(weights.canonical_blob(weights_np, 'coarse')), weights.pack_bwd(weights.canonical_blob(weights_np, 'fine')))
    hp.can_dev = {lvl: torch.from_numpy(weights.canonical_blob(weights_np, lvl)).cuda() for lvl in ('coarse', 'fine')}
    return hp


def _feat_close(a, b):
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-9


@pytest.fixture(params=['float_atomics', 'fixed_point'])
def feat_same(request):
    """Sets the feature-gradient mode of the HotPath objects the test creates (a per-call option, HotPath.default_options); yields the
    comparison two runs' feature-map gradients must pass."""
    from graspnerf_amd.hotpath import HotPath
    bit = _lib.OPTIONS['feature_grad_fixed']
    prev = HotPath.default_options
    HotPath.default_options = (prev | bit) if request.param == 'fixed_point' else (prev & ~bit)

    def same(a, b):
        if request.param == 'fixed_point':
            assert torch.equal(a, b)
        else:
            _feat_close(a, b)
    same.fixed = request.param == 'fixed_point'
    yield same
    HotPath.default_options = prev


def _float_mode(hp, fn):
    """fn() with hp's float scatter (reference values for the fixed-point runs)."""
    prev = hp.feature_grad_mode(False)
    try:
        return fn()
    finally:
        hp.feature_grad_mode(prev)


def _not_clamped(hp, prep=None):
    assert hp.range_status(prep) & 8 == 0, 'a feature-gradient contribution left the fixed-point range'


@pytest.fixture()
def poison():
    """Runs the test body with the partial buffers poisoned (NaN patterns) before every backward kernel."""
    from graspnerf_amd.hotpath import HotPath
    prev = HotPath.default_options
    HotPath.default_options = prev | _lib.OPTIONS['poison_partials']
    yield
    HotPath.default_options = prev


def _volume_case(hp, V, res, B, H=96, W=128):
    from graspnerf_amd.hotpath import batch_scenes
    scenes = [make_scene(90 + i, dict(CONFIGS['cfg1' if H == 96 else 'cfg2'], V=V, rn=4)) for i in range(B)]
    bref, _ = batch_scenes(scenes)
    hp.sample_volume_train(bref, res)
    dvol = torch.from_numpy(np.random.default_rng(V + res).standard_normal((B, 1, res, res, res)).astype(np.float32)).cuda()
    return dvol


@pytest.mark.parametrize('V,res,B,H,W', [(3, 16, 2, 96, 128), (6, 40, 2, 288, 512)])
def test_sample_volume_bwd_is_bit_reproducible(V, res, B, H, W, weights_np, poison, feat_same):
    """All five stages, the second case at the benchmark's scene size (6 views 288x512, 40^3: every CU's four wavefronts hold
    tiles): parameter gradients equal bit for bit, no entry of a slot left unstored."""
    hp = _hot(weights_np)
    dvol = _volume_case(hp, V, res, B, H, W)
    runs = [hp.sample_volume_bwd(dvol, hp.can_dev['coarse']) for _ in range(3)]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(runs[0][0]).all())
    assert float(runs[0][0].abs().max()) > 0
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0])
        feat_same(r[1], runs[0][1]); feat_same(r[2], runs[0][2])
    if feat_same.fixed:                                            # the same values as the float path, to that path's own spread
        ref = _float_mode(hp, lambda: hp.sample_volume_bwd(dvol, hp.can_dev['coarse']))
        assert torch.equal(ref[0], runs[0][0])
        _feat_close(runs[0][1], ref[1]); _feat_close(runs[0][2], ref[2])
        assert float(runs[0][1].abs().max()) > 0 and float(runs[0][2].abs().max()) > 0
        _not_clamped(hp)


@pytest.mark.parametrize('V,rn,dn', [(3, 33, 16), (6, 512, 40)])
def test_render_pass_bwd_is_bit_reproducible(V, rn, dn, weights_np, poison, feat_same):
    """One training render pass backwards: per-view chain (k_red2 / k_view2<true> / k_hoist / k_view1), geometry_fc on dual numbers,
    the per-ray tail with its second-order path and the compositing -- every parameter gradient equal bit for bit."""
    from graspnerf_amd.hotpath import batch_scenes
    hp = _hot(weights_np)
    cfgs = dict(CONFIGS['cfg1'] if V == 3 else CONFIGS['cfg2'], V=V, rn=rn)
    bref, bque = batch_scenes([make_scene(70 + i, cfgs) for i in range(2)])
    B = 2
    prep = hp.prepare(bref, 1, rn, dn)
    rng = np.random.default_rng(rn)
    depth = torch.sort(torch.from_numpy(rng.uniform(0.25, 0.75, (B, rn, dn)).astype(np.float32)), -1)[0].cuda()
    cfg = {'depth_sample_num': min(dn, 64), 'fine_depth_sample_num': min(dn, 64), 'ray_mask_view_num': 2, 'ray_mask_point_num': 8}
    bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
    stats, colors, geo, ctx = hp.render_chain_train(bq, depth, 'fine', cfg, prep)
    fw = hp.render_tail_train(ctx, bq, depth, colors)
    N = B * rn * dn
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
    ds, dc, a, gamma = t(B, rn * dn, 65), t(B, rn * dn, 3), t(B * rn, dn), t(B * rn, dn, 3)
    st = stats.reshape(N, 66)
    canon = hp.can_dev['fine']
    dpix, wg = t(B * rn, 3), torch.full((B * rn,), 1e-3, device='cuda')

    def once():
        dcan, dray, dimg = hp.render_chain_bwd(ctx, ds, dc)
        g, gd = hp.geo_dual_fwd(canon, st, geo['pts'], gamma.reshape(-1, 3))
        gbar, gdbar, dt = hp.ray_tail_dual_bwd('fine', g.reshape(B * rn, dn, 16), gd.reshape(B * rn, dn, 16), a, st[:, 65].reshape(B * rn, dn))
        dstats, dgeo = hp.geo_dual_bwd(canon, st, geo['pts'], gamma.reshape(-1, 3), gbar.reshape(-1, 16), gdbar.reshape(-1, 16))
        comp = hp.composite_bwd('fine', fw['sdf_values'].reshape(B * rn, dn), fw['sdf_gradient'].reshape(B * rn, dn, 3),
                                colors.reshape(B * rn, dn, 3), depth.reshape(B * rn, dn), geo['qdir'], dpix, None, wg, a, None)
        return dict(dcan=dcan.clone(), dt=dt.clone(), dgeo=dgeo.clone(), dvar=comp[3].clone(), gbar=gbar.clone(), dstats=dstats.clone(),
                    a=comp[0].clone(), gamma=comp[1].clone()), (dray.clone(), dimg.clone())
    (p0, f0), (p1, f1), (p2, f2) = once(), once(), once()
    torch.cuda.synchronize()
    for k, v in p0.items():
        assert bool(torch.isfinite(v).all()), k
        assert float(v.abs().max()) > 0, k
        assert torch.equal(p1[k], v) and torch.equal(p2[k], v), k
    feat_same(f1[0], f0[0]); feat_same(f1[1], f0[1]); feat_same(f2[0], f0[0]); feat_same(f2[1], f0[1])
    if feat_same.fixed:
        pr, fr = _float_mode(hp, once)
        assert torch.equal(pr['dcan'], p0['dcan'])
        _feat_close(f0[0], fr[0]); _feat_close(f0[1], fr[1])
        assert float(f0[0].abs().max()) > 0 and float(f0[1].abs().max()) > 0
        _not_clamped(hp, prep)


def test_depth_mean_bwd_is_bit_reproducible(weights_np, poison, feat_same):
    from graspnerf_amd.hotpath import batch_scenes
    hp = _hot(weights_np)
    bref, _ = batch_scenes([make_scene(s, 'cfg2') for s in (0, 1)])
    rng = np.random.default_rng(4)
    pn = 8192
    coords = np.stack([rng.uniform(-1, 512, (2, pn)), rng.uniform(-1, 288, (2, pn))], -1).astype(np.float32)
    dmean = rng.standard_normal((2, 6, pn, 2)).astype(np.float32)
    prep = hp.prepare(bref, 1)
    runs = [hp.depth_mean_bwd(bref, coords, dmean, 'coarse', prepared=prep) for _ in range(3)]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(runs[0][0]).all()) and float(runs[0][0].abs().max()) > 0
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0])
        feat_same(r[1], runs[0][1])
    if feat_same.fixed:
        ref = _float_mode(hp, lambda: hp.depth_mean_bwd(bref, coords, dmean, 'coarse', prepared=prep))
        _feat_close(runs[0][1], ref[1])
        assert float(runs[0][1].abs().max()) > 0
        _not_clamped(hp, prep)
        zero = hp.depth_mean_bwd(bref, coords, np.zeros_like(dmean), 'coarse', prepared=prep)      # no upstream gradient: quantum falls back, nothing added
        assert float(zero[1].abs().max()) == 0.0 and bool(torch.isfinite(zero[0]).all())


def test_fixed_point_feature_gradients_over_forty_orders_of_magnitude(weights_np, feat_same):
    """The quantum follows the launch's largest upstream gradient: upstream gradients scaled by 1e-20 ... 1e+18 give feature-map
    gradients that scale with them (a fixed quantum would return zeros at one end and clamp at the other)."""
    if not feat_same.fixed:
        pytest.skip('fixed-point mode only')
    from graspnerf_amd.hotpath import batch_scenes
    hp = _hot(weights_np)
    bref, _ = batch_scenes([make_scene(s, 'cfg1') for s in (0, 1)])
    rng = np.random.default_rng(9)
    pn = 2048
    coords = np.stack([rng.uniform(-1, 128, (2, pn)), rng.uniform(-1, 96, (2, pn))], -1).astype(np.float32)
    dmean = rng.standard_normal((2, 3, pn, 2)).astype(np.float32)
    prep = hp.prepare(bref, 1)
    base = hp.depth_mean_bwd(bref, coords, dmean, 'coarse', prepared=prep)[1].double()
    for k in (1e-20, 1e-6, 1e+9, 1e+18):
        g = hp.depth_mean_bwd(bref, coords, (dmean.astype(np.float64) * k).astype(np.float32), 'coarse', prepared=prep)[1].double()
        assert float((g / k - base).abs().max()) <= 1e-6 * float(base.abs().max()), k
    _not_clamped(hp, prep)


def test_train_step_path_gradients_are_bit_reproducible(feat_same):
    """A training step of BASELINE configs[4]'s kind (render, volume, depth-mean head, grasp head, losses; two scenes stacked)
    run twice from the same state, the same RNG stream and the same feature maps: every output of the forward and the gradient
    of every parameter of the volumetric path (dist decoders, aggregation nets: 122 tensors) come out bit-identical -- their
    upstream gradients pass through the losses and the grasp head, deterministic kernels as well.  The 2D feature extractors
    are held fixed here (their outputs are recorded once and replayed): MIOpen's convolutions are outside this library and not
    bit-reproducible from call to call on this stack; the feature-map gradients the path hands them are compared to rounding in
    the float-atomic mode and bit for bit in the fixed-point mode (sums over the volume, coarse, fine and depth-mean backward)."""
    from test_train_step import build, scene_data
    from graspnerf_amd.trainer import train_losses_stacked
    from graspnerf_amd import losses
    net = build('cuda')
    net.nr_net.cfg['ray_batch_num'] = 4096                     # all 64 rays of a scene in one chunk: the batched forward
    net.train()
    datas = [dict(scene_data('cuda', scene_id=i, loss_seed=5 + i), step=0) for i in range(2)]
    nr = net.nr_net
    with torch.no_grad():                                      # record the feature maps once
        imgs = torch.cat([d['ref_imgs_info']['imgs'] for d in datas])
        f_img = nr.image_encoder(imgs)
        f_ray = nr.vis_encoder(nr.init_net({'imgs': imgs}, None, True), f_img)
    leaves = {}

    class Replay(torch.nn.Module):
        def __init__(self, name, value):
            super().__init__()
            self.name, self.value = name, value

        def forward(self, *a, **k):
            leaves[self.name] = self.value.clone().requires_grad_(True)
            return leaves[self.name]
    nr.image_encoder = Replay('img', f_img)
    nr.init_net = Replay('init', f_ray)
    nr.vis_encoder = Replay('ray', f_ray)
    grads, fwd, feat = [], [], []
    for it in range(2):
        torch.manual_seed(11)
        net.zero_grad(set_to_none=True)
        st = net.forward_scenes(datas, stacked=True)
        assert st is not None
        losses.total_loss(train_losses_stacked(st, datas), scenes=2).backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
        fwd.append({k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)})
        feat.append({k: v.grad.clone() for k, v in leaves.items() if v.grad is not None})
    for k in fwd[0]:                                               # the forward of the path has no atomics at all
        assert torch.equal(fwd[0][k], fwd[1][k]), ('forward', k)
    path = [k for k in grads[0] if 'dist_decoder' in k or 'agg_net' in k]
    assert len(path) >= 120
    bad = [k for k in path if not torch.equal(grads[0][k], grads[1][k])]
    assert not bad, (len(bad), bad[:6], [float((grads[0][k] - grads[1][k]).abs().max()) for k in bad[:6]])
    assert len(feat[0]) >= 2
    for k in feat[0]:
        feat_same(feat[1][k], feat[0][k])


@pytest.mark.parametrize('pattern', [0x7fc00000, 0x7149f2ca])          # NaN, 1e30
def test_backward_entry_points_ignore_stale_lds(pattern, weights_np):
    """LDS is not cleared between kernels: every backward entry point after gnr_debug_fill_lds(0) and after a NaN / 1e30 flood of
    every CU's LDS (ragged sizes: partly filled tiles and workgroups whose spare lanes shadow valid ones) -- the same bits, in the
    fixed-point feature-gradient mode also for the feature maps.  (The forward's twin: tests/test_range_guard.py::test_stale_lds_...)"""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    prev = HotPath.default_options
    HotPath.default_options = prev | _lib.OPTIONS['feature_grad_fixed']
    try:
        hp = _hot(weights_np)

        def fill(p):
            _lib.check(_lib.lib().gnr_debug_fill_lds(p, torch.cuda.current_stream().cuda_stream), 'gnr_debug_fill_lds')
        # ---- sample_volume backwards (3 views, 16^3, 2 scenes)
        dvol = _volume_case(hp, 3, 16, 2)
        vols = []
        for p in (0, pattern):
            fill(p)
            vols.append([x.clone() for x in hp.sample_volume_bwd(dvol, hp.can_dev['coarse'])])
        for a, b in zip(*vols):
            assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
        # ---- one fine render pass backwards (3 views, 33 rays x 16 samples, 2 scenes)
        rn, dn, B = 33, 16, 2
        bref, bque = batch_scenes([make_scene(70 + i, dict(CONFIGS['cfg1'], V=3, rn=rn)) for i in range(B)])
        prep = hp.prepare(bref, 1, rn, dn)
        rng = np.random.default_rng(rn)
        depth = torch.sort(torch.from_numpy(rng.uniform(0.25, 0.75, (B, rn, dn)).astype(np.float32)), -1)[0].cuda()
        cfg = {'depth_sample_num': dn, 'fine_depth_sample_num': dn, 'ray_mask_view_num': 2, 'ray_mask_point_num': 8}
        bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
        t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
        ds, dc, a_, gamma = t(B, rn * dn, 65), t(B, rn * dn, 3), t(B * rn, dn), t(B * rn, dn, 3)
        dpix, wg = t(B * rn, 3), torch.full((B * rn,), 1e-3, device='cuda')
        canon = hp.can_dev['fine']

        def once(p):
            fill(p); stats, colors, geo, ctx = hp.render_chain_train(bq, depth, 'fine', cfg, prep)
            fill(p); fw = hp.render_tail_train(ctx, bq, depth, colors)
            st = stats.reshape(B * rn * dn, 66)
            fill(p); dcan, dray, dimg = hp.render_chain_bwd(ctx, ds, dc)
            fill(p); g, gd = hp.geo_dual_fwd(canon, st, geo['pts'], gamma.reshape(-1, 3))
            fill(p); gbar, gdbar, dt = hp.ray_tail_dual_bwd('fine', g.reshape(B * rn, dn, 16), gd.reshape(B * rn, dn, 16), a_, st[:, 65].reshape(B * rn, dn))
            fill(p); dstats, dgeo = hp.geo_dual_bwd(canon, st, geo['pts'], gamma.reshape(-1, 3), gbar.reshape(-1, 16), gdbar.reshape(-1, 16))
            fill(p); comp = hp.composite_bwd('fine', fw['sdf_values'].reshape(B * rn, dn), fw['sdf_gradient'].reshape(B * rn, dn, 3),
                                             colors.reshape(B * rn, dn, 3), depth.reshape(B * rn, dn), geo['qdir'], dpix, None, wg, a_, None)
            return dict(stats=stats.clone(), colors=colors.clone(), sdf=fw['sdf_values'].clone(), grad=fw['sdf_gradient'].clone(), dcan=dcan.clone(),
                        dray=dray.clone(), dimg=dimg.clone(), g=g.clone(), gd=gd.clone(), gbar=gbar.clone(), dt=dt.clone(), dstats=dstats.clone(),
                        dgeo=dgeo.clone(), a=comp[0].clone(), gamma=comp[1].clone(), dvar=comp[3].clone())
        r0, r1 = once(0), once(pattern)
        torch.cuda.synchronize()
        for k, v in r0.items():
            assert bool(torch.isfinite(v).all()), k
            assert torch.equal(v, r1[k]), k
    finally:
        HotPath.default_options = prev


def _fill_allocations(monkeypatch, byte):
    """torch.empty hands out memory filled with `byte` (0xff: every float a NaN, every index -1) for the rest of the test: workspaces,
    output arrays, scratch -- whatever graspnerf_amd/hotpath.py allocates without initialising."""
    real = torch.empty

    def filled(*a, **k):
        t = real(*a, **k)
        if t.is_cuda and t.numel():
            t.view(torch.uint8).fill_(byte) if t.is_contiguous() else None
        return t
    monkeypatch.setattr(torch, 'empty', filled)


def test_nothing_reads_memory_it_did_not_write(weights_np, monkeypatch):
    """The forward (volume + both render passes with their debug outputs) and a training render pass with its backward on freshly
    allocated workspaces / outputs / scratch filled with 0x00 and with 0xff bytes (NaNs): the same bits, everything finite -- no kernel
    reads a workspace region, a partial slot or an output array before something wrote it."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    prev = HotPath.default_options
    HotPath.default_options = prev | _lib.OPTIONS['feature_grad_fixed']
    try:
        rn, dn, B = 50, 16, 2
        bref, bque = batch_scenes([make_scene(70 + i, dict(CONFIGS['cfg1'], V=3, rn=rn)) for i in range(B)])
        rng = np.random.default_rng(5)
        depth_np = np.sort(rng.uniform(0.25, 0.75, (B, rn, dn)).astype(np.float32), -1)
        up = {k: rng.standard_normal(s).astype(np.float32) for k, s in (('dvol', (B, 1, 16, 16, 16)), ('ds', (B, rn * dn, 65)), ('dc', (B, rn * dn, 3)))}
        cfg = {'depth_sample_num': dn, 'fine_depth_sample_num': dn, 'ray_mask_view_num': 2, 'ray_mask_point_num': 8}

        def run(byte):
            _fill_allocations(monkeypatch, byte)
            hp = _hot(weights_np)
            out = {}
            prep = hp.prepare(bref, 16, rn, dn)
            out['volume'] = hp.sample_volume(bref, 16, prepared=prep)
            co, fi = hp.render(bref, bque, cfg, debug=True, prepared=prep)[:2]
            out.update({'coarse ' + k: v for k, v in co.items()}); out.update({'fine ' + k: v for k, v in fi.items()})
            hp.sample_volume_train(bref, 16)
            dcan, dray, dimg = hp.sample_volume_bwd(torch.from_numpy(up['dvol']).cuda(), hp.can_dev['coarse'])
            out.update(vol_dcan=dcan, vol_dray=dray, vol_dimg=dimg)
            prep2 = hp.prepare(bref, 1, rn, dn)
            bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
            depth = torch.from_numpy(depth_np).cuda()
            stats, colors, geo, ctx = hp.render_chain_train(bq, depth, 'fine', cfg, prep2)
            fw = hp.render_tail_train(ctx, bq, depth, colors)
            dcan2, dray2, dimg2 = hp.render_chain_bwd(ctx, torch.from_numpy(up['ds']).cuda(), torch.from_numpy(up['dc']).cuda())
            out.update(stats=stats, colors=colors, sdf=fw['sdf_values'], grad=fw['sdf_gradient'], dcan=dcan2, dray=dray2, dimg=dimg2)
            torch.cuda.synchronize()
            assert hp.range_status(prep) == 0 and hp.range_status(prep2) & 7 == 0
            return {k: v.clone() for k, v in out.items() if v is not None}
        a, b = run(0x00), run(0xff)
        for k, v in a.items():
            if v.dtype.is_floating_point:
                assert bool(torch.isfinite(v).all()), k
            assert torch.equal(v, b[k]), k
    finally:
        HotPath.default_options = prev
