"""Parity of the HIP hot path (through the C ABI, libgnr.so) against the CPU oracle and the
golden vectors produced by the imported reference.  Needs a real MI355X:  pytest -m gpu

Tolerances (BASELINE.json north_star): 1e-3 relative on SDF / alpha: |a-b| <= ATOL + RTOL*|b| with RTOL = 1e-3 and
ATOL = 2e-5 on the volume / sdf (measured 3e-6 .. 3e-5), ATOL_A = 6e-5 on alpha-derived quantities (alpha, hit
probability, composites: measured <= 5e-5 where alpha ~ 0, where a relative bound says nothing), 3e-4 on per-sample colours; bit-exact on index-valued outputs
(in-image view masks, ray masks, voxel index map).  Resampling indices (row F1): EQUAL to the reference's except where a
cdf edge provably moved across the sample (per-sample margins in the fixtures), and the resampler itself is checked in
isolation against the oracle on the kernel's own coarse hit probabilities (test_f1_*)."""
import numpy as np
import pytest
import torch

from graspnerf_amd import weights
from graspnerf_amd.synth import make_scene, CONFIGS
from oracle import graspnerf_oracle as O
from conftest import check_resampling_inds, PARITY_LOG

pytestmark = pytest.mark.gpu

RTOL, ATOL, ATOL_A = 1e-3, 2e-5, 6e-5
# per-sample colours are a softmax blend over the views of logits from a 37->16->8->1 MLP on unnormalised features: with the
# seeded-random weights the logits are O(10) and carry the fp32 noise of every layer before them (measured 1.9e-4 at most)
ATOLS = {'sdf_values': ATOL, 'sdf_gradient_error': ATOL, 'colors_nr': 3e-4}          # everything else: ATOL_A
DN = {'cfg1': 16, 'cfg2': 40}


def close(a, b, what, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64).reshape(a.shape)
    assert not np.isnan(a).any(), f'{what}: NaN'
    if a.size == 0:
        return
    err = np.abs(a - b)
    PARITY_LOG.append({'what': what, 'max_abs': float(err.max()), 'max_over_tol': float((err / (atol + rtol * np.abs(b))).max()),
                       'atol': atol, 'rtol': rtol, 'n': int(a.size)})
    assert (err - (atol + rtol * np.abs(b))).max() <= 0, f'{what}: max abs diff {err.max():.3e} exceeds tolerance (atol {atol}, rtol {rtol})'


@pytest.fixture(scope='module')
def hot(weights_np):
    from graspnerf_amd.hotpath import HotPath
    return HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))


@pytest.fixture(scope='module')
def W(weights_np):
    return {k: torch.from_numpy(v) for k, v in weights_np.items()}


def _batched(name, seeds=(0,)):
    from graspnerf_amd.hotpath import batch_scenes
    scenes = [make_scene(s, name) for s in seeds]
    return scenes, batch_scenes(scenes)


@pytest.mark.parametrize('name', ['cfg1', 'cfg2'])
def test_volume_matches_reference_and_oracle(name, hot, W, golden):
    G = golden(name)
    res, V = CONFIGS[name]['res'], CONFIGS[name]['V']
    scenes, (bref, bque) = _batched(name)
    vol, vm = hot.sample_volume(bref, res, want_mask=True)
    torch.cuda.synchronize()
    vol = vol.cpu().numpy()
    assert vol.shape == (1, 1, res, res, res)
    close(vol[0], G['volume'][0], 'volume vs reference golden')
    close(vol[0], O.sample_volume(W, O.to_torch(scenes[0][0]), res).numpy()[0], 'volume vs oracle')
    # index-valued: per-view in-image masks, bit exact vs the reference (golden stores [V, column, top->down])
    gm = np.unpackbits(G['volume_mask_bits']).reshape(V, res * res, res).astype(bool)
    mine = vm.cpu().numpy()[0]
    for v in range(V):
        mv = ((mine >> v) & 1).astype(bool).reshape(res * res, res)[:, ::-1]
        assert np.array_equal(mv, gm[v]), f'view {v} mask differs'


def _f1_check(co, fi, inds, que, fdn, u, what):
    """Row F1 in isolation (render_ops.py:172-229): the kernel's resampler against the oracle's (= the reference's op sequence)
    on the KERNEL'S OWN coarse depths and hit probabilities.  Indices must be equal wherever the sample is further than float
    noise (5e-7: the sum of the pdf normalisation has no defined add order in torch) from a cdf edge; the sorted resampled
    depths must agree to 1e-6 + the sample's own conditioning |d depth / d cdf| x that noise (sorting is 1-Lipschitz in the
    sup norm, so the bound of a ray is the largest bound of its samples).  -> the oracle's details (cdf ...)."""
    det = {}
    depth, hp = co['depth'][0].cpu(), co['hit_prob_nr'][0].cpu()
    fd_o, inds_o = O.sample_fine_depth(depth, hp, torch.from_numpy(np.asarray(que['depth_range'])).reshape(-1)[:2], fdn,
                                       None if u is None else torch.from_numpy(np.asarray(u)), details=det)
    ii = inds.cpu().numpy()[0]
    margin = det['margin'].numpy()
    bad = ii != inds_o.numpy()
    assert not (bad & (margin > 5e-7)).any(), f'{what}: resampling index differs away from a cdf edge'
    assert bad.mean() < 2e-3, f'{what}: {bad.sum()} samples within 5e-7 of a cdf edge?'
    fd = fi['depth'][0].cpu().numpy()
    assert np.all(np.diff(fd, axis=1) >= 0), f'{what}: fine depths not sorted'
    # the 1e-5 guard on the bin width (render_ops.py:221) is a branch: a sample whose raw width sits within noise of it
    # may take the other side in fp32 arithmetic of a different add order -- excluded from the value check, and rare
    guard = (np.abs(det['den_raw'].numpy() - 1e-5) < 3e-7) | bad
    tol = 1e-6 + det['sens'].numpy() * 5e-7
    ray_ok = ~guard.any(1)
    assert ray_ok.mean() > 0.9, f'{what}: {100 * (1 - ray_ok.mean()):.1f} % of the rays touch the bin-width guard'
    err = np.abs(fd - np.sort(fd_o.numpy(), -1))
    worst = (err.max(1) - tol.max(1))[ray_ok]
    if worst.max() > 0:
        r = np.flatnonzero(ray_ok)[int(np.argmax(worst))]
        k = int(np.argmax(err[r] - tol[r].max()))
        raise AssertionError(f'{what}: resampled depth off by {err[r].max():.3e} on ray {r} (bound {tol[r].max():.3e}, sens max {det["sens"].numpy()[r].max():.3e}, '
                             f'sample {k}: err {err[r, k]:.3e}, den_raw {np.sort(det["den_raw"].numpy()[r])[:3]}, margins {np.sort(margin[r])[:3]})')
    return det


@pytest.mark.parametrize('name', ['cfg1', 'cfg2'])
def test_render_matches_reference(name, hot, W, golden):
    G = golden(name)
    dn = DN[name]
    cfg = {'depth_sample_num': dn, 'fine_depth_sample_num': dn}
    scenes, (bref, bque) = _batched(name)
    # fine pass teacher-forced on the reference's resampled depths: tight VALUE parity of the fine level needs identical
    # sample positions; the resampler itself is checked exactly below and in test_f1_*
    co, fi, inds = hot.render(bref, bque, cfg, fine_depth_in=G['fine_depth_sorted'][None], debug=True)
    torch.cuda.synchronize()
    for k in ['sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr', 'pixel_colors_gt',
              'render_depth', 'sdf_gradient_error']:
        close(co[k].cpu().numpy(), G['render.' + k], 'coarse ' + k, atol=ATOLS.get(k, ATOL_A))
        close(fi[k].cpu().numpy(), G['render.' + k + '_fine'], 'fine ' + k, atol=ATOLS.get(k, ATOL_A))
    assert np.array_equal(co['ray_mask'].cpu().numpy(), G['render.ray_mask'])
    assert np.array_equal(fi['ray_mask'].cpu().numpy(), G['render.ray_mask_fine'])
    # resampling indices vs the REFERENCE: equal except where a cdf edge moved across the sample
    det = _f1_check(co, hot.render(bref, bque, cfg)[1], inds, scenes[0][1], dn, None, f'{name} F1')
    n_bad = check_resampling_inds(inds.cpu().numpy()[0], det['cdf'].numpy(), G['fine_inds'], G['fine_cdf'], G['fine_inds_margin'],
                                  f'{name} inds vs reference')
    assert n_bad <= (G['fine_inds_margin'] < 3e-5).sum()
    # SDF gradient (the in-forward VJP) against the oracle's autograd
    dbo = {}
    O.render(W, O.to_torch(scenes[0][0]), O.to_torch(scenes[0][1]), cfg, debug=dbo,
             fine_depth_override=torch.from_numpy(G['fine_depth_sorted']))
    close(co['sdf_gradient'].cpu().numpy()[0], dbo['coarse']['grad'].numpy(), 'coarse sdf gradient', atol=2e-4)
    close(fi['sdf_gradient'].cpu().numpy()[0], dbo['fine']['grad'].numpy(), 'fine sdf gradient', atol=2e-4)


def test_train_mode_render_matches_reference(hot, golden):
    """is_train=True forward: caller-drawn random inverse-CDF samples (GnrRays.fine_u) and per-chunk
    sdf_gradient_error (GnrRays.ray_batch_num = 24 -> 3 chunks of the 64 rays)."""
    G = golden('train_cfg1')
    cfg = {'depth_sample_num': 16, 'fine_depth_sample_num': 16, 'ray_batch_num': int(G['ray_batch_num'])}
    scenes, (bref, bque) = _batched('cfg1')
    bque = dict(bque, fine_u=G['fine_u'][None])
    co, fi, inds = hot.render(bref, bque, cfg, fine_depth_in=G['fine_depth_sorted'][None], debug=True)
    torch.cuda.synchronize()
    for k in ['sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr', 'render_depth',
              'sdf_gradient_error']:
        assert co[k].shape[1:] == G['render.' + k].shape[1:], k
        close(co[k].cpu().numpy(), G['render.' + k], 'coarse ' + k, atol=ATOLS.get(k, ATOL_A))
        close(fi[k].cpu().numpy(), G['render.' + k + '_fine'], 'fine ' + k, atol=ATOLS.get(k, ATOL_A))
    assert co['sdf_gradient_error'].shape == (1, 3)
    # free-running: the kernel's own resampled + sorted depths on the reference's random draws
    co2, fi2, inds2 = hot.render(bref, bque, cfg, debug=True)
    det = _f1_check(co2, fi2, inds2, scenes[0][1], 16, G['fine_u'], 'train-mode F1')
    check_resampling_inds(inds2.cpu().numpy()[0], det['cdf'].numpy(), G['fine_inds'], G['fine_cdf'], G['fine_inds_margin'],
                          'train-mode inds vs reference')


@pytest.mark.parametrize('name', ['cfg1', 'cfg2'])
def test_f1_resampler_exact_on_margin_scenes(name, hot, golden):
    """golden_f1.npz: scenes whose every inverse-CDF sample keeps >= 1e-4 (16^3 case) / >= 1e-6 (40^3 case) from every cdf
    edge of the REFERENCE (tools/make_goldens.py::run_f1).  End to end and free-running (no teacher forcing):
    the kernel's indices are the reference's unless its cdf is further than the margin away, and the resampler is
    checked in isolation on the kernel's own coarse pass."""
    G = {k.split('.', 1)[1]: v for k, v in golden('f1').items() if k.startswith(name + '.')}
    dn = DN[name]
    cfg = {'depth_sample_num': dn, 'fine_depth_sample_num': dn}
    scenes, (bref, bque) = _batched(name, seeds=(int(G['seed']),))
    co, fi, inds = hot.render(bref, bque, cfg, debug=True)
    torch.cuda.synchronize()
    assert np.array_equal(co['depth'][0].cpu().numpy(), G['depth']), 'coarse depths (S1) are bit-exact'
    close(co['hit_prob_nr'][0].cpu().numpy(), G['hit_prob'], 'coarse hit_prob', atol=ATOL_A)
    det = _f1_check(co, fi, inds, scenes[0][1], dn, None, f'f1 {name}')
    dc = np.abs(det['cdf'].numpy() - G['cdf']).max()
    n_bad = check_resampling_inds(inds.cpu().numpy()[0], det['cdf'].numpy(), G['inds'], G['cdf'], G['margin'], f'f1 {name} vs reference')
    if dc < G['margin'].min():
        assert n_bad == 0, f'{name}: cdf within {dc:.2e} of the reference, margin {G["margin"].min():.2e}, yet {n_bad} indices differ'
    if name == 'cfg1':
        assert dc < 1e-4 and n_bad == 0, f'16^3 margin scene: every index must be the reference\'s (cdf off by {dc:.2e})'


def test_free_running_fine_depths(hot, golden):
    """End to end (no teacher forcing) at full size: resampler exact on the kernel's own coarse pass; depths sorted ascending
    and inside the depth range."""
    scenes, (bref, bque) = _batched('cfg2')
    co, fi, inds = hot.render(bref, bque, {}, debug=True)
    torch.cuda.synchronize()
    fd = fi['depth'].cpu().numpy()[0]
    assert fd.min() >= 0.2 - 1e-6 and fd.max() <= 0.8 + 1e-6
    _f1_check(co, fi, inds, scenes[0][1], 40, None, 'cfg2 free-running')


def test_batch_equals_per_scene(hot):
    """Scenes are independent: a batched launch is bitwise the per-scene result."""
    name = 'cfg1'
    res, dn = CONFIGS[name]['res'], DN[name]
    cfg = {'depth_sample_num': dn, 'fine_depth_sample_num': dn}
    scenes, (bref, bque) = _batched(name, seeds=(0, 1, 2))
    vol = hot.sample_volume(bref, res).cpu().numpy()
    co, fi = hot.render(bref, bque, cfg)
    co = {k: v.cpu().numpy() for k, v in co.items()}
    from graspnerf_amd.hotpath import batch_scenes
    for i, sc in enumerate(scenes):
        r1, q1 = batch_scenes([sc])
        v1 = hot.sample_volume(r1, res).cpu().numpy()
        assert np.array_equal(v1[0], vol[i])
        c1, f1 = hot.render(r1, q1, cfg)
        for k in ('sdf_values', 'alpha_values', 'hit_prob_nr', 'render_depth'):
            assert np.array_equal(c1[k].cpu().numpy()[0], co[k][i]), k


def test_full_size_properties(hot):
    """BASELINE config 3 shape (batch of 6-view 40^3 scenes): size-independent invariants."""
    scenes, (bref, bque) = _batched('cfg2', seeds=(3, 4, 5, 6))
    vol = hot.sample_volume(bref, 40).cpu().numpy()
    assert vol.shape == (4, 1, 40, 40, 40) and np.isfinite(vol).all()
    assert vol.min() >= -1.0 and vol.max() <= 1.0                      # clip(-1,1)  ibrnet.py:494
    co, fi = hot.render(bref, bque, {})
    for o in (co, fi):
        a, h = o['alpha_values'].cpu().numpy(), o['hit_prob_nr'].cpu().numpy()
        assert a.min() >= 0 and a.max() <= 1
        assert h.min() >= 0 and h.sum(-1).max() <= 1 + 1e-4           # transmittance partition
        # hit_k = alpha_k * prod_{j<k}(1 - alpha_j + 1e-10)  (render_ops.py:72-80), recomputed in float64
        T = np.cumprod(np.concatenate([np.ones_like(a[..., :1]), 1 - a + 1e-10], -1).astype(np.float64), -1)[..., :-1]
        np.testing.assert_allclose(h, a * T, rtol=1e-4, atol=1e-6)
        z = o['depth'].cpu().numpy()
        np.testing.assert_allclose(o['render_depth'].cpu().numpy(), (h * z).sum(-1), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(o['pixel_colors_nr'].cpu().numpy(), (h[..., None] * o['colors_nr'].cpu().numpy()).sum(-2),
                                   rtol=1e-4, atol=1e-5)


def test_error_codes(hot):
    """The ABI reports bad shapes / tiny workspaces instead of crashing."""
    import ctypes as C
    from graspnerf_amd import _lib
    scenes, (bref, bque) = _batched('cfg1')
    scene, keep, ws = hot.prepare(bref, 16)
    bad = _lib.GnrScene(scene.B, 9, scene.H, scene.W, scene.fh, scene.fw, scene.imgs, scene.img_feats, scene.ray_feats,
                        scene.poses, scene.Ks, scene.depth_range)
    assert hot.L.gnr_prepare(C.byref(bad), ws.data_ptr(), ws.numel(), None) == -2
    assert hot.L.gnr_prepare(C.byref(scene), ws.data_ptr(), 16, None) == -4
    assert hot.L.gnr_prepare(C.byref(scene), None, 0, None) == -1


@pytest.mark.parametrize('V', [2, 4, 5, 7, 8])
def test_other_view_counts(V, hot, W):
    """k_chain is instantiated for 2..8 views; masks include views that see nothing (ragged validity)."""
    from graspnerf_amd.hotpath import batch_scenes
    cfgV = dict(CONFIGS['cfg1'], V=V)
    sc = make_scene(V, cfgV)
    bref, bque = batch_scenes([sc])
    res, dn = 16, 16
    vol = hot.sample_volume(bref, res).cpu().numpy()
    close(vol[0], O.sample_volume(W, O.to_torch(sc[0]), res).numpy()[0], f'V={V} volume vs oracle')
    depth = O.sample_depth(torch.from_numpy(sc[1]['depth_range']), 64, dn)
    o = hot.render_by_depth(bref, bque, depth[None], 'coarse', debug=False)
    ref_o = O.render_by_depth(W, O.to_torch(sc[0]), O.to_torch(sc[1]), depth, 'dist_decoder.', 'agg_net.',
                              O.DEFAULT_RENDER_CFG)
    for k in ('sdf_values', 'alpha_values', 'hit_prob_nr', 'render_depth', 'colors_nr'):
        close(o[k].cpu().numpy(), ref_o[k].numpy(), f'V={V} {k}', atol=ATOLS.get(k, ATOL_A))
    assert np.array_equal(o['ray_mask'].cpu().numpy(), ref_o['ray_mask'].numpy())


def _shifted_scene(seed, V, bbox_shift, res_rn=(16, 64)):
    """cfg1-sized scene whose workspace box is pushed partly out of the cameras' frusta: many points are seen by 0 or 1
    views (n_valid < 1 -> sdf := 1, ibrnet.py:495; n_valid <= 1 -> uniform attention row)."""
    cfgV = dict(CONFIGS['cfg1'], V=V)
    ref, que = make_scene(seed, cfgV)
    ref['bbox3d'] = (ref['bbox3d'] + np.asarray(bbox_shift, np.float32)).astype(np.float32)
    return ref, que


@pytest.mark.parametrize('shift', [(0.35, 0.0, 0.0), (0.0, 0.0, 0.9), (0.6, 0.6, 0.0)])
def test_points_outside_most_views(shift, hot, W):
    from graspnerf_amd.hotpath import batch_scenes
    ref, que = _shifted_scene(11, 3, shift)
    bref, bque = batch_scenes([(ref, que)])
    dbg = {}
    vol_o = O.sample_volume(W, O.to_torch(ref), 16, debug=dbg).numpy()
    nv = dbg['mask'].sum(0).numpy()
    assert (nv == 0).any() or (nv == 1).any(), 'test scene should contain unseen / singly-seen voxels'
    vol, vm = hot.sample_volume(bref, 16, want_mask=True)
    close(vol.cpu().numpy()[0], vol_o[0], f'shifted volume {shift}')
    mine = vm.cpu().numpy()[0]
    for v in range(3):                                             # masks bit-exact vs the oracle
        mv = ((mine >> v) & 1).astype(bool).reshape(256, 16)[:, ::-1]
        assert np.array_equal(mv, dbg['mask'][v].numpy().reshape(256, 16))
    if (nv == 0).any():
        assert np.all(vol.cpu().numpy()[0, 0].reshape(256, 16)[:, ::-1][nv.reshape(256, 16) == 0] == 1.0)


@pytest.mark.parametrize('res,rn,dn', [(10, 3, 17), (12, 5, 3), (24, 2, 64)])
def test_ragged_and_extreme_sizes(res, rn, dn, hot, W):
    """Point counts that are not multiples of the 16-point MFMA tile, the smallest / largest samples-per-ray, and
    grid sizes that disable (res % 8 != 0) or enable the brick-ordered traversal."""
    from graspnerf_amd.hotpath import batch_scenes
    cfgV = dict(CONFIGS['cfg1'], V=4, rn=rn)
    scs = [make_scene(s, cfgV) for s in (21, 22)]
    bref, bque = batch_scenes(scs)
    vol = hot.sample_volume(bref, res).cpu().numpy()
    depth = O.sample_depth(torch.from_numpy(scs[0][1]['depth_range']), rn, dn)
    o = hot.render_by_depth(bref, bque, depth[None].repeat(2, 1, 1), 'fine')
    for i, sc in enumerate(scs):
        close(vol[i], O.sample_volume(W, O.to_torch(sc[0]), res).numpy()[0], f'res {res} scene {i}')
        ref_o = O.render_by_depth(W, O.to_torch(sc[0]), O.to_torch(sc[1]), depth, 'fine_dist_decoder.', 'fine_agg_net.',
                                  O.DEFAULT_RENDER_CFG)
        for k in ('sdf_values', 'alpha_values', 'hit_prob_nr', 'render_depth'):
            close(o[k].cpu().numpy()[i], ref_o[k].numpy()[0], f'rn {rn} dn {dn} scene {i} {k}', atol=ATOLS.get(k, ATOL_A))
        assert np.array_equal(o['ray_mask'].cpu().numpy()[i], ref_o['ray_mask'].numpy()[0])


@pytest.mark.parametrize('seed', range(24))
def test_random_geometry_sweep(seed, hot, W):
    """Randomised view counts, cameras (also close to / inside the workspace), per-view intrinsics and depth ranges,
    image / feature-map sizes without the 1/4 ratio, shifted boxes, fractional and out-of-image ray coordinates.
    Values are compared where the index-valued decisions (in-image masks) are not within float noise of a border."""
    from graspnerf_amd.synth import random_scene
    from graspnerf_amd.hotpath import batch_scenes
    ref, que, m = random_scene(seed)
    bref, bque = batch_scenes([(ref, que)])
    V, res, rn, dn = m['V'], m['res'], m['rn'], m['dn']

    def safe_points(dbg):                                    # points whose every view is >= 1e-3 px away from a border
        uv, z = dbg['uv'].numpy(), dbg['z'].numpy()
        marg = np.minimum.reduce([np.abs(uv[..., 0] + 0.5), np.abs(uv[..., 0] - (m['W'] - 0.5)),
                                  np.abs(uv[..., 1] + 0.5), np.abs(uv[..., 1] - (m['H'] - 0.5))])
        return ((marg > 1e-3) & (np.abs(np.abs(z) - 1e-4) > 1e-6)).all(0)
    dbg = {}
    vol_o = O.sample_volume(W, O.to_torch(ref), res, debug=dbg).numpy()
    vol, vm = hot.sample_volume(bref, res, want_mask=True)
    ok = safe_points(dbg).reshape(res * res, res)
    cols = ok.all(1)                                        # a column's samples interact through the attention
    mine = vol.cpu().numpy()[0, 0].reshape(res * res, res)[:, ::-1]
    close(mine[cols], vol_o[0, 0].reshape(res * res, res)[:, ::-1][cols], f'seed {seed} volume {m}')
    vmn = vm.cpu().numpy()[0].reshape(res * res, res)[:, ::-1]
    for v in range(V):
        assert np.array_equal(((vmn >> v) & 1).astype(bool)[ok], dbg['mask'][v].numpy().reshape(res * res, res)[ok])
    depth = O.sample_depth(torch.from_numpy(que['depth_range']), rn, dn)
    dbr = {}
    ref_o = O.render_by_depth(W, O.to_torch(ref), O.to_torch(que), depth, 'dist_decoder.', 'agg_net.',
                              O.DEFAULT_RENDER_CFG, debug=dbr)
    o = hot.render_by_depth(bref, bque, depth[None], 'coarse')
    # border margins of the ray points from the oracle's projection
    pts, _ = O.ray_points(torch.from_numpy(que['coords']), torch.from_numpy(que['pose']), torch.from_numpy(que['K']), depth)
    uv, z, _, _ = O.project_points(pts.reshape(-1, 3), torch.from_numpy(ref['poses']), torch.from_numpy(ref['Ks']), m['H'], m['W'])
    rays = safe_points({'uv': uv, 'z': z}).reshape(rn, dn).all(1)
    ref_o['pixel_colors_gt'] = O.query_pixel_colors(O.to_torch(que))
    for k in ('sdf_values', 'alpha_values', 'hit_prob_nr', 'render_depth', 'pixel_colors_nr', 'pixel_colors_gt'):
        close(o[k].cpu().numpy()[0][rays], ref_o[k].numpy()[0][rays], f'seed {seed} {k} {m}', atol=ATOLS.get(k, ATOL_A))
    assert np.array_equal(o['ray_mask'].cpu().numpy()[0][rays], ref_o['ray_mask'].numpy()[0][rays])


@pytest.mark.parametrize('scale', [6.0, 40.0, 160.0])
def test_attention_with_peaked_softmax(scale, weights_np):
    """Attention projections scaled up: logits of +-hundreds (one-hot softmax; x160: +-thousands).  The kernel shifts the softmax by the
    Cauchy-Schwarz bound |q||k|/2 and must fall back to the exact row maximum when that bound underflows a whole row or exceeds 2^12
    (round 6: such `exact` rows, and the workgroups that hold one, keep the plain FMA chains in all three sweeps of k_ray)."""
    from graspnerf_amd.hotpath import HotPath
    wn = dict(weights_np)
    for lvl in ('agg_net', 'fine_agg_net'):
        for m in ('w_qs', 'w_ks'):
            k = f'{lvl}.agg_impl.ray_attention.{m}.weight'
            wn[k] = wn[k] * np.float32(scale)
    hp = HotPath(weights.pack_state_dict(wn, 'coarse'), weights.pack_state_dict(wn, 'fine'))
    Wt = {k: torch.from_numpy(v) for k, v in wn.items()}
    scenes, (bref, bque) = _batched('cfg1')
    vol = hp.sample_volume(bref, 16).cpu().numpy()
    assert np.isfinite(vol).all()
    close(vol[0], O.sample_volume(Wt, O.to_torch(scenes[0][0]), 16).numpy()[0], f'volume, attention x{scale}', atol=5e-4)
    cfg = {'depth_sample_num': 16, 'fine_depth_sample_num': 16}
    co, fi = hp.render(bref, bque, cfg)
    ref_o = O.render(Wt, O.to_torch(scenes[0][0]), O.to_torch(scenes[0][1]), cfg, fine_depth_override=fi['depth'][0].cpu())
    for k in ('sdf_values', 'alpha_values', 'hit_prob_nr'):
        a = co[k].cpu().numpy()
        assert np.isfinite(a).all()
        close(a, ref_o[k].numpy(), f'coarse {k}, attention x{scale}', atol=2e-3)


def test_bad_sizes_are_refused(hot):
    from graspnerf_amd import _lib
    scenes, (bref, bque) = _batched('cfg1')
    with pytest.raises(_lib.GnrError):
        hot.sample_volume(bref, 65)                                # volume_res > 64
    with pytest.raises(_lib.GnrError):
        hot.render(bref, bque, {'depth_sample_num': 2, 'fine_depth_sample_num': 16})
    with pytest.raises(_lib.GnrError):
        hot.render(bref, bque, {'depth_sample_num': 16, 'fine_depth_sample_num': 65})
