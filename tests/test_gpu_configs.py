"""BASELINE.json configurations and numerical regimes under test on the MI355X (pytest -m gpu), all through the C ABI:

* configs[2]: batch 32 full-size scenes (6 views 288x512, 40^3, 512 rays) in ONE launch -- scene 0 against the reference
  golden, batched == per-scene bitwise;
* the full-frame render (renderer.py:201-220): 147 456 rays = 36 chunks of ray_batch_num 4096, every output against the
  oracle on a strided subset of rays, per-chunk sdf_gradient_error against the kernel's own per-ray gradients;
* weight regimes a trained checkpoint sits in and seeded-random weights do not: decoder / geometry weights x3 and x0.3
  (saturated tanh cdfs, softplus > 20 linear branch, clip(sdf, +-1) -> zero VJP seed), NeuS variance in
  {-1.4, 0, 0.3, 1.4} (inv_s = exp(10 s) hits both clips 1e-6 / 1e6, neus.py:15-19)."""
import numpy as np
import pytest
import torch

from graspnerf_amd import weights
from graspnerf_amd.synth import make_scene, CONFIGS
from oracle import graspnerf_oracle as O
from conftest import PARITY_LOG
from test_gpu_parity import close, ATOL, ATOL_A, ATOLS

pytestmark = pytest.mark.gpu

VALUE_KEYS = ['sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr', 'render_depth']


@pytest.fixture(scope='module')
def hot(weights_np):
    from graspnerf_amd.hotpath import HotPath
    return HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))


def test_batch32_full_size(hot, golden):
    """BASELINE.json configs[2] as bench.py runs it: 32 scenes per launch."""
    from graspnerf_amd.hotpath import batch_scenes
    G = golden('cfg2')
    scenes = [make_scene(i, 'cfg2') for i in range(32)]
    bref, bque = batch_scenes(scenes)
    dev = hot.device
    bref = {k: torch.from_numpy(v).to(dev) for k, v in bref.items()}
    bque = {k: torch.from_numpy(v).to(dev) for k, v in bque.items()}
    prep = hot.prepare(bref, 40, 512, 40)
    vol, vm = hot.sample_volume(bref, 40, want_mask=True, prepared=prep)
    co, fi = hot.render(bref, bque, prepared=prep)
    torch.cuda.synchronize()
    assert vol.shape == (32, 1, 40, 40, 40)
    close(vol[0].cpu().numpy(), G['volume'][0], 'B=32 scene 0 volume vs reference golden')
    gm = np.unpackbits(G['volume_mask_bits']).reshape(6, 1600, 40).astype(bool)
    mine = vm[0].cpu().numpy()
    for v in range(6):
        assert np.array_equal(((mine >> v) & 1).astype(bool).reshape(1600, 40)[:, ::-1], gm[v]), f'view {v} mask'
    for k in VALUE_KEYS + ['sdf_gradient_error']:
        close(co[k][0].cpu().numpy(), G['render.' + k][0], f'B=32 scene 0 coarse {k}', atol=ATOLS.get(k, ATOL_A))
    assert np.array_equal(co['ray_mask'][0].cpu().numpy(), G['render.ray_mask'][0])
    assert np.array_equal(fi['ray_mask'][0].cpu().numpy(), G['render.ray_mask_fine'][0])
    # scenes are independent: the batched launch is bitwise the per-scene launch, volume and both render levels
    for i in (0, 13, 31):
        r1 = {k: v[i:i + 1].contiguous() for k, v in bref.items()}
        q1 = {k: v[i:i + 1].contiguous() for k, v in bque.items()}
        v1 = hot.sample_volume(r1, 40)
        c1, f1 = hot.render(r1, q1)
        assert torch.equal(v1[0], vol[i]), f'scene {i} volume'
        for k in VALUE_KEYS + ['depth', 'ray_mask']:
            assert torch.equal(c1[k][0], co[k][i]), f'scene {i} coarse {k}'
            assert torch.equal(f1[k][0], fi[k][i]), f'scene {i} fine {k}'


def test_full_frame_render_vs_oracle(hot, weights_np):
    """renderer.py:201-220 at the planner's size: every pixel of the 288x512 query view, ray_batch_num 4096 -> 36 chunks."""
    from graspnerf_amd.hotpath import batch_scenes
    W = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    ref, que = make_scene(0, 'cfg2')
    H, Wd = ref['imgs'].shape[-2:]
    ys, xs = np.meshgrid(np.arange(H), np.arange(Wd), indexing='ij')
    que = dict(que, coords=np.stack([xs, ys], -1).reshape(-1, 2).astype(np.float32))
    rn = H * Wd
    bref, bque = batch_scenes([(ref, que)])
    cfg = {'ray_batch_num': 4096}
    co, fi, inds = hot.render(bref, bque, cfg, debug=True)
    torch.cuda.synchronize()
    nch = (rn + 4095) // 4096
    assert co['sdf_gradient_error'].shape == (1, nch) and fi['sdf_gradient_error'].shape == (1, nch)
    for o, lvl in ((co, 'coarse'), (fi, 'fine')):
        g = o['sdf_gradient'][0].cpu().numpy().astype(np.float64)
        per_ray = ((np.linalg.norm(g, axis=-1) - 1.0) ** 2).mean(-1)
        chunk_mean = np.array([per_ray[c * 4096:(c + 1) * 4096].mean() for c in range(nch)])
        np.testing.assert_allclose(o['sdf_gradient_error'][0].cpu().numpy(), chunk_mean, rtol=2e-5, err_msg=f'{lvl} per-chunk eikonal term')
        assert np.isfinite(o['sdf_values'].cpu().numpy()).all()
    sel = np.arange(0, rn, 577)                                  # 256 rays spread over all 36 chunks
    assert len(np.unique(sel // 4096)) == nch
    sque = dict(que, coords=que['coords'][sel])
    dbg = {}
    ref_o = O.render(W, O.to_torch(ref), O.to_torch(sque), {}, debug=dbg, fine_depth_override=fi['depth'][0, sel].cpu())
    for k in VALUE_KEYS + ['pixel_colors_gt']:
        close(co[k][0, sel].cpu().numpy(), ref_o[k].numpy()[0], f'full frame coarse {k}', atol=ATOLS.get(k, ATOL_A))
        close(fi[k][0, sel].cpu().numpy(), ref_o[k + '_fine'].numpy()[0], f'full frame fine {k}', atol=ATOLS.get(k, ATOL_A))
    close(co['sdf_gradient'][0, sel].cpu().numpy(), dbg['coarse']['grad'].numpy(), 'full frame coarse sdf gradient', atol=2e-4)
    close(fi['sdf_gradient'][0, sel].cpu().numpy(), dbg['fine']['grad'].numpy(), 'full frame fine sdf gradient', atol=2e-4)
    assert np.array_equal(co['ray_mask'][0, sel].cpu().numpy(), ref_o['ray_mask'].numpy()[0])
    assert np.array_equal(fi['ray_mask'][0, sel].cpu().numpy(), ref_o['ray_mask_fine'].numpy()[0])
    # the resampler on the kernel's own coarse pass, for the subset
    fd_o, inds_o = O.sample_fine_depth(co['depth'][0, sel].cpu(), co['hit_prob_nr'][0, sel].cpu(), torch.from_numpy(que['depth_range']), 40,
                                       details=(det := {}))
    bad = inds[0, sel].cpu().numpy() != inds_o.numpy()
    assert not (bad & (det['margin'].numpy() > 5e-7)).any()


def _scaled(weights_np, scale=None, variance=None):
    wn = {k: v.copy() for k, v in weights_np.items()}
    if scale is not None:
        for k in wn:
            if k.endswith('.weight') and any(s in k for s in ('mean_decoder', 'var_decoder', 'aw_decoder', 'geometry_fc', 'out_geometry_fc')):
                wn[k] = wn[k] * np.float32(scale)
    if variance is not None:
        for k in wn:
            if k.endswith('deviation_network.variance'):
                wn[k] = np.asarray(variance, np.float32).reshape(wn[k].shape)
    return wn


@pytest.mark.parametrize('scale,variance', [(3.0, None), (0.3, None), (None, -1.4), (None, 0.0), (None, 0.3), (None, 1.4), (3.0, 1.4)])
def test_weight_regimes(scale, variance, weights_np):
    """Numerical regimes of trained checkpoints (dist_decoder.py:99-142, aggregate_net.py:105-121, neus.py:15-19, ibrnet.py:494)."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    wn = _scaled(weights_np, scale, variance)
    hp = HotPath(weights.pack_state_dict(wn, 'coarse'), weights.pack_state_dict(wn, 'fine'))
    Wt = {k: torch.from_numpy(v) for k, v in wn.items()}
    ref, que = make_scene(0, 'cfg1')
    bref, bque = batch_scenes([(ref, que)])
    tag = f'regime x{scale} s={variance}'
    vol = hp.sample_volume(bref, 16).cpu().numpy()
    vol_o = O.sample_volume(Wt, O.to_torch(ref), 16).numpy()
    if scale == 3.0:
        assert (np.abs(vol_o) == 1.0).mean() > 0.02, 'regime should saturate clip(sdf, -1, 1)'
    # |sdf| == 1 is a branch (clip): a pre-clip value within noise of +-1 may land on either side; values are compared all the same
    close(vol[0], vol_o[0], f'{tag} volume', atol=ATOL * (3 if scale == 3.0 else 1))
    cfg = {'depth_sample_num': 16, 'fine_depth_sample_num': 16}
    co, fi = hp.render(bref, bque, cfg, debug=False)
    dbg = {}
    ref_o = O.render(Wt, O.to_torch(ref), O.to_torch(que), cfg, debug=dbg, fine_depth_override=fi['depth'][0].cpu())
    for k in VALUE_KEYS:
        a = co[k].cpu().numpy()
        assert np.isfinite(a).all() and np.isfinite(fi[k].cpu().numpy()).all(), (tag, k)
        m = 3 if scale == 3.0 else 1
        close(a, ref_o[k].numpy(), f'{tag} coarse {k}', atol=m * ATOLS.get(k, ATOL_A))
        close(fi[k].cpu().numpy(), ref_o[k + '_fine'].numpy(), f'{tag} fine {k}', atol=m * ATOLS.get(k, ATOL_A))
    if variance is not None and abs(variance) == 1.4:                     # both clips of neus.py:17 are hit
        inv_s = float(torch.exp(torch.tensor(variance * 10.0)).clip(1e-6, 1e6))
        assert inv_s == float(torch.tensor(1e-6 if variance < 0 else 1e6))


@pytest.mark.parametrize('fscale', [1e-3, 30.0])
def test_feature_magnitude_regimes(fscale, weights_np):
    """The chain multiplies fp32 operands carried as fp16 pairs.  Feature maps far from unit scale exercise the
    ends of that representation: at 1e-3 the first-layer operands and their residuals sit in / near the fp16 subnormal range
    (the pair's 2^11 scaling and the MFMA's subnormal support keep them to 1 ulp), at 30 the activations reach the hundreds
    (fp16 range 65 504).  Tolerances relative to the oracle on the same scaled inputs.  Measured, pair build vs the fp32-MFMA
    build (-DGNR_SPLIT16=0) against the same oracle: x1e-3 volume 7.0e-7 vs 7.6e-7, coarse sdf 2.9e-6 vs 3.1e-6; x30 volume
    6.9e-4 vs 6.4e-4, fine sdf 9.0e-4 vs 9.6e-4 -- the two multiply paths are indistinguishable over 4.5 decades of scale."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    Wt = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    ref, que = make_scene(0, 'cfg1')
    ref = dict(ref, ray_feats=(ref['ray_feats'] * np.float32(fscale)), img_feats=(ref['img_feats'] * np.float32(fscale)))
    bref, bque = batch_scenes([(ref, que)])
    tag = f'features x{fscale}'
    vol = hp.sample_volume(bref, 16).cpu().numpy()
    vol_o = O.sample_volume(Wt, O.to_torch(ref), 16).numpy()
    assert np.isfinite(vol).all()
    # x30 inputs make the network 30x steeper in its first layers: the rounding differences between ANY two fp32 implementations
    # grow alike (the fp32-MFMA build of -DGNR_SPLIT16=0 shows the same 7e-4 on the volume against the CPU oracle)
    m = 30 if fscale > 1 else 1
    close(vol[0], vol_o[0], f'{tag} volume', atol=m * ATOL)
    cfg = {'depth_sample_num': 16, 'fine_depth_sample_num': 16}
    co, fi = hp.render(bref, bque, cfg, debug=False)
    ref_o = O.render(Wt, O.to_torch(ref), O.to_torch(que), cfg, debug={}, fine_depth_override=fi['depth'][0].cpu())
    for k in VALUE_KEYS:
        a = co[k].cpu().numpy()
        assert np.isfinite(a).all() and np.isfinite(fi[k].cpu().numpy()).all(), (tag, k)
        close(a, ref_o[k].numpy(), f'{tag} coarse {k}', atol=m * ATOLS.get(k, ATOL_A))
        close(fi[k].cpu().numpy(), ref_o[k + '_fine'].numpy(), f'{tag} fine {k}', atol=m * ATOLS.get(k, ATOL_A))


def test_fine_depth_use_all_matches_reference(weights_np, golden):
    """cfg `fine_depth_use_all: true` (renderer.py:145-146): the fine pass renders sort(cat(coarse, resampled depths)), 16 + 16 = 32
    samples at cfg1, against the reference's own outputs (golden_cfg1_use_all.npz, tools/make_goldens.py run_use_all).
    Free-running: the merged depths contain the coarse ones bit for bit and are sorted; teacher-forced on the reference's
    depths: every fine key within the tolerances of the other parity tests."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    G = golden('cfg1_use_all')
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    ref, que = make_scene(0, 'cfg1')
    bref, bque = batch_scenes([(ref, que)])
    cfg = {'depth_sample_num': 16, 'fine_depth_sample_num': 16, 'fine_depth_use_all': True}
    co, fi = hp.render(bref, bque, cfg)
    fd, cd = fi['depth'][0].cpu().numpy(), co['depth'][0].cpu().numpy()
    assert fd.shape == (64, 32) and np.all(np.diff(fd, axis=1) >= 0)
    assert all(np.isin(cd[r], fd[r]).all() for r in range(64))
    assert np.mean(np.abs(fd - G['fine_depth_sorted']) > 1e-3) < 0.02           # resampling is ill-conditioned where the pdf is ~0
    for k in VALUE_KEYS:
        close(co[k].cpu().numpy(), G['render.' + k], f'use_all coarse {k}', atol=ATOLS.get(k, ATOL_A))
    co2, fi2 = hp.render(bref, bque, cfg, fine_depth_in=G['fine_depth_sorted'][None])
    for k in VALUE_KEYS + ['pixel_colors_nr', 'render_depth']:
        close(fi2[k].cpu().numpy(), G['render.' + k + '_fine'], f'use_all fine {k}', atol=ATOLS.get(k, ATOL_A))
    assert np.array_equal(fi2['ray_mask'].cpu().numpy(), G['render.ray_mask_fine'])


def test_fine_depth_use_all_full_size(weights_np):
    """40 + 40 = 80 samples per ray in the fine pass (beyond the 64 the resampler / backward twins hold; the forward render
    pass takes up to 128) against the oracle, B = 2 scenes, plus the model mirror's config handling."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    from graspnerf_amd.renderer import NeuralRayRenderer
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    Wt = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    scenes = [make_scene(i, 'cfg2') for i in range(2)]
    for s in scenes:
        s[1]['coords'] = s[1]['coords'][:96]
    bref, bque = batch_scenes(scenes)
    cfg = {'depth_sample_num': 40, 'fine_depth_sample_num': 40, 'fine_depth_use_all': True}
    co, fi = hp.render(bref, bque, cfg)
    assert fi['sdf_values'].shape == (2, 96, 80)
    for b, (ref, que) in enumerate(scenes):
        o = O.render(Wt, O.to_torch(ref), O.to_torch(que), cfg, fine_depth_override=fi['depth'][b].cpu())
        for k in VALUE_KEYS + ['pixel_colors_nr', 'render_depth']:
            close(fi[k][b].cpu().numpy(), o[k + '_fine'].numpy()[0], f'use_all 80 scene {b} fine {k}', atol=ATOLS.get(k, ATOL_A))
        assert np.array_equal(fi['ray_mask'][b].cpu().numpy(), o['ray_mask_fine'].numpy()[0])
    base = {'network': 'grasp_nerf', 'init_net_type': 'cost_volume', 'agg_net_type': 'neus', 'use_hierarchical_sampling': True,
            'fine_depth_use_all': True, 'depth_sample_num': 40, 'fine_depth_sample_num': 40, 'volume_type': ['sdf'],
            'agg_net_cfg': {'sample_num': 40}, 'dist_decoder_cfg': {'use_vis': False}, 'fine_dist_decoder_cfg': {'use_vis': False}}
    with pytest.raises(ValueError):
        NeuralRayRenderer({**base, 'fine_agg_net_cfg': {'sample_num': 40}})
    net = NeuralRayRenderer({**base, 'fine_agg_net_cfg': {'sample_num': 80}})
    assert net._dn_max() == 80


def test_use_vis_matches_reference(weights_np, golden):
    """cfg `use_vis: true` on both decoders (dist_decoder.py:89-97,103-104,133-134; the reference's own default, switched off by
    nrvgn_sdf.yaml): a fourth decoder branch whose sigmoid output multiplies both cdfs.  Volume and render against the
    reference's outputs (golden_cfg1_use_vis.npz carries the twelve vis_decoder tensors); the k_chain<.., USEVIS> kernels."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    G = golden('cfg1_use_vis')
    wn = {**weights_np, **{k[len('weights.'):]: v for k, v in G.items() if k.startswith('weights.')}}
    hp = HotPath(weights.pack_state_dict(wn, 'coarse'), weights.pack_state_dict(wn, 'fine'))
    assert hp.use_vis
    ref, que = make_scene(0, 'cfg1')
    bref, bque = batch_scenes([(ref, que)])
    vol = hp.sample_volume(bref, 16).cpu().numpy()
    close(vol[0], G['volume'][0], 'use_vis volume vs reference golden')
    plain = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    assert not plain.use_vis and np.abs(plain.sample_volume(bref, 16).cpu().numpy() - G['volume']).max() > 1e-3, 'the branch must matter'
    cfg = {'depth_sample_num': 16, 'fine_depth_sample_num': 16}
    co, fi = hp.render(bref, bque, cfg, fine_depth_in=G['fine_depth_sorted'][None])
    for k in VALUE_KEYS:
        close(co[k].cpu().numpy(), G['render.' + k], f'use_vis coarse {k}', atol=ATOLS.get(k, ATOL_A))
        close(fi[k].cpu().numpy(), G['render.' + k + '_fine'], f'use_vis fine {k}', atol=ATOLS.get(k, ATOL_A))
    assert np.array_equal(co['ray_mask'].cpu().numpy(), G['render.ray_mask'])
    with pytest.raises(Exception):                                           # levels must agree
        HotPath(weights.pack_state_dict(wn, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))


def test_use_vis_model_mirror(weights_np, golden):
    """NeuralRayRenderer with use_vis: the state dict gains the reference's twelve vis_decoder keys, sample_volume through the
    mirror equals the reference's volume, and training is on (its gradients: tests/test_train_step.py, fixture vis)."""
    from graspnerf_amd.renderer import NeuralRayRenderer
    G = golden('cfg1_use_vis')
    base = {'network': 'grasp_nerf', 'init_net_type': 'cost_volume', 'agg_net_type': 'neus', 'use_hierarchical_sampling': True,
            'volume_type': ['sdf'], 'volume_resolution': 16, 'depth_sample_num': 16, 'fine_depth_sample_num': 16,
            'agg_net_cfg': {'sample_num': 16}, 'fine_agg_net_cfg': {'sample_num': 16}}
    with pytest.raises(NotImplementedError):
        NeuralRayRenderer({**base, 'dist_decoder_cfg': {'use_vis': True}, 'fine_dist_decoder_cfg': {'use_vis': False}})
    net = NeuralRayRenderer({**base, 'dist_decoder_cfg': {'use_vis': True}, 'fine_dist_decoder_cfg': {'use_vis': True}})
    keys = set(net.state_dict())
    vis = {k[len('weights.'):] for k in G if k.startswith('weights.')}
    assert len(vis) == 12 and vis <= keys
    sd = {k: torch.from_numpy(v) for k, v in {**weights_np, **{k[len('weights.'):]: v for k, v in G.items() if k.startswith('weights.')}}.items()}
    missing = net.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if 'decoder' in k or 'agg_net' in k] and not missing.unexpected_keys
    net = net.cuda().eval()
    ref, _ = make_scene(0, 'cfg1')
    info = {k: torch.from_numpy(v).cuda() for k, v in ref.items()}
    with torch.no_grad():
        vol = net.sample_volume(info).cpu().numpy()
    close(vol, G['volume'], 'use_vis volume through the model mirror')
    assert net._use_autograd(True)                         # trains too: tests/test_train_step.py fixture 'vis'


@pytest.mark.gpu
@pytest.mark.parametrize('cfg_name,extra', [('cfg1', {}), ('cfg2', {}), ('cfg1', {'fine_depth_use_all': True}), ('cfg1', {'ray_batch_num': 24})])
def test_ray_traversal_order_is_invisible(cfg_name, extra, weights_np):
    """GNR_OPT_RAY_ORDER_MORTON: the inference render passes lay their internal per-point arrays out in the Morton order of the rays'
    pixels (k_ray_order) and write every output through the permutation -- all outputs, both levels, the resampling indices and the
    per-chunk eikonal terms equal bit for bit to the caller-order run."""
    from graspnerf_amd import _lib
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    L = _lib.lib()
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    bref, bque = batch_scenes([make_scene(s, cfg_name) for s in (0, 1, 2)])
    n = 16 if cfg_name == 'cfg1' else 40
    cfg = dict({'depth_sample_num': n, 'fine_depth_sample_num': n}, **extra)
    outs = []
    prev = hp.set_option('ray_order_morton', False)
    try:
        for on in (0, 1):
            hp.set_option('ray_order_morton', bool(on))
            outs.append(hp.render(bref, bque, cfg, debug=True))
    finally:
        hp.set_option('ray_order_morton', prev)
    torch.cuda.synchronize()
    for lvl in (0, 1):
        for k in outs[0][lvl]:
            assert torch.equal(outs[0][lvl][k], outs[1][lvl][k]), (lvl, k)
    assert torch.equal(outs[0][2], outs[1][2])


@pytest.mark.gpu
def test_a_view_that_sees_nothing(weights_np):
    """A reference view whose principal point lies 10 000 pixels outside its image (every projection out of bounds: mask 0 everywhere,
    render_ops.py:24-31; a camera that merely looks away would not do, :101 keeps points BEHIND a camera valid) at the benchmark's scene
    size: on the ray points all its (tile, view) pairs take k_chain's skip, on the volume points the masked arithmetic.  Outputs against
    the oracle on the same seven views (volume, both render levels on a subset of rays, masks bit-exact) -- and NOT equal to the six-view
    run: ibrnet.py:488 feeds weight.mean over ALL views (1 / V) into geometry_fc, the count of views is an input of the network."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    W = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    ref, que = make_scene(3, 'cfg2')
    r7 = {k: v.copy() for k, v in ref.items()}
    for k in ('imgs', 'img_feats', 'ray_feats', 'Ks', 'depth_range', 'poses'):
        r7[k] = np.concatenate([ref[k], ref[k][:1]], 0)
    r7['Ks'][6, 0, 2] = r7['Ks'][6, 1, 2] = -1.0e4
    res = 24                                                           # the oracle's volume on the CPU: 24^3 x 7 views in a few seconds
    bref, bque = batch_scenes([(r7, que)])
    vol, vm = hp.sample_volume(bref, res, want_mask=True)
    co, fi, inds = hp.render(bref, bque, {'depth_sample_num': 40, 'fine_depth_sample_num': 40}, debug=True)
    b6, q6 = batch_scenes([(ref, que)])
    vol6 = hp.sample_volume(b6, res)
    torch.cuda.synchronize()
    assert int((vm >> 6).max()) == 0                                   # the seventh view's mask bit is never set
    assert float((vol - vol6).abs().max()) > 1e-3                      # ... and yet the view counts: 1 / V (ibrnet.py:488)
    close(vol.cpu().numpy()[0], O.sample_volume(W, O.to_torch(r7), res).numpy()[0], 'volume with an unseeing seventh view')
    sel = np.arange(0, que['coords'].shape[0], 4)                      # 128 of the 512 rays
    sque = dict(que, coords=que['coords'][sel])
    ref_o = O.render(W, O.to_torch(r7), O.to_torch(sque), {}, fine_depth_override=fi['depth'][0, sel].cpu())
    for k in VALUE_KEYS:
        close(co[k][0, sel].cpu().numpy(), ref_o[k].numpy()[0], f'unseeing view, coarse {k}', atol=ATOLS.get(k, ATOL_A))
        close(fi[k][0, sel].cpu().numpy(), ref_o[k + '_fine'].numpy()[0], f'unseeing view, fine {k}', atol=ATOLS.get(k, ATOL_A))
    assert np.array_equal(co['ray_mask'][0, sel].cpu().numpy(), ref_o['ray_mask'].numpy()[0])
    assert np.array_equal(fi['ray_mask'][0, sel].cpu().numpy(), ref_o['ray_mask_fine'].numpy()[0])
