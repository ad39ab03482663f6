"""Generate golden vectors by importing and running the REFERENCE (/root/reference) on CPU.

Runs only in the build container (the reference cannot travel to the GPU box).  Emits
  tests/golden/weights_seed0.npz   hot-path parameters (reference state-dict keys)
  tests/golden/golden_cfg1.npz     V=3, 16^3, 96x128: reference outputs (+ sha256 of inputs)
  tests/golden/golden_cfg2.npz     V=6, 40^3, 288x512, 512 rays: reference outputs
Inputs are NOT stored: tests regenerate them from graspnerf_amd.synth.make_scene(seed) and
check the recorded SHA-256.

Weights: `torch.manual_seed(0)` construction of the reference NeuralRayRenderer, then every
hot-path bias / LayerNorm parameter gets + 0.1*N(0,1) (the reference zero-initialises most
biases, ibrnet.py:104-109, which would leave bias handling untested).

Usage:  python tools/make_goldens.py
"""
import hashlib
import os
import sys

import numpy as np
import torch
import yaml

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import import_reference, REF  # noqa: E402
from graspnerf_amd.synth import make_scene, CONFIGS, synth_state_dict, synth_loss_case  # noqa: E402

HOT = ('dist_decoder.', 'fine_dist_decoder.', 'agg_net.', 'fine_agg_net.')


def sha(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def build_net(renderer, res, dn):
    cfg = yaml.load(open(REF + '/src/nr/configs/nrvgn_sdf.yaml'), Loader=yaml.FullLoader)
    cfg['volume_resolution'] = res
    cfg['depth_sample_num'] = dn
    cfg['fine_depth_sample_num'] = dn
    cfg['agg_net_cfg']['sample_num'] = dn
    cfg['fine_agg_net_cfg']['sample_num'] = dn
    torch.manual_seed(0)
    net = renderer.NeuralRayRenderer(cfg)
    net.eval()
    return net, cfg


def perturb_and_export(net):
    g = torch.Generator().manual_seed(1234)
    out = {}
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if not k.startswith(HOT):
                continue
            if k.endswith('.bias') or 'layer_norm' in k:
                v.add_(0.1 * torch.randn(v.shape, generator=g))
            out[k] = v.detach().clone().numpy()
    return out


def load_weights(net, weights):
    sd = net.state_dict()
    for k, v in weights.items():
        sd[k].copy_(torch.from_numpy(v))


def run_reference(renderer, net, cfg, ref, que, res):
    """-> dict of numpy outputs of the reference sample_volume + render (eval mode)."""
    import network.render_ops as rops
    import utils.field_utils as fu
    # 16^3 support: the grid is a module constant (field_utils.py:12,27)
    if res != 40:
        fu.RESOLUTION = res
        fu.VOXEL_SIZE = fu.VOLUME_SIZE / res
        fu.HALF_VOXEL_SIZE = fu.VOXEL_SIZE / 2
        renderer.TSDF_SAMPLE_POINTS = fu.generate_grid_points()
    else:
        fu.RESOLUTION = 40
        fu.VOXEL_SIZE = fu.VOLUME_SIZE / 40
        fu.HALF_VOXEL_SIZE = fu.VOXEL_SIZE / 2
        renderer.TSDF_SAMPLE_POINTS = fu.generate_grid_points()
    t = lambda a: torch.from_numpy(a.copy())
    ref_info = {k: t(v) for k, v in ref.items()}
    que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None],
                'Ks': t(que['K'])[None], 'depth_range': t(que['depth_range'])[None],
                'imgs': t(que['imgs'])}
    out = {}
    with torch.no_grad():
        vol = net.sample_volume(ref_info)
    out['volume'] = vol.numpy()
    # volume-path masks (bit-exact check): re-run the reference projection on the same points
    pts = (torch.from_numpy(renderer.TSDF_SAMPLE_POINTS) + ref_info['bbox3d'][0]).reshape(1, res * res, res, 3)
    pts = torch.flip(pts, (2,))
    prj = rops.project_points_dict(ref_info, pts)
    out['volume_mask_bits'] = np.packbits(prj['mask'].numpy().astype(bool).reshape(-1))
    uv = prj['pts'].numpy().reshape(ref['poses'].shape[0], -1, 2)
    h, w = ref['imgs'].shape[-2:]
    marg = np.minimum.reduce([np.abs(uv[..., 0] + 0.5), np.abs(uv[..., 0] - (w - 0.5)),
                              np.abs(uv[..., 1] + 0.5), np.abs(uv[..., 1] - (h - 0.5))])
    out['volume_mask_min_margin_px'] = np.float32(marg.min())

    # render (coarse + fine), capturing searchsorted indices and the fine depths
    captured = {}
    orig_ss = torch.searchsorted
    orig_sort = torch.sort

    def ss(cdf, u, **kw):
        r = orig_ss(cdf, u, **kw)
        captured['inds'] = r.clone()
        captured['cdf'] = cdf.clone()
        captured['u'] = u.clone()
        return r

    def srt(x, *a, **kw):
        r = orig_sort(x, *a, **kw)
        captured['fine_depth_sorted'] = r[0].clone()
        return r
    torch.searchsorted = ss
    torch.sort = srt
    try:
        with torch.no_grad():
            rend = net.render(que_info, ref_info, False)
    finally:
        torch.searchsorted = orig_ss
        torch.sort = orig_sort
    for k, v in rend.items():
        out['render.' + k] = v.numpy()
    out['fine_inds'] = captured['inds'].numpy()[0].astype(np.int64)
    out['fine_depth_sorted'] = captured['fine_depth_sorted'].numpy()[0]
    cdf, u = captured['cdf'].numpy()[0], captured['u'].numpy()[0]
    out['fine_inds_margin'] = inds_margin(cdf, u)          # per sample: min_j |u - cdf_j|
    out['fine_cdf'] = cdf
    out['fine_inds_min_margin'] = np.float32(out['fine_inds_margin'].min())
    return out


def inds_margin(cdf, u):
    """cdf [rn,dn+1], u [rn,fdn] -> [rn,fdn] float32: distance of every inverse-CDF sample to the nearest cdf edge.  A
    searchsorted index can only differ between two implementations where this is below their cdf difference."""
    return np.abs(u[:, :, None].astype(np.float64) - cdf[:, None, :].astype(np.float64)).min(-1).astype(np.float32)


def run_train_mode(renderer, weights):
    """render(..., is_train=True) of the reference on cfg1 with ray_batch_num=24 (64 rays -> chunks 24/24/16):
    pins the random inverse-CDF sampling (torch.rand on the CPU generator, render_ops.py:204-205) and the
    per-chunk outputs `sdf_gradient_error` / `s` of shape [1,n_chunks] (renderer.py:203-218)."""
    import network.render_ops as rops   # noqa: F401
    net, cfg = build_net(renderer, 16, 16)
    load_weights(net, weights)
    net.cfg['ray_batch_num'] = 24
    ref, que = make_scene(0, 'cfg1')
    t = lambda a: torch.from_numpy(a.copy())
    ref_info = {k: t(v) for k, v in ref.items()}
    que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
                'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    cap = {'u': [], 'inds': [], 'fds': []}
    orig_ss, orig_sort = torch.searchsorted, torch.sort

    def ss(cdf, u, **kw):
        r = orig_ss(cdf, u, **kw)
        cap['u'].append(u.clone()); cap['inds'].append(r.clone()); cap.setdefault('cdf', []).append(cdf.clone())
        return r

    def srt(x, *a, **kw):
        r = orig_sort(x, *a, **kw)
        cap['fds'].append(r[0].clone())
        return r
    torch.searchsorted, torch.sort = ss, srt
    try:
        torch.manual_seed(7)
        with torch.no_grad():
            rend = net.render(que_info, ref_info, True)
    finally:
        torch.searchsorted, torch.sort = orig_ss, orig_sort
    out = {'render.' + k: v.numpy() for k, v in rend.items()}
    out['fine_u'] = torch.cat(cap['u'], 1).numpy()[0]
    out['fine_inds'] = torch.cat(cap['inds'], 1).numpy()[0].astype(np.int64)
    out['fine_depth_sorted'] = torch.cat(cap['fds'], 1).numpy()[0]
    out['fine_cdf'] = torch.cat(cap['cdf'], 1).numpy()[0]
    out['fine_inds_margin'] = inds_margin(out['fine_cdf'], out['fine_u'])
    out['seed'] = np.int64(7)
    out['ray_batch_num'] = np.int64(24)
    np.savez_compressed(ROOT + '/tests/golden/golden_train_cfg1.npz', **out)
    for k, v in out.items():
        print('   ', k, getattr(v, 'shape', None), getattr(v, 'dtype', None))


def run_use_all(renderer, weights):
    """render() of the reference with `fine_depth_use_all: true` (renderer.py:145-146: the fine pass renders the coarse and the
    resampled depths of a ray together, dn + fdn samples; the fine aggregation net's positional table is built for that
    length: fine_agg_net_cfg.sample_num = dn + fdn) on cfg1 (16 + 16 = 32 samples) -> golden_cfg1_use_all.npz."""
    import network.render_ops as rops   # noqa: F401
    cfg = yaml.load(open(REF + '/src/nr/configs/nrvgn_sdf.yaml'), Loader=yaml.FullLoader)
    cfg.update(volume_resolution=16, depth_sample_num=16, fine_depth_sample_num=16, fine_depth_use_all=True)
    cfg['agg_net_cfg']['sample_num'] = 16
    cfg['fine_agg_net_cfg']['sample_num'] = 32
    torch.manual_seed(0)
    net = renderer.NeuralRayRenderer(cfg)
    net.eval()
    load_weights(net, weights)
    ref, que = make_scene(0, 'cfg1')
    t = lambda a: torch.from_numpy(a.copy())
    ref_info = {k: t(v) for k, v in ref.items()}
    que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
                'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    cap = {}
    orig_sort = torch.sort

    def srt(x, *a, **kw):
        r = orig_sort(x, *a, **kw)
        cap['sorted'] = r[0].clone()
        return r
    torch.sort = srt
    try:
        with torch.no_grad():
            rend = net.render(que_info, ref_info, False)
    finally:
        torch.sort = orig_sort
    out = {'render.' + k: v.numpy() for k, v in rend.items()}
    out['fine_depth_sorted'] = cap['sorted'].numpy()[0]                # [rn, 32]
    np.savez_compressed(ROOT + '/tests/golden/golden_cfg1_use_all.npz', **out)
    for k, v in out.items():
        print('   ', k, getattr(v, 'shape', None), getattr(v, 'dtype', None))


def run_use_vis(renderer, weights):
    """sample_volume + render of the reference with `use_vis: true` on both levels (dist_decoder.py:89-97,103-104,133-134: a
    fourth decoder branch whose sigmoid output multiplies both cdfs) on cfg1 -> golden_cfg1_use_vis.npz, which also carries
    the twelve vis_decoder tensors (the net of weights_seed0.npz has none): seeded init + the same bias perturbation."""
    import network.render_ops as rops   # noqa: F401
    import utils.field_utils as fu
    cfg = yaml.load(open(REF + '/src/nr/configs/nrvgn_sdf.yaml'), Loader=yaml.FullLoader)
    cfg.update(volume_resolution=16, depth_sample_num=16, fine_depth_sample_num=16)
    cfg['agg_net_cfg']['sample_num'] = 16
    cfg['fine_agg_net_cfg']['sample_num'] = 16
    cfg['dist_decoder_cfg']['use_vis'] = True
    cfg['fine_dist_decoder_cfg']['use_vis'] = True
    torch.manual_seed(0)
    net = renderer.NeuralRayRenderer(cfg)
    net.eval()
    load_weights(net, weights)
    g = torch.Generator().manual_seed(4321)
    extra = {}
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if 'vis_decoder' in k:
                if k.endswith('.bias'):
                    v.add_(0.1 * torch.randn(v.shape, generator=g))
                extra[k] = v.detach().clone().numpy()
    assert len(extra) == 12
    fu.RESOLUTION = 16
    fu.VOXEL_SIZE = fu.VOLUME_SIZE / 16
    fu.HALF_VOXEL_SIZE = fu.VOXEL_SIZE / 2
    renderer.TSDF_SAMPLE_POINTS = fu.generate_grid_points()
    ref, que = make_scene(0, 'cfg1')
    t = lambda a: torch.from_numpy(a.copy())
    ref_info = {k: t(v) for k, v in ref.items()}
    que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
                'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    cap = {}
    orig_sort = torch.sort

    def srt(x, *a, **kw):
        r = orig_sort(x, *a, **kw)
        cap['sorted'] = r[0].clone()
        return r
    torch.sort = srt
    try:
        with torch.no_grad():
            vol = net.sample_volume(ref_info)
            rend = net.render(que_info, ref_info, False)
    finally:
        torch.sort = orig_sort
    out = {'render.' + k: v.numpy() for k, v in rend.items()}
    out['volume'] = vol.numpy()
    out['fine_depth_sorted'] = cap['sorted'].numpy()[0]
    out.update({'weights.' + k: v for k, v in extra.items()})
    np.savez_compressed(ROOT + '/tests/golden/golden_cfg1_use_vis.npz', **out)
    for k, v in out.items():
        print('   ', k, getattr(v, 'shape', None), getattr(v, 'dtype', None))


class _Stop(Exception):
    pass


def run_f1(renderer, weights, min_margin={'cfg1': 1e-4, 'cfg2': 1e-6}, max_tries=400):
    """Row F1 in isolation (render_ops.py:172-229): for each shape, the first scene seed >= 1 whose inverse-CDF samples all
    keep `min_margin` from every cdf edge, so that `inds` must be reproduced EXACTLY by any implementation
    whose cdf is closer than that to the reference's.  Stores the reference's coarse hit_prob (the resampler's input), its
    cdf, u, inds, per-sample margins and the resampled depths before and after the sort.  Only the coarse pass runs during
    the search (the hook on sample_fine_depth stops the forward)."""
    out = {}
    for name, res, dn in (('cfg1', 16, 16), ('cfg2', 40, 40)):
        net, cfg = build_net(renderer, res, dn)
        load_weights(net, weights)
        cap = {}
        orig_sfd, orig_ss = renderer.sample_fine_depth, torch.searchsorted

        def ss(cdf, u, **kw):
            r = orig_ss(cdf, u, **kw)
            cap.update(cdf=cdf.clone(), u=u.clone(), inds=r.clone())
            return r

        def sfd(depth, hit_prob, *a, **k):
            cap.update(depth=depth.clone(), hit_prob=hit_prob.clone())
            fd = orig_sfd(depth, hit_prob, *a, **k)
            cap['fine_depth'] = fd.clone()
            raise _Stop()
        renderer.sample_fine_depth, torch.searchsorted = sfd, ss
        try:
            for seed in range(1, max_tries):
                ref, que = make_scene(seed, name)
                t = lambda a: torch.from_numpy(a.copy())
                ref_info = {k: t(v) for k, v in ref.items()}
                que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
                            'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
                try:
                    with torch.no_grad():
                        net.render(que_info, ref_info, False)
                except _Stop:
                    pass
                marg = inds_margin(cap['cdf'].numpy()[0], cap['u'].numpy()[0])
                print(f'   f1 {name} seed {seed}: min margin {marg.min():.3e}', flush=True)
                if marg.min() >= min_margin[name]:
                    break
            else:
                raise RuntimeError('no seed found')
        finally:
            renderer.sample_fine_depth, torch.searchsorted = orig_sfd, orig_ss
        out.update({f'{name}.seed': np.int64(seed), f'{name}.hit_prob': cap['hit_prob'].numpy()[0],
                    f'{name}.depth': cap['depth'].numpy()[0], f'{name}.cdf': cap['cdf'].numpy()[0],
                    f'{name}.inds': cap['inds'].numpy()[0].astype(np.int64), f'{name}.margin': marg,
                    f'{name}.fine_depth': cap['fine_depth'].numpy()[0],
                    f'{name}.input_sha256': np.frombuffer(sha([ref[k] for k in sorted(ref)] + [que[k] for k in sorted(que)]).encode(), np.uint8)})
    np.savez_compressed(ROOT + '/tests/golden/golden_f1.npz', **out)
    print('f1 golden:', {k: (v.shape if hasattr(v, 'shape') and v.shape else v) for k, v in out.items()})


def run_ckpt_keys(renderer):
    """The layout of a reference checkpoint (`torch.save({'network_state_dict': net.state_dict(), ...})`, trainer side;
    loaded at main.py:153-155): every key of the reference's own GraspNeRF(cfg).state_dict() with its shape, in order.
    A 4.66 M-parameter checkpoint cannot be a fixture; its key / shape list can (tests build a `model_best.pth` from it)."""
    cfg = yaml.load(open(REF + '/src/nr/configs/nrvgn_sdf.yaml'), Loader=yaml.FullLoader)
    net = renderer.GraspNeRF(cfg)
    sd = net.state_dict()
    names = np.array(list(sd.keys()))
    shapes = np.array([','.join(str(int(d)) for d in v.shape) for v in sd.values()])
    np.savez_compressed(ROOT + '/tests/golden/golden_ckpt_keys.npz', names=names, shapes=shapes,
                        dtypes=np.array([str(v.dtype) for v in sd.values()]))
    print('checkpoint layout golden:', len(names), 'tensors,', sum(v.numel() for v in sd.values()), 'values')


def run_full_forward(renderer):
    """GraspNeRF.forward (backbones + render + sample_volume + depth-mean head + grasp head), eval mode,
    cfg1 shape, parameters from synth_state_dict.  ref: renderer.py:268-331."""
    cfg = yaml.load(open(REF + '/src/nr/configs/nrvgn_sdf.yaml'), Loader=yaml.FullLoader)
    for k, v in (('volume_resolution', 16), ('depth_sample_num', 16), ('fine_depth_sample_num', 16)):
        cfg[k] = v
    cfg['agg_net_cfg']['sample_num'] = 16
    cfg['fine_agg_net_cfg']['sample_num'] = 16
    import utils.field_utils as fu
    fu.RESOLUTION, fu.VOXEL_SIZE = 16, fu.VOLUME_SIZE / 16
    fu.HALF_VOXEL_SIZE = fu.VOXEL_SIZE / 2
    renderer.TSDF_SAMPLE_POINTS = fu.generate_grid_points()
    net = renderer.GraspNeRF(cfg)
    net.eval()
    sd = net.state_dict()
    syn = synth_state_dict({k: tuple(v.shape) for k, v in sd.items()})
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
    ref, que = make_scene(0, 'cfg1')
    t = lambda a: torch.from_numpy(a.copy())
    ref_info = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
    que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
                'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    data = {'step': 0, 'eval': True, 'full_vol': True, 'ref_imgs_info': ref_info, 'que_imgs_info': que_info,
            'src_imgs_info': dict(ref_info)}
    torch.manual_seed(123)                      # fixes randperm of the depth-loss coordinates
    with torch.no_grad():
        out = net(data)
        img_feats = net.nr_net.image_encoder(ref_info['imgs'])
        ray_feats = net.nr_net.vis_encoder(net.nr_net.init_net(ref_info, None, False), img_feats)
    g = {'volume': out['volume'].numpy(), 'img_feats_sub': img_feats.numpy()[:, :, ::2, ::2],
         'ray_feats_sub': ray_feats.numpy()[:, :, ::2, ::2],
         'depth_coords': out['depth_coords'][0].numpy().astype(np.int16)}
    for k in ('depth_mean', 'depth_mean_2', 'depth_mean_fine', 'depth_mean_fine_2'):
        g[k] = out[k].numpy()[:, ::4]
    for k in ('sdf_values', 'alpha_values', 'hit_prob_nr', 'render_depth', 'pixel_colors_nr', 'ray_mask'):
        g['render.' + k] = out[k].numpy()
    q, r, w = out['vgn_pred']
    g['vgn_qual_sub'], g['vgn_rot_sub'], g['vgn_width_sub'] = q.numpy()[..., ::2, ::2, ::2], r.numpy()[..., ::2, ::2, ::2], w.numpy()[..., ::2, ::2, ::2]
    np.savez_compressed(ROOT + '/tests/golden/golden_full_cfg1.npz', **g)
    print('full forward golden:', {k: v.shape for k, v in g.items()})


def run_no_hier(renderer):
    """cfg use_hierarchical_sampling: false (the reference's base_cfg default, renderer.py:22; render_impl :153-162 returns the coarse
    pass's outputs only): GraspNeRF.forward in eval mode and NeuralRayRenderer.render in training mode (values), cfg1 shape.  Also
    records what the reference does with volume_type ['alpha'] (renderer.py:189-191)."""
    cfg = yaml.load(open(REF + '/src/nr/configs/nrvgn_sdf.yaml'), Loader=yaml.FullLoader)
    for k, v in (('volume_resolution', 16), ('depth_sample_num', 16), ('fine_depth_sample_num', 16), ('use_hierarchical_sampling', False)):
        cfg[k] = v
    cfg['agg_net_cfg']['sample_num'] = 16
    cfg['fine_agg_net_cfg']['sample_num'] = 16
    import utils.field_utils as fu
    fu.RESOLUTION, fu.VOXEL_SIZE = 16, fu.VOLUME_SIZE / 16
    fu.HALF_VOXEL_SIZE = fu.VOXEL_SIZE / 2
    renderer.TSDF_SAMPLE_POINTS = fu.generate_grid_points()
    net = renderer.GraspNeRF(cfg)
    net.eval()
    sd = net.state_dict()
    syn = synth_state_dict({k: tuple(v.shape) for k, v in sd.items()})
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
    ref, que = make_scene(0, 'cfg1')
    t = lambda a: torch.from_numpy(a.copy())
    ref_info = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
    que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
                'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    data = {'step': 0, 'eval': True, 'full_vol': True, 'ref_imgs_info': ref_info, 'que_imgs_info': que_info,
            'src_imgs_info': dict(ref_info)}
    torch.manual_seed(123)
    with torch.no_grad():
        out = net(data)
    assert not any(k.endswith('_fine') and not k.startswith('depth_mean') for k in out), sorted(out)
    g = {'volume': out['volume'].numpy(), 'keys': np.array(sorted(k for k in out if k != 'vgn_pred'))}
    for k in ('sdf_values', 'alpha_values', 'hit_prob_nr', 'render_depth', 'pixel_colors_nr', 'ray_mask', 'pixel_colors_gt', 'sdf_gradient_error'):
        g['render.' + k] = out[k].numpy()
    # training-mode values of the render (coarse pass only: no torch.rand is drawn, only the coarse aggregation net counts a step)
    nr = net.nr_net
    nr.train()
    with torch.no_grad():
        ri = dict(ref_info)
        ri['img_feats'] = nr.image_encoder(ri['imgs'])
        ri['ray_feats'] = nr.vis_encoder(nr.init_net(ri, None, True), ri['img_feats'])
        torch.manual_seed(7)
        before = torch.rand(1).item()
        torch.manual_seed(7)
        tr = nr.render(que_info, ri, True)
        after = torch.rand(1).item()
    g['train_rng_untouched'] = np.array(before == after)
    assert not hasattr(nr, 'fine_agg_net') and not any(k.startswith('nr_net.fine_') for k in sd)      # renderer.py:56-58
    g['train_steps'] = np.array([nr.agg_net.step])
    for k in ('sdf_values', 'alpha_values', 'hit_prob_nr', 'render_depth', 'pixel_colors_nr'):
        g['train.' + k] = tr[k].numpy()
    # volume_type ['alpha'] (renderer.py:189-191): does the reference's own call run?
    try:
        cfg2 = dict(cfg, volume_type=['alpha'])
        net2 = renderer.GraspNeRF(cfg2).eval()
        net2.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
        with torch.no_grad():
            net2(data)
        g['volume_type_alpha'] = np.array('runs')
    except Exception as e:
        import traceback
        g['volume_type_alpha'] = np.array('raises ' + type(e).__name__ + ': ' + str(e)[:200] + ' @ ' + traceback.format_exc().strip().splitlines()[-3].strip()[:160])
    print('volume_type alpha in the reference:', g['volume_type_alpha'])
    np.savez_compressed(ROOT + '/tests/golden/golden_full_cfg1_nohier.npz', **g)
    print('no-hier golden:', {k: getattr(v, 'shape', v) for k, v in g.items()})


def run_train_step(renderer, scene_id=0, weight_seed=7, torch_seed=321, loss_seed=5, out_name='golden_train_step.npz', use_vis=False, use_all=False):
    """One training step of the reference (trainer.py:142-158): GraspNeRF.forward in train mode on a cfg1 scene with
    synthetic supervision (synth_loss_case targets), the configured losses (loss: [render, depth, sdf, vgn]), backward.
    -> tests/golden/<out_name>: every loss term, the gradient of every hot-path parameter (full arrays) and the
    L2 norm of every parameter's gradient.  Two fixtures are kept: the defaults (golden_train_step.npz) and a
    second scene / parameter draw / RNG stream (golden_train_step_b.npz: scene 3, weight seed 11, torch seed 77, loss case 9); a third with
    `use_vis: true` on both decoders (golden_train_step_vis.npz: the fourth decoder branch, dist_decoder.py:89-97, under training); a fourth with
    `fine_depth_use_all: true` (golden_train_step_all.npz: 16 + 16 samples per ray in the fine pass, renderer.py:145-146)."""
    _stub_optional_modules()
    import network.loss as L
    cfg = yaml.load(open(REF + '/src/nr/configs/nrvgn_sdf.yaml'), Loader=yaml.FullLoader)
    for k, v in (('volume_resolution', 16), ('depth_sample_num', 16), ('fine_depth_sample_num', 16), ('ray_batch_num', 40)):
        cfg[k] = v
    cfg['agg_net_cfg']['sample_num'] = cfg['fine_agg_net_cfg']['sample_num'] = 16
    if use_vis:
        cfg['dist_decoder_cfg']['use_vis'] = cfg['fine_dist_decoder_cfg']['use_vis'] = True
    if use_all:                                       # renderer.py:145-146; the fine level's positional table must hold dn + fdn rows (ibrnet.py:491)
        cfg['fine_depth_use_all'] = True
        cfg['fine_agg_net_cfg']['sample_num'] = 32
    import utils.field_utils as fu
    fu.RESOLUTION, fu.VOXEL_SIZE = 16, fu.VOLUME_SIZE / 16
    fu.HALF_VOXEL_SIZE = fu.VOXEL_SIZE / 2
    renderer.TSDF_SAMPLE_POINTS = fu.generate_grid_points()
    net = renderer.GraspNeRF(cfg)
    net.train()
    syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=weight_seed)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
    ref, que = make_scene(scene_id, 'cfg1')
    _, gt = synth_loss_case(seed=loss_seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    ref_info = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
    ref_info.update(true_depth=t(gt['true_depth']), sdf_gt=t(gt['sdf_gt']))
    que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
                'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    data = {'step': 0, 'ref_imgs_info': ref_info, 'que_imgs_info': que_info, 'src_imgs_info': dict(ref_info),
            'grasp_info': tuple(t(x) for x in gt['grasp_info']), 'scene_name': 'vgn_syn/train/pile/x'}
    torch.manual_seed(torch_seed)
    out = net(data)
    terms = {}
    for loss in (L.RenderLoss({'use_nr_fine_loss': True}), L.DepthLoss({}), L.SDFLoss({}), L.VGNLoss({})):
        terms.update(loss(out, data, 0))
    total = sum(torch.mean(v) for k, v in terms.items() if k.startswith('loss'))
    total.backward()
    g = {'loss.' + k: np.asarray(v.detach().numpy(), np.float64).reshape(-1) for k, v in terms.items()}
    g['total'] = np.float64(total.item())
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
        if p.grad is not None and any(s in k for s in ('dist_decoder', 'agg_net')):
            g['grad.' + k] = p.grad.numpy()
    g['param_names'] = np.array(names)
    g['grad_norms'] = np.asarray(norms)
    g['no_grad'] = np.array([k for k, p in net.named_parameters() if p.grad is None])
    g['setup'] = np.array([scene_id, weight_seed, torch_seed, loss_seed])
    g['use_vis'] = np.int64(1 if use_vis else 0)
    g['fine_depth_use_all'] = np.int64(1 if use_all else 0)
    np.savez_compressed(ROOT + '/tests/golden/' + out_name, **g)
    print('train step golden: total', total.item(), {k: float(v.mean()) for k, v in terms.items() if k.startswith('loss')})
    print('  params', len(names), 'without grad', len(g['no_grad']), 'hot-path grads stored', sum(k.startswith('grad.') for k in g))


def run_losses():
    """Reference losses (loss.py) on synth_loss_case tensors -> tests/golden/golden_losses.npz."""
    import types
    for name in ('pyquaternion', 'torchmetrics', 'cv2', 'h5py', 'plyfile', 'skimage', 'skimage.io', 'skimage.metrics',
                 'transforms3d', 'transforms3d.axangles', 'transforms3d.euler', 'open3d'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    class _Any(types.ModuleType):
        def __getattr__(self, k):
            return None
    for name in list(sys.modules):
        if name.split('.')[0] in ('pyquaternion', 'torchmetrics', 'cv2', 'h5py', 'plyfile', 'skimage', 'transforms3d', 'open3d') and not isinstance(sys.modules[name], _Any):
            sys.modules[name] = _Any(name)
    sys.modules['skimage.io'].imread = sys.modules['skimage.io'].imsave = None
    sys.modules['transforms3d.axangles'].mat2axangle = None
    sys.modules['transforms3d.euler'].mat2euler = sys.modules['transforms3d.euler'].euler2mat = None
    sys.modules['plyfile'].PlyData = None
    sys.modules['skimage.metrics'].structural_similarity = None
    import network.loss as L
    pr, gt = synth_loss_case()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    data_pr = {k: (tuple(t(x) for x in v) if isinstance(v, tuple) else t(v)) for k, v in pr.items()}
    data_gt = {'ref_imgs_info': {'true_depth': t(gt['true_depth']), 'depth_range': t(gt['depth_range']), 'sdf_gt': t(gt['sdf_gt'])},
               'grasp_info': tuple(t(x) for x in gt['grasp_info']), 'scene_name': 'vgn_syn/train/pile/x'}
    out = {}
    out.update(L.RenderLoss({'use_nr_fine_loss': True})(data_pr, data_gt, 0, True))
    out.update(L.DepthLoss({})(data_pr, data_gt, 0, True))
    out.update(L.SDFLoss({})(data_pr, data_gt, 0, True))
    out.update(L.VGNLoss({})(data_pr, data_gt, 0, True))
    g = {k: np.asarray(v.detach().numpy(), np.float64).reshape(-1) for k, v in out.items()}
    np.savez_compressed(ROOT + '/tests/golden/golden_losses.npz', **g)
    print('losses golden:', {k: float(v[0]) for k, v in g.items()})


def _stub_optional_modules():
    import types

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            return None
    for name in ('pyquaternion', 'torchmetrics', 'cv2', 'h5py', 'plyfile', 'skimage', 'skimage.io', 'skimage.metrics',
                 'transforms3d', 'transforms3d.axangles', 'transforms3d.euler', 'open3d'):
        if not isinstance(sys.modules.get(name), _Any):
            sys.modules[name] = _Any(name)


def run_post():
    """The reference planner's own process() / select() (main.py:23-84, scipy.ndimage inside) on synthetic head
    outputs -> tests/golden/golden_post.npz (planner thresholds main.py:93-94 and the function defaults)."""
    from graspnerf_amd.synth import synth_head_outputs
    _stub_optional_modules()
    import main as refmain                     # /root/reference/src/nr/main.py
    out = {}
    for seed, (hi, lo) in ((0, (0.0, -0.85)), (1, (0.0, -0.85)), (2, (0.5, 1e-3))):
        tsdf, qual, rot, width = synth_head_outputs(seed)
        q, r, w = refmain.process(tsdf.copy(), qual.copy(), rot.copy(), width.copy(), tsdf_thres_high=hi, tsdf_thres_low=lo)
        grasps, scores, indexs = refmain.select(q.copy(), r, w)
        out[f's{seed}.qual'] = q.astype(np.float32)
        out[f's{seed}.index'] = np.asarray(indexs, np.int64).reshape(-1, 3)
        out[f's{seed}.score'] = np.asarray(scores, np.float32)
        out[f's{seed}.quat'] = np.asarray([g.pose.rotation.as_quat() for g in grasps], np.float64).reshape(-1, 4)
        out[f's{seed}.pos'] = np.asarray([g.pose.translation for g in grasps], np.float64).reshape(-1, 3)
        out[f's{seed}.width'] = np.asarray([g.width for g in grasps], np.float32)
        out[f's{seed}.thres'] = np.asarray([hi, lo], np.float64)
        print('post golden seed', seed, 'nonzero qual', int((q != 0).sum()), 'grasps', len(grasps))
    np.savez_compressed(ROOT + '/tests/golden/golden_post.npz', **out)


def main():
    renderer = import_reference()
    if '--post-only' in sys.argv:
        return run_post()
    if '--train-step-only' in sys.argv:
        if '--vis' in sys.argv:
            return run_train_step(renderer, 1, 13, 99, 7, 'golden_train_step_vis.npz', use_vis=True)
        if '--use-all' in sys.argv:
            return run_train_step(renderer, 2, 17, 55, 3, 'golden_train_step_all.npz', use_all=True)
        run_train_step(renderer)
        return run_train_step(renderer, 3, 11, 77, 9, 'golden_train_step_b.npz')
    if '--losses-only' in sys.argv:
        return run_losses()
    if '--full-only' in sys.argv:
        return run_full_forward(renderer)
    if '--no-hier-only' in sys.argv:
        return run_no_hier(renderer)
    if '--ckpt-keys-only' in sys.argv:
        return run_ckpt_keys(renderer)
    if '--use-vis-only' in sys.argv:
        return run_use_vis(renderer, dict(np.load(ROOT + '/tests/golden/weights_seed0.npz')))
    if '--use-all-only' in sys.argv:
        return run_use_all(renderer, dict(np.load(ROOT + '/tests/golden/weights_seed0.npz')))
    if '--f1-only' in sys.argv:
        return run_f1(renderer, dict(np.load(ROOT + '/tests/golden/weights_seed0.npz')))
    if '--train-only' in sys.argv:
        return run_train_mode(renderer, dict(np.load(ROOT + '/tests/golden/weights_seed0.npz')))
    os.makedirs(ROOT + '/tests/golden', exist_ok=True)
    net40, cfg40 = build_net(renderer, 40, 40)
    weights = perturb_and_export(net40)
    np.savez_compressed(ROOT + '/tests/golden/weights_seed0.npz', **weights)
    print('weights:', sum(v.size for v in weights.values()), 'floats')

    for name, res, dn, net_cfg in (('cfg2', 40, 40, (net40, cfg40)), ('cfg1', 16, 16, None)):
        if net_cfg is None:
            net, cfg = build_net(renderer, res, dn)
            load_weights(net, weights)
        else:
            net, cfg = net_cfg
        ref, que = make_scene(0, name)
        out = run_reference(renderer, net, cfg, ref, que, res)
        out['input_sha256'] = np.frombuffer(sha([ref[k] for k in sorted(ref)] +
                                                 [que[k] for k in sorted(que)]).encode(), np.uint8)
        out['dn'] = np.int64(dn)
        np.savez_compressed(ROOT + f'/tests/golden/golden_{name}.npz', **out)
        print(name, 'volume range', out['volume'].min(), out['volume'].max(),
              'mask margin', out['volume_mask_min_margin_px'], 'inds margin', out['fine_inds_min_margin'])
        for k, v in out.items():
            print('   ', k, getattr(v, 'shape', None), getattr(v, 'dtype', None))
    run_train_mode(renderer, weights)
    run_f1(renderer, weights)
    run_use_all(renderer, weights)
    run_use_vis(renderer, weights)
    run_ckpt_keys(renderer)
    run_full_forward(renderer)
    run_losses()
    run_post()
    run_train_step(renderer)
    run_train_step(renderer, 3, 11, 77, 9, 'golden_train_step_b.npz')
    run_train_step(renderer, 1, 13, 99, 7, 'golden_train_step_vis.npz', use_vis=True)
    run_train_step(renderer, 2, 17, 55, 3, 'golden_train_step_all.npz', use_all=True)


if __name__ == '__main__':
    main()
