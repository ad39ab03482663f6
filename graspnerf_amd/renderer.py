"""Drop-in mirror of the reference's model API for the volumetric path:
`NeuralRayRenderer(cfg).forward / sample_volume / render / render_by_depth / predict_mean_for_depth_loss`
and `GraspNeRF(cfg).forward / select` with the reference's names, dict schemas, output keys and
state-dict keys (ref: src/nr/network/renderer.py:13-335), so `src/gd`'s consumers and reference
checkpoints plug in unchanged.  The 2D backbones and the grasp head are PyTorch-ROCm modules
(backbone.py); everything between `ray_feats` and `volume` / the render dict runs in the HIP
kernels behind libgnr.so (hotpath.py).  With autograd enabled in training mode the same kernels run behind
autograd.Functions whose backward calls the backward twins (csrc/gnr_bwd.inc): ray geometry, coarse depths,
inverse-CDF resampling and the sort are done by the forward kernels, this module only routes tensors between the twin pairs
and builds the output dicts.  There is no PyTorch statement of the path in the product: without a GPU, training raises
like inference does (the differentiable reference statement the backward kernels are tested against lives in
tests/reference_autograd.py).
"""
import numpy as np
import torch
import torch.nn as nn

from . import weights as _w
from .backbone import ResUNetLight, CostVolumeInitNet, DefaultVisEncoder, ConvNet
from .hotpath import HotPath
from .grasp_head import GraspHead
from . import _lib
from . import ray_tail as _rt


def _kaiming(mods):
    for m in mods.modules():
        if isinstance(m, nn.Linear):
            nn.init.kaiming_normal_(m.weight.data)
            if m.bias is not None:
                nn.init.zeros_(m.bias.data)


def _mlp(dims, skip_act_index=True):
    """Linear layers at the reference's nn.Sequential indices (0, 2, 4, ...): parameter holders."""
    layers = []
    for i in range(len(dims) - 1):
        layers += [nn.Linear(dims[i], dims[i + 1]), nn.Identity()]
    return nn.Sequential(*layers[:-1])


class _DistDecoderParams(nn.Module):
    """ref: dist_decoder.py:53-97 (parameter container; `use_vis` adds the fourth branch, :89-97)"""

    def __init__(self, use_vis=False):
        super().__init__()
        self.mean_decoder, self.var_decoder, self.aw_decoder = _mlp([32, 32, 32, 2]), _mlp([32, 32, 32, 2]), _mlp([32, 32, 32, 1])
        if use_vis:
            self.vis_decoder = _mlp([32, 32, 32, 1])


class _Attention(nn.Module):
    def __init__(self):
        super().__init__()
        self.w_qs, self.w_ks, self.w_vs, self.fc = (nn.Linear(16, 16, bias=False) for _ in range(4))
        self.layer_norm = nn.LayerNorm(16, eps=1e-6)


class _AggImplParams(nn.Module):
    """ref: ibrnet.py:373-435 (IBRNetWithNeuRayNeus)"""

    def __init__(self):
        super().__init__()
        self.ray_dir_fc = _mlp([4, 16, 35])
        self.base_fc = _mlp([207, 64, 32])
        self.vis_fc = _mlp([32, 32, 33])
        self.vis_fc2 = _mlp([32, 32, 1])
        self.geometry_fc = _mlp([86, 64, 16])
        self.ray_attention = _Attention()
        self.out_geometry_fc = nn.Sequential(nn.Linear(16, 16), nn.Linear(16, 1))
        self.rgb_fc = _mlp([37, 16, 8, 1])
        self.neuray_fc = _mlp([32, 8, 1])
        for m in (self.base_fc, self.vis_fc2, self.vis_fc, self.geometry_fc, self.rgb_fc, self.neuray_fc):
            _kaiming(m)                                                   # ibrnet.py:430-435


class _Deviation(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.variance = nn.Parameter(torch.tensor(float(init_val)), requires_grad=False)    # neus.py:9-10


class _AggNetParams(nn.Module):
    """ref: aggregate_net.py:19-33,87-101"""

    def __init__(self, cfg):
        super().__init__()
        self.prob_embed = _mlp([34, 32, 32])
        self.agg_impl = _AggImplParams()
        self.deviation_network = _Deviation(cfg.get('init_s', 0.3))
        self.fix_s = cfg.get('fix_s', False)                               # aggregate_net.py:91
        self.step = 0                                                      # aggregate_net.py:99

    def train_step_bookkeeping(self):
        """aggregate_net.py:135-137 + neus.py:13-18: one render pass in training mode advances the step counter and,
        once step > fix_s, makes the NeuS variance trainable."""
        self.step += 1
        if self.fix_s != -1 and self.step > self.fix_s:
            self.deviation_network.variance.requires_grad_(True)


_DM_PARAMS = ('0.weight', '0.bias', '2.weight', '2.bias', '4.weight', '4.bias')


class _DepthMeanFn(torch.autograd.Function):
    """predict_mean_for_depth_loss for one level and one scene on the HIP path in both directions
    (gnr_depth_mean_fwd / gnr_depth_mean_bwd).  Differentiable inputs: ray_feats [V,32,fh,fw] and the six
    mean_decoder parameters; `prep` must stay untouched between forward and backward."""

    @staticmethod
    def forward(ctx, hot, bref, prep, xy, level, ray_feats, *params):      # xy [B,pn,2], ray_feats [B,V,32,fh,fw]
        ctx.args = (hot, bref, prep, xy, level)
        ctx.gen = hot.generation
        return hot.depth_mean(bref, xy, level, prepared=prep)              # [B,V,pn,2]

    @staticmethod
    def backward(ctx, dmean):
        hot, bref, prep, xy, level = ctx.args
        hot.check_generation(ctx.gen)
        dcan, dray = hot.depth_mean_bwd(bref, xy, dmean.contiguous(), level, prepared=prep)
        dec = _w.LEVELS[level][0]
        g = _w.split_canonical(dcan, level)
        return (None, None, None, None, None, dray) + tuple(g[dec + 'mean_decoder.' + n] for n in _DM_PARAMS)


class _SampleVolumeFn(torch.autograd.Function):
    """sample_volume of one scene on the HIP path in both directions (gnr_sample_volume_fwd_train / gnr_sample_volume_bwd,
    csrc/gnr_bwd.inc).  Differentiable inputs: ray_feats, img_feats [V,32,fh,fw] and every coarse-level parameter in
    state-dict order; the workspaces of `hot` must stay untouched between forward and backward."""

    @staticmethod
    def forward(ctx, hot, bref, prep, res, ray_feats, img_feats, *params):
        ctx.hot, ctx.gen = hot, hot.generation
        return hot.sample_volume_train(bref, res, prepared=prep)

    @staticmethod
    def backward(ctx, dvol):
        hot = ctx.hot
        hot.check_generation(ctx.gen)
        dcan, dray, dimg = hot.sample_volume_bwd(dvol.contiguous(), hot.can_dev['coarse'])
        g = _w.split_canonical(dcan, 'coarse', use_vis=hot.use_vis)
        return (None, None, None, None, dray, dimg) + tuple(g[k] for k, _ in _w.level_keys('coarse', use_vis=hot.use_vis))


_FW_KEYS = ('sdf_values', 'sdf_gradient', 'alpha_values', 'hit_prob_nr', 'pixel_colors_nr', 'render_depth', 'ray_mask',
            'sdf_gradient_error')
_GEO_KEYS = ('depth', 'pts', 'qdir')


def _pass_value_names(with_gt, with_fine_depth):
    """Names of the non-differentiable outputs _RenderChainFn returns after (stats, colours)."""
    return _GEO_KEYS + _FW_KEYS + (('pixel_colors_gt',) if with_gt else ()) + (('fine_depth',) if with_fine_depth else ())


class _RenderChainFn(torch.autograd.Function):
    """The per-view chain of one render pass of B scenes on the HIP path in both directions
    (gnr_render_chain_fwd_train / gnr_render_chain_bwd): everything up to the cross-view statistics and the colour blend.
    `depth` None = the coarse pass (sample_depth on the device).  Differentiable inputs: ray_feats, img_feats and every
    parameter of the pass's level in state-dict order; differentiable outputs stats [B,P,66] (mean, var, wbar, n_valid) and
    colours [B,P,3].  The forward also runs the pass's per-ray tail, NeuS alpha and compositing (k_ray<true>) on the records
    the chain left in the workspace and returns their VALUES (they become the outputs of _RayTailFn / _CompositeFn, which
    own the backward), the pass's ray geometry (depth, pts, qdir) and -- for the coarse pass -- the sorted inverse-CDF
    resampling (render_ops.py:172-229) as non-differentiable outputs, in _pass_value_names() order."""

    @staticmethod
    def forward(ctx, hot, prep, que, depth, level, cfg, want_fine_depth, ray_feats, img_feats, *params):
        stats, colors, geo, hctx = hot.render_chain_train(que, depth, level, cfg, prep)
        ctx.hot, ctx.hctx, ctx.level, ctx.gen = hot, hctx, level, hot.generation
        fw = hot.render_tail_train(hctx, que, geo['depth'], colors, want_fine_depth)
        fw.update(geo)
        extra = tuple(fw[k] for k in _pass_value_names('pixel_colors_gt' in fw, want_fine_depth))
        ctx.mark_non_differentiable(*extra)
        return (stats, colors) + extra

    @staticmethod
    def backward(ctx, dstats, dcolors, *_):
        ctx.hot.check_generation(ctx.gen)
        dcan, dray, dimg = ctx.hot.render_chain_bwd(ctx.hctx, dstats[..., :65].contiguous(), dcolors.contiguous())
        g = _w.split_canonical(dcan, ctx.level, use_vis=ctx.hot.use_vis)
        return (None,) * 7 + (dray, dimg) + tuple(g[k] for k, _ in _w.level_keys(ctx.level, use_vis=ctx.hot.use_vis))


def randperm_prefix(n, k):
    """torch.randperm(n)[:k] on the default CPU generator -- same values, same generator state afterwards -- through
    gnr_host_randperm_prefix (csrc/gnr_host_rng.cpp: k swaps on a sparse identity + skipping the other draws) instead of n
    random-access swaps.  None when the helper does not know the generator's state layout (caller uses torch.randperm)."""
    import ctypes
    from . import _lib
    st = torch.get_rng_state()
    out = torch.empty(k, dtype=torch.int64)
    rc = _lib.lib().gnr_host_randperm_prefix(ctypes.c_void_p(st.data_ptr()), st.numel(), n, k, ctypes.c_void_p(out.data_ptr()))
    if rc != 0:
        return None
    torch.set_rng_state(st)
    return out


class _CompositeFn(torch.autograd.Function):
    """NeuS alpha + compositing of one training render pass (aggregate_net.py:105-121, render_ops.py:72-80,
    renderer.py:110-123) for R = B*rn rays.  Forward values are k_ray<true>'s (computed next to the chain); the backward is
    k_composite_bwd and hands a = dL/d sdf and gamma = dL/d grad to _RayTailFn.  Differentiable inputs: sdf [R,dn],
    grad [R,dn,3], colours [R,dn,3], deviation_network.variance; outputs alpha, hit_prob [R,dn], pixel colours [R,3],
    render depth [R], sdf_gradient_error [B,1]."""

    @staticmethod
    def forward(ctx, hot, level, depth, qdir, fw, sdf, grad, col, variance):
        assert fw['sdf_gradient_error'].shape[1] == 1, 'one ray chunk per training call (renderer.py:207-215 loops outside)'
        ctx.save_for_backward(sdf, grad, col, depth, qdir)
        ctx.meta = (hot, level, fw['sdf_gradient_error'].shape[0], hot.generation, variance.shape)
        R, dn = sdf.shape
        return (fw['alpha_values'].reshape(R, dn).clone(), fw['hit_prob_nr'].reshape(R, dn).clone(),
                fw['pixel_colors_nr'].reshape(R, 3).clone(), fw['render_depth'].reshape(R).clone(), fw['sdf_gradient_error'].clone())

    @staticmethod
    def backward(ctx, dalpha, dhit, dpix, ddepth, dgerr):
        hot, level, B, gen, vshape = ctx.meta
        hot.check_generation(gen)
        sdf, grad, col, depth, qdir = ctx.saved_tensors
        R, dn = sdf.shape
        if dpix is None:
            dpix = torch.zeros(R, 3, device=sdf.device)
        wgerr = None if dgerr is None else (dgerr.reshape(B, -1)[:, 0] / float((R // B) * dn)).repeat_interleave(R // B)
        a, gamma, dcol, dvar = hot.composite_bwd(level, sdf, grad, col, depth, qdir, dpix, ddepth, wgerr, dalpha, dhit)
        return None, None, None, None, None, a, gamma, dcol, dvar.reshape(vshape)


class _RayTailFn(torch.autograd.Function):
    """The per-ray tail of one training render pass (geometry_fc on [stats, embed(p)], attention, LayerNorm,
    out_geometry_fc, clip, and the in-forward gradient of sdf w.r.t. the points; ibrnet.py:485-504).  Forward values are
    k_ray<true>'s (computed next to the chain, _RenderChainFn); the backward takes dL/d sdf AND dL/d grad: a reverse pass
    on dual numbers: k_geo_dual_fwd -> k_ray_dual_bwd -> k_geo_dual_bwd (csrc/gnr_bwd.inc), no double-backward graph.
    Differentiable inputs: stats [N,66] and the 14 tail parameters of the level (ray_tail.TAIL_KEYS order)."""

    @staticmethod
    def forward(ctx, hot, level, agg, pts, rn, dn, sdf, grad, stats, *params):
        ctx.save_for_backward(stats, pts, *params)
        ctx.meta = (hot, level, agg, rn, dn, hot.generation)
        return sdf.reshape(rn, dn).clone(), grad.reshape(rn, dn, 3).clone()

    @staticmethod
    def backward(ctx, a, gamma):
        hot, level, agg, rn, dn, gen = ctx.meta
        hot.check_generation(gen)
        stats, pts, *params = ctx.saved_tensors
        P = {agg + 'agg_impl.' + k: p for k, p in zip(_rt.TAIL_KEYS, params)}
        a = torch.zeros(rn, dn, device=stats.device) if a is None else a
        gamma = torch.zeros(rn, dn, 3, device=stats.device) if gamma is None else gamma
        with torch.no_grad():
            # the device kernels read / write the 66-column layout directly (column 65 = n_valid, gradient 0)
            dstats, G = _rt.tail_backward(hot, level, P, agg, stats, pts, rn, dn, a.contiguous(), gamma.contiguous())
        return (None,) * 8 + (dstats,) + tuple(G[agg + 'agg_impl.' + k] for k in _rt.TAIL_KEYS)


class NeuralRayRenderer(nn.Module):
    base_cfg = {
        'vis_encoder_type': 'default', 'vis_encoder_cfg': {}, 'dist_decoder_type': 'mixture_logistics',
        'dist_decoder_cfg': {}, 'agg_net_type': 'default', 'agg_net_cfg': {}, 'use_hierarchical_sampling': False,
        'fine_agg_net_cfg': {}, 'fine_dist_decoder_cfg': {}, 'fine_depth_sample_num': 64, 'fine_depth_use_all': False,
        'ray_batch_num': 2048, 'depth_sample_num': 64, 'alpha_value_ground_state': -15, 'use_dr_prediction': False,
        'use_nr_color_for_dr': False, 'use_self_hit_prob': False, 'use_ray_mask': True, 'ray_mask_view_num': 2,
        'ray_mask_point_num': 8, 'render_depth': False, 'disable_view_dir': False, 'render_rgb': False,
        'init_net_type': 'depth', 'init_net_cfg': {}, 'depth_loss_coords_num': 8192,
    }                                                                       # renderer.py:14-45

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.base_cfg, **cfg}
        c = self.cfg
        unsupported = [k for k, ok in (
            ('agg_net_type', c['agg_net_type'] == 'neus'), ('init_net_type', c['init_net_type'] == 'cost_volume'),
            # the reference evaluates the fine level with the COARSE decoder's compute_prob (renderer.py:70-72): mixed settings
            # either crash there (None * tensor) or silently ignore the fine branch
            ('dist_decoder_cfg.use_vis != fine_dist_decoder_cfg.use_vis',
             not c['use_hierarchical_sampling'] or
             bool(c['dist_decoder_cfg'].get('use_vis', True)) == bool(c['fine_dist_decoder_cfg'].get('use_vis', True))),
            ('disable_view_dir', not c['disable_view_dir']),
            ('volume_type', list(c.get('volume_type', ['sdf'])) == ['sdf'])) if not ok]
        if unsupported:
            # Not configured by the reference's only yaml.  agg_net_type 'default' (the density branch) does not run in the reference either:
            # network_rendering (renderer.py:93) passes every aggregation net five arguments, DefaultAggregationNet.forward (aggregate_net.py:78)
            # takes four -> TypeError at the first render (tools/probe_agg_default.py, tests/golden/ref_agg_default_probe.json).  disable_view_dir cannot run in the reference either with
            # agg_net_type neus (aggregate_net.py:126 unpacks que_dir.shape of None), and volume_type ['alpha'] raises in the
            # reference's own call (renderer.py:190 -> network_rendering :100 "ValueError: too many values to unpack (expected 2)":
            # the neus aggregation returns five values; recorded by tools/make_goldens.py --no-hier-only in
            # tests/golden/golden_full_cfg1_nohier.npz 'volume_type_alpha'); 'image' reads prj_dict before it exists (:185).
            raise NotImplementedError(f'config options outside configs/nrvgn_sdf.yaml are not built: {unsupported}')
        if c['fine_depth_use_all']:
            # renderer.py:145-146: the fine pass renders the coarse and the resampled depths together.  The reference adds its
            # positional table [1, sample_num, 16] to [rn, dn + fdn, 16] (ibrnet.py:491): the lengths must agree there too.
            n = c['depth_sample_num'] + c['fine_depth_sample_num']
            if c['fine_agg_net_cfg'].get('sample_num', 64) != n:
                raise ValueError(f"fine_depth_use_all needs fine_agg_net_cfg.sample_num = depth_sample_num + fine_depth_sample_num = {n}")
            if n > 128:
                raise NotImplementedError('fine_depth_use_all: at most 128 samples per ray in the fine pass')
        self.vis_encoder = DefaultVisEncoder(c['vis_encoder_cfg'])
        self.use_vis = bool(c['dist_decoder_cfg'].get('use_vis', True))      # dist_decoder.py:57: the reference's default is True
        self.dist_decoder = _DistDecoderParams(self.use_vis)
        self.image_encoder = ResUNetLight(3, [1, 2, 6, 4], 32, inplanes=16)
        self.init_net = CostVolumeInitNet(c['init_net_cfg'])
        self.agg_net = _AggNetParams(c['agg_net_cfg'])
        # renderer.py:56-58: the fine level's modules (and state-dict keys) exist only with hierarchical sampling
        self.levels = ('coarse', 'fine') if c['use_hierarchical_sampling'] else ('coarse',)
        if c['use_hierarchical_sampling']:
            self.fine_dist_decoder = _DistDecoderParams(self.use_vis)
            self.fine_agg_net = _AggNetParams(c['fine_agg_net_cfg'])
        self.use_sdf = True
        self._hot = None

    def train(self, mode=True):
        if not mode and self._hot is not None:             # leaving training: the gigabyte-sized per-pass workspaces go back to the allocator
            self._hot.release_training_workspaces()
        return super().train(mode)

    # ---- HIP hot path handle (re-packed when parameters change device or values) -----------------
    def _apply(self, fn, *a, **k):
        self._hot = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._hot = None
        return super().load_state_dict(*a, **k)

    def _hot_versions(self):
        """Autograd version counters of the hot-path parameters: every in-place update (optimizer.step, copy_, load_state_dict)
        bumps them, so a change means the packed copies in the HotPath are stale."""
        ps = getattr(self, '_hot_params', None)
        if ps is None or self._hot is None:                  # (re)collected whenever the HotPath is rebuilt (_apply / load_state_dict)
            P = self._params()
            ps = self._hot_params = [P[k] for lvl in self.levels for k, _ in _w.level_keys(lvl)]
            if self.use_vis:
                ps += [P[_w.LEVELS[lvl][0] + k] for lvl in self.levels for k, _ in _w.VIS_KEYS]
        # (version, storage address): optimizer.step / copy_ / load_state_dict bump the version, `p.data = tensor` moves the
        # storage.  A write through `p.data.copy_()` / `p.data.mul_()` changes neither: call invalidate_packed() after such surgery.
        return tuple((p._version, p.data_ptr()) for p in ps)

    def invalidate_packed(self):
        """Forget the packed HIP copies of the hot-path weights: the next forward re-packs from the current parameter values.
        Needed only after in-place writes that bypass autograd's version counters (p.data.copy_, EMA swaps through .data)."""
        self._hot_ver = None
        self._bwd_ver = None
        self.__dict__.pop('_param_dict', None)

    def _vis_dev(self, sd, lvl):
        """The vis_decoder's six tensors of a level (use_vis) as one flat device tensor, state-dict order."""
        return torch.cat([sd[_w.LEVELS[lvl][0] + k].detach().reshape(-1).to(torch.float32) for k, _ in _w.VIS_KEYS])

    def _repack_on_device(self, with_bwd):
        """Packed copies of the CURRENT parameter values, built on the device (csrc/gnr_pack_dev.hip: the host packer's own
        arithmetic, bit-identical blobs): the parameters are gathered into the level's canonical blob by one torch.cat and a
        handful of stream-ordered kernels rewrite the packed blobs in place -- no device-to-host copy, no host-side pack, no
        upload, and nothing the host waits for (the reference's parameters never leave the device either, trainer.py:146-158)."""
        sd = self._params()
        hot, L = self._hot, _lib.lib()
        st = hot._stream()
        can_dev = {lvl: _w.canonical_blob_device(sd, lvl, as_tensor=True) for lvl in self.levels}
        vis = {lvl: self._vis_dev(sd, lvl) if self.use_vis else None for lvl in can_dev}
        for lvl, blob in (('coarse', hot.wc), ('fine', hot.wf))[:len(self.levels)]:
            _lib.check(L.gnr_pack_weights_device(can_dev[lvl].data_ptr(), blob.data_ptr(), st), 'gnr_pack_weights_device')
            if vis[lvl] is not None:
                _lib.check(L.gnr_pack_vis_decoder_device(vis[lvl].data_ptr(), blob.data_ptr(), st), 'gnr_pack_vis_decoder_device')
        if with_bwd:
            wb = getattr(hot, 'wb', None)
            if wb is None or wb.get(self.levels[-1]) is None:
                # first training forward: the backward blobs are created from host packs once (their structural zeros stay; the
                # device packer rewrites the rest every step)
                host = {lvl: t.cpu().numpy() for lvl, t in can_dev.items()}
                hv = {lvl: None if vis[lvl] is None else vis[lvl].cpu().numpy() for lvl in vis}
                hot.set_bwd_weights(*[_w.pack_bwd(host[lvl], hv[lvl]) for lvl in self.levels])
            else:
                for lvl in self.levels:
                    _lib.check(L.gnr_pack_weights_bwd_device(can_dev[lvl].data_ptr(), wb[lvl].data_ptr(), st), 'gnr_pack_weights_bwd_device')
                    if vis[lvl] is not None:
                        _lib.check(L.gnr_pack_vis_decoder_bwd_device(vis[lvl].data_ptr(), wb[lvl].data_ptr(), st), 'gnr_pack_vis_decoder_bwd_device')
            hot.can_dev = can_dev

    def hot(self):
        """The HIP path with weights packed from the CURRENT parameter values: re-packed (on the device) whenever a parameter was
        updated in place since the last packing (an eval forward after optimizer.step() must not run on the previous weights)."""
        ver = self._hot_versions()
        if self._hot is None:
            sd = self.state_dict()
            dev = next(self.parameters()).device
            self._hot = HotPath(*[_w.pack_state_dict(sd, lvl) for lvl in self.levels], device=dev)
            self._hot_ver = ver
        elif getattr(self, '_hot_ver', None) != ver:
            self._repack_on_device(with_bwd=False)
            self._hot_ver = ver
        if self.__dict__.get('_hot_option_bits') is not None:
            self._hot.options = self._hot_option_bits
        return self._hot

    def set_hot_option(self, name, on=True):
        """A per-call option of the HIP path (include/gnr.h GNR_OPT_*; hotpath.HotPath.set_option) for every call THIS module makes --
        kept here because the HotPath is rebuilt whenever the parameters move (`.to()`, load_state_dict); -> previous setting."""
        from . import _lib
        bit = _lib.OPTIONS[name]
        bits = self.__dict__.get('_hot_option_bits')
        if bits is None:
            bits = self._hot.options if self._hot is not None else HotPath.default_options
        self.__dict__['_hot_option_bits'] = (bits | bit) if on else (bits & ~bit)
        if self._hot is not None:
            self._hot.options = self._hot_option_bits
        return bool(bits & bit)

    def _upload(self, name, host_tensor, dev):
        """Host tensor -> device through a persistent pinned staging buffer (one per name): `x.pin_memory()` every step
        goes through the caching host allocator, whose occasional fresh hipHostMalloc showed up as 30-60 ms outlier steps
        (1 in 20).  The buffer is rewritten only after the previous copy out of it has completed."""
        bufs = self.__dict__.setdefault('_stage_bufs', {})
        ent = bufs.get(name)
        if ent is None or ent[0].shape != host_tensor.shape or ent[0].dtype != host_tensor.dtype:
            ent = [torch.empty(host_tensor.shape, dtype=host_tensor.dtype, pin_memory=True), None]
            bufs[name] = ent
        if ent[1] is not None:
            ent[1].synchronize()
        ent[0].copy_(host_tensor)
        out = ent[0].to(dev, non_blocking=True)
        ent[1] = torch.cuda.Event()
        ent[1].record()
        return out

    def repack_begin(self):
        """Kept for callers of the round-2..4 interface (the re-pack used to start its device-to-host copies here, ahead of the 2D
        backbones, and finish on the host in hot_for_training()).  The packer runs on the device now: nothing to start early."""
        return None

    def hot_for_training(self):
        """The HIP path with weights re-packed from the CURRENT parameter values (they move every optimiser step), plus
        the transposed fragments of the backward twins -- all on the device, stream-ordered, without a host wait."""
        ver = self._hot_versions()
        if self._hot is None:
            sd = self.state_dict()
            dev = next(self.parameters()).device
            self._hot = HotPath(*[_w.pack_state_dict(sd, lvl) for lvl in self.levels], device=dev)
            ver = self._hot_versions()
            self._repack_on_device(with_bwd=True)            # creates the backward blobs (host pack, once) and can_dev
        elif getattr(self, '_hot_ver', None) != ver or getattr(self._hot, 'wb', None) is None or self._hot.wb.get(self.levels[-1]) is None \
                or getattr(self._hot, 'can_dev', None) is None or getattr(self, '_bwd_ver', None) != ver:
            self._repack_on_device(with_bwd=True)
        self._hot_ver = self._bwd_ver = ver
        if self.__dict__.get('_hot_option_bits') is not None:
            self._hot.options = self._hot_option_bits
        return self._hot

    def _train_prep(self, ref_imgs_info, rn=0):
        """Once per training forward on the GPU: re-packed weights and the prepared (channel-last) feature maps shared
        by the HIP twin pairs of this forward."""
        hot = self.hot_for_training()
        bref = self._batched_ref({**ref_imgs_info, 'ray_feats': ref_imgs_info['ray_feats'].detach(),
                                  'img_feats': ref_imgs_info['img_feats'].detach()})
        prep = hot.prepare(bref, self.cfg.get('volume_resolution', 40), min(rn, self.cfg['ray_batch_num']), self._dn_max())
        return hot, bref, prep

    @staticmethod
    def _batched_ref(ref):
        out = {k: ref[k][None] for k in ('imgs', 'img_feats', 'ray_feats', 'poses', 'Ks', 'depth_range')}
        bb = ref['bbox3d']
        out['bbox3d'] = (bb if torch.is_tensor(bb) else torch.as_tensor(np.asarray(bb), dtype=torch.float32))[None]
        return out

    def _prepare(self, ref_imgs_info, rn=0):
        """Feature-map repack + view blocks once per forward; shared by volume / render / depth-mean."""
        c = self.cfg
        bref = self._batched_ref(ref_imgs_info)
        return bref, self.hot().prepare(bref, c.get('volume_resolution', 40), rn, self._dn_max())

    @staticmethod
    def _batched_que(que):
        out = {'coords': que['coords'], 'pose': que['poses'], 'K': que['Ks'], 'depth_range': que['depth_range']}
        if 'imgs' in que:
            out['imgs'] = que['imgs']
        return out

    def _render_cfg(self):
        c = self.cfg
        return {'depth_sample_num': c['depth_sample_num'], 'fine_depth_sample_num': c['fine_depth_sample_num'],
                'use_hierarchical_sampling': bool(c['use_hierarchical_sampling']),
                'ray_mask_view_num': c['ray_mask_view_num'], 'ray_mask_point_num': c['ray_mask_point_num'],
                'ray_batch_num': c['ray_batch_num'], 'fine_depth_use_all': bool(c['fine_depth_use_all'])}

    def _dn_max(self):
        """Most samples per ray of any render pass (workspace sizing)."""
        c = self.cfg
        if not c['use_hierarchical_sampling']:
            return c['depth_sample_num']
        fine = c['fine_depth_sample_num'] + (c['depth_sample_num'] if c['fine_depth_use_all'] else 0)
        return max(c['depth_sample_num'], fine)

    def _use_autograd(self, is_train):
        """Training (autograd on, parameters trainable): the HIP twin pairs behind autograd.Functions."""
        on = bool(is_train) and torch.is_grad_enabled()
        return on

    @staticmethod
    def _need_gpu(t):
        if not t.is_cuda:
            raise _lib.GnrError('graspnerf_amd: the volumetric path trains on a ROCm GPU only (HIP kernels in both directions); '
                                'there is no PyTorch / CPU fallback')

    def _params(self):
        """name -> Parameter of this module tree.  Cached (named_parameters() walks ~1 000 modules, several times per training step:
        2 ms of a step's host time); the Parameter OBJECTS are stable across optimizer steps, and whatever replaces them -- _apply
        (device / dtype moves), load_state_dict, invalidate_packed() -- drops the cache together with the packed weights."""
        d = self.__dict__.get('_param_dict')
        if d is None or self._hot is None:
            d = self.__dict__['_param_dict'] = dict(self.named_parameters())
        return d

    def _train_pass(self, hot, prep, que_b, depth, level, want_fine_depth, ray_feats, img_feats, P):
        """One render pass (renderer.py:90-138) of B scenes in training mode: per-view chain -> per-ray tail -> NeuS alpha /
        compositing, each a HIP twin pair behind its autograd.Function; tensors are only routed here.
        -> (dict of [B*rn, ...] results of the pass, dict of its non-differentiable values incl. geometry / fine depths)."""
        agg = _w.LEVELS[level][1]
        st, co, *extra = _RenderChainFn.apply(hot, prep, que_b, depth, level, self._render_cfg(), want_fine_depth, ray_feats, img_feats,
                                              *[P[k] for k, _ in _w.level_keys(level, use_vis=self.use_vis)])
        ex = dict(zip(_pass_value_names('imgs' in que_b, want_fine_depth), extra))
        B, rn, dn = ex['depth'].shape
        R = B * rn
        sdf, grad = _RayTailFn.apply(hot, level, agg, ex['pts'], R, dn, ex['sdf_values'], ex['sdf_gradient'], st.reshape(-1, 66),
                                     *[P[agg + 'agg_impl.' + k] for k in _rt.TAIL_KEYS])
        col = co.reshape(R, dn, 3)
        alpha, hit, pix, rdepth, gerr = _CompositeFn.apply(hot, level, ex['depth'].reshape(R, dn), ex['qdir'], ex, sdf, grad, col,
                                                           P[agg + 'deviation_network.variance'])
        return {'sdf': sdf, 'col': col, 'alpha': alpha, 'hit': hit, 'pix': pix, 'rdepth': rdepth, 'gerr': gerr,
                'rmask': ex['ray_mask'].reshape(R), 'gt': ex['pixel_colors_gt'].reshape(R, 3) if 'pixel_colors_gt' in ex else None,
                's': P[agg + 'deviation_network.variance'].reshape(1, 1)}, ex

    _SHARED_KEYS = ('s', 's_fine')                                         # one value for all scenes of a forward
    _PER_VIEW_KEYS = ('depth_mean', 'depth_coords', 'depth_mean_2', 'depth_mean_fine', 'depth_mean_fine_2')   # [rfn,...], no leading 1

    @staticmethod
    def _stacked(o, B, rn, suffix=''):
        """A pass's [B*rn, ...] results as scene-major stacks: the keys of renderer.py:110-138 with a leading B where the
        reference has its leading 1 (qn)."""
        un = lambda x: x.reshape(B, rn, *x.shape[1:])
        d = {'sdf_values': un(o['sdf']), 'alpha_values': un(o['alpha']), 'colors_nr': un(o['col']), 'hit_prob_nr': un(o['hit']),
             'pixel_colors_nr': un(o['pix']), 'sdf_gradient_error': o['gerr'], 's': o['s'], 'render_depth': un(o['rdepth']),
             'ray_mask': un(o['rmask'])}
        if o['gt'] is not None:
            d['pixel_colors_gt'] = un(o['gt'])
        return {k + suffix: v for k, v in d.items()}

    @classmethod
    def unstack(cls, st, B):
        """Scene-major stacks -> list of B per-scene output dicts with the reference's shapes (views of the stacks)."""
        one = lambda k, v, b: v if k in cls._SHARED_KEYS else v[b] if k in cls._PER_VIEW_KEYS else v[b:b + 1]
        return [{k: one(k, v, b) for k, v in st.items()} for b in range(B)]

    def _render_train(self, hot, prep, que_b, fine_u, ray_feats, img_feats):
        """renderer.py:140-162 + 201-220 for B scenes in training mode: ray chunks of ray_batch_num (the reference's loop,
        outputs concatenated along the ray axis, the [1,1] scalars become [1,n_chunks]); per chunk the coarse pass (coarse
        depths sampled, fine depths resampled from `fine_u` and sorted inside the forward kernels), then the fine pass.
        que_b: coords [B,rn,2], pose [B,3,4], K [B,3,3], depth_range [B,2] (+ imgs [B,3,H,W]); fine_u [B,rn,fdn] on the device.
        -> scene-major stacks ([B, ...], `unstack` makes the per-scene dicts) with '' and '_fine' keys."""
        P = self._params()
        B, rn = que_b['coords'].shape[:2]
        chunk = self.cfg['ray_batch_num']
        parts = []
        for r0 in range(0, rn, chunk):
            q = dict(que_b, coords=que_b['coords'][:, r0:r0 + chunk].contiguous())
            if fine_u is not None:
                q['fine_u'] = fine_u[:, r0:r0 + chunk].contiguous()
            n = q['coords'].shape[1]
            if not self.cfg['use_hierarchical_sampling']:                   # renderer.py:153-162: the coarse pass's outputs only
                coarse, ex = self._train_pass(hot, prep, q, None, 'coarse', False, ray_feats, img_feats, P)
                parts.append(self._stacked(coarse, B, n))
                continue
            coarse, ex = self._train_pass(hot, prep, q, None, 'coarse', True, ray_feats, img_feats, P)
            fine_depth = ex['fine_depth']
            if self.cfg['fine_depth_use_all']:                              # renderer.py:145-146: coarse and resampled depths together
                fine_depth = hot.merge_depths(ex['depth'].reshape(B, n, -1), fine_depth)     # (a rank merge of two ascending lists in HIP: no ATen arithmetic on the path)
            fine, _ = self._train_pass(hot, prep, q, fine_depth, 'fine', False, ray_feats, img_feats, P)
            parts.append(dict(self._stacked(coarse, B, n), **self._stacked(fine, B, n, '_fine')))
        st = {k: torch.cat([p[k] for p in parts], 1) for k in parts[0]} if len(parts) > 1 else parts[0]
        if not self.cfg['render_depth']:
            st.pop('render_depth', None), st.pop('render_depth_fine', None)
        if not self.cfg['use_ray_mask']:                                    # renderer.py:129-132
            st.pop('ray_mask', None), st.pop('ray_mask_fine', None)
        return st

    def _render_autograd(self, que, ref, _prep=None):
        """One scene in training mode (the reference's API): the is_train inverse-CDF samples are drawn exactly as the
        reference draws them (one torch.rand([1,chunk_rn,fdn]) on the CPU generator per ray chunk, render_ops.py:204-205),
        the NeuS step counters advance once per chunk (aggregate_net.py:135-137)."""
        self._need_gpu(ref['imgs'])
        hot, bref, prep = _prep if _prep is not None and len(_prep) == 3 else self._train_prep(ref, que['coords'].shape[1])
        rn, chunk, fdn = que['coords'].shape[1], self.cfg['ray_batch_num'], self.cfg['fine_depth_sample_num']
        us = []
        hier = bool(self.cfg['use_hierarchical_sampling'])                  # without it the reference neither draws (sample_fine_depth is
        for r0 in range(0, rn, chunk):                                      # not called) nor runs the fine aggregation net (its step stays)
            if hier:
                us.append(torch.rand([1, min(chunk, rn - r0), fdn]))
            for net in ((self.agg_net, self.fine_agg_net) if hier else (self.agg_net,)):
                net.train_step_bookkeeping()
        dev = ref['imgs'].device
        fine_u = self._upload('fine_u1', torch.cat(us, 1), dev) if hier else None
        bq = {'coords': que['coords'], 'pose': que['poses'], 'K': que['Ks'], 'depth_range': que['depth_range']}
        if 'imgs' in que:
            bq['imgs'] = que['imgs']
        return self._render_train(hot, prep, bq, fine_u, ref['ray_feats'][None], ref['img_feats'][None])    # B = 1: the stacks ARE the reference's shapes

    @staticmethod
    def draw_fine_u(rn, fdn, chunk):
        """The is_train inverse-CDF samples exactly as the reference draws them: one torch.rand([1,chunk_rn,fdn])
        on the default (CPU) generator per ray chunk, in chunk order (render_ops.py:204-205, renderer.py:207-209)."""
        return torch.cat([torch.rand([1, min(chunk, rn - r0), fdn]) for r0 in range(0, rn, chunk)], 1)

    # ---- the reference's methods ------------------------------------------------------------------
    def sample_volume(self, ref_imgs_info, _prep=None, is_train=False):     # renderer.py:164-199
        if self._use_autograd(is_train):
            self._need_gpu(ref_imgs_info['imgs'])
            hot, bref, prep = _prep if _prep is not None and len(_prep) == 3 else self._train_prep(ref_imgs_info)
            P = self._params()
            return _SampleVolumeFn.apply(hot, bref, prep, self.cfg['volume_resolution'], ref_imgs_info['ray_feats'][None],
                                         ref_imgs_info['img_feats'][None], *[P[k] for k, _ in _w.level_keys('coarse', use_vis=self.use_vis)])
        bref, prep = _prep or self._prepare(ref_imgs_info)
        if not self.cfg.get('warn_low_valid_ratio', False):
            return self.hot().sample_volume(bref, self.cfg['volume_resolution'], prepared=prep)
        # renderer.py:174-176: the reference reads the share of (view, voxel) pairs that project into their image back to the
        # host on every call and prints when it is below one half.  Opt-in here (cfg warn_low_valid_ratio): the read is a
        # host synchronisation (and not capturable in a hipGraph); the in-image bits come out of the chain kernel anyway.
        vol, vmask = self.hot().sample_volume(bref, self.cfg['volume_resolution'], want_mask=True, prepared=prep)
        V = bref['imgs'].shape[1]
        bits = sum(((vmask >> v) & 1).float() for v in range(V))
        valid_ratio = bits.reshape(vmask.shape[0], -1).sum(1) / float(V * vmask[0].numel())
        if float(valid_ratio.mean()) < 0.5:
            print("!! too low ratio", valid_ratio)
        return vol

    def _out_dict(self, o, suffix, level_net):
        keys = ['sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr']
        if self.cfg['use_ray_mask']:                                        # renderer.py:129-132
            keys.append('ray_mask')
        if 'pixel_colors_gt' in o:
            keys.append('pixel_colors_gt')
        if self.cfg['render_depth']:
            keys.append('render_depth')
        out = {k + suffix: o[k] for k in keys}
        # the reference renders chunks of ray_batch_num rays and concatenates one [1,1] value per chunk
        # (renderer.py:203-218); here all rays go in one launch and the per-chunk means come back as [1,n_chunks]
        ge = o['sdf_gradient_error'].reshape(1, -1)
        out['sdf_gradient_error' + suffix] = ge
        out['s' + suffix] = level_net.deviation_network.variance.reshape(1, 1).expand(1, ge.shape[1])
        return out

    def render(self, que_imgs_info, ref_imgs_info, is_train, _prep=None):   # renderer.py:201-220 (+140-162)
        rn = que_imgs_info['coords'].shape[1]
        if self._use_autograd(is_train):
            return self._render_autograd(que_imgs_info, ref_imgs_info, _prep)
        bref, prep = _prep or self._prepare(ref_imgs_info, rn)
        bque = self._batched_que(que_imgs_info)
        hier = bool(self.cfg['use_hierarchical_sampling'])
        if is_train:
            # forward values only (no autograd through the HIP path).  The reference draws the
            # inverse-CDF samples per chunk with torch.rand on the CPU generator (render_ops.py:204-208): same
            # draws, same order, so a seeded run samples the same fine depths.
            fdn, chunk = self.cfg['fine_depth_sample_num'], self.cfg['ray_batch_num']
            if hier:
                bque['fine_u'] = self.draw_fine_u(rn, fdn, chunk)
            for net in ((self.agg_net, self.fine_agg_net) if hier else (self.agg_net,)):          # aggregate_net.py:135-137 bookkeeping
                for _ in range((rn + chunk - 1) // chunk):
                    net.train_step_bookkeeping()
        co, fi = self.hot().render(bref, bque, self._render_cfg(), prepared=prep)
        out = self._out_dict(co, '', self.agg_net)
        if hier:                                                            # renderer.py:157-161
            out.update(self._out_dict(fi, '_fine', self.fine_agg_net))
        return out

    def render_by_depth(self, que_depth, que_imgs_info, ref_imgs_info, is_train, is_fine):   # renderer.py:110-138
        o = self.hot().render_by_depth(self._batched_ref(ref_imgs_info), self._batched_que(que_imgs_info), que_depth,
                                       'fine' if is_fine else 'coarse', self._render_cfg())
        return self._out_dict(o, '', self.fine_agg_net if is_fine else self.agg_net)

    def gen_depth_loss_coords(self, h, w, device, keep_on_host=False):      # renderer.py:222-228
        # default: CPU generator like the reference (same RNG stream -> identical coordinates for a given seed);
        # cfg['depth_coords_rng'] = 'device' draws on the GPU instead (randperm of 147 456 costs 5-15 ms on the host)
        gen_dev = device if self.cfg.get('depth_coords_rng', 'cpu') == 'device' else 'cpu'
        idx = None
        if gen_dev == 'cpu':
            idx = randperm_prefix(h * w, min(self.cfg['depth_loss_coords_num'], h * w))
        if idx is None:
            idx = torch.randperm(h * w, device=gen_dev)[:self.cfg['depth_loss_coords_num']]
        rc = torch.stack([idx // w, idx % w], -1)                           # (row, col)
        return rc if keep_on_host else rc.to(device)

    def predict_mean_for_depth_loss(self, ref_imgs_info, _prep=None, is_train=False):       # renderer.py:230-266
        h, w = ref_imgs_info['imgs'].shape[-2:]
        rfn = ref_imgs_info['imgs'].shape[0]
        coords = self.gen_depth_loss_coords(h, w, ref_imgs_info['imgs'].device)
        if self._use_autograd(is_train):
            P = self._params()
            self._need_gpu(ref_imgs_info['imgs'])
            # HIP forward + HIP backward behind an autograd.Function (csrc/gnr_bwd.inc)
            hot, bref, prep = _prep if _prep is not None and len(_prep) == 3 else self._train_prep(ref_imgs_info)
            xy = coords.to(torch.float32)[None]
            ms = [_DepthMeanFn.apply(hot, bref, prep, xy, lvl, ref_imgs_info['ray_feats'][None],
                                     *[P[_w.LEVELS[lvl][0] + 'mean_decoder.' + n] for n in _DM_PARAMS])[0] for lvl in self.levels]
            out = {'depth_mean': ms[0][..., 0], 'depth_coords': coords[None].repeat(rfn, 1, 1), 'depth_mean_2': ms[0][..., 1]}
            if len(ms) > 1:                                                 # renderer.py:247-251,259-264: the fine means only with hierarchical sampling
                out.update({'depth_mean_fine': ms[1][..., 0], 'depth_mean_fine_2': ms[1][..., 1]})
            return out
        # the reference feeds (row, col) where (x, y) is expected; kept
        xy = coords.to(torch.float32)[None]
        bref, prep = _prep or self._prepare(ref_imgs_info)
        hot = self.hot()
        mc = hot.depth_mean(bref, xy, 'coarse', prepared=prep)[0]
        out = {'depth_mean': mc[..., 0], 'depth_coords': coords[None].repeat(rfn, 1, 1), 'depth_mean_2': mc[..., 1]}
        if len(self.levels) > 1:
            mf = hot.depth_mean(bref, xy, 'fine', prepared=prep)[0]
            out.update({'depth_mean_fine': mf[..., 0], 'depth_mean_fine_2': mf[..., 1]})
        return out

    def forward(self, data):                                                # renderer.py:268-291
        ref = dict(data['ref_imgs_info'])
        que = dict(data['que_imgs_info'])
        is_train = 'eval' not in data
        if self._use_autograd(is_train) and ref['imgs'].is_cuda:
            self.repack_begin()                                             # finished in _train_prep, behind the backbones
        ref['img_feats'] = self.image_encoder(ref['imgs'])
        ref['ray_feats'] = self.init_net(ref, data.get('src_imgs_info'), is_train)
        ref['ray_feats'] = self.vis_encoder(ref['ray_feats'], ref['img_feats'])
        out = {}
        if self._use_autograd(is_train):
            self._need_gpu(ref['imgs'])
            prep = self._train_prep(ref, que['coords'].shape[1] if self.cfg['render_rgb'] else 0)
        else:
            prep = self._prepare(ref, que['coords'].shape[1] if self.cfg['render_rgb'] else 0)
        if self.cfg['render_rgb']:
            out = self.render(que, ref, is_train, _prep=prep)
        if self.cfg.get('sample_volume', False):
            out['volume'] = self.sample_volume(ref, _prep=prep, is_train=is_train)
        if (self.cfg.get('use_depth_loss', False) and 'true_depth' in ref) or (not is_train):
            out.update(self.predict_mean_for_depth_loss(ref, _prep=prep, is_train=is_train))
        return out


    def forward_scenes(self, datas, stacked=False):
        """Training forward of several scenes in ONE pass (trainer-internal; the reference's API is one scene per forward):
        batched backbones, one weight re-pack, batched HIP twin pairs, one per-ray tail over the rays of all scenes.  Same
        values and the same RNG stream as `[self.forward(d) for d in datas]` (which it falls back to whenever the batch is
        not uniform, off the GPU, or autograd is off).  -> list of per-scene output dicts (stacked=True: ONE dict of
        scene-major stacks, leading B where the reference has its leading 1; `unstack` gives the list), or None when the scenes cannot
        be batched (the caller must then run forward + backward scene by scene: the HIP twin pairs keep their saved states
        in per-module workspaces, so a second training forward before the first one's backward would overwrite them)."""
        c = self.cfg
        refs = [d['ref_imgs_info'] for d in datas]
        ques = [d['que_imgs_info'] for d in datas]
        same = all(r['imgs'].shape == refs[0]['imgs'].shape and q['coords'].shape == ques[0]['coords'].shape for r, q in zip(refs, ques))
        if not (len(datas) > 1 and same and all('eval' not in d for d in datas) and self._use_autograd(True) and refs[0]['imgs'].is_cuda
                and c['render_rgb'] and c.get('sample_volume', False) and ques[0]['coords'].shape[1] <= c['ray_batch_num']):
            return None
        B, V = len(datas), refs[0]['imgs'].shape[0]
        h, w = refs[0]['imgs'].shape[-2:]
        rn, fdn, R = ques[0]['coords'].shape[1], c['fine_depth_sample_num'], c['volume_resolution']
        dev = refs[0]['imgs'].device
        # The backbones are queued first; the weights are re-packed on the device behind them (hot_for_training: stream-ordered
        # kernels, nothing waits), and the reference's CPU random draws reach the device in one pinned copy each.
        imgs = torch.cat([r['imgs'] for r in refs])
        img_feats = self.image_encoder(imgs)
        ray_feats = self.vis_encoder(self.init_net({'imgs': imgs}, None, True), img_feats)
        img_feats, ray_feats = img_feats.reshape(B, V, *img_feats.shape[1:]), ray_feats.reshape(B, V, *ray_feats.shape[1:])
        hot = self.hot_for_training()
        want_depth = c.get('use_depth_loss', False) and 'true_depth' in refs[0]
        us, coords = [], []
        hier = bool(c['use_hierarchical_sampling'])
        for _ in range(B):                                                  # the per-scene draw order of forward()
            if hier:
                us.append(torch.rand([1, rn, fdn]))
            for net in ((self.agg_net, self.fine_agg_net) if hier else (self.agg_net,)):
                net.train_step_bookkeeping()
            if want_depth:
                coords.append(self.gen_depth_loss_coords(h, w, dev, keep_on_host=True))
        fine_u = self._upload('fine_u', torch.cat(us), dev) if hier else None
        if want_depth:
            coords = torch.stack(coords)                                    # [B,8192,2]
            coords = coords if coords.is_cuda else self._upload('depth_coords', coords, dev)
        stack = lambda k, src: torch.stack([torch.as_tensor(x[k], dtype=torch.float32, device=dev) for x in src])
        bref = {'imgs': imgs.reshape(B, V, 3, h, w), 'img_feats': img_feats.detach(), 'ray_feats': ray_feats.detach(),
                'poses': stack('poses', refs), 'Ks': stack('Ks', refs), 'depth_range': stack('depth_range', refs),
                'bbox3d': stack('bbox3d', refs)}
        prep = hot.prepare(bref, R, rn, self._dn_max())
        P = self._params()
        que_b = {'coords': torch.cat([q['coords'] for q in ques]), 'pose': torch.cat([q['poses'] for q in ques]),
                 'K': torch.cat([q['Ks'] for q in ques]), 'depth_range': torch.cat([q['depth_range'] for q in ques])}
        if 'imgs' in ques[0]:
            que_b['imgs'] = torch.cat([q['imgs'] for q in ques])
        st = self._render_train(hot, prep, que_b, fine_u, ray_feats, img_feats)
        st['volume'] = _SampleVolumeFn.apply(hot, bref, prep, R, ray_feats, img_feats, *[P[k] for k, _ in _w.level_keys('coarse', use_vis=self.use_vis)])
        if want_depth:
            xy = coords.to(torch.float32)
            ms = [_DepthMeanFn.apply(hot, bref, prep, xy, lvl, ray_feats, *[P[_w.LEVELS[lvl][0] + 'mean_decoder.' + n] for n in _DM_PARAMS])
                  for lvl in self.levels]
            st.update({'depth_mean': ms[0][..., 0], 'depth_coords': coords[:, None].expand(B, V, *coords.shape[1:]), 'depth_mean_2': ms[0][..., 1]})
            if len(ms) > 1:
                st.update({'depth_mean_fine': ms[1][..., 0], 'depth_mean_fine_2': ms[1][..., 1]})
        return st if stacked else self.unstack(st, B)


class GraspNeRF(nn.Module):
    default_cfg_vgn = {'nr_initial_training_steps': 0, 'freeze_nr_after_init': False}

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg_vgn, **cfg}
        if self.cfg['nr_initial_training_steps'] or self.cfg['freeze_nr_after_init']:
            # renderer.py:314-321: both branches call `super().forward(data)` = nn.Module.forward, i.e. they raise in the
            # reference too (GraspNeRF does not derive from the renderer); nrvgn_sdf.yaml leaves them at their defaults
            raise NotImplementedError('nr_initial_training_steps / freeze_nr_after_init: dead branches of the reference '
                                      '(renderer.py:314-321 call nn.Module.forward), not built')
        self.nr_net = NeuralRayRenderer(self.cfg)
        self.vgn_net = ConvNet()                                            # gd.networks.get_network("conv"): parameters
        self._head = None                                                   # HIP kernels built from vgn_net's weights

    def _apply(self, fn, *a, **k):
        self._head = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._head = None
        return super().load_state_dict(*a, **k)

    def grasp_head(self, volume):
        """gd.networks.ConvNet.forward: HIP implicit-GEMM kernels for inference on the GPU (csrc/gnr_head.hip);
        the PyTorch module (same parameters) when autograd is needed."""
        if volume.is_cuda and not torch.is_grad_enabled():
            # the packed copy follows the parameters: optimizer.step() updates them in place (version counters move)
            ver = tuple((p._version, p.data_ptr()) for p in self.vgn_net.parameters())
            if self._head is None or self._head_ver != ver:
                self._head = GraspHead(self.vgn_net.state_dict(), device=volume.device)
                self._head_ver = ver
            return self._head(volume)
        return self.vgn_net(volume)

    @staticmethod
    def select(out, index):                                                 # renderer.py:305-311
        qual, rot, width = out
        b = torch.arange(qual.shape[0], device=qual.device)                 # on the device: a host index tensor is a blocking copy
        i, j, k = index[:, 0], index[:, 1], index[:, 2]
        return qual[b, :, i, j, k].squeeze(), rot[b, :, i, j, k], width[b, :, i, j, k].squeeze()

    def forward(self, data):                                                # renderer.py:313-331
        render_outputs = self.nr_net(data)
        vgn_pred = self.grasp_head(render_outputs['volume'])
        render_outputs['vgn_pred'] = vgn_pred if 'full_vol' in data else self.select(vgn_pred, data['grasp_info'][0])
        return render_outputs

    def forward_scenes(self, datas, stacked=False):
        """Several scenes in one training forward (NeuralRayRenderer.forward_scenes + one batched grasp-head call).
        stacked=True: one dict of scene-major stacks, 'vgn_pred' = (label [B,N], rot [B,N,4], width [B,N]) gathered at every
        scene's GT voxels in one indexing call; declined (None, before anything runs) when a scene wants the full volumes or
        the scenes carry different numbers of grasps."""
        if stacked and (any('full_vol' in d for d in datas) or len({tuple(d['grasp_info'][0].shape) for d in datas}) != 1):
            return None
        st = self.nr_net.forward_scenes(datas, stacked=True)
        if st is None:
            return None
        B = len(datas)
        q, r, w = self.grasp_head(st['volume'])
        if stacked:
            idx = torch.stack([d['grasp_info'][0] for d in datas])          # [B,N,3]
            b = torch.arange(B, device=q.device)[:, None]
            i, j, k = idx[..., 0], idx[..., 1], idx[..., 2]
            st['vgn_pred'] = (q[b, 0, i, j, k], r[b, :, i, j, k], w[b, 0, i, j, k])
            return st
        outs = self.nr_net.unstack(st, B)
        for b, (o, d) in enumerate(zip(outs, datas)):
            pred = (q[b:b + 1], r[b:b + 1], w[b:b + 1])
            o['vgn_pred'] = pred if 'full_vol' in d else self.select(pred, d['grasp_info'][0])
        return outs


name2network = {'grasp_nerf': GraspNeRF}
