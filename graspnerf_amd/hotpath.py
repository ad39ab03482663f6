"""Batched device front-end of libgnr.so.  PyTorch is used for device memory and streams only;
every number on the hot path is produced by the HIP kernels behind the C ABI (include/gnr.h).

Tensors are float32, contiguous, on one `cuda` device, with a leading scene-batch dim B:
  ref : imgs[B,V,3,H,W] img_feats[B,V,32,fh,fw] ray_feats[B,V,32,fh,fw] poses[B,V,3,4] Ks[B,V,3,3]
        depth_range[B,V,2] bbox3d[B,2,3]
  que : coords[B,rn,2] pose[B,3,4] K[B,3,3] depth_range[B,2] (imgs[B,3,H,W])
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import GnrScene, GnrRays, GnrRenderOut

RENDER_KEYS = ['sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr', 'pixel_colors_gt',
               'render_depth', 'ray_mask', 'sdf_gradient_error']


def _f32(t, device):
    if not torch.is_tensor(t):
        t = torch.as_tensor(np.asarray(t))
    return t.to(device=device, dtype=torch.float32).contiguous()


class HotPath:
    default_options = 0          # options a new HotPath starts with (a Python-side default for test fixtures; the library itself keeps no switch)

    def __init__(self, packed_coarse, packed_fine=None, device='cuda:0'):
        self.L = _lib.lib()                      # raises if the HIP library is missing
        if not torch.cuda.is_available():
            raise _lib.GnrError('graspnerf_amd hot path needs a ROCm GPU (torch.cuda.is_available() is False); '
                                'there is no CPU fallback')
        self.device = torch.device(device)
        self.wc = torch.from_numpy(np.ascontiguousarray(packed_coarse, np.float32)).to(self.device)
        self.wf = None if packed_fine is None else torch.from_numpy(np.ascontiguousarray(packed_fine, np.float32)).to(self.device)
        # cfg use_vis: the packed blobs say whether they carry the fourth decoder branch (gnr_pack_vis_decoder sets the flag)
        flag = self.L.gnr_layout_offset(b'T_VIS') + 1
        self.use_vis = bool(np.asarray(packed_coarse).reshape(-1)[flag] != 0)
        if packed_fine is not None and bool(np.asarray(packed_fine).reshape(-1)[flag] != 0) != self.use_vis:
            raise _lib.GnrError('use_vis must be the same for both levels (the reference evaluates the fine level with the coarse '
                                "decoder's compute_prob, renderer.py:70-72)")
        self.options = HotPath.default_options   # per-call options of this HotPath's calls (set_option; include/gnr.h GNR_OPT_*)
        self._ws = None
        self._keep = []

    # ---- plumbing ------------------------------------------------------------------------
    def _stream(self):
        # the library launches on the CURRENT device: refuse a mismatch instead of launching on the wrong GPU
        if self.device.index is not None and self.device.index != torch.cuda.current_device():
            raise _lib.GnrError(f'HotPath on {self.device} but the current device is cuda:{torch.cuda.current_device()}: '
                                f'wrap the call in `with torch.cuda.device({self.device.index})`')
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _sc(self, scene):
        """byref(scene) with this HotPath's per-call options (GnrScene.options, include/gnr.h GNR_OPT_*): the library keeps no switch,
        so the options travel with every call; a prepared scene that is re-used picks up the current ones."""
        scene.options = self.options
        return C.byref(scene)

    def set_option(self, name, on=True):
        """Switch one of the per-call options (_lib.OPTIONS: fp32_chain, feature_grad_fixed, view1_one_wavefront, view2_one_wavefront,
        ray_order_morton, poison_partials, direct_scatter, geo_dual_fp32, test_lose_partner, static_tiles) for the calls of THIS HotPath; -> previous."""
        bit = _lib.OPTIONS[name]
        prev = bool(self.options & bit)
        self.options = (self.options | bit) if on else (self.options & ~bit)
        return prev

    def _scene(self, ref):
        d = self.device
        t = {k: _f32(ref[k], d) for k in ('imgs', 'img_feats', 'ray_feats', 'poses', 'Ks', 'depth_range')}
        B, V, _, H, W = t['imgs'].shape
        fh, fw = t['img_feats'].shape[-2:]
        assert t['img_feats'].shape == (B, V, 32, fh, fw) and t['ray_feats'].shape == (B, V, 32, fh, fw)
        assert t['poses'].shape == (B, V, 3, 4) and t['Ks'].shape == (B, V, 3, 3) and t['depth_range'].shape == (B, V, 2)
        s = GnrScene(B, V, H, W, fh, fw, t['imgs'].data_ptr(), t['img_feats'].data_ptr(), t['ray_feats'].data_ptr(),
                     t['poses'].data_ptr(), t['Ks'].data_ptr(), t['depth_range'].data_ptr(), 1 if self.use_vis else 0, self.options)
        return s, t

    def _workspace(self, scene, res, rn, dn):
        need = max(self.L.gnr_workspace_bytes(self._sc(scene), res, rn, dn), 0 if self._ws is None else self._ws.numel())
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _scratch(self, name, nbytes):
        """Persistent per-purpose scratch of the backward twins (the partial sums of their deterministic parameter-gradient
        reductions; stream-ordered, so one buffer per purpose serves every call)."""
        bufs = self.__dict__.setdefault('_scratch_bufs', {})
        if name not in bufs or bufs[name].numel() < nbytes:
            bufs[name] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
        return bufs[name]

    generation = 0

    def check_generation(self, gen):
        """The backward twins recompute from the prepared feature maps and the saved states in this object's workspaces:
        a later prepare() (= another forward) before the backward would silently corrupt them."""
        if gen != self.generation:
            raise _lib.GnrError('HIP backward after the workspaces were re-prepared: run forward and backward of a scene '
                                '(or of one batch of scenes) before the next training forward')

    def prepare(self, ref, res=40, rn=0, dn=0):
        """Repack feature maps + per-view projection blocks (timed part of a forward)."""
        self.generation += 1
        self._pass_seq = 0                               # render passes of this forward, counted for their persistent training workspaces
        scene, keep = self._scene(ref)
        ws = self._workspace(scene, res, rn, dn)
        _lib.check(self.L.gnr_prepare(self._sc(scene), ws.data_ptr(), ws.numel(), self._stream()), 'gnr_prepare')
        self._prepared = (scene, keep, ws)
        return self._prepared

    def range_status(self, prepared=None):
        """Watch word of the range guard for the last prepare() (include/gnr.h gnr_range_status; synchronises the stream):
        0 = every chain launch ran in the fp16-pair form; bit 0 / bit 1 = a feature / an activation left its range and the
        launches were recomputed by the fp32-MFMA twin."""
        scene, keep, ws = prepared or self._prepared
        flags = C.c_uint(0)
        _lib.check(self.L.gnr_range_status(self._sc(scene), ws.data_ptr(), ws.numel(), C.byref(flags), self._stream()), 'gnr_range_status')
        return int(flags.value)

    def status_words(self, prepared=None):
        """The 64 status words of the prepared scene as an int32 DEVICE view of the workspace (no copy, no synchronisation):
        `(words & 16).any()` = GNR_STATUS_LOST_PARTNER of the backward calls since that prepare (include/gnr.h)."""
        scene, keep, ws = prepared or self._prepared
        off = self.L.gnr_status_words_offset(C.byref(scene))
        return ws[off:off + 256].view(torch.int32)

    def feature_grad_mode(self, fixed_point):
        """Feature-map gradients of this HotPath's backward calls through 64-bit fixed-point adds (bit-reproducible) instead of float
        sums in arrival order (GNR_OPT_FEATURE_GRAD_FIXED; -> previous setting).  range_status() bit 3 tells a clamped contribution."""
        return self.set_option('feature_grad_fixed', fixed_point)

    def force_fp32_chain(self, on):
        """Test / measurement switch: every chain launch of this HotPath -- forward and backward -- on the fp32-input MFMA
        (GNR_OPT_FP32_CHAIN; -> previous setting)."""
        return self.set_option('fp32_chain', on)

    # ---- sample_volume (ref: renderer.py:164-199) ------------------------------------------
    def sample_volume(self, ref, res=40, want_mask=False, prepared=None):
        scene, keep, ws = prepared or self.prepare(ref, res)
        B = scene.B
        bbox_min = _f32(ref['bbox3d'], self.device)[:, 0].contiguous()
        vol = torch.empty(B, 1, res, res, res, dtype=torch.float32, device=self.device)
        vmask = torch.empty(B, res, res, res, dtype=torch.uint8, device=self.device) if want_mask else None
        _lib.check(self.L.gnr_sample_volume_fwd(self._sc(scene), bbox_min.data_ptr(), res, self.wc.data_ptr(),
                                                vol.data_ptr(), vmask.data_ptr() if want_mask else None,
                                                ws.data_ptr(), ws.numel(), self._stream()), 'gnr_sample_volume_fwd')
        return (vol, vmask) if want_mask else vol

    def debug_volume_chain(self, ref, res=40, prepared=None):
        scene, keep, ws = prepared or self.prepare(ref, res)
        bbox_min = _f32(ref['bbox3d'], self.device)[:, 0].contiguous()
        dbg = torch.zeros(scene.B, res ** 3, 32, dtype=torch.float32, device=self.device)
        _lib.check(self.L.gnr_debug_volume_chain(self._sc(scene), bbox_min.data_ptr(), res, self.wc.data_ptr(),
                                                 dbg.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                   'gnr_debug_volume_chain')
        return dbg

    # ---- render (ref: renderer.py:140-162, 201-220) ----------------------------------------
    def _rays(self, que, dn, fdn, cfg, H, W):
        d = self.device
        t = {'coords': _f32(que['coords'], d), 'pose': _f32(que['pose'], d), 'K': _f32(que['K'], d),
             'depth_range': _f32(que['depth_range'], d)}
        B, rn, _ = t['coords'].shape
        imgs = _f32(que['imgs'], d) if 'imgs' in que else None
        if imgs is not None:
            assert imgs.shape == (B, 3, H, W)
            t['imgs'] = imgs
        fine_u = None
        if que.get('fine_u') is not None:            # is_train: caller-drawn inverse-CDF samples (render_ops.py:204-205)
            fine_u = t['fine_u'] = _f32(que['fine_u'], d)
            assert fine_u.shape == (B, rn, fdn)
        chunk = int(cfg.get('ray_batch_num', 0) or 0)
        r = GnrRays(rn, dn, fdn, cfg.get('ray_mask_view_num', 2), cfg.get('ray_mask_point_num', 8),
                    t['coords'].data_ptr(), t['pose'].data_ptr(), t['K'].data_ptr(), t['depth_range'].data_ptr(),
                    imgs.data_ptr() if imgs is not None else None,
                    fine_u.data_ptr() if fine_u is not None else None, chunk, 1 if cfg.get('fine_depth_use_all', False) else 0)
        return r, t

    def _alloc_out(self, B, rn, dn, with_gt, debug, chunk=0):
        d = self.device
        nch = (rn + chunk - 1) // chunk if chunk > 0 else 1
        o = {'depth': torch.empty(B, rn, dn, device=d), 'sdf_values': torch.empty(B, rn, dn, device=d),
             'alpha_values': torch.empty(B, rn, dn, device=d), 'colors_nr': torch.empty(B, rn, dn, 3, device=d),
             'hit_prob_nr': torch.empty(B, rn, dn, device=d), 'pixel_colors_nr': torch.empty(B, rn, 3, device=d),
             'render_depth': torch.empty(B, rn, device=d), 'ray_mask': torch.empty(B, rn, dtype=torch.uint8, device=d),
             'sdf_gradient_error': torch.empty(B, nch, device=d) if chunk > 0 else torch.empty(B, device=d)}
        if with_gt:
            o['pixel_colors_gt'] = torch.empty(B, rn, 3, device=d)
        if debug:
            o['sdf_gradient'] = torch.empty(B, rn, dn, 3, device=d)
            o['view_mask'] = torch.empty(B, rn, dn, dtype=torch.uint8, device=d)
        s = GnrRenderOut(*[o[k].data_ptr() if k in o else None for k in _lib.RENDER_OUT_FIELDS])
        return s, o

    def render(self, ref, que, cfg=None, fine_depth_in=None, debug=False, prepared=None):
        """-> (coarse dict, fine dict[, fine_inds]) of device tensors with a leading batch dim.  cfg use_hierarchical_sampling false
        (renderer.py:153-162): the coarse pass only, fine dict = None."""
        cfg = cfg or {}
        dn, fdn = cfg.get('depth_sample_num', 40), cfg.get('fine_depth_sample_num', 40)
        if not cfg.get('use_hierarchical_sampling', True):
            if fine_depth_in is not None or debug:
                raise _lib.GnrError('render(): fine depths / resampling indices need use_hierarchical_sampling')
            B, rn = que['coords'].shape[:2]
            scene, keep, ws = prepared or self.prepare(ref, 1, rn, dn)
            rays, rkeep = self._rays(que, dn, fdn, cfg, scene.H, scene.W)
            co_s, co = self._alloc_out(B, rn, dn, 'imgs' in que, False, rays.ray_batch_num)
            _lib.check(self.L.gnr_render_rays_fwd(self._sc(scene), C.byref(rays), self.wc.data_ptr(), None, C.byref(co_s), None, None, None,
                                                  ws.data_ptr(), ws.numel(), self._stream()), 'gnr_render_rays_fwd')
            co['ray_mask'] = co['ray_mask'].view(torch.bool)
            return co, None
        if self.wf is None:
            raise _lib.GnrError('render() needs the fine-level weights')
        B, rn = que['coords'].shape[:2]
        # fine_depth_use_all (renderer.py:145-146): the fine pass renders the coarse and the resampled depths together
        fine_dn = dn + fdn if cfg.get('fine_depth_use_all', False) else fdn
        scene, keep, ws = prepared or self.prepare(ref, 1, rn, max(dn, fine_dn))
        rays, rkeep = self._rays(que, dn, fdn, cfg, scene.H, scene.W)
        co_s, co = self._alloc_out(B, rn, dn, 'imgs' in que, debug, rays.ray_batch_num)
        fi_s, fi = self._alloc_out(B, rn, fine_dn, 'imgs' in que, debug, rays.ray_batch_num)
        fd_in = _f32(fine_depth_in, self.device) if fine_depth_in is not None else None
        inds = torch.empty(B, rn, fdn, dtype=torch.int32, device=self.device) if debug else None
        _lib.check(self.L.gnr_render_rays_fwd(self._sc(scene), C.byref(rays), self.wc.data_ptr(), self.wf.data_ptr(),
                                              C.byref(co_s), C.byref(fi_s),
                                              fd_in.data_ptr() if fd_in is not None else None,
                                              inds.data_ptr() if debug else None,
                                              ws.data_ptr(), ws.numel(), self._stream()), 'gnr_render_rays_fwd')
        for o in (co, fi):
            o['ray_mask'] = o['ray_mask'].view(torch.bool)       # the kernels write 0 / 1 bytes: a view, not a conversion kernel
        return (co, fi, inds) if debug else (co, fi)

    def render_by_depth(self, ref, que, depth, level='coarse', cfg=None, debug=False, prepared=None):
        cfg = cfg or {}
        depth = _f32(depth, self.device)
        B, rn, dn = depth.shape
        scene, keep, ws = prepared or self.prepare(ref, 1, rn, dn)
        rays, rkeep = self._rays(que, dn, dn, cfg, scene.H, scene.W)
        o_s, o = self._alloc_out(B, rn, dn, 'imgs' in que, debug, rays.ray_batch_num)
        w = self.wc if level == 'coarse' else self.wf
        _lib.check(self.L.gnr_render_by_depth_fwd(self._sc(scene), C.byref(rays), depth.data_ptr(), dn, w.data_ptr(),
                                                  C.byref(o_s), ws.data_ptr(), ws.numel(), self._stream()),
                   'gnr_render_by_depth_fwd')
        o['ray_mask'] = o['ray_mask'].view(torch.bool)       # the kernels write 0 / 1 bytes: a view, not a conversion kernel
        return o

    def merge_depths(self, a, b):
        """sort(cat([a, b], -1), -1) of two per-ray ascending depth lists [..., na], [..., nb] on the device (gnr_merge_depths;
        renderer.py:145-146, fine_depth_use_all under training)."""
        a, b = _f32(a, self.device), _f32(b, self.device)
        assert a.shape[:-1] == b.shape[:-1]
        out = torch.empty(*a.shape[:-1], a.shape[-1] + b.shape[-1], dtype=torch.float32, device=self.device)
        _lib.check(self.L.gnr_merge_depths(a.data_ptr(), a.shape[-1], b.data_ptr(), b.shape[-1], out.data_ptr(), a.numel() // a.shape[-1],
                                           self._stream()), 'gnr_merge_depths')
        return out

    def depth_mean(self, ref, coords, level='coarse', prepared=None):   # noqa: D401
        """predict_mean_for_depth_loss for one level: coords [B,pn,2] (x,y) -> mean [B,V,pn,2]."""
        scene, keep, ws = prepared or self.prepare(ref, 1)
        coords = _f32(coords, self.device)
        B, pn, _ = coords.shape
        out = torch.empty(B, scene.V, pn, 2, dtype=torch.float32, device=self.device)
        w = self.wc if level == 'coarse' else self.wf
        _lib.check(self.L.gnr_depth_mean_fwd(self._sc(scene), coords.data_ptr(), pn, w.data_ptr(), out.data_ptr(),
                                             ws.data_ptr(), ws.numel(), self._stream()), 'gnr_depth_mean_fwd')
        return out

    def depth_mean_bwd(self, ref, coords, dmean, level='coarse', prepared=None, want_feat_grad=True):
        """Backward of depth_mean: dmean [B,V,pn,2] -> (d_canonical [36958] gradient blob of the level in state-dict
        order (mean_decoder entries), d_ray_feats [B,V,32,fh,fw] or None).  Packed weights must be current
        (`set_bwd_weights`)."""
        if getattr(self, 'wb', None) is None or self.wb.get(level) is None:
            raise _lib.GnrError('depth_mean_bwd: call set_bwd_weights() first')
        scene, keep, ws = prepared or self.prepare(ref, 1)
        need = self.L.gnr_depth_mean_bwd_workspace_bytes(self._sc(scene))
        if getattr(self, '_dm_scratch', None) is None or self._dm_scratch.numel() < need:
            self._dm_scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        coords = _f32(coords, self.device)
        dmean = _f32(dmean, self.device)
        B, pn, _ = coords.shape
        assert dmean.shape == (B, scene.V, pn, 2)
        dcan = torch.zeros(self.L.gnr_canonical_weights_floats(), dtype=torch.float32, device=self.device)
        dray = torch.empty(B, scene.V, 32, scene.fh, scene.fw, dtype=torch.float32, device=self.device) if want_feat_grad else None
        w = self.wc if level == 'coarse' else self.wf
        _lib.check(self.L.gnr_depth_mean_bwd(self._sc(scene), coords.data_ptr(), pn, w.data_ptr(), self.wb[level].data_ptr(),
                                             dmean.data_ptr(), dcan.data_ptr(), dray.data_ptr() if want_feat_grad else None,
                                             ws.data_ptr(), ws.numel(), self._dm_scratch.data_ptr(), self._dm_scratch.numel(),
                                             self._stream()), 'gnr_depth_mean_bwd')
        return dcan, dray

    # ---- sample_volume for training: forward with saved states + staged backward (csrc/gnr_bwd.inc) --------------
    def sample_volume_train(self, ref, res=40, prepared=None):
        scene, keep, ws = prepared or self.prepare(ref, res)
        need = self.L.gnr_sample_volume_train_workspace_bytes(self._sc(scene), res)
        if getattr(self, '_tws', None) is None or self._tws.numel() < need:
            self._tws = torch.empty(need, dtype=torch.uint8, device=self.device)
        bbox_min = _f32(ref['bbox3d'], self.device)[:, 0].contiguous()
        vol = torch.empty(scene.B, 1, res, res, res, dtype=torch.float32, device=self.device)
        _lib.check(self.L.gnr_sample_volume_fwd_train(self._sc(scene), bbox_min.data_ptr(), res, self.wc.data_ptr(), vol.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), self._tws.data_ptr(), self._tws.numel(),
                                                      self._stream()), 'gnr_sample_volume_fwd_train')
        self._train_ctx = (scene, keep, ws, res, self._tws)     # the saved states travel with the context: a release_training_workspaces()
        return vol                                               # (net.eval()) between this forward and its backward cannot take them away

    def train_ws_section(self, name, scene, res):
        """Float view of one section of the training workspace (tests): save1 save2 saveG dg16 dS2 dG dS1 dfeat64 dtail."""
        names = ['save1', 'save2', 'saveG', 'dg16', 'dS2', 'dG', 'dS1', 'dfeat64', 'dtail']
        off = (C.c_size_t * 9)()
        _lib.check(self.L.gnr_train_workspace_layout(self._sc(scene), res, off), 'gnr_train_workspace_layout')
        i = names.index(name)
        end = off[i + 1] if i + 1 < 9 else self._tws.numel()
        return self._tws[off[i]:end].view(torch.float32)

    def sample_volume_bwd(self, dvol, canonical_dev, stages=0x1f, want_feat_grads=True):
        """Backward of the last sample_volume_train: dvol [B,1,R,R,R] -> (d_canonical [36958], d_ray_feats, d_img_feats)."""
        scene, keep, ws, res, tws = self._train_ctx
        dvol = _f32(dvol, self.device)
        dcan = torch.zeros(self.L.gnr_canonical_weights_floats() + (self.L.gnr_canonical_vis_floats() if self.use_vis else 0), dtype=torch.float32, device=self.device)
        shp = (scene.B, scene.V, 32, scene.fh, scene.fw)
        dray = torch.zeros(shp, dtype=torch.float32, device=self.device) if want_feat_grads else None
        dimg = torch.zeros(shp, dtype=torch.float32, device=self.device) if want_feat_grads else None
        _lib.check(self.L.gnr_sample_volume_bwd(self._sc(scene), res, self.wc.data_ptr(), self.wb['coarse'].data_ptr(),
                                                canonical_dev.data_ptr(), dvol.data_ptr(), dcan.data_ptr(),
                                                dray.data_ptr() if want_feat_grads else None,
                                                dimg.data_ptr() if want_feat_grads else None, ws.data_ptr(), ws.numel(),
                                                tws.data_ptr(), tws.numel(), stages, self._stream()),
                   'gnr_sample_volume_bwd')
        return dcan, dray, dimg

    # ---- render path for training: the per-view chain of one pass in both directions (csrc/gnr_bwd.inc) ---------
    def render_chain_train(self, que, depth, level, cfg, prepared):
        """que: batched ray dict (coords [B,rn,2], pose, K, depth_range[, fine_u]); depth [B,rn,dn], or None for the coarse
        pass (sample_depth on the device, render_ops.py:146-170) -> (stats [B,rn*dn,66], colours [B,rn*dn,3], the pass's
        ray geometry {depth [B,rn,dn], pts [B*rn*dn,3], qdir [B*rn,3]}, ctx for render_tail_train / render_chain_bwd).
        The training workspace of the pass travels in ctx."""
        scene, keep, ws = prepared
        dn0, fdn = cfg.get('depth_sample_num', 40), cfg.get('fine_depth_sample_num', 40)
        B, rn = que['coords'].shape[:2]
        if depth is not None:
            depth = _f32(depth, self.device)
            assert depth.shape[:2] == (B, rn)
        dn = dn0 if depth is None else depth.shape[2]
        rays, rkeep = self._rays(que, dn0, fdn, cfg, scene.H, scene.W)
        need = self.L.gnr_workspace_bytes(self._sc(scene), 1, rn, dn)
        if ws.numel() < need:
            raise _lib.GnrError('render_chain_train: prepare() the workspace for the ray count first')
        # the pass's training workspace (saved per-view states + gradient staging: ~1 GB at 8 scenes x 512 rays x 40 samples) is kept
        # per (level, shape) across steps: allocating and releasing gigabyte blocks every step next to the feature extractors'
        # activations left the caching allocator re-carving its pool, which showed up as occasional 50-80 ms host stalls inside the
        # forward.  Keyed by the pass's position in its forward (every ray chunk's passes are alive until the backward) and its
        # shape; safe to re-use across forwards: a training forward before the previous one's backward is refused (check_generation).
        need_t = self.L.gnr_render_chain_train_workspace_bytes(self._sc(scene), rn, dn)
        pool = self.__dict__.setdefault('_pass_tws', {})
        seq = self._pass_seq = getattr(self, '_pass_seq', 0) + 1
        key = (seq, level)                               # one buffer per pass position, grown in place when a shape needs more
        if key not in pool or pool[key].numel() < need_t:
            pool[key] = None                             # drop the smaller buffer before allocating its replacement
            pool[key] = torch.empty(need_t, dtype=torch.uint8, device=self.device)
        tws = pool[key]
        stats = torch.empty(B, rn * dn, 66, dtype=torch.float32, device=self.device)
        colors = torch.empty(B, rn * dn, 3, dtype=torch.float32, device=self.device)
        geo = {'depth': torch.empty(B, rn, dn, dtype=torch.float32, device=self.device) if depth is None else depth,
               'pts': torch.empty(B * rn * dn, 3, dtype=torch.float32, device=self.device),
               'qdir': torch.empty(B * rn, 3, dtype=torch.float32, device=self.device)}
        w = self.wc if level == 'coarse' else self.wf
        _lib.check(self.L.gnr_render_chain_fwd_train(self._sc(scene), C.byref(rays), None if depth is None else depth.data_ptr(), dn,
                                                     w.data_ptr(), stats.data_ptr(), colors.data_ptr(),
                                                     geo['depth'].data_ptr() if depth is None else None, geo['pts'].data_ptr(),
                                                     geo['qdir'].data_ptr(), ws.data_ptr(), ws.numel(), tws.data_ptr(), tws.numel(),
                                                     self._stream()), 'gnr_render_chain_fwd_train')
        return stats, colors, geo, (scene, keep, ws, tws, rn, dn, level, rays, rkeep)

    def render_chain_bwd(self, ctx, dstats, dcolors):
        """-> (d_canonical [36958] of the pass's level, d_ray_feats, d_img_feats [B,V,32,fh,fw])."""
        scene, keep, ws, tws, rn, dn, level = ctx[:7]
        dstats = _f32(dstats, self.device)
        dcolors = _f32(dcolors, self.device)
        assert dstats.shape[-1] == 65 and dcolors.shape[-1] == 3
        dcan = torch.zeros(self.L.gnr_canonical_weights_floats() + (self.L.gnr_canonical_vis_floats() if self.use_vis else 0), dtype=torch.float32, device=self.device)
        shp = (scene.B, scene.V, 32, scene.fh, scene.fw)
        dray = torch.empty(shp, dtype=torch.float32, device=self.device)
        dimg = torch.empty(shp, dtype=torch.float32, device=self.device)
        w = self.wc if level == 'coarse' else self.wf
        _lib.check(self.L.gnr_render_chain_bwd(self._sc(scene), rn, dn, w.data_ptr(), self.wb[level].data_ptr(), dstats.data_ptr(),
                                               dcolors.data_ptr(), dcan.data_ptr(), dray.data_ptr(), dimg.data_ptr(), ws.data_ptr(),
                                               ws.numel(), tws.data_ptr(), tws.numel(), self._stream()), 'gnr_render_chain_bwd')
        return dcan, dray, dimg

    # ---- per-ray tail of a training render pass (k_ray<true> forward, k_ray_dual_bwd backward) -----------------
    def render_tail_train(self, ctx, que, depth, colors, want_fine_depth=False):
        """Right after render_chain_train (same `ctx`): the per-ray tail, NeuS alpha and compositing of the pass (k_ray<true>).
        -> dict of device tensors: sdf_values, alpha_values, hit_prob_nr [B,rn,dn], sdf_gradient [B,rn,dn,3],
        pixel_colors_nr [B,rn,3], render_depth [B,rn], ray_mask [B,rn] bool, sdf_gradient_error [B,n_chunks]
        (+ pixel_colors_gt [B,rn,3] when `que` carries the query images; + fine_depth [B,rn,fdn], the sorted inverse-CDF
        resampling of this pass on que['fine_u'] / the eval midpoints, render_ops.py:172-229, when want_fine_depth)."""
        scene, keep, ws, tws, rn, dn, level, rays, rkeep = ctx
        depth = _f32(depth, self.device)
        o_s, o = self._alloc_out(scene.B, rn, dn, 'imgs' in que, True, rays.ray_batch_num or rn)
        o['colors_nr'] = _f32(colors, self.device).reshape(scene.B, rn, dn, 3)
        o_s.colors_nr = o['colors_nr'].data_ptr()
        o_s.depth = None
        o_s.view_mask = None
        fd = torch.empty(scene.B, rn, rays.fdn, dtype=torch.float32, device=self.device) if want_fine_depth else None
        w = self.wc if level == 'coarse' else self.wf
        _lib.check(self.L.gnr_render_tail_fwd_train(self._sc(scene), C.byref(rays), depth.data_ptr(), dn, w.data_ptr(), C.byref(o_s),
                                                    fd.data_ptr() if want_fine_depth else None, ws.data_ptr(), ws.numel(),
                                                    tws.data_ptr(), tws.numel(), self._stream()), 'gnr_render_tail_fwd_train')
        o['ray_mask'] = o['ray_mask'].view(torch.bool)       # the kernels write 0 / 1 bytes: a view, not a conversion kernel
        o.pop('depth'), o.pop('view_mask')
        if want_fine_depth:
            o['fine_depth'] = fd
        return o

    def geo_dual_fwd(self, canon, stats, pts, gamma):
        """geometry_fc on dual numbers (k_geo_dual_fwd_mm; k_geo_dual_fwd with the option geo_dual_fp32): canon = the level's canonical blob on the device, stats [P,66],
        pts, gamma [P,3] -> g, gd [P,16] (value / tangent of geometry_fc's output)."""
        stats, pts, gamma = (_f32(x, self.device) for x in (stats, pts, gamma))
        P = stats.shape[0]
        assert stats.shape == (P, 66) and pts.shape == (P, 3) and gamma.shape == (P, 3)
        g = torch.empty(P, 16, dtype=torch.float32, device=self.device)
        gd = torch.empty_like(g)
        need = self.L.gnr_geo_dual_fwd_workspace_bytes()
        if getattr(self, '_gdf_scratch', None) is None or self._gdf_scratch.numel() < need:
            self._gdf_scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.check(self.L.gnr_geo_dual_fwd(canon.data_ptr(), stats.data_ptr(), pts.data_ptr(), gamma.data_ptr(), g.data_ptr(),
                                           gd.data_ptr(), P, self._gdf_scratch.data_ptr(), self._gdf_scratch.numel(), self.options,
                                           self._stream()), 'gnr_geo_dual_fwd')
        return g, gd

    def geo_dual_bwd(self, canon, stats, pts, gamma, gbar, gdbar):
        """-> d stats [P,66] and d_canonical [36958] holding the gradients of geometry_fc.{0,2} (k_geo_dual_bwd)."""
        stats, pts, gamma, gbar, gdbar = (_f32(x, self.device) for x in (stats, pts, gamma, gbar, gdbar))
        P = stats.shape[0]
        dstats = torch.empty(P, 66, dtype=torch.float32, device=self.device)
        dcan = torch.zeros(self.L.gnr_canonical_weights_floats(), dtype=torch.float32, device=self.device)
        need = self.L.gnr_geo_dual_bwd_workspace_bytes(P)
        if getattr(self, '_gd_scratch', None) is None or self._gd_scratch.numel() < need:
            self._gd_scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.check(self.L.gnr_geo_dual_bwd(canon.data_ptr(), stats.data_ptr(), pts.data_ptr(), gamma.data_ptr(), gbar.data_ptr(),
                                           gdbar.data_ptr(), dstats.data_ptr(), dcan.data_ptr(), P, self._gd_scratch.data_ptr(),
                                           self._gd_scratch.numel(), self.options, self._stream()), 'gnr_geo_dual_bwd')
        return dstats, dcan

    def composite_bwd(self, level, sdf, grad, col, depth, qdir, dpix, ddepth=None, wgerr=None, dalpha=None, dhit=None):
        """Backward of NeuS alpha + compositing for R rays (k_composite_bwd): sdf, depth [R,dn], grad, col [R,dn,3],
        qdir [R,3]; upstream dpix [R,3], ddepth, wgerr [R], dalpha, dhit [R,dn] (None = zero).
        -> a = dL/d sdf [R,dn], gamma = dL/d grad [R,dn,3], dcol [R,dn,3], dvar [1]."""
        f = lambda x: None if x is None else _f32(x, self.device)
        sdf, grad, col, depth, qdir, dpix, ddepth, wgerr, dalpha, dhit = (f(x) for x in (sdf, grad, col, depth, qdir, dpix, ddepth,
                                                                                        wgerr, dalpha, dhit))
        R, dn = sdf.shape
        a, gamma, dcol = torch.empty_like(sdf), torch.empty_like(grad), torch.empty_like(col)
        dvar = torch.empty(1, dtype=torch.float32, device=self.device)
        ptr = lambda x: None if x is None else x.data_ptr()
        w = self.wc if level == 'coarse' else self.wf
        scr = self._scratch('comp', self.L.gnr_composite_bwd_workspace_bytes(R))
        _lib.check(self.L.gnr_composite_bwd(w.data_ptr(), sdf.data_ptr(), grad.data_ptr(), col.data_ptr(), depth.data_ptr(),
                                            qdir.data_ptr(), dpix.data_ptr(), ptr(ddepth), ptr(wgerr), ptr(dalpha), ptr(dhit),
                                            a.data_ptr(), gamma.data_ptr(), dcol.data_ptr(), dvar.data_ptr(), R, dn,
                                            scr.data_ptr(), scr.numel(), self._stream()), 'gnr_composite_bwd')
        return a, gamma, dcol, dvar

    def ray_tail_dual_bwd(self, level, g, gd, a, nvalid):
        """Attention / LayerNorm core of the tail's backward incl. the second-order path (ray_tail.attn_core in HIP).
        g, gd [R,dn,16]; a, nvalid [R,dn] -> gbar, gdbar [R,dn,16], dtail [gnr_ray_tail_grad_floats()]."""
        g, gd, a, nvalid = (_f32(x, self.device) for x in (g, gd, a, nvalid))
        R, dn, _ = g.shape
        gbar, gdbar = torch.empty_like(g), torch.empty_like(g)
        dtail = torch.empty(self.L.gnr_ray_tail_grad_floats(), dtype=torch.float32, device=self.device)
        w = self.wc if level == 'coarse' else self.wf
        scr = self._scratch('tail', self.L.gnr_ray_tail_dual_bwd_workspace_bytes())
        _lib.check(self.L.gnr_ray_tail_dual_bwd(w.data_ptr(), g.data_ptr(), gd.data_ptr(), a.data_ptr(), nvalid.data_ptr(),
                                                gbar.data_ptr(), gdbar.data_ptr(), dtail.data_ptr(), R, dn, scr.data_ptr(), scr.numel(),
                                                self.options, self._stream()), 'gnr_ray_tail_dual_bwd')
        return gbar, gdbar, dtail

    def release_training_workspaces(self):
        """Give the per-pass training workspaces (about 1 GB each at 8 scenes x 512 rays x 40 samples), the volume's training
        workspace and the backward scratch buffers back to the allocator (after training, before a long evaluation)."""
        self.__dict__.pop('_pass_tws', None)
        self.__dict__.pop('_scratch_bufs', None)
        self._tws = None
        self._dm_scratch = self._gd_scratch = None

    def set_bwd_weights(self, packed_bwd_coarse, packed_bwd_fine=None):
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)
        self.wb = {'coarse': t(packed_bwd_coarse), 'fine': t(packed_bwd_fine)}

    def time_chain_kernel(self, ref, res=40, iters=10):
        """Average ms per launch of the dominant kernel (k_chain on the volume points), HIP events
        recorded on the launch stream inside the library."""
        scene, keep, ws = self.prepare(ref, res)
        ms = C.c_float(0)
        _lib.check(self.L.gnr_time_chain_kernel(self._sc(scene), res, self.wc.data_ptr(), ws.data_ptr(), ws.numel(),
                                                iters, C.byref(ms), self._stream()), 'gnr_time_chain_kernel')
        return ms.value


def batch_scenes(scenes):
    """list of (ref, que) numpy dicts from synth.make_scene -> batched numpy dicts."""
    ref = {k: np.stack([s[0][k] for s in scenes]) for k in scenes[0][0]}
    que = {k: np.stack([s[1][k] for s in scenes]) for k in scenes[0][1]}
    if 'imgs' in que:
        que['imgs'] = que['imgs'][:, 0]
    return ref, que
