"""Device-side grasp post-processing: the reference planner's `process` + `select` (ref: src/nr/main.py:23-84) as HIP
kernels (csrc/gnr_post.hip) instead of scipy.ndimage on the host, plus the host-side tail of `GraspNeRFPlanner.__call__`
(main.py:197-209: seeded permutation, voxel -> metric coordinates)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def gaussian_weights(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d (order 0): fp64, radius int(truncate*sigma + .5); centre first."""
    r = int(truncate * float(sigma) + 0.5)
    if r > _lib.GNR_GAUSS_MAX_RADIUS:
        raise ValueError(f'gaussian radius {r} > {_lib.GNR_GAUSS_MAX_RADIUS}')
    x = np.arange(-r, r + 1)
    w = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    w = w / w.sum()
    return r, w[r:]


class GraspSelector:
    """process() + select() for B scenes at once.  Defaults are the reference functions' defaults; the planner passes
    tsdf_thres_high=0, tsdf_thres_low=-0.85 (main.py:93-94,199)."""

    def __init__(self, device='cuda:0', max_grasps=2048):
        self.L = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.GnrError('the HIP grasp post-processing needs a ROCm GPU; there is no CPU fallback')
        self.device = torch.device(device)
        self.max_grasps = int(max_grasps)
        self._ws = None

    def __call__(self, tsdf, qual, rot, width, gaussian_filter_sigma=1.0, min_width=1.33, max_width=9.33,
                 tsdf_thres_high=0.5, tsdf_thres_low=1e-3, threshold=0.90, max_filter_size=4):
        """tsdf, qual, width [B,1,R,R,R] (or [B,R,R,R]); rot [B,4,R,R,R]  ->  dict of device tensors:
        qual [B,R,R,R] processed quality; count [B]; index [B,max,3] int32; score [B,max]; quat [B,max,4]; width [B,max]
        (entries beyond count[b] are undefined)."""
        d = self.device
        f = lambda a: torch.as_tensor(a, dtype=torch.float32, device=d).contiguous()
        tsdf, qual, rot, width = f(tsdf), f(qual), f(rot), f(width)
        B, R = rot.shape[0], rot.shape[-1]
        assert rot.shape == (B, 4, R, R, R) and tsdf.numel() == qual.numel() == width.numel() == B * R ** 3
        p = _lib.GnrSelectParams()
        p.gauss_radius, w = gaussian_weights(gaussian_filter_sigma)
        for k, v in enumerate(w):
            p.gauss_w[k] = float(v)
        p.tsdf_thres_high, p.tsdf_thres_low = float(tsdf_thres_high), float(tsdf_thres_low)
        p.min_width, p.max_width, p.threshold = float(min_width), float(max_width), float(threshold)
        p.dilate_iterations, p.max_filter_size = 2, int(max_filter_size)
        need = self.L.gnr_grasp_select_workspace_bytes(B, R)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=d)
        M = self.max_grasps
        out = {'qual': torch.empty(B, R, R, R, device=d), 'count': torch.empty(B, dtype=torch.int32, device=d),
               'index': torch.empty(B, M, 3, dtype=torch.int32, device=d), 'score': torch.empty(B, M, device=d),
               'quat': torch.empty(B, M, 4, device=d), 'width': torch.empty(B, M, device=d)}
        rc = self.L.gnr_grasp_select_fwd(tsdf.data_ptr(), qual.data_ptr(), rot.data_ptr(), width.data_ptr(), B, R,
                                         C.byref(p), out['qual'].data_ptr(), out['count'].data_ptr(),
                                         out['index'].data_ptr(), out['score'].data_ptr(), out['quat'].data_ptr(),
                                         out['width'].data_ptr(), M, self._ws.data_ptr(), self._ws.numel(),
                                         C.c_void_p(torch.cuda.current_stream(d).cuda_stream))
        if rc:
            raise _lib.GnrError(f'gnr_grasp_select_fwd failed: {rc} ({self.L.gnr_post_last_error().decode()})')
        return out


def grasps_from_selection(sel, b=0, voxel_size=0.3 / 40, seed=None):
    """Scene b of a GraspSelector result -> numpy dict in the reference's conventions (main.py:79-84,201-209):
    `pos` = voxel index * voxel_size (metres, bbox-local), `quat` normalised (x,y,z,w; scipy Rotation.from_quat),
    `width` in metres, `score`, `index`; permuted with np.random.seed(seed) like the planner when seed is given."""
    n = int(sel['count'][b].item())
    if n > sel['index'].shape[1]:
        # the reference's select() returns EVERY non-maximum-suppression survivor (main.py:70-84); a truncated list (in
        # index order, before the seeded permutation) would silently be a different result
        raise _lib.GnrError(f'{n} grasps selected but the buffers hold {sel["index"].shape[1]}: raise GraspSelector(max_grasps=...)')
    idx = sel['index'][b, :n].cpu().numpy().astype(np.int64)
    quat = sel['quat'][b, :n].cpu().numpy().astype(np.float64)
    quat = quat / np.linalg.norm(quat, axis=1, keepdims=True) if n else quat
    score = sel['score'][b, :n].cpu().numpy()
    width = sel['width'][b, :n].cpu().numpy()
    if seed is not None and n > 0:
        np.random.seed(seed)
        p = np.random.permutation(n)
        idx, quat, score, width = idx[p], quat[p], score[p], width[p]
    return {'index': idx, 'pos': idx.astype(np.float64) * voxel_size, 'quat': quat, 'width': width * voxel_size, 'score': score}
