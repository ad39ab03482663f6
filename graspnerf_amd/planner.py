"""Inference wrapper with the call shape of the reference planner's `core()` (ref: src/nr/main.py:211-253):
numpy images / extrinsics / intrinsics in, (volume, qual, rot, width, seconds) out, and of its `__call__`
(main.py:185-209) from arrays: `plan()` adds the grasp post-processing (`process`, `select`, main.py:23-84) on the
device (graspnerf_amd/grasp_post.py).  Loads reference checkpoints (`network_state_dict`, main.py:153-155) unchanged.
`GraspNeRFPlanner` below is the file-I/O half of the reference class (main.py:87-209: rendered PNGs, camera poses, intrinsics,
fixed depth range -> core -> process/select), with PIL / numpy only.  The simulator / Blender loop that produces those files
is outside the volumetric path and not rebuilt."""
import os
import time

import numpy as np
import torch

from .renderer import GraspNeRF
from .grasp_post import GraspSelector, grasps_from_selection


def load_model(cfg, checkpoint=None, device='cuda:0', depth_coords_rng='device'):
    """cfg: the reference's yaml as a dict.  checkpoint: path to `model_best.pth` or a state dict."""
    cfg = {**cfg, 'render_rgb': False, 'depth_coords_rng': depth_coords_rng}     # main.py:150: no rendering when grasping
    net = GraspNeRF(cfg)
    if checkpoint is not None:
        sd = torch.load(checkpoint, map_location='cpu') if isinstance(checkpoint, (str, bytes)) else checkpoint
        net.load_state_dict(sd.get('network_state_dict', sd), strict=True)
    return net.to(device).eval()


def core(net, images, extrinsics, intrinsics, depth_range=(0.2, 0.8),
         bbox3d=((-0.15, -0.15, -0.05), (0.15, 0.15, 0.25)), que_id=0):
    """images [V,3,H,W] in [0,1]; extrinsics [V,3|4,4] world->camera; intrinsics [V,3,3]; H, W multiples of 32.
    -> volume [1,1,R,R,R], qual [1,1,40^3], rot [1,4,40^3], width [1,1,40^3] (numpy) and the forward seconds."""
    V, _, h, w = images.shape
    assert h % 32 == 0 and w % 32 == 0                                          # main.py:226
    dev = next(net.parameters()).device
    t = lambda a: torch.as_tensor(np.array(a, np.float32), device=dev)
    ext = np.asarray(extrinsics, np.float32)[:, :3, :]
    dr = np.broadcast_to(np.asarray(depth_range, np.float32), (V, 2)) if np.ndim(depth_range) == 1 else np.asarray(depth_range, np.float32)
    ref = {'imgs': t(images), 'poses': t(ext), 'Ks': t(intrinsics), 'depth_range': t(dr), 'bbox3d': t(bbox3d)}
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')               # imgs_info.py:126-135
    que = {'poses': t(ext[que_id])[None], 'Ks': t(np.asarray(intrinsics, np.float32)[que_id])[None],
           'coords': t(np.stack([xs, ys], -1).reshape(1, -1, 2)), 'depth_range': t(dr[que_id])[None]}
    data = {'step': 0, 'eval': True, 'full_vol': True, 'ref_imgs_info': ref, 'que_imgs_info': que, 'src_imgs_info': dict(ref)}
    with torch.no_grad():
        torch.cuda.synchronize(dev)
        t0 = time.time()
        out = net(data)
        torch.cuda.synchronize(dev)                                             # the reference times without a sync (main.py:244-247)
        dt = time.time() - t0
    q, r, wd = out['vgn_pred']
    return out['volume'].cpu().numpy(), q.cpu().numpy(), r.cpu().numpy(), wd.cpu().numpy(), dt


def plan(net, images, extrinsics, intrinsics, depth_range=(0.2, 0.8), bbox3d=((-0.15, -0.15, -0.0503), (0.15, 0.15, 0.2497)),
         seed=None, selector=None, tsdf_thres_high=0.0, tsdf_thres_low=-0.85, voxel_size=0.3 / 40, return_volumes=False):
    """`GraspNeRFPlanner.__call__` from arrays (main.py:185-209): forward, process + select on the device, seeded
    permutation, voxel -> metric.  -> (grasps dict of numpy arrays: pos, quat, width, score, index; forward seconds)."""
    dev = next(net.parameters()).device
    V, _, h, w = images.shape
    t = lambda a: torch.as_tensor(np.array(a, np.float32), device=dev)
    ext = np.asarray(extrinsics, np.float32)[:, :3, :]
    dr = np.broadcast_to(np.asarray(depth_range, np.float32), (V, 2)) if np.ndim(depth_range) == 1 else np.asarray(depth_range, np.float32)
    ref = {'imgs': t(images), 'poses': t(ext), 'Ks': t(intrinsics), 'depth_range': t(dr), 'bbox3d': t(bbox3d)}
    que = {'poses': t(ext[0])[None], 'Ks': t(np.asarray(intrinsics, np.float32)[0])[None],
           'coords': torch.zeros(1, 1, 2, device=dev), 'depth_range': t(dr[0])[None]}
    data = {'step': 0, 'eval': True, 'full_vol': True, 'ref_imgs_info': ref, 'que_imgs_info': que, 'src_imgs_info': dict(ref)}
    selector = selector or GraspSelector(dev)
    with torch.no_grad():
        torch.cuda.synchronize(dev)
        t0 = time.time()
        out = net(data)
        q, r, wd = out['vgn_pred']
        sel = selector(out['volume'], q, r, wd, tsdf_thres_high=tsdf_thres_high, tsdf_thres_low=tsdf_thres_low)
        torch.cuda.synchronize(dev)
        dt = time.time() - t0
    grasps = grasps_from_selection(sel, 0, voxel_size, seed)
    if return_volumes:
        grasps['volumes'] = tuple(x.cpu().numpy() for x in (out['volume'], q, r, wd, sel['qual']))
    return grasps, dt


# ---- the planner's file I/O (ref: src/nr/main.py:87-209) -----------------------------------------------
BLENDER2OPENCV = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], np.float64)      # main.py:104
SRC_WH = {'vgn_syn': (640, 360)}                                                                          # main.py:100-102


def resize_bilinear_u8(img, wh):
    """cv2.resize(img, wh) for uint8 HxWxC images with its default INTER_LINEAR (main.py:171): half-pixel centres, border
    replication, OpenCV's fixed-point arithmetic (11-bit coefficients, (.. + 2) >> 2 rounding in the vertical pass), no
    anti-aliasing -- PIL's BILINEAR filters when downscaling and gives different pixels.  (cv2 is not in this image: written
    from OpenCV's resize.cpp, checked against float bilinear to 1 LSB in tests/test_planner_io.py.)"""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    sh, sw = img.shape[:2]
    dw, dh = int(wh[0]), int(wh[1])
    if (dw, dh) == (sw, sh):
        return img.copy()

    def axis(dn, sn):
        f = (np.arange(dn, dtype=np.float64) + 0.5) * (sn / dn) - 0.5
        i0 = np.floor(f).astype(np.int64)
        w = (f - i0).astype(np.float32)
        lo = i0 < 0
        i0[lo], w[lo] = 0, 0.0
        hi = i0 >= sn - 1
        i0[hi], w[hi] = sn - 1, 0.0
        i1 = np.minimum(i0 + 1, sn - 1)
        c1 = np.clip(np.rint(w.astype(np.float64) * 2048), -32768, 32767).astype(np.int64)      # saturate_cast<short>(w * 2^11)
        c0 = np.clip(np.rint((1.0 - w.astype(np.float64)) * 2048), -32768, 32767).astype(np.int64)
        return i0, i1, c0, c1
    x0, x1, a0, a1 = axis(dw, sw)
    y0, y1, b0, b1 = axis(dh, sh)
    src = img.astype(np.int64)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]                       # horizontal pass: int, scale 2^11
    s0, s1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def read_rgb_png(path):
    """skimage.io.imread(path)[:, :, :3] (main.py:169-170) through PIL."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGBA' if im.mode in ('RGBA', 'LA', 'P') else 'RGB'))[:, :, :3].copy()


class _Rotation:
    """The two accessors of scipy's Rotation the reference's callers use on a grasp pose (clutter_removal.py:198-199,
    simulation.execute_grasp): as_quat() (x, y, z, w) and as_matrix()."""

    def __init__(self, quat):
        q = np.asarray(quat, np.float64)
        self._q = q / max(np.linalg.norm(q), 1e-12)

    def as_quat(self):
        return self._q.copy()

    def as_matrix(self):
        x, y, z, w = self._q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class _Pose:
    def __init__(self, quat, translation):
        self.rotation, self.translation = _Rotation(quat), np.asarray(translation, np.float64)


class Grasp:
    """gd.grasp.Grasp(Transform(Rotation, t), width) as the planner's callers read it (main.py:79-84, 205;
    clutter_removal.py:198-199): `.pose.rotation.as_quat()` / `.as_matrix()`, `.pose.translation`, `.width` -- plus the flat
    `.quat` (x, y, z, w) and `.translation` this package uses itself."""

    def __init__(self, quat, translation, width):
        self.quat, self.translation, self.width = np.asarray(quat, np.float64), np.asarray(translation, np.float64), float(width)
        self.pose = _Pose(self.quat, self.translation)

    def __repr__(self):
        return f'Grasp(t={self.translation.round(4).tolist()}, q={self.quat.round(4).tolist()}, w={self.width:.4f})'


class GraspNeRFPlanner:
    """Mirror of the reference planner (main.py:87-209) on files: `rgb/%04d.png` renderings and `camera_pose.npy`.

    cfg: the reference's yaml as a dict;  checkpoint: `model_best.pth` path or state dict ({'network_state_dict': ...});
    renderer_root_dir: holds camera_pose.npy (main.py:174);  rgb_dir: the directory of the rendered `%04d.png` images
    (main.py:168);  database_name: as in the reference's args, e.g. 'vgn_syn/test/packed/packed_170-220/032cd891d9be4a16be5ea4be9f7eca2b/w_0.8'
    (the trailing `<background>_<size>` sets the down-sampling, main.py:96-103)."""

    def __init__(self, cfg, checkpoint, renderer_root_dir, rgb_dir, database_name='vgn_syn/test/x/x/x/w_0.8', seed=0, device='cuda:0'):
        tp, _split, _stype, _ssplit, _sid, background_size = database_name.split('/')                 # main.py:96
        self.tp, self.down_sample = tp, float(background_size.split('_')[1])
        self.img_wh = (np.array(SRC_WH[tp]) * self.down_sample).astype(int)                            # main.py:103
        K = np.array([[892.62, 0.0, 639.5], [0.0, 892.62, 359.5], [0.0, 0.0, 1.0]])                    # main.py:105-111
        K[:2] = K[:2] * self.down_sample
        if tp == 'vgn_syn':
            K[:2] /= 2
        self.K = K
        self.voxel_size, self.bbox3d = 0.3 / 40, [[-0.15, -0.15, -0.0503], [0.15, 0.15, 0.2497]]       # main.py:90-91
        self.tsdf_thres_high, self.tsdf_thres_low = 0.0, -0.85                                         # main.py:92-93
        self.renderer_root_dir, self.rgb_dir, self.seed = renderer_root_dir, rgb_dir, seed
        ckpt = torch.load(checkpoint, map_location='cpu') if isinstance(checkpoint, (str, bytes)) else checkpoint   # read once
        self.net = load_model(cfg, ckpt, device)                                                       # main.py:150-157
        self.step = int(ckpt.get('step', 0)) if isinstance(ckpt, dict) else 0                          # main.py:155
        self.selector = GraspSelector(next(self.net.parameters()).device)

    def get_image(self, img_id, round_idx=0):                                                          # main.py:167-172
        img = read_rgb_png(os.path.join(self.rgb_dir, '%04d.png' % img_id))
        return resize_bilinear_u8(img, self.img_wh).astype(np.float32)

    def get_pose(self, img_id):                                                                        # main.py:174-177
        # read on every call like the reference: the simulator rewrites camera_pose.npy between rounds
        ori = np.load(os.path.join(self.renderer_root_dir, 'camera_pose.npy'))[img_id]
        return np.linalg.inv(ori @ BLENDER2OPENCV)[:3, :].astype(np.float32)

    def get_K(self, img_id):
        return self.K.astype(np.float32).copy()

    def get_depth_range(self, img_id, round_idx=0, fixed=True):                                        # main.py:182-187
        if not fixed:
            raise NotImplementedError('depth-map based ranges need the simulator\'s depth renderings (main.py:185-187)')
        return np.array([0.2, 0.8])

    def __call__(self, test_view_id, round_idx=0, n_grasp=0, gt_tsdf=None):                            # main.py:189-209
        images = np.stack([self.get_image(i, round_idx) for i in test_view_id], 0)
        images = (images.astype(np.float32) / 255).transpose([0, 3, 1, 2])                             # color_map_forward
        extrinsics = np.stack([self.get_pose(i) for i in test_view_id], 0)
        intrinsics = np.stack([self.get_K(i) for i in test_view_id], 0)
        depth_range = np.asarray([self.get_depth_range(i, round_idx, fixed=True) for i in test_view_id], dtype=np.float32)
        g, toc = plan(self.net, images, extrinsics, intrinsics, depth_range, self.bbox3d, seed=self.seed + round_idx + n_grasp,
                      selector=self.selector, tsdf_thres_high=self.tsdf_thres_high, tsdf_thres_low=self.tsdf_thres_low,
                      voxel_size=self.voxel_size)
        grasps = [Grasp(q, t, w) for q, t, w in zip(g['quat'], g['pos'], g['width'])]
        return grasps, g['score'], toc
