"""Inference wrapper with the call shape of the reference planner's `core()` (ref: src/nr/main.py:211-253):
numpy images / extrinsics / intrinsics in, (volume, qual, rot, width, seconds) out, and of its `__call__`
(main.py:185-209) from arrays: `plan()` adds the grasp post-processing (`process`, `select`, main.py:23-84) on the
device (graspnerf_amd/grasp_post.py).  Loads reference checkpoints (`network_state_dict`, main.py:153-155) unchanged.
The simulator / Blender loop and image I/O are outside the volumetric path and not rebuilt."""
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
