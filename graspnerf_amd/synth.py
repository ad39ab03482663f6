"""Synthetic scene generator for benchmarks, tests and golden fixtures.

Everything is drawn from `numpy.random.default_rng(1000 + scene_id)` (PCG64, platform
stable) in a fixed order, so the GPU box regenerates bit-identical inputs from a seed and the
fixtures only need to store outputs.  Cameras sit on a ring looking at the workspace centre
(OpenCV axes, world->camera poses), intrinsics follow the reference planner
(ref: src/nr/main.py:105-112), depth range [0.2, 0.8] (main.py:181-183), workspace bbox
[[-0.15,-0.15,-0.05],[0.15,0.15,0.25]] (ref: src/nr/dataset/database.py:122-123).
"""
import numpy as np

CONFIGS = {
    # BASELINE.json configs[0]: 1 scene, 3 views, 16^3, small images (CPU plumbing case)
    'cfg1': dict(V=3, H=96, W=128, res=16, rn=64,
                 K=[[100.0, 0, 63.5], [0, 100.0, 47.5], [0, 0, 1]]),
    # BASELINE.json configs[1..3]: 6 views, 288x512, 40^3, 512 rays
    'cfg2': dict(V=6, H=288, W=512, res=40, rn=512,
                 K=[[357.048, 0, 255.8], [0, 357.048, 143.8], [0, 0, 1]]),
}


def ring_cameras(V, radius=0.5, theta=np.pi / 3, target=(0.0, 0.0, 0.1)):
    """[V,3,4] world->camera, OpenCV convention (z forward, x right, y down)."""
    poses = []
    tgt = np.asarray(target, np.float64)
    for i in range(V):
        phi = 2 * np.pi * i / V
        eye = np.array([radius * np.sin(theta) * np.cos(phi),
                        radius * np.sin(theta) * np.sin(phi),
                        radius * np.cos(theta) + tgt[2]])
        zf = tgt - eye
        zf /= np.linalg.norm(zf)
        xr = np.cross(zf, np.array([0.0, 0.0, 1.0]))
        xr /= np.linalg.norm(xr)
        yd = np.cross(zf, xr)
        R = np.stack([xr, yd, zf], 0)
        poses.append(np.concatenate([R, (-R @ eye)[:, None]], 1))
    return np.asarray(poses, np.float32)


def make_scene(scene_id=0, cfg='cfg2', with_query_image=True):
    """-> (ref dict, que dict) of float32 numpy arrays.
    ref: imgs[V,3,H,W] in [0,1], img_feats/ray_feats[V,32,H/4,W/4], poses[V,3,4], Ks[V,3,3],
         depth_range[V,2], bbox3d[2,3]
    que: coords[rn,2] (x,y) float, pose[3,4], K[3,3], depth_range[2], imgs[1,3,H,W] (= view 0)."""
    c = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    V, H, W, rn = c['V'], c['H'], c['W'], c['rn']
    fh, fw = H // 4, W // 4
    rng = np.random.default_rng(1000 + scene_id)
    imgs = rng.random((V, 3, H, W)).astype(np.float32)
    img_feats = (0.5 * rng.standard_normal((V, 32, fh, fw))).astype(np.float32)
    ray_feats = (0.5 * rng.standard_normal((V, 32, fh, fw))).astype(np.float32)
    cx = rng.integers(0, W, rn)
    cy = rng.integers(0, H, rn)
    coords = np.stack([cx, cy], -1).astype(np.float32)
    poses = ring_cameras(V)
    K = np.asarray(c['K'], np.float32)
    ref = dict(imgs=imgs, img_feats=img_feats, ray_feats=ray_feats, poses=poses,
               Ks=np.repeat(K[None], V, 0).copy(),
               depth_range=np.repeat(np.asarray([[0.2, 0.8]], np.float32), V, 0).copy(),
               bbox3d=np.asarray([[-0.15, -0.15, -0.05], [0.15, 0.15, 0.25]], np.float32))
    que = dict(coords=coords, pose=poses[0].copy(), K=K.copy(),
               depth_range=np.asarray([0.2, 0.8], np.float32))
    if with_query_image:
        que['imgs'] = imgs[:1].copy()
    return ref, que


def random_scene(seed):
    """A scene with randomised geometry for parity sweeps: 2..8 views on a jittered ring (one camera may sit close to
    or inside the workspace), per-view intrinsics and depth ranges, image and feature-map sizes without the fixed 1/4
    ratio, a shifted workspace box, fractional (partly out-of-image) ray coordinates.
    -> (ref, que, meta) with meta = dict(V, H, W, fh, fw, res, rn, dn)."""
    rng = np.random.default_rng(5000 + seed)
    V = int(rng.integers(2, 9))
    H, W = int(rng.integers(24, 80)), int(rng.integers(32, 120))
    fh, fw = max(3, int(H / rng.uniform(2.0, 6.0))), max(3, int(W / rng.uniform(2.0, 6.0)))
    res, rn, dn = int(rng.integers(5, 15)), int(rng.integers(1, 40)), int(rng.integers(3, 49))
    poses = []
    for i in range(V):
        phi = 2 * np.pi * (i + rng.uniform(-0.3, 0.3)) / V
        theta = rng.uniform(0.5, 1.3)
        radius = rng.uniform(0.08, 0.25) if (i == 0 and seed % 3 == 0) else rng.uniform(0.35, 0.7)
        tgt = np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), 0.1 + rng.uniform(-0.05, 0.05)])
        eye = tgt + radius * np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])
        zf = (tgt - eye) / np.linalg.norm(tgt - eye)
        xr = np.cross(zf, np.array([0.0, 0.0, 1.0]))
        xr /= np.linalg.norm(xr)
        R = np.stack([xr, np.cross(zf, xr), zf], 0)
        poses.append(np.concatenate([R, (-R @ eye)[:, None]], 1))
    poses = np.asarray(poses, np.float32)
    f = rng.uniform(0.7, 1.3, V) * 0.75 * W
    Ks = np.zeros((V, 3, 3), np.float32)
    Ks[:, 0, 0], Ks[:, 1, 1], Ks[:, 2, 2] = f, f * rng.uniform(0.9, 1.1, V), 1.0
    Ks[:, 0, 2], Ks[:, 1, 2] = W / 2 + rng.uniform(-4, 4, V), H / 2 + rng.uniform(-4, 4, V)
    dr = np.stack([rng.uniform(0.1, 0.3, V), rng.uniform(0.6, 1.2, V)], 1).astype(np.float32)
    bb0 = np.array([-0.15, -0.15, -0.05]) + rng.uniform(-0.1, 0.1, 3)
    ref = dict(imgs=rng.random((V, 3, H, W)).astype(np.float32),
               img_feats=(0.5 * rng.standard_normal((V, 32, fh, fw))).astype(np.float32),
               ray_feats=(0.5 * rng.standard_normal((V, 32, fh, fw))).astype(np.float32),
               poses=poses, Ks=Ks, depth_range=dr, bbox3d=np.stack([bb0, bb0 + 0.3]).astype(np.float32))
    coords = np.stack([rng.uniform(-3, W + 2, rn), rng.uniform(-3, H + 2, rn)], -1).astype(np.float32)
    que = dict(coords=coords, pose=poses[0].copy(), K=Ks[0].copy(), depth_range=dr[0].copy(), imgs=ref['imgs'][:1].copy())
    return ref, que, dict(V=V, H=H, W=W, fh=fh, fw=fw, res=res, rn=rn, dn=dn)


def synth_state_dict(shapes, seed=7):
    """Deterministic, construction-order independent parameters for a whole GraspNeRF model:
    every tensor is drawn from its own PCG64 stream keyed by crc32(name).  Used by the full-forward
    golden (tools/make_goldens.py) and its tests, because 4.66 M parameters cannot be a fixture.
    shapes: {state-dict key: shape}.  Weights ~ N(0, 1/fan_in), biases ~ N(0, 0.05^2), norm scales ~ 1 +- 0.1."""
    import zlib
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        if k.endswith('variance'):
            v = np.asarray(0.3, np.float32)
        elif k.endswith(('imagenet_mean', 'imagenet_std')):
            v = np.asarray([0.485, 0.456, 0.406] if k.endswith('mean') else [0.229, 0.224, 0.225], np.float32).reshape(shp)
        elif len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            v = (rng.standard_normal(shp) / np.sqrt(fan_in)).astype(np.float32)
        elif 'bn' in k.split('.')[-2] or 'layer_norm' in k or k.endswith('.weight'):
            # 1-D weights are normalisation scales
            v = (1.0 + 0.1 * rng.standard_normal(shp)).astype(np.float32) if k.endswith('.weight') \
                else (0.05 * rng.standard_normal(shp)).astype(np.float32)
        else:
            v = (0.05 * rng.standard_normal(shp)).astype(np.float32)
        out[k] = v
    return out


def synth_loss_case(seed=5, rfn=3, h=96, w=128, rn=64, pn=256, R=16, N=12):
    """Random (prediction, ground truth) tensors in the shapes the reference's losses consume (loss.py); the golden
    stores the reference's loss values for exactly these tensors."""
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.random(s).astype(np.float32)
    pr = {'pixel_colors_gt': f(1, rn, 3), 'pixel_colors_nr': f(1, rn, 3), 'pixel_colors_nr_fine': f(1, rn, 3),
          'ray_mask': rng.random((1, rn)) > 0.3,
          'depth_coords': np.repeat(np.stack([rng.integers(0, h, pn), rng.integers(0, w, pn)], -1)[None], rfn, 0).astype(np.int64),
          'depth_mean': f(rfn, pn), 'depth_mean_fine': f(rfn, pn),
          'volume': (f(1, 1, R, R, R) * 2 - 1), 'sdf_gradient_error': f(1, 1), 's': np.full((1, 1), 0.3, np.float32)}
    q = rng.standard_normal((N, 4)).astype(np.float32)
    pr['vgn_pred'] = (f(N) * 0.98 + 0.01, q / np.linalg.norm(q, axis=1, keepdims=True), f(N) * 0.08)
    sdf_gt = (f(R, R, R) * 2 - 1)
    sdf_gt[rng.random((R, R, R)) < 0.3] = -1.0
    r2 = rng.standard_normal((N, 2, 4)).astype(np.float32)
    gt = {'true_depth': 0.2 + 0.6 * f(rfn, 1, h, w), 'depth_range': np.repeat(np.asarray([[0.2, 0.8]], np.float32), rfn, 0),
          'sdf_gt': sdf_gt,
          'grasp_info': (rng.integers(0, 40, (N, 3)).astype(np.int64), (rng.random(N) > 0.5).astype(np.float32),
                         r2 / np.linalg.norm(r2, axis=2, keepdims=True), f(N) * 0.08)}
    return pr, gt


def synth_head_outputs(seed, R=40):
    """Smooth random (tsdf, qual, rot, width) volumes in the value ranges the grasp head produces, for the
    grasp post-processing path (main.py:23-74): a blobby surface, broad high-quality regions that survive the
    Gaussian smoothing, unit quaternions, widths in voxel units.  -> float32 [1,1,R,R,R], [1,1,..], [1,4,..], [1,1,..]"""
    rng = np.random.default_rng(7000 + seed)

    def field(n_ch=1, coarse=6):
        c = rng.standard_normal((n_ch, coarse, coarse, coarse))
        x = np.linspace(0, coarse - 1, R)
        i0 = np.clip(np.floor(x).astype(int), 0, coarse - 2)
        t = x - i0
        for ax in (1, 2, 3):                                     # separable linear upsampling
            a = np.take(c, i0, axis=ax)
            b = np.take(c, i0 + 1, axis=ax)
            shp = [1, 1, 1, 1]
            shp[ax] = R
            tt = t.reshape(shp)
            c = a * (1 - tt) + b * tt
        return c
    tsdf = np.clip(0.8 * field()[0] - 0.2, -1, 1)
    qual = 1.0 / (1.0 + np.exp(-(3.0 * field()[0] + 1.0)))
    rot = field(4) + 0.05 * rng.standard_normal((4, R, R, R))
    rot /= np.linalg.norm(rot, axis=0, keepdims=True)
    width = 5.0 + 4.5 * field()[0]
    f = lambda a: np.ascontiguousarray(a, np.float32)
    return f(tsdf)[None, None], f(qual)[None, None], f(rot)[None], f(width)[None, None]
