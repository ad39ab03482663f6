"""the file-I/O half of the reference planner (src/nr/main.py:87-209) and the checkpoint loader
(main.py:153-155): rendered PNGs -> resize to 512x288 -> /255, camera_pose.npy -> OpenCV world->camera poses, scaled
intrinsics, fixed depth range, `model_best.pth`-shaped checkpoints with the REFERENCE's key layout."""
import os

import numpy as np
import pytest
import torch
import yaml

from graspnerf_amd import planner
from graspnerf_amd.synth import synth_state_dict, ring_cameras

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = yaml.safe_load("""
network: grasp_nerf
init_net_type: cost_volume
agg_net_type: neus
use_hierarchical_sampling: true
use_depth_loss: true
dist_decoder_cfg: {use_vis: false}
fine_dist_decoder_cfg: {use_vis: false}
ray_batch_num: 4096
sample_volume: true
render_rgb: false
volume_type: [sdf]
volume_resolution: 40
depth_sample_num: 40
fine_depth_sample_num: 40
agg_net_cfg: {sample_num: 40, init_s: 0.3, fix_s: 0}
fine_agg_net_cfg: {sample_num: 40, init_s: 0.3, fix_s: 0}
""")


def _float_bilinear(img, wh):
    """Plain float bilinear with half-pixel centres and border replication (what cv2.INTER_LINEAR computes before its
    fixed-point rounding)."""
    sh, sw = img.shape[:2]
    dw, dh = wh
    fx = np.clip((np.arange(dw) + 0.5) * sw / dw - 0.5, 0, sw - 1)
    fy = np.clip((np.arange(dh) + 0.5) * sh / dh - 0.5, 0, sh - 1)
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    x1, y1 = np.minimum(x0 + 1, sw - 1), np.minimum(y0 + 1, sh - 1)
    wx, wy = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
    s = img.astype(np.float64)
    return (s[y0][:, x0] * (1 - wx) + s[y0][:, x1] * wx) * (1 - wy) + (s[y1][:, x0] * (1 - wx) + s[y1][:, x1] * wx) * wy


def test_resize_is_cv2_inter_linear_not_pil():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (360, 640, 3), dtype=np.uint8)
    out = planner.resize_bilinear_u8(img, (512, 288))                      # main.py:103,171: 640x360 * 0.8
    assert out.shape == (288, 512, 3) and out.dtype == np.uint8
    assert np.abs(out.astype(np.float64) - _float_bilinear(img, (512, 288))).max() <= 1.0      # fixed point: within 1 LSB
    assert np.array_equal(planner.resize_bilinear_u8(img, (640, 360)), img)
    const = np.full((360, 640, 3), 137, np.uint8)
    assert np.array_equal(planner.resize_bilinear_u8(const, (512, 288)), np.full((288, 512, 3), 137, np.uint8))
    up = planner.resize_bilinear_u8(img[:45, :80], (160, 90))                                  # upscaling, border replication
    assert np.abs(up.astype(np.float64) - _float_bilinear(img[:45, :80], (160, 90))).max() <= 1.0
    from PIL import Image
    pil = np.asarray(Image.fromarray(img).resize((512, 288), Image.BILINEAR))
    assert np.abs(pil.astype(int) - out.astype(int)).max() > 2, 'PIL anti-aliases when downscaling; cv2 does not'


def test_png_reader(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (36, 64, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / 'a.png')
    rgba = np.concatenate([rgb, rng.integers(0, 256, (36, 64, 1), dtype=np.uint8)], -1)
    Image.fromarray(rgba, 'RGBA').save(tmp_path / 'b.png')
    assert np.array_equal(planner.read_rgb_png(tmp_path / 'a.png'), rgb)
    assert np.array_equal(planner.read_rgb_png(tmp_path / 'b.png'), rgb)                       # imread(...)[:, :, :3]


def _write_scene_files(root, n_views=6, seed=2):
    """rgb/%04d.png renderings (640x360) and camera_pose.npy (Blender camera-to-world, main.py:174-176) of a ring of cameras."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, 'rgb'), exist_ok=True)
    imgs = rng.integers(0, 256, (n_views, 360, 640, 3), dtype=np.uint8)
    for i in range(n_views):
        Image.fromarray(imgs[i]).save(os.path.join(root, 'rgb', '%04d.png' % i))
    w2c = ring_cameras(n_views).astype(np.float64)                                             # OpenCV world->camera [V,3,4]
    c2w_blender = []
    for P in w2c:
        M = np.eye(4)
        M[:3] = P
        c2w_blender.append(np.linalg.inv(M) @ np.linalg.inv(planner.BLENDER2OPENCV))          # pose = inv(p @ b2o)  =>  p = inv(pose) b2o^-1
    np.save(os.path.join(root, 'camera_pose.npy'), np.asarray(c2w_blender))
    return imgs, w2c


def _checkpoint(tmp_path):
    """A `model_best.pth` with the reference's own state-dict layout (tests/golden/golden_ckpt_keys.npz, made by
    tools/make_goldens.py from the reference's GraspNeRF(cfg).state_dict()) and synthetic values."""
    K = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_ckpt_keys.npz'))
    shapes = {str(n): tuple(int(d) for d in str(s).split(',') if d) for n, s in zip(K['names'], K['shapes'])}
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth_state_dict(shapes).items()}
    sd = {str(n): sd[str(n)] for n in K['names']}                                              # the reference's key order
    sd['vgn_net.conv_qual.bias'] = sd['vgn_net.conv_qual.bias'] + 2.5                          # some qualities above 0.9
    path = os.path.join(tmp_path, 'model_best.pth')
    torch.save({'network_state_dict': sd, 'step': 4321, 'best_para': 0.5}, path)              # trainer.py: what the reference saves
    return path, sd


def test_checkpoint_layout_is_the_references(tmp_path):
    """Our mirror's state dict has the reference's 348 keys, shapes and order; a checkpoint file in the reference's layout
    loads with strict=True through planner.load_model (no GPU needed to load)."""
    from graspnerf_amd.renderer import GraspNeRF
    K = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_ckpt_keys.npz'))
    mine = GraspNeRF(dict(CFG)).state_dict()
    assert [str(n) for n in K['names']] == list(mine.keys())
    assert [tuple(int(d) for d in str(s).split(',') if d) for s in K['shapes']] == [tuple(v.shape) for v in mine.values()]
    path, sd = _checkpoint(tmp_path)
    net = planner.load_model(dict(CFG), path, device='cpu')
    got = net.state_dict()
    assert all(torch.equal(got[k], v) for k, v in sd.items()) and not net.training
    with pytest.raises(RuntimeError):                                                          # strict: a missing key is an error
        planner.load_model(dict(CFG), {'network_state_dict': {k: v for k, v in sd.items() if 'conv_qual' not in k}}, device='cpu')


def test_poses_and_intrinsics_follow_the_reference(tmp_path):
    imgs, w2c = _write_scene_files(str(tmp_path))
    P = planner.GraspNeRFPlanner.__new__(planner.GraspNeRFPlanner)                             # I/O half only: no model
    P.renderer_root_dir, P.rgb_dir, P._poses = str(tmp_path), os.path.join(tmp_path, 'rgb'), None
    P.img_wh = (np.array(planner.SRC_WH['vgn_syn']) * 0.8).astype(int)
    for i in range(6):
        np.testing.assert_allclose(P.get_pose(i), w2c[i], atol=1e-6)
        assert P.get_pose(i).dtype == np.float32 and P.get_pose(i).shape == (3, 4)
    im = P.get_image(3)
    assert im.shape == (288, 512, 3) and im.dtype == np.float32 and im.max() <= 255
    assert np.array_equal(im, planner.resize_bilinear_u8(imgs[3], (512, 288)).astype(np.float32))
    assert list(P.get_depth_range(0)) == [0.2, 0.8]


@pytest.mark.gpu
def test_planner_from_files_end_to_end(tmp_path):
    """GraspNeRFPlanner.__call__ (main.py:189-209) on files: checkpoint in the reference's layout, six PNG renderings,
    camera_pose.npy -> grasps; equals planner.plan() on the arrays the I/O half must have produced."""
    path, sd = _checkpoint(str(tmp_path))
    imgs, w2c = _write_scene_files(str(tmp_path))
    P = planner.GraspNeRFPlanner(dict(CFG), path, str(tmp_path), os.path.join(tmp_path, 'rgb'),
                                 database_name='vgn_syn/test/packed/packed_170-220/scene/w_0.8', seed=11)
    assert P.step == 4321 and tuple(P.img_wh) == (512, 288)
    np.testing.assert_allclose(P.K, [[357.048, 0, 255.8], [0, 357.048, 143.8], [0, 0, 1]], rtol=1e-6)   # main.py:105-112
    grasps, scores, toc = P([0, 1, 2, 3, 4, 5], round_idx=1, n_grasp=2)
    images = np.stack([planner.resize_bilinear_u8(i, (512, 288)) for i in imgs]).astype(np.float32).transpose(0, 3, 1, 2) / 255
    K = np.repeat(P.K[None].astype(np.float32), 6, 0)
    g, _ = planner.plan(P.net, images, w2c.astype(np.float32), K, np.tile(np.float32([0.2, 0.8]), (6, 1)), P.bbox3d, seed=11 + 1 + 2,
                        tsdf_thres_high=0.0, tsdf_thres_low=-0.85)
    assert len(grasps) == len(g['index']) and len(scores) == len(grasps)
    for gr, q, t, w in zip(grasps, g['quat'], g['pos'], g['width']):
        np.testing.assert_allclose(gr.quat, q, atol=1e-4)          # MIOpen's 2D backbones are not run-to-run deterministic (~1e-5)
        np.testing.assert_allclose(gr.translation, t, atol=1e-9)
        assert abs(gr.width - w) < 1e-5
    assert 0 < toc < 60


def test_grasp_exposes_the_pose_interface_of_the_reference():
    """gd.grasp.Grasp as its consumers read it (clutter_removal.py:198-199, simulation.execute_grasp): pose.rotation.as_quat()
    / as_matrix(), pose.translation, width."""
    from graspnerf_amd.planner import Grasp
    q = np.array([0.1, -0.3, 0.2, 0.9])
    q /= np.linalg.norm(q)
    g = Grasp(q, [0.1, 0.2, 0.3], 0.05)
    assert np.allclose(g.pose.rotation.as_quat(), q) and np.allclose(g.pose.translation, [0.1, 0.2, 0.3]) and g.width == 0.05
    R = g.pose.rotation.as_matrix()
    from scipy.spatial.transform import Rotation
    assert np.allclose(R, Rotation.from_quat(q).as_matrix(), atol=1e-12)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
