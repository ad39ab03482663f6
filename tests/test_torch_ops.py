"""torch.ops.graspnerf.* (csrc/gnr_torch_ops.cpp, TORCH_LIBRARY over the C ABI): registered, GPU-only, and the same
bits as the ctypes route (HotPath) for sample_volume (renderer.py:164-199), render (renderer.py:201-220) and the volume's backward."""
import numpy as np
import pytest
import torch

import os

from graspnerf_amd import weights, torch_ops
from graspnerf_amd.synth import make_scene

pytestmark = pytest.mark.skipif(not os.path.exists(torch_ops.LIB_PATH), reason='libgnr_torch.so is optional and was not built (csrc/build.sh)')


def test_operators_are_registered_and_refuse_cpu_tensors():
    ops = torch_ops.load()
    for name in torch_ops.OPS:
        assert hasattr(ops, name), name
    s = str(ops.render_rays.default._schema)
    assert 'Tensor? que_imgs' in s and '-> Tensor[]' in s
    z = torch.zeros(1)
    with pytest.raises(NotImplementedError):               # no CPU kernel is registered: the dispatcher refuses, nothing falls back
        ops.sample_volume(z, z, z, z, z, z, z, z, 16)


@pytest.mark.gpu
def test_operators_return_the_bits_of_the_ctypes_route(weights_np):
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    ops = torch_ops.load()
    wc_np, wf_np = weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine')
    hp = HotPath(wc_np, wf_np)
    bref, bque = batch_scenes([make_scene(s, 'cfg1') for s in (0, 1)])
    r = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
    q = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
    scene = (r['imgs'], r['img_feats'], r['ray_feats'], r['poses'], r['Ks'], r['depth_range'])
    bbox_min = r['bbox3d'][:, 0].contiguous()
    vol = ops.sample_volume(*scene, bbox_min, hp.wc, 16)
    assert torch.equal(vol, hp.sample_volume(bref, 16))
    cfg = {'depth_sample_num': 16, 'fine_depth_sample_num': 16, 'ray_batch_num': 24}
    out = ops.render_rays(*scene, q['coords'], q['pose'], q['K'], q['depth_range'], q['imgs'], hp.wc, hp.wf, 16, 16, 2, 8, 24)
    co, fi = hp.render(bref, bque, cfg)
    assert len(out) == 20
    for lvl, ref in ((0, co), (1, fi)):
        for i, k in enumerate(torch_ops.RENDER_KEYS):
            got = out[10 * lvl + i]
            assert torch.equal(got.bool() if k == 'ray_mask' else got, ref[k]), (lvl, k)
    # the training pair: forward with saved states, then the backward twins (deterministic: bit-equal to the ctypes route)
    can = weights.canonical_blob(weights_np, 'coarse')
    hp.set_bwd_weights(weights.pack_bwd(can))
    can_dev = torch.from_numpy(can).cuda()
    v2, ws, tws = ops.sample_volume_train(*scene, bbox_min, hp.wc, 16)
    assert torch.equal(v2, vol)
    dvol = torch.from_numpy(np.random.default_rng(3).standard_normal((2, 1, 16, 16, 16)).astype(np.float32)).cuda()
    dcan, dray, dimg = ops.sample_volume_bwd(*scene, ws, tws, dvol, hp.wc, hp.wb['coarse'], can_dev, 16)
    hp.sample_volume_train(bref, 16)
    dcan2, dray2, dimg2 = hp.sample_volume_bwd(dvol, can_dev)
    torch.cuda.synchronize()
    assert torch.equal(dcan, dcan2) and float(dcan.abs().max()) > 0
    for a, b in ((dray, dray2), (dimg, dimg2)):            # feature-map gradients: float-atomic scatter, equal to rounding
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.gpu
def test_operators_check_what_they_hand_to_the_c_abi(weights_np):
    """The C ABI takes raw device pointers: the operators must turn a short, misplaced or mis-shaped tensor into an error, not into an
    out-of-bounds device read -- and an input that requires grad must not come back as an output that silently carries none."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    ops = torch_ops.load()
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    bref, bque = batch_scenes([make_scene(0, 'cfg1')])
    r = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
    q = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
    scene = (r['imgs'], r['img_feats'], r['ray_feats'], r['poses'], r['Ks'], r['depth_range'])
    bb = r['bbox3d'][:, 0].contiguous()
    with pytest.raises(RuntimeError, match='weights'):
        ops.sample_volume(*scene, bb, hp.wc[:-8], 16)                       # a truncated packed blob
    with pytest.raises(RuntimeError, match='bbox_min'):
        ops.sample_volume_train(*scene, bb[:, :2].contiguous(), hp.wc, 16)
    with pytest.raises(RuntimeError, match='que_pose'):
        ops.render_rays(*scene, q['coords'], q['pose'][:, :2].contiguous(), q['K'], q['depth_range'], None, hp.wc, hp.wf, 16, 16)
    with pytest.raises(RuntimeError, match='que_imgs'):
        ops.render_rays(*scene, q['coords'], q['pose'], q['K'], q['depth_range'], q['imgs'][:, :, ::2].contiguous(), hp.wc, hp.wf, 16, 16)
    with pytest.raises(RuntimeError):
        ops.sample_volume(r['imgs'], r['img_feats'], r['ray_feats'], r['poses'].cpu(), r['Ks'], r['depth_range'], bb, hp.wc, 16)   # a scene tensor on the host
    can = weights.canonical_blob(weights_np, 'coarse')
    hp.set_bwd_weights(weights.pack_bwd(can))
    vol, ws, tws = ops.sample_volume_train(*scene, bb, hp.wc, 16)
    dvol = torch.ones_like(vol)
    with pytest.raises(RuntimeError, match='canonical'):
        ops.sample_volume_bwd(*scene, ws, tws, dvol, hp.wc, hp.wb['coarse'], torch.from_numpy(can[:100]).cuda(), 16)
    with pytest.raises(RuntimeError, match='weights_bwd'):
        ops.sample_volume_bwd(*scene, ws, tws, dvol, hp.wc, hp.wc, torch.from_numpy(can).cuda(), 16)
    with pytest.raises(RuntimeError, match='ws / tws'):
        ops.sample_volume_bwd(*scene, ws, tws[:1000], dvol, hp.wc, hp.wb['coarse'], torch.from_numpy(can).cuda(), 16)
    # autograd: the forward runs, a backward through its outputs raises (the differentiable route is renderer.py's autograd.Functions)
    rf = r['ray_feats'].clone().requires_grad_(True)
    v = ops.sample_volume_train(r['imgs'], r['img_feats'], rf, r['poses'], r['Ks'], r['depth_range'], bb, hp.wc, 16)[0]
    assert v.requires_grad
    with pytest.raises(RuntimeError, match='not implemented|derivative'):
        v.sum().backward()
