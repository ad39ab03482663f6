"""graspnerf_amd.losses vs the reference's loss.py evaluated on the same tensors (golden from tools/make_goldens.py)."""
import os

import numpy as np
import torch

from graspnerf_amd import losses
from graspnerf_amd.synth import synth_loss_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_losses_match_reference():
    G = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_losses.npz')))
    pr, gt = synth_loss_case()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    p = {k: (tuple(t(x) for x in v) if isinstance(v, tuple) else t(v)) for k, v in pr.items()}
    out = {}
    out.update(losses.render_loss(p))
    out.update(losses.depth_loss(p, t(gt['true_depth']), t(gt['depth_range'])))
    out.update(losses.sdf_loss(p, t(gt['sdf_gt'])))
    out.update(losses.vgn_loss(p['vgn_pred'], tuple(t(x) for x in gt['grasp_info'])))
    assert set(G) == set(out)
    for k, v in G.items():
        np.testing.assert_allclose(out[k].detach().numpy().reshape(-1), v, rtol=2e-5, atol=1e-7, err_msg=k)
    tot = losses.total_loss(out)
    ref_tot = sum(float(v.mean()) for k, v in G.items() if k.startswith('loss'))
    assert abs(float(tot) - ref_tot) < 1e-5


def test_stacked_losses_equal_the_per_scene_ones():
    """losses.* with scenes=B (what Trainer.step runs on forward_scenes(..., stacked=True)): entry b of every term equals the
    one-scene term of scene b, the total is the sum of the per-scene totals, and so are the gradients."""
    from graspnerf_amd.renderer import NeuralRayRenderer
    from graspnerf_amd.trainer import train_losses, train_losses_stacked
    B = 3
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    scenes, datas = [], []
    for b in range(B):
        pr, gt = synth_loss_case(seed=20 + b)
        if b == 1:
            gt['grasp_info'] = (gt['grasp_info'][0], np.zeros_like(gt['grasp_info'][1]), *gt['grasp_info'][2:])   # no positive label
        p = {k: (tuple(t(x) for x in v) if isinstance(v, tuple) else t(v)) for k, v in pr.items()}
        scenes.append(p)
        datas.append({'ref_imgs_info': {k: t(gt[k]) for k in ('true_depth', 'depth_range', 'sdf_gt')}, 'grasp_info': tuple(t(x) for x in gt['grasp_info'])})
    leaf_keys = ('pixel_colors_nr', 'pixel_colors_nr_fine', 'depth_mean', 'depth_mean_fine', 'volume', 'sdf_gradient_error')
    shared, per_view = NeuralRayRenderer._SHARED_KEYS, NeuralRayRenderer._PER_VIEW_KEYS
    st = {}
    for k in scenes[0]:
        if k == 'vgn_pred':
            st[k] = tuple(torch.stack([s[k][i] for s in scenes]).requires_grad_(True) for i in range(3))
        elif k in shared:
            st[k] = scenes[0][k]
        else:
            st[k] = (torch.stack if k in per_view else torch.cat)([s[k] for s in scenes])
            if k in leaf_keys:
                st[k].requires_grad_(True)
    outs = NeuralRayRenderer.unstack({k: v for k, v in st.items() if k != 'vgn_pred'}, B)
    for b, o in enumerate(outs):
        o['vgn_pred'] = tuple(x[b] for x in st['vgn_pred'])
        assert all(o[k].shape == scenes[b][k].shape for k in o if k != 'vgn_pred'), 'unstack gives the one-scene shapes'
    terms = train_losses_stacked(st, datas)
    per = [train_losses(o, d) for o, d in zip(outs, datas)]
    assert set(terms) == set(per[0])
    for k, v in terms.items():
        assert v.shape[0] == B
        for b in range(B):
            np.testing.assert_allclose(v[b].detach().float().mean().numpy(), per[b][k].detach().float().mean().numpy(), rtol=2e-6, atol=1e-7, err_msg=k)
    tot = losses.total_loss(terms, scenes=B)
    ref = sum(losses.total_loss(x) for x in per)
    assert abs(float(tot.detach()) - float(ref.detach())) < 1e-5 * abs(float(ref.detach()))
    leaves = [st[k] for k in leaf_keys] + list(st['vgn_pred'])
    g1 = torch.autograd.grad(tot, leaves)
    g2 = torch.autograd.grad(ref, leaves)
    for k, a, b in zip(list(leaf_keys) + ['q', 'r', 'w'], g1, g2):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-9, err_msg=k)


def test_render_loss_without_ray_mask():
    """Renderer cfg use_ray_mask false drops the key (renderer.py:129-132): the train step's render loss is then the plain mean
    (loss.py:70-74, RenderLoss use_ray_mask false), per scene and stacked."""
    from graspnerf_amd.trainer import train_losses, train_losses_stacked
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    pr, gt = synth_loss_case(seed=3)
    p = {k: (tuple(t(x) for x in v) if isinstance(v, tuple) else t(v)) for k, v in pr.items() if k != 'ray_mask'}
    d = {'ref_imgs_info': {k: t(gt[k]) for k in ('true_depth', 'depth_range', 'sdf_gt')}, 'grasp_info': tuple(t(x) for x in gt['grasp_info'])}
    terms = train_losses(p, d)
    want = 0.01 * ((p['pixel_colors_nr'] - p['pixel_colors_gt']) ** 2).sum(-1).mean(1)
    assert torch.allclose(terms['loss_rgb_nr'], want)
