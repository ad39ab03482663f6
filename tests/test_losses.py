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
