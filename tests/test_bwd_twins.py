"""Backward twins of the HIP path against PyTorch autograd over graspnerf_amd/autograd_path.py (which is itself pinned
to the reference's backward by tests/test_train_step.py).  Round 1: gnr_depth_mean_bwd."""
import numpy as np
import pytest
import torch

from graspnerf_amd import weights
from graspnerf_amd.synth import make_scene


def test_bwd_blob_layout(weights_np):
    """CPU: the transposed fragments hold W^T where the chained-MFMA convention expects it."""
    from graspnerf_amd import _lib
    can = weights.canonical_blob(weights_np, 'coarse')
    pb = weights.pack_bwd(can)
    assert pb.size == _lib.lib().gnr_packed_bwd_floats() == 2048
    W2 = weights_np['dist_decoder.mean_decoder.2.weight']
    W1 = weights_np['dist_decoder.mean_decoder.0.weight']
    nat = lambda j, g: 16 * (j // 4) + 4 * g + j % 4
    for j in range(8):
        for nb in range(2):
            for lane in range(64):
                r16, g = lane & 15, lane >> 4
                # frag[(j, nb, lane)] = W^T[out(nb, lane&15)][in = nat(j, g)] = W[nat(j, g)][out]
                assert pb[(j * 64 + lane) * 2 + nb] == W2[nat(j, g), 16 * nb + r16]
                ch = 8 * (r16 // 4) + 4 * nb + r16 % 4                  # gather layout of the output rows
                assert pb[1024 + (j * 64 + lane) * 2 + nb] == W1[nat(j, g), ch]
    g = np.arange(36958, dtype=np.float32)
    parts = weights.split_canonical(g, 'fine')
    assert parts['fine_dist_decoder.mean_decoder.2.weight'][0, 0] == 1056 and len(parts) == 63


@pytest.mark.gpu
@pytest.mark.parametrize('level,pn', [('coarse', 333), ('fine', 64)])
def test_depth_mean_bwd_matches_autograd(level, pn, weights_np):
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    from graspnerf_amd import autograd_path as ag
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    hp.set_bwd_weights(weights.pack_bwd(weights.canonical_blob(weights_np, 'coarse')),
                       weights.pack_bwd(weights.canonical_blob(weights_np, 'fine')))
    scenes = [make_scene(s, 'cfg1') for s in (0, 1)]
    bref, _ = batch_scenes(scenes)
    rng = np.random.default_rng(2)
    H, Wd = bref['imgs'].shape[-2:]
    coords = np.stack([rng.uniform(-1, Wd, (2, pn)), rng.uniform(-1, H, (2, pn))], -1).astype(np.float32)
    dmean = rng.standard_normal((2, 3, pn, 2)).astype(np.float32)
    prep = hp.prepare(bref, 1)
    dcan, dray = hp.depth_mean_bwd(bref, coords, dmean, level, prepared=prep)
    torch.cuda.synchronize()
    got = weights.split_canonical(dcan.cpu().numpy(), level)
    dec = 'dist_decoder.' if level == 'coarse' else 'fine_dist_decoder.'
    # autograd reference, scene by scene
    P = {k: torch.from_numpy(v).cuda().requires_grad_(k.startswith(dec + 'mean_decoder')) for k, v in weights_np.items()}
    ray = torch.from_numpy(bref['ray_feats']).cuda().requires_grad_(True)
    tot = 0
    for b in range(2):
        ref = {'imgs': torch.from_numpy(bref['imgs'][b]).cuda(), 'ray_feats': ray[b]}
        m = ag.depth_mean(P, ref, torch.from_numpy(coords[b]).cuda(), dec)           # coords are used as (x, y)
        tot = tot + (m * torch.from_numpy(dmean[b]).cuda()).sum()
    tot.backward()
    for name in ('0.weight', '0.bias', '2.weight', '2.bias', '4.weight', '4.bias'):
        k = dec + 'mean_decoder.' + name
        r = P[k].grad.cpu().numpy()
        assert np.abs(got[k] - r).max() <= 2e-4 * np.abs(r).max() + 1e-6, k
    others = [k for k in got if 'mean_decoder' not in k]
    assert all(np.all(got[k] == 0) for k in others)
    r = ray.grad.cpu().numpy()
    assert np.abs(dray.cpu().numpy() - r).max() <= 2e-4 * np.abs(r).max() + 1e-6
    # forward values of the twin pair agree as well
    fwd = hp.depth_mean(bref, coords, level, prepared=prep).cpu().numpy()
    with torch.no_grad():
        m0 = ag.depth_mean(P, {'imgs': torch.from_numpy(bref['imgs'][0]).cuda(), 'ray_feats': ray[0]},
                           torch.from_numpy(coords[0]).cuda(), dec).cpu().numpy()
    assert np.abs(fwd[0] - m0).max() < 1e-4
