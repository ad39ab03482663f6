"""Row-by-row check of the CPU oracle against the reference's own functions on random inputs.
Only runs where the read-only reference is mounted (/root/reference, i.e. the build container): the reference cannot
travel to the GPU box, there the committed goldens stand in for it.  Never imported by -m gpu tests, smoke() or bench."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import graspnerf_oracle as O

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF + '/src/nr/network'), reason='reference checkout not mounted')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def R():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from ref_import import import_reference
    import_reference()
    import network.render_ops as rops
    import network.dist_decoder as dd
    import network.ibrnet as ib
    import network.aggregate_net as an
    import network.neus as neus
    return dict(rops=rops, dd=dd, ib=ib, an=an, neus=neus)


def rand_cams(rng, V):
    from graspnerf_amd.synth import random_scene
    ref, que, m = random_scene(int(rng.integers(0, 10 ** 6)))
    return ref, que, m


@pytest.mark.parametrize('seed', range(8))
def test_projection_rows_P1_P3(seed, R):
    rng = np.random.default_rng(seed)
    ref, que, m = rand_cams(rng, None)
    N = 257
    pts = torch.from_numpy((rng.uniform(-0.3, 0.3, (N, 3)) + [0, 0, 0.1]).astype(np.float32))
    info = {'poses': torch.from_numpy(ref['poses']), 'Ks': torch.from_numpy(ref['Ks']), 'imgs': torch.from_numpy(ref['imgs'])}
    uv, z, mask, dirv = O.project_points(pts, info['poses'], info['Ks'], m['H'], m['W'])
    rfn = ref['poses'].shape[0]
    pts4 = pts.reshape(1, N, 1, 3)
    uv_r, valid_r, z_r = R['rops'].project_points_coords(pts4.reshape(-1, 3), info['poses'], info['Ks'])
    # same formula, different contraction order (einsum vs batched matmul): fp32 rounding only
    wc = z.reshape(rfn, N).abs() > 1e-2                        # u = x/z is ill-conditioned next to the camera plane
    assert torch.allclose(uv.reshape(rfn, N, 2)[wc], uv_r.reshape(rfn, N, 2)[wc], rtol=1e-4, atol=1e-3)
    assert torch.allclose(z.reshape(rfn, N), z_r.reshape(rfn, N), rtol=1e-6, atol=1e-7)
    dir_r = R['rops'].project_points_directions(info['poses'], pts4.reshape(-1, 3))
    assert torch.allclose(dirv, dir_r.reshape(rfn, N, 3), atol=1e-6)
    _, _, _, mk = R['rops'].project_points_ref_views(info, pts4.reshape(-1, 3))
    u, v = uv[..., 0], uv[..., 1]
    marg = torch.minimum(torch.minimum((u + 0.5).abs(), (u - (m['W'] - 0.5)).abs()), torch.minimum((v + 0.5).abs(), (v - (m['H'] - 0.5)).abs()))
    safe = (marg > 1e-2) & wc
    assert torch.equal(mask.reshape(rfn, N)[safe], mk.reshape(rfn, N).bool()[safe]) and safe.float().mean() > 0.9


@pytest.mark.parametrize('seed', range(8))
def test_bilinear_row_I1(seed, R):
    rng = np.random.default_rng(100 + seed)
    V, C = int(rng.integers(1, 5)), int(rng.integers(1, 9))
    h, w = int(rng.integers(8, 40)), int(rng.integers(8, 40))
    same = seed % 2 == 0                                        # full-res map (align_corners True) or another resolution
    fh, fw = (h, w) if same else (int(rng.integers(3, 20)), int(rng.integers(3, 20)))
    feat = torch.from_numpy(rng.standard_normal((V, C, fh, fw)).astype(np.float32))
    uv = torch.from_numpy(np.stack([rng.uniform(-2, w + 1, (V, 99)), rng.uniform(-2, h + 1, (V, 99))], -1).astype(np.float32))
    mask = torch.ones(V, 99)
    ref = R['rops'].interpolate_feature_map(feat, uv, mask, h, w)
    got = O.bilinear_border(feat, uv, h, w)
    assert torch.allclose(got, ref, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize('seed', range(6))
def test_depth_sampling_rows_S1_S3_C0_F1(seed, R):
    rng = np.random.default_rng(200 + seed)
    rn, dn, fdn = int(rng.integers(1, 30)), int(rng.integers(3, 48)), int(rng.integers(3, 48))
    dr = torch.tensor([[rng.uniform(0.1, 0.3), rng.uniform(0.6, 1.2)]], dtype=torch.float32)
    coords = torch.zeros(1, rn, 2)
    d_ref, _ = R['rops'].sample_depth(dr, coords, dn, False)
    d = O.sample_depth(dr[0], rn, dn)
    assert torch.equal(d, d_ref[0])
    alpha = torch.from_numpy(rng.random((1, rn, dn)).astype(np.float32))
    assert torch.allclose(O.alpha_to_hit_prob(alpha[0]), R['rops'].alpha_values2hit_prob(alpha)[0], atol=1e-7)
    hp = torch.from_numpy(rng.random((1, rn, dn)).astype(np.float32) ** 3)
    torch.manual_seed(seed)
    fd_ref = R['rops'].sample_fine_depth(d_ref, hp, dr, fdn, False)
    fd, _ = O.sample_fine_depth(d, hp[0], dr[0], fdn)
    assert torch.allclose(fd, fd_ref[0], atol=1e-6)
    u = torch.rand(rn, fdn)
    torch.manual_seed(seed + 50)
    fd_ref = R['rops'].sample_fine_depth(d_ref, hp, dr, fdn, True)
    torch.manual_seed(seed + 50)
    fd, _ = O.sample_fine_depth(d, hp[0], dr[0], fdn, u=torch.rand(1, rn, fdn)[0])
    assert torch.allclose(fd, fd_ref[0], atol=1e-6)
    lo, hi = O.ray_half_intervals(d, dr[0])
    inv = R['rops'].depth2inv_dists(d_ref, dr)
    near, far = R['dd'].get_near_far_points(torch.ones(1, 1, rn, dn), inv[None], torch.tensor([[1.0, 2.0]]), True)
    base = (-1.0 - (-1.0)) / ((-0.5) - (-1.0))
    assert torch.allclose(base - near[0, 0].reshape(-1), lo, atol=1e-7) and torch.allclose(far[0, 0].reshape(-1) - base, hi, atol=1e-7)


@pytest.mark.parametrize('seed', range(6))
def test_decoder_rows_D1_D3(seed, R, weights_np):
    rng = np.random.default_rng(300 + seed)
    V, N = int(rng.integers(1, 5)), 77
    dec = R['dd'].MixtureLogisticsDistDecoder({'use_vis': False})
    dec.load_state_dict({k[len('dist_decoder.'):]: torch.from_numpy(v) for k, v in weights_np.items() if k.startswith('dist_decoder.')})
    f = torch.from_numpy((0.7 * rng.standard_normal((V, 1, N, 1, 32))).astype(np.float32))
    z = torch.from_numpy(rng.uniform(0.05, 1.0, (V, 1, N, 1)).astype(np.float32))
    dr = torch.from_numpy(np.stack([rng.uniform(0.1, 0.3, V), rng.uniform(0.6, 1.2, V)], 1).astype(np.float32))
    with torch.no_grad():
        prj_mean, prj_var, prj_vis, prj_aw = dec(f)
        alpha_r, vis_r, hit_r = dec.compute_prob(z, torch.empty(1, 0), prj_mean, prj_var, prj_vis, prj_aw, True, dr)
    W = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    hit, vis = O.decode_hit_vis(W, 'dist_decoder.', f.reshape(V, N, 32), z.reshape(V, N), torch.ones(V, N, dtype=torch.bool), dr, 0.005, 0.005)
    assert torch.allclose(hit, hit_r.reshape(V, N), atol=2e-6) and torch.allclose(vis, vis_r.reshape(V, N), atol=2e-6)


@pytest.mark.parametrize('seed', range(4))
def test_mean_variance_embedding_and_posenc_rows_A2_A3(seed, R):
    rng = np.random.default_rng(400 + seed)
    x = torch.from_numpy(rng.standard_normal((5, 9, 4, 7)).astype(np.float32))
    w = torch.from_numpy(rng.random((5, 9, 4, 1)).astype(np.float32))
    mr, vr = R['ib'].fused_mean_variance(x, w)
    m, v = O.weighted_mean_var(x.permute(2, 0, 1, 3).reshape(4, 45, 7), w.permute(2, 0, 1, 3).reshape(4, 45, 1))
    assert torch.allclose(m, mr.reshape(45, 7), atol=1e-6) and torch.allclose(v, vr.reshape(45, 7), atol=1e-6)
    embed_fn, dim = R['neus'].get_embedder(3, 3)
    p = torch.from_numpy(rng.uniform(-0.5, 0.5, (33, 3)).astype(np.float32))
    assert dim == 21 and torch.equal(O.embed_points(p), embed_fn(p))
    for n in (16, 40):
        net = R['ib'].IBRNetWithNeuRayNeus(32, n_samples=n)
        assert torch.equal(O.sinusoid_table(n), net.pos_encoding[0].cpu())


def test_neus_alpha_row_N1(R, weights_np):
    rng = np.random.default_rng(7)
    rn, dn = 11, 13
    net = R['an'].NeusAggregationNet({'sample_num': dn, 'init_s': 0.3, 'fix_s': 0})
    sdf = torch.from_numpy(rng.uniform(-0.3, 0.3, (rn, dn)).astype(np.float32))
    grad = torch.from_numpy(rng.standard_normal((1, rn, dn, 3)).astype(np.float32))
    qdir = torch.from_numpy(rng.standard_normal((rn, 3)).astype(np.float32))
    qdir = qdir / qdir.norm(dim=1, keepdim=True)
    depth = torch.sort(torch.from_numpy(rng.uniform(0.2, 0.8, (rn, dn)).astype(np.float32)), -1)[0]
    with torch.no_grad():
        ar = net._get_alpha_from_sdf(sdf, grad, qdir[None, :, None].expand(1, rn, dn, 3), R['rops'].depth2dists(depth[None]))
    a = O.neus_alpha(sdf, grad[0], qdir, depth, torch.tensor(0.3))
    assert torch.allclose(a, ar[0], atol=1e-6)
