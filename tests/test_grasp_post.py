"""Grasp post-processing (reference planner process()/select(), src/nr/main.py:23-84): the numpy oracle against the
golden produced by the reference's own functions (scipy.ndimage inside), and the HIP kernels against both."""
import os

import numpy as np
import pytest

from graspnerf_amd.synth import synth_head_outputs
from oracle import grasp_post_oracle as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def G():
    return dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_post.npz')))


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_oracle_matches_reference_functions(seed, G):
    tsdf, qual, rot, width = synth_head_outputs(seed)
    hi, lo = G[f's{seed}.thres']
    q = P.process(tsdf[0, 0], qual[0, 0], rot[0], width[0, 0], thres_high=hi, thres_low=lo)
    assert np.array_equal(q, G[f's{seed}.qual'])                     # bit-exact: same accumulation order as scipy
    idx, score, quat, w = P.select(q, rot[0], width[0, 0])
    assert np.array_equal(idx, G[f's{seed}.index'])
    assert np.array_equal(score, G[f's{seed}.score']) and np.array_equal(w, G[f's{seed}.width'])
    # the reference wraps the quaternion in scipy's Rotation (normalises, x y z w order kept)
    qn = quat.astype(np.float64) / np.linalg.norm(quat.astype(np.float64), axis=1, keepdims=True)
    assert np.abs(qn - G[f's{seed}.quat']).max() < 1e-6
    assert np.array_equal(idx.astype(np.float64), G[f's{seed}.pos'])


def test_filters_against_scipy_on_random_input():
    """The three restated filters vs scipy.ndimage itself (the dependency the reference calls), ragged sizes."""
    ndi = pytest.importorskip('scipy.ndimage')
    rng = np.random.default_rng(3)
    x = rng.random((7, 12, 9)).astype(np.float32)
    assert np.array_equal(P.gaussian_filter_nearest(x, 1.0), ndi.gaussian_filter(x, sigma=1.0, mode='nearest'))
    assert np.array_equal(P.maximum_filter_reflect(x, 4), ndi.maximum_filter(x, size=4))
    a, m = rng.random(x.shape) > 0.8, rng.random(x.shape) > 0.3
    assert np.array_equal(P.masked_dilation(a, m, 2), ndi.binary_dilation(a, iterations=2, mask=m))


@pytest.mark.gpu
def test_hip_post_processing_is_bit_exact(G):
    """Through the C ABI (gnr_grasp_select_fwd), three scenes in one batch with the planner thresholds, and the third
    golden case with the function defaults: processed volume, selected voxels, scores, quaternions, widths."""
    import torch
    from graspnerf_amd.grasp_post import GraspSelector, grasps_from_selection
    sel = GraspSelector()
    vols = [synth_head_outputs(s) for s in (0, 1, 2)]
    cat = [torch.from_numpy(np.concatenate([v[i] for v in vols])).cuda() for i in range(4)]
    for seeds, kw in (((0, 1), dict(tsdf_thres_high=0.0, tsdf_thres_low=-0.85)), ((2,), {})):
        out = sel(*cat, **kw)
        torch.cuda.synchronize()
        for s in seeds:
            assert np.array_equal(out['qual'][s].cpu().numpy(), G[f's{s}.qual']), f'scene {s} processed quality'
            g = grasps_from_selection(out, s, voxel_size=1.0)
            assert int(out['count'][s]) == len(G[f's{s}.index'])
            assert np.array_equal(g['index'], G[f's{s}.index'])
            assert np.array_equal(g['score'], G[f's{s}.score']) and np.array_equal(g['width'], G[f's{s}.width'])
            assert np.abs(g['quat'] - G[f's{s}.quat']).max() < 1e-6
            assert np.array_equal(g['pos'], G[f's{s}.pos'])


@pytest.mark.gpu
def test_hip_post_processing_ragged_and_truncated():
    """R not a multiple of anything, more survivors than max_grasps, against the oracle."""
    import torch
    from graspnerf_amd.grasp_post import GraspSelector
    rng = np.random.default_rng(11)
    R = 13
    tsdf = (rng.random((2, 1, R, R, R)) * 2 - 1).astype(np.float32)
    qual = rng.random((2, 1, R, R, R)).astype(np.float32) ** 0.05          # mostly > 0.9 after smoothing
    rot = rng.standard_normal((2, 4, R, R, R)).astype(np.float32)
    width = (rng.random((2, 1, R, R, R)) * 10).astype(np.float32)
    sel = GraspSelector(max_grasps=8)
    out = sel(tsdf, qual, rot, width, tsdf_thres_high=0.0, tsdf_thres_low=-0.85, max_filter_size=3)
    torch.cuda.synchronize()
    for b in range(2):
        q = P.process(tsdf[b, 0], qual[b, 0], rot[b], width[b, 0], thres_high=0.0, thres_low=-0.85)
        assert np.array_equal(out['qual'][b].cpu().numpy(), q)
        idx, score, quat, w = P.select(q, rot[b], width[b, 0], size=3)
        n = int(out['count'][b])
        assert n == len(idx) and n > 8
        assert np.array_equal(out['index'][b].cpu().numpy(), idx[:8])
        assert np.array_equal(out['score'][b].cpu().numpy(), score[:8])
        assert np.array_equal(out['quat'][b].cpu().numpy(), quat[:8])


@pytest.mark.gpu
def test_truncated_selection_is_reported():
    """The reference's select() returns every NMS survivor (main.py:70-84): a selection that does not fit the buffers raises
    instead of silently returning the first max_grasps candidates in index order."""
    import torch
    from graspnerf_amd import _lib
    from graspnerf_amd.grasp_post import GraspSelector, grasps_from_selection
    from graspnerf_amd.synth import synth_head_outputs
    tsdf, qual, rot, width = synth_head_outputs(0)
    sel = GraspSelector(max_grasps=2048)(tsdf, qual, rot, width, tsdf_thres_high=0.0, tsdf_thres_low=-0.85)
    n = int(sel['count'][0])
    assert n > 3
    assert len(grasps_from_selection(sel, 0)['index']) == n
    small = GraspSelector(max_grasps=3)(tsdf, qual, rot, width, tsdf_thres_high=0.0, tsdf_thres_low=-0.85)
    assert int(small['count'][0]) == n
    with pytest.raises(_lib.GnrError):
        grasps_from_selection(small, 0)
