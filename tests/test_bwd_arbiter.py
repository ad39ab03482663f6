"""Round 6: the float64 arbiter of the BACKWARD at the benched training shape (pytest -m gpu, through the C ABI).

The view kernels of the backward recompute the forward and run their dX chains on the f16 matrix cores, every fp32 operand an
fp16 pair, the upstream gradient of a (tile, view) normalised by a power of two (csrc/gnr_bwd.inc, GNR_BWD_PAIRS); the per-ray tails
use the same form.  The forward's licence for "f32" is tests/test_range_guard.py::test_fp64_arbiter_benched_shape; this is the same
arbiter for the gradients: on BASELINE configs[4]'s shape -- 8 scenes x 6 views of 288 x 512, the 40^3 volume of the coarse level,
a coarse pass of 512 rays x 40 samples, a fine pass of 512 x 40 on its own depths --

    |HIP gradient - float64|   against   |torch fp32 autograd - float64|

for EVERY hot-path parameter tensor of both levels and for d_ray_feats / d_img_feats, where "float64" is autograd over
tests/reference_autograd.py (the differentiable statement of the path, pinned by the reference's golden training steps) evaluated
in double and "torch fp32" the same statement in float (the reference's semantics: fp32 autograd, ibrnet.py:497-504,
train/trainer.py:146-158).  Three upstream scales: a mean-loss-like one (1 / number of outputs), x 2^10 (a loss scale), x 2^-30
(gradients of ~1e-14: the power-of-two normalisation at work; powers of two so that the two references scale exactly).
Requirement (the forward's): rms ratio <= 1.5, 99th-percentile ratio <= 2 -- met by the feature-map gradients and by three quarters of
the 110 parameter tensors (median ratio 0.96: the HIP path is as close to float64 as torch's fp32 autograd is); what the rest are is said
below and recorded tensor by tensor (gpurun_out/bwd_arbiter.json -> profiles/).  NOT the pair arithmetic: with GNR_OPT_FP32_CHAIN (every
product on fp32 instructions, ARBITER_FP32=1) the table is the same to two digits.
 * cancellation-dominated tensors: the var / aw decoders of the coarse level (gradient norms 10 - 50x below the mean decoder's: their
   upstream is the DIFFERENCE of the two logistic edges of a sample, (far - mean) sech^2(u1) - (near - mean) sech^2(u0) over an interval
   of 1 / 39 of the range, dist_decoder.py:109-142, and the sum over ~1e6 (view, sample) terms cancels to 1e-8 .. 1e-6): ratios 2.5 - 13 on
   ten tensors whose relative error is 3e-5.  These are ratios of two rounding errors of ill-conditioned sums, not a deficit of the path:
   over 1 / 2 / 4 / 8 scenes of the coarse pass the var decoder's bias reads hip 7e-6 / 9e-6 / 2.4e-4 / 2.7e-4 against torch fp32's
   3.2e-5 / 2.7e-6 / 1.5e-4 / 4.8e-5 (torch is the FARTHER one at one scene), both at an absolute 1e-12 .. 3e-11, while well-conditioned
   tensors stay at 2e-6 .. 1e-5 on both sides at every size.  (Round 6 first blamed the path's tanh, 1 - 2 rcp(e^2x + 1) on v_exp_f32 /
   v_rcp_f32 at ~2e-7 against libm's 6e-8; a ~1-ulp tanh with sech^2 = fma(-y, y, 1) in the backward kernels changed no ratio --
   profiles/r06_h_tanh_experiment.json, docs/experiments/r06_tanh_acc.patch.)
 * one-to-35-element bias tensors at 1.5 - 5: a ratio of two rounding errors of single numbers.
Gate: every tensor within 16 (a wrong kernel, a lost tile or a bad scale shows as hundreds: the border samples below did), three quarters
within the forward's (1.5, 2), the median within 1.2, the feature maps within (1.5, 2).

Kinks.  The path has ONE non-smooth point between the feature maps and the statistics: the ReLU of prob_embed.0 (aggregate_net.py:46-50).
A sample whose pre-activation lies within fp32 rounding of zero (|pre| ~ 1e-7 .. 1e-6: a handful of the 980 000 (view, sample) rows of a
scene) gets derivative 0 in one fp32 evaluation and 1 in another -- torch's fp32 autograd flips against float64 on such rows exactly as
the HIP path does, on OTHER rows (tools/dbg/coarse_pass_rows.py lists them: 26 rows where only the HIP path differs from float64, 15
where only torch fp32 does, 32 where both do, over the 8 scenes' coarse passes).  One flipped row with a large upstream moves every
parameter gradient of the level by ~1e-4 of its norm, i.e. decides an rms RATIO of two fp32 evaluations by itself and says nothing
about arithmetic.  The arbiter therefore runs twice:
 * `arithmetic`: prob_embed.0.bias + 8 on both levels -- every pre-activation positive (asserted), the path is smooth, the ratios
   measure arithmetic alone: the strict gate, all three upstream scales;
 * `as packed`: the test weights as they are -- the per-tensor ratios are RECORDED (gpurun_out/bwd_arbiter.json -> profiles/), the
   gate is the robust part: the median ratio over the tensors, and rms / p99 of the feature-map gradients with the flipped rows'
   pixels inside the quantile."""
import numpy as np
import pytest
import torch

from graspnerf_amd import weights
from graspnerf_amd.synth import make_scene
from conftest import PARITY_LOG

pytestmark = pytest.mark.gpu

B, RES, RN, DN = 8, 40, 512, 40
CFG = {'depth_sample_num': DN, 'fine_depth_sample_num': DN, 'ray_mask_view_num': 2, 'ray_mask_point_num': 8}
import os
PASSES = os.environ.get('ARBITER_PASSES', 'all')     # debugging aid: 'volume' / 'coarse' / 'fine' alone
MIN_E1 = {'v': float('inf')}
RECORD = {}          # variant / regime -> {tensor: (rms ratio, p99 ratio)}: dumped into PARITY_LOG and gpurun_out/bwd_arbiter.json


def _chain(ag, P, tref, q1, depth, dec, agg, rn, dn, hw):
    """statistics [rn*dn,65] (mean 32, var 32, wbar) and colours [rn*dn,3] of one render pass, as the HIP chain defines them"""
    pts, qdir = ag.ray_points(q1, depth)
    uv, z, mask, dirv = ag.project(pts, tref['poses'], tref['Ks'], *hw)
    f_ray, rgb, f_img = ag._gather(tref, uv, mask)
    near, far = -1 / q1['depth_range'][0], -1 / q1['depth_range'][1]
    di = (-1 / depth - near) / (far - near)
    half = torch.cat([di[:, 1:] - di[:, :-1], torch.full_like(di[:, :1], 1e6)], -1) / 2
    ext = torch.cat([half[:, :1], half], -1)
    hit, vis = ag.decode_hit_vis(P, dec, f_ray, z, mask, tref['depth_range'], ext[:, :-1].reshape(-1), ext[:, 1:].reshape(-1))
    taps = {}
    qd = qdir[:, None].expand(rn, dn, 3).reshape(-1, 3)
    _, _, col = ag.aggregate(P, agg, f_ray, rgb, f_img, hit, vis, mask, dirv, qd, pts, rn, dn, False, True, taps)
    v2 = taps['v2']
    wbar = (v2 / (v2.sum(0, keepdim=True) + 1e-8)).mean(0)
    if depth.dtype == torch.float64:
        MIN_E1['v'] = min(MIN_E1['v'], float(taps['e1'][mask.bool()].min()))        # smallest ReLU output over the valid rows
    return torch.cat([taps['mean'], taps['var'], wbar], -1), col.reshape(-1, 3)


@pytest.fixture(scope='module', params=['arithmetic', 'as packed'])
def case(request, weights_np):
    """HIP training forwards of the three passes (their contexts stay alive), the upstream gradients at scale 1, and the two
    references' gradients at scale 1 (accumulated over the scenes; linear in the upstream)."""
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    import reference_autograd as ag
    smooth = request.param == 'arithmetic'
    if smooth:
        weights_np = dict(weights_np)
        for k in ('agg_net.prob_embed.0.bias', 'fine_agg_net.prob_embed.0.bias'):
            weights_np[k] = weights_np[k] + np.float32(8.0)
    hp = HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine'))
    can = {lvl: weights.canonical_blob(weights_np, lvl) for lvl in ('coarse', 'fine')}
    hp.set_bwd_weights(weights.pack_bwd(can['coarse']), weights.pack_bwd(can['fine']))
    import os
    if os.environ.get('ARBITER_FP32'):
        hp.force_fp32_chain(True)
    can_dev = {lvl: torch.from_numpy(can[lvl]).cuda() for lvl in can}
    scenes = [make_scene(i, 'cfg2', with_query_image=False) for i in range(B)]
    bref, bque = batch_scenes(scenes)
    bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
    bq = {k: torch.from_numpy(v).cuda() for k, v in bque.items() if k != 'imgs'}
    g = torch.Generator().manual_seed(2026)
    up = {'dvol': torch.randn(B, 1, RES, RES, RES, generator=g).cuda() / (B * RES ** 3),
          'ds_c': torch.randn(B, RN * DN, 65, generator=g).cuda() / (B * RN * DN * 65), 'dc_c': torch.randn(B, RN * DN, 3, generator=g).cuda() / (B * RN * DN * 3),
          'ds_f': torch.randn(B, RN * DN, 65, generator=g).cuda() / (B * RN * DN * 65), 'dc_f': torch.randn(B, RN * DN, 3, generator=g).cuda() / (B * RN * DN * 3)}
    fine_depth = torch.sort(torch.rand(B, RN, DN, generator=g) * 0.5 + 0.25, -1)[0].cuda()
    prep = hp.prepare(bref, RES, RN, DN)
    st_c, _, geo_c, ctx_c = hp.render_chain_train(bq, None, 'coarse', CFG, prep)
    st_f, _, _, ctx_f = hp.render_chain_train(bq, fine_depth, 'fine', CFG, prep)
    hp.sample_volume_train(bref, RES, prepared=prep)      # last, as in the training forward (renderer.py): the volume's backward reads the
                                                          # point records of the regular workspace, which a later pass would overwrite (include/gnr.h)
    assert hp.range_status(prep) == 0
    coarse_depth = geo_c['depth']
    refs = {}
    hw = scenes[0][0]['imgs'].shape[-2:]
    # In-image tests on the border.  A sample whose projection into a view lies within rounding of the image border is inside for one
    # evaluation order and outside for another: the HIP path follows the REFERENCE's order (render_ops.py:98-104,126-128; its masks are
    # bit-exact against the reference's goldens, tests/test_gpu_parity.py), tests/reference_autograd.py on the GPU its own, in either
    # precision (scene 5's coarse pass has such a sample: a statistic that differs by 0.2).  That is not arithmetic: every sample whose
    # number of valid views differs between the HIP forward and either evaluation of the statement takes no part (zero upstream).
    dropped = 0
    for dt in (torch.float64, torch.float32):
        for b in range(B):
            tref = {k: torch.from_numpy(scenes[b][0][k]).cuda().to(dt) for k in ('poses', 'Ks')}
            q1 = {'coords': bq['coords'][b].to(dt), 'pose': bq['pose'][b].to(dt), 'K': bq['K'][b].to(dt), 'depth_range': bq['depth_range'][b].to(dt)}
            for depth, st, dsk, dck in ((coarse_depth, st_c, 'ds_c', 'dc_c'), (fine_depth, st_f, 'ds_f', 'dc_f')):
                pts, _ = ag.ray_points(q1, depth[b].to(dt))
                nvalid = ag.project(pts, tref['poses'], tref['Ks'], *hw)[2].sum(0).float()
                differ = nvalid != st[b, :, 65]
                dropped += int(differ.sum())
                up[dsk][b, differ] = 0
                up[dck][b, differ] = 0
    assert dropped <= 64, f'{dropped} samples on an image border: too many for rounding'
    for name, dt in (('f64', torch.float64), ('f32', torch.float32)):
        P = {k: torch.from_numpy(v).cuda().to(dt).requires_grad_(True) for k, v in weights_np.items()}
        dray, dimg = [], []
        for b in range(B):
            tref = {k: (torch.from_numpy(v).cuda().to(dt) if v.dtype.kind == 'f' else torch.from_numpy(v).cuda()) for k, v in scenes[b][0].items()}
            tref['ray_feats'].requires_grad_(True); tref['img_feats'].requires_grad_(True)
            q1 = {'coords': bq['coords'][b].to(dt), 'pose': bq['pose'][b].to(dt), 'K': bq['K'][b].to(dt), 'depth_range': bq['depth_range'][b].to(dt)}
            if PASSES in ('all', 'volume'):
                (ag.sample_volume(P, tref, RES) * up['dvol'][b:b + 1].to(dt)).sum().backward()
            if PASSES in ('all', 'coarse'):
                st, col = _chain(ag, P, tref, q1, coarse_depth[b].to(dt), 'dist_decoder.', 'agg_net.', RN, DN, hw)
                ((st * up['ds_c'][b].to(dt)).sum() + (col * up['dc_c'][b].to(dt)).sum()).backward()
            if PASSES in ('all', 'fine'):
                st, col = _chain(ag, P, tref, q1, fine_depth[b].to(dt), 'fine_dist_decoder.', 'fine_agg_net.', RN, DN, hw)
                ((st * up['ds_f'][b].to(dt)).sum() + (col * up['dc_f'][b].to(dt)).sum()).backward()
            dray.append(tref['ray_feats'].grad.double()); dimg.append(tref['img_feats'].grad.double())
            del tref
        refs[name] = {k: p.grad.double() for k, p in P.items() if p.grad is not None}
        refs[name]['d_ray_feats'] = torch.stack(dray); refs[name]['d_img_feats'] = torch.stack(dimg)
        del P
        torch.cuda.empty_cache()
    if smooth:
        assert MIN_E1['v'] > 1e-3, f"prob_embed.0 + 8 should keep every ReLU input positive (min {MIN_E1['v']})"
    return dict(hp=hp, can_dev=can_dev, up=up, ctx_c=ctx_c, ctx_f=ctx_f, refs=refs, smooth=smooth, variant=request.param, dropped=dropped)


def _hip_gradients(c, scale):
    hp, up = c['hp'], c['up']
    vol = hp.sample_volume_bwd(up['dvol'] * scale, c['can_dev']['coarse'])
    co = hp.render_chain_bwd(c['ctx_c'], up['ds_c'] * scale, up['dc_c'] * scale)
    fi = hp.render_chain_bwd(c['ctx_f'], up['ds_f'] * scale, up['dc_f'] * scale)
    torch.cuda.synchronize()
    if PASSES != 'all':
        z = lambda t: [torch.zeros_like(x) for x in t]
        vol, co, fi = (vol if PASSES == 'volume' else z(vol)), (co if PASSES == 'coarse' else z(co)), (fi if PASSES == 'fine' else z(fi))
    out = {}
    for k, v in weights.split_canonical(vol[0] + co[0], 'coarse').items():
        out[k] = v.double()
    for k, v in weights.split_canonical(fi[0], 'fine').items():
        out[k] = v.double()
    out['d_ray_feats'] = vol[1].double() + co[1].double() + fi[1].double()
    out['d_img_feats'] = vol[2].double() + co[2].double() + fi[2].double()
    return out


def _q(e):
    e = e.abs().reshape(-1)
    return float(e.pow(2).mean().sqrt()), float(torch.quantile(e[:4_000_000] if e.numel() > 4_000_000 else e, 0.99))


@pytest.mark.parametrize('regime,scale', [('mean loss', 1.0), ('x 2^10', 2.0 ** 10), ('x 2^-30', 2.0 ** -30)])
def test_fp64_arbiter_on_the_backward_at_the_benched_shape(regime, scale, case):
    if not case['smooth'] and scale != 1.0:
        pytest.skip('the kinked variant is recorded at one scale (the ratios do not depend on it: powers of two)')
    got = _hip_gradients(case, scale)
    r64, r32 = case['refs']['f64'], case['refs']['f32']
    rows, bad = {}, []
    for k, g64 in r64.items():
        if k.endswith('rgb_fc.4.bias'):                 # in front of a softmax over views: exactly zero, both sides hold rounding noise
            continue
        assert k in got, k
        want = g64 * scale
        e_hip, e_32 = got[k] - want, r32[k] * scale - want
        (rms_h, p99_h), (rms_o, p99_o) = _q(e_hip), _q(e_32)
        size = float(want.pow(2).mean().sqrt())
        assert bool(torch.isfinite(got[k]).all()) and size > 0, k
        # the floor: half an fp32 ulp of the tensor's own rms -- below it a ratio of two rounding errors says nothing (tensors of a few elements)
        floor = size * 2.0 ** -24
        ratio_rms, ratio_p99 = rms_h / (rms_o + floor), p99_h / (p99_o + floor)
        rows[k] = {'rms_ratio': round(ratio_rms, 3), 'p99_ratio': round(ratio_p99, 3), 'hip_rms_err_over_rms': rms_h / size, 'torch_fp32_rms_err_over_rms': rms_o / size, 'numel': int(want.numel())}
        if ratio_rms > 1.5 or ratio_p99 > 2.0:
            bad.append((k, rows[k]))
    tag = f"{case['variant']}, {regime}"
    RECORD[tag] = rows
    worst = max(rows.values(), key=lambda r: r['rms_ratio'])
    median = float(np.median([r['rms_ratio'] for r in rows.values()]))
    PARITY_LOG.append({'what': f'backward fp64 arbiter, benched shape, {tag}', 'tensors': len(rows),
                       'worst_rms_ratio': worst['rms_ratio'], 'worst_p99_ratio': max(r['p99_ratio'] for r in rows.values()),
                       'median_rms_ratio': median, 'tensors_beyond_the_gate': len(bad), 'samples_dropped_on_image_borders': case['dropped'],
                       'feature_maps': {k: rows[k] for k in ('d_ray_feats', 'd_img_feats')},
                       'max_abs_err': worst['hip_rms_err_over_rms'], 'max_over_tol': (worst['rms_ratio'] if case['smooth'] else median) / 1.5})
    try:
        import json
        from conftest import ROOT
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        json.dump(RECORD, open(os.path.join(ROOT, 'gpurun_out', 'bwd_arbiter.json'), 'w'), indent=1)
    except OSError:
        pass
    assert len(rows) >= 100
    if case['smooth']:
        far = [(k, r) for k, r in rows.items() if r['rms_ratio'] > 16 or r['p99_ratio'] > 16]
        assert not far, (tag, far[:8])
        assert len(bad) <= len(rows) // 4, (tag, len(bad), bad[:8])
        assert median <= 1.2, (tag, median)
        for k in ('d_ray_feats', 'd_img_feats'):
            assert rows[k]['rms_ratio'] <= 1.5 and rows[k]['p99_ratio'] <= 2.0, (tag, k, rows[k])
    else:
        assert median <= 1.5, (tag, median)
        for k in ('d_ray_feats', 'd_img_feats'):
            assert rows[k]['p99_ratio'] <= 2.0, (tag, k, rows[k])
