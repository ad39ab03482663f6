"""Training step: gradients of the differentiable path + backbones + grasp head + losses against the
reference's own backward (tests/golden/golden_train_step.npz, tools/make_goldens.py --train-step-only), the flat-buffer
gradient all-reduce over gloo, one optimiser step."""
import os

import numpy as np
import pytest
import torch
import yaml

from graspnerf_amd.synth import make_scene, synth_state_dict, synth_loss_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = yaml.safe_load("""
network: grasp_nerf
init_net_type: cost_volume
agg_net_type: neus
use_hierarchical_sampling: true
use_depth_loss: true
dist_decoder_cfg: {use_vis: false}
fine_dist_decoder_cfg: {use_vis: false}
ray_batch_num: 40
sample_volume: true
render_rgb: true
volume_type: [sdf]
volume_resolution: 16
depth_sample_num: 16
fine_depth_sample_num: 16
agg_net_cfg: {sample_num: 16, init_s: 0.3, fix_s: 0}
fine_agg_net_cfg: {sample_num: 16, init_s: 0.3, fix_s: 0}
render_depth: true
""")


FIXTURES = {'a': ('golden_train_step.npz', dict(scene_id=0, weight_seed=7, torch_seed=321, loss_seed=5)),
            'b': ('golden_train_step_b.npz', dict(scene_id=3, weight_seed=11, torch_seed=77, loss_seed=9)),
            # dist_decoder_cfg.use_vis: true on both levels (the fourth decoder branch, dist_decoder.py:89-97,103-104,133-134, under training)
            'vis': ('golden_train_step_vis.npz', dict(scene_id=1, weight_seed=13, torch_seed=99, loss_seed=7, use_vis=True)),
            # fine_depth_use_all: true (renderer.py:145-146): the fine pass renders the 16 coarse and the 16 resampled depths of a ray together
            'all': ('golden_train_step_all.npz', dict(scene_id=2, weight_seed=17, torch_seed=55, loss_seed=3, use_all=True))}   # tools/make_goldens.py run_train_step


def build(device='cpu', reference_statement=None, weight_seed=7, use_vis=False, use_all=False, samples=16):
    """The model mirror with synthetic parameters.  On the CPU (or with reference_statement=True) its training forward
    runs the differentiable PyTorch statement of the path (tests/reference_autograd.py: test infrastructure -- the product
    trains through its HIP twin pairs only and raises without a GPU)."""
    from graspnerf_amd.renderer import GraspNeRF
    from reference_autograd import use_reference_statement
    cfg = dict(CFG, dist_decoder_cfg={'use_vis': bool(use_vis)}, fine_dist_decoder_cfg={'use_vis': bool(use_vis)})
    if samples != 16:                                  # samples per ray of the coarse pass = resampled depths of the fine pass
        cfg.update(depth_sample_num=samples, fine_depth_sample_num=samples, agg_net_cfg=dict(CFG['agg_net_cfg'], sample_num=samples),
                   fine_agg_net_cfg=dict(CFG['fine_agg_net_cfg'], sample_num=samples))
    if use_all:
        cfg.update(fine_depth_use_all=True, fine_agg_net_cfg=dict(CFG['fine_agg_net_cfg'], sample_num=2 * samples))
    net = GraspNeRF(cfg)
    syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=weight_seed)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()}, strict=True)
    if reference_statement if reference_statement is not None else (str(device) == 'cpu'):
        use_reference_statement(net)
    return net.to(device)


def scene_data(device='cpu', scene_id=0, loss_seed=5):
    ref, que = make_scene(scene_id, 'cfg1')
    _, gt = synth_loss_case(seed=loss_seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    ref_info = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
    ref_info.update(true_depth=t(gt['true_depth']), sdf_gt=t(gt['sdf_gt']))
    que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
                'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    return {'step': 0, 'ref_imgs_info': ref_info, 'que_imgs_info': que_info, 'src_imgs_info': dict(ref_info),
            'grasp_info': tuple(t(x) for x in gt['grasp_info'])}


def check_against_golden(net, terms, G, rtol_loss, rtol_grad, rtol_backbone=None):
    for k in ('loss_rgb_nr', 'loss_rgb_nr_fine', 'loss_depth', 'loss_depth_fine', 'loss_sdf', 'loss_eikonal', 'loss_vgn'):
        np.testing.assert_allclose(float(terms[k].detach().mean()), G['loss.' + k].mean(), rtol=rtol_loss, err_msg=k)
    norms = dict(zip(G['param_names'].tolist(), G['grad_norms'].tolist()))
    worst, worst_path = (0.0, ''), (0.0, '')
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        n = float(p.grad.double().norm())
        # biases in front of an InstanceNorm / a softmax over views have an exactly-zero gradient: both sides hold
        # rounding noise of ~1e-8 there, hence the absolute floor
        e = (abs(n - norms[k]) - 1e-7) / (norms[k] + 1e-12)
        worst = max(worst, (e, k))
        if any(s in k for s in ('dist_decoder', 'agg_net', 'vgn_net')):
            worst_path = max(worst_path, (e, k))
        if 'grad.' + k in G:                                   # hot-path parameters: full gradient arrays
            g, r = p.grad.cpu().numpy(), G['grad.' + k]
            assert np.abs(g - r).max() <= rtol_grad * max(np.abs(r).max(), 1e-8) + 1e-9, k
    # the volumetric path's and the grasp head's own parameters (this repo's kernels) sit at ~5e-6 of the reference's gradient
    # norms; the 2D backbones run through MIOpen (other convolution algorithms than the CPU reference): rtol_backbone
    assert worst_path[0] < rtol_grad, f'gradient-norm mismatch on the path {worst_path}'
    assert worst[0] < (rtol_backbone or rtol_grad), f'gradient-norm mismatch {worst}'


@pytest.mark.parametrize('fx', ['a', 'b', 'vis', 'all'])
def test_train_step_gradients_match_reference(fx):
    """CPU, bitwise-same RNG draws as the reference: random fine samples, depth-loss pixels.  Three fixtures made by the imported
    reference: four scenes, parameter draws and RNG streams, the third with `use_vis: true`, the fourth with `fine_depth_use_all: true`."""
    from graspnerf_amd.trainer import train_losses
    from graspnerf_amd import losses
    name, su = FIXTURES[fx]
    G = dict(np.load(os.path.join(ROOT, 'tests', 'golden', name)))
    net = build(weight_seed=su['weight_seed'], use_vis=su.get('use_vis', False), use_all=su.get('use_all', False)).train()
    data = scene_data(scene_id=su['scene_id'], loss_seed=su['loss_seed'])
    torch.manual_seed(su['torch_seed'])
    out = net(data)
    assert out['s'].shape == (1, 2) and out['sdf_gradient_error_fine'].shape == (1, 2)      # 64 rays, chunks of 40
    terms = train_losses(out, data)
    total = losses.total_loss(terms)
    np.testing.assert_allclose(float(total.detach()), float(G['total']), rtol=2e-5)
    total.backward()
    check_against_golden(net, terms, G, rtol_loss=5e-5, rtol_grad=2e-3)
    assert net.nr_net.agg_net.deviation_network.variance.grad is not None                   # trainable from step 1 (fix_s 0)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from graspnerf_amd.trainer import Trainer
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 1))
    for p in net.parameters():
        torch.nn.init.constant_(p, 0.1)
    net[1].bias.requires_grad_(rank == 0)                      # a parameter without a gradient on one rank
    tr = Trainer(net)
    xs = [torch.full((2, 3), float(rank + 1)), torch.full((2, 3), float(rank + 3))][:rank + 1]   # 1 scene on rank 0, 2 on rank 1
    for x in xs:
        net(x).sum().backward()
    tr._allreduce_grads(len(xs))
    q.put((rank, [p.grad.numpy().copy() for p in net.parameters()]))       # plain arrays: no shared-memory handles
    dist.destroy_process_group()


def test_flat_gradient_allreduce_gloo():
    """world_size 2 over gloo: summed per-scene gradients divided by the GLOBAL scene count (3), missing gradients
    count as zeros, every rank ends with the same buffer."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:                                # a free port chosen by the OS
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    # expectation on one process: 3 "scenes" with inputs 1, 2(rank1's first), 4
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 1))
    for p in net.parameters():
        torch.nn.init.constant_(p, 0.1)
    for v, with_bias in ((1.0, True), (2.0, False), (4.0, False)):
        net[1].bias.requires_grad_(with_bias)
        net(torch.full((2, 3), v)).sum().backward()
    for got, p in zip(res[0], net.parameters()):
        np.testing.assert_allclose(got, (p.grad / 3.0).numpy(), rtol=1e-6)


def _step_worker(rank, world, port, q):
    import torch.distributed as dist
    from graspnerf_amd.trainer import Trainer
    torch.set_num_threads(2 if world < 8 else 1)
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    net = build()
    tr = Trainer(net, {'lr_init': 1e-3})
    if world == 8:                                                         # one node of 8 GPUs: contiguous shards of the global batch
        from graspnerf_amd.sharding import scene_shard
        shards = list(range(*scene_shard(8, rank, 8)))
    else:
        shards = {2: [[0], [1, 2]], 3: [[0], [1, 2], []]}[world][rank]    # ragged / empty shards
    torch.manual_seed(100 + rank)
    log = tr.step([scene_data(scene_id=i) for i in shards])
    sd = {k: v.detach().numpy().copy() for k, v in net.state_dict().items() if 'dist_decoder' in k or 'vgn_net.conv_qual' in k or 'agg_net.prob_embed' in k}
    q.put((rank, sd, log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_trainer_step_end_to_end_gloo(world):
    """Trainer.step over gloo with ragged shards (1 + 2 scenes; world 3 adds a rank with an EMPTY shard, as scene_shard
    produces when the global batch is smaller than the world): every rank takes part in the flat-gradient all-reduce and
    ends the step with identical parameters, which moved; the empty rank returns a log without loss terms instead of failing.
    world 8 = the rank layout of one MI355X node (BASELINE.json configs[4] at 8 GPUs), one tiny scene per rank."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ps = [ctx.Process(target=_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = {}
    for _ in ps:
        r, sd, log = q.get(timeout=600)
        res[r] = (sd, log)
    for p in ps:
        p.join(120)
    ref0 = build().state_dict()
    moved = 0
    for k, v in res[0][0].items():
        for r in range(1, world):
            assert np.array_equal(v, res[r][0][k]), f'{k}: ranks 0 and {r} diverged'
        moved += int(not np.array_equal(v, ref0[k].numpy()))
    assert moved > len(res[0][0]) // 2, 'the optimiser step should have moved the parameters'
    assert all(np.isfinite(v) for k, v in res[0][1].items() if k.startswith('loss'))
    if world == 3:
        assert [k for k in res[2][1] if k.startswith('loss')] == [] and res[2][1]['lr'] == 1e-3


def test_optimizer_step_changes_parameters_and_reduces_loss():
    from graspnerf_amd.trainer import Trainer, exp_decay_lr
    assert exp_decay_lr(0) == 1e-4 and exp_decay_lr(100000) == 5e-5 and exp_decay_lr(10 ** 7) == 1e-5
    net = build()
    tr = Trainer(net, {'lr_init': 1e-3})
    data = scene_data()
    torch.manual_seed(1)
    l0 = tr.step([data])
    torch.manual_seed(1)
    l1 = tr.step([data])
    tot = lambda l: sum(v for k, v in l.items() if k.startswith('loss'))
    assert np.isfinite(tot(l0)) and tot(l1) < tot(l0)
    assert tr.step_id == 2 and l0['lr'] == 1e-3


def test_loss_terms_leave_the_device_on_logging_steps_only():
    """Trainer(log_every=N), the reference's train_log_step (trainer.py:31,159): step() returns the terms as floats every N-th step and
    only `lr` in between (no device-to-host copy: the host is free to queue the next step); last_log() reads the latest terms on demand,
    and the parameter updates do not depend on the cadence."""
    from graspnerf_amd.trainer import Trainer
    data = scene_data()
    runs = {}
    for every in (1, 3):
        net = build()
        tr = Trainer(net, {'lr_init': 1e-3}, log_every=every)
        logs = []
        for i in range(3):
            torch.manual_seed(10 + i)
            logs.append(tr.step([data]))
        runs[every] = (logs, tr.last_log(), [p.detach().clone() for p in net.parameters()])
    l1, l3 = runs[1][0], runs[3][0]
    assert all(any(k.startswith('loss') for k in l) for l in l1)
    assert set(l3[0]) == {'lr'} and set(l3[1]) == {'lr'} and set(l3[2]) == set(l1[2])
    assert l3[2] == l1[2] == runs[3][1] == runs[1][1]
    assert all(torch.equal(a, b) for a, b in zip(runs[1][2], runs[3][2]))


def test_trainer_with_several_scenes_on_cpu_uses_the_per_scene_loop():
    """Off the GPU forward_scenes() declines (None) and the trainer runs forward + backward scene by scene."""
    from graspnerf_amd.trainer import Trainer
    net = build()
    assert net.forward_scenes([scene_data(scene_id=0), scene_data(scene_id=1)]) is None
    tr = Trainer(net, {'lr_init': 1e-3})
    torch.manual_seed(3)
    log = tr.step([scene_data(scene_id=0), scene_data(scene_id=1)])
    assert np.isfinite(sum(v for k, v in log.items() if k.startswith('loss'))) and tr.step_id == 1
    assert all(p.grad is not None for p in net.parameters())


def test_product_refuses_to_train_without_a_gpu():
    """No PyTorch / CPU fallback in the product: a training forward on host tensors raises (inference raises in HotPath)."""
    from graspnerf_amd import _lib
    net = build(reference_statement=False).train()
    with pytest.raises(_lib.GnrError):
        net(scene_data())


@pytest.mark.gpu
@pytest.mark.parametrize('fx', ['a', 'b', 'vis', 'all'])
def test_train_step_on_gpu_matches_reference_gradients(fx):
    """Same check with the model on the MI355X: the volumetric path in HIP in both directions (renderer.py autograd.Functions over
    csrc/gnr_bwd.inc), backbones / grasp head / losses under PyTorch autograd."""
    from graspnerf_amd.trainer import train_losses
    from graspnerf_amd import losses
    name, su = FIXTURES[fx]
    G = dict(np.load(os.path.join(ROOT, 'tests', 'golden', name)))
    net = build('cuda', weight_seed=su['weight_seed'], use_vis=su.get('use_vis', False), use_all=su.get('use_all', False)).train()
    data = scene_data('cuda', scene_id=su['scene_id'], loss_seed=su['loss_seed'])
    torch.manual_seed(su['torch_seed'])
    terms = train_losses(net(data), data)
    losses.total_loss(terms).backward()
    torch.cuda.synchronize()
    check_against_golden(net, terms, G, rtol_loss=2e-4, rtol_grad=1e-3, rtol_backbone=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize('variant', ['default', 'fine_depth_use_all 40+40', 'use_vis + fine_depth_use_all 64+64'])
def test_hip_and_autograd_training_paths_agree(variant, monkeypatch):
    """The same train-mode forward + backward through the differentiable PyTorch statement (tests/reference_autograd.py, pure
    autograd incl. the double backward) and through the product's HIP twin pairs: outputs and every parameter gradient.
    The fine_depth_use_all variants put 80 and 128 samples per ray (the reference's own default sizes, renderer.py:22-24) through
    the fine pass's backward twins -- past one wavefront per ray; the statement itself is pinned by the reference's golden train
    steps (fixtures above, 16 + 16 samples)."""
    from graspnerf_amd.trainer import train_losses
    from graspnerf_amd import losses
    from reference_autograd import use_reference_statement
    kw = {'default': {}, 'fine_depth_use_all 40+40': dict(use_all=True, samples=40),
          'use_vis + fine_depth_use_all 64+64': dict(use_all=True, use_vis=True, samples=64)}[variant]
    net = build('cuda', **kw).train()
    data = scene_data('cuda')
    # With 40 / 64 resampled depths on each of 64 rays a handful of the inverse-CDF draws sit within rounding of a cdf edge, where
    # the device resampler and torch.searchsorted pick neighbouring bins (a different sample on that ray; the indices' parity is
    # the business of tests/test_gpu_parity.py).  For the large variants the statement is therefore run on the fine depths the
    # HIP pass drew (teacher forcing): what is compared is the arithmetic of the passes, sample for sample.
    forced, rec = variant != 'default', []
    if forced:
        import reference_autograd
        from graspnerf_amd.renderer import NeuralRayRenderer
        orig_pass = NeuralRayRenderer._train_pass

        def spy(self, hot, prep, q, depth, level, *a, **k):
            out = orig_pass(self, hot, prep, q, depth, level, *a, **k)
            if level == 'coarse':
                rec.append(out[1]['fine_depth'][0].detach().clone())
            return out
        monkeypatch.setattr(NeuralRayRenderer, '_train_pass', spy)
    res = {}
    for hip in ((True, False) if forced else (False, True)):
        use_reference_statement(net, on=not hip)
        if forced and not hip:
            assert len(rec) == 2                                          # 64 rays in chunks of 40
            monkeypatch.setattr(reference_autograd, 'sample_fine_depth', lambda *a, **k: rec.pop(0))
        for a in (net.nr_net.agg_net, net.nr_net.fine_agg_net):
            a.step = 0
        net.zero_grad(set_to_none=True)
        torch.manual_seed(11)
        out = net(data)
        losses.total_loss(train_losses(out, data)).backward()
        torch.cuda.synchronize()
        res[hip] = ({k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v) and v.dtype.is_floating_point},
                    {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    assert set(res[False][0]) == set(res[True][0])
    for k, v in res[False][0].items():
        assert res[True][0][k].shape == v.shape, k
        assert (res[True][0][k] - v).abs().max() <= 2e-4 + 1e-3 * v.abs().max(), k
    for k, g in res[False][1].items():
        if any(s in k for s in ('dist_decoder', 'agg_net', 'vgn_net')):
            d = (res[True][1][k] - g).abs().max().item()
            # (absolute floor: the NeuS variance's gradient is one scalar summed over every sample of the pass with heavy
            # cancellation -- 3e-6 at 128 samples per ray, its two evaluations 1.6e-7 apart)
            assert d <= 3e-3 * g.abs().max().item() + 5e-7, (k, d, g.abs().max().item())
        else:
            # the 2D backbones run through MIOpen, whose kernels are not run-to-run deterministic (~1e-5 on the feature maps, which
            # the ill-conditioned fine samples amplify): the two passes do not even see identical backbone arithmetic
            d = (res[True][1][k] - g).norm().item()
            assert d <= 2e-2 * g.norm().item() + 1e-6, (k, d, g.norm().item())


@pytest.mark.gpu
def test_batched_training_forward_equals_per_scene_loop():
    """Trainer: all scenes of a rank in one forward / backward (forward_scenes: per-scene dicts + per-scene losses, and the
    scene-major stacks + stacked losses Trainer.step runs) against the reference-style loop of per-scene forwards -- same
    RNG stream, same losses, same accumulated gradients."""
    from graspnerf_amd.trainer import train_losses, train_losses_stacked
    from graspnerf_amd import losses
    net = build('cuda').train()
    datas = [scene_data('cuda', 0), scene_data('cuda', 1), scene_data('cuda', 2)]
    assert net.forward_scenes(datas) is None                   # 64 rays > ray_batch_num 40: several chunks, not batchable
    net.nr_net.cfg['ray_batch_num'] = 4096
    res = {}
    for batched in (False, True, 'stacked'):
        net.zero_grad(set_to_none=True)
        for a in (net.nr_net.agg_net, net.nr_net.fine_agg_net):
            a.step = 0
        torch.manual_seed(21)
        if batched == 'stacked':
            st = net.forward_scenes(datas, stacked=True)
            assert isinstance(st, dict) and st['pixel_colors_nr'].shape == (3, 64, 3) and st['vgn_pred'][1].shape[:1] == (3,)
            tv = train_losses_stacked(st, datas)
            losses.total_loss(tv, scenes=3).backward()
            terms = [{k: v[b] for k, v in tv.items()} for b in range(3)]
        elif batched:
            outs = net.forward_scenes(datas)
            assert outs is not None
            terms = [train_losses(o, d) for o, d in zip(outs, datas)]
            sum(losses.total_loss(t) for t in terms).backward()
        else:
            terms = []
            for d in datas:
                t = train_losses(net(d), d)
                losses.total_loss(t).backward()
                terms.append(t)
        torch.cuda.synchronize()
        res[batched] = ([{k: float(v.detach().mean()) for k, v in t.items() if k.startswith('loss')} for t in terms],
                        {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    for mode in (True, 'stacked'):
        for la, lb in zip(res[False][0], res[mode][0]):
            for k in la:
                assert abs(la[k] - lb[k]) <= 1e-4 * abs(la[k]) + 1e-7, (mode, k, la[k], lb[k])
        for k, g in res[False][1].items():
            if any(s in k for s in ('dist_decoder', 'agg_net', 'vgn_net')):
                d = (res[mode][1][k] - g).abs().max().item()
                assert d <= 3e-3 * g.abs().max().item() + 1e-6, (mode, k, d, g.abs().max().item())
            else:
                # the 2D backbones run through MIOpen, which picks other (Winograd) algorithms for the batched call
                d = (res[mode][1][k] - g).norm().item()
                assert d <= 3e-2 * g.norm().item() + 1e-6, (mode, k, d, g.norm().item())
    uneven = [dict(datas[0]), dict(datas[1], grasp_info=tuple(x[:-1] for x in datas[1]['grasp_info']))]
    assert net.forward_scenes(uneven, stacked=True) is None, 'different grasp counts: declined before anything runs'


@pytest.mark.gpu
def test_backward_after_another_forward_is_refused():
    """The HIP twin pairs keep their states in per-module workspaces: a second training forward before the first
    backward must be reported, not silently produce wrong gradients."""
    from graspnerf_amd.trainer import train_losses
    from graspnerf_amd import losses, _lib
    net = build('cuda').train()
    d0, d1 = scene_data('cuda', 0), scene_data('cuda', 1)
    l0 = losses.total_loss(train_losses(net(d0), d0))
    l1 = losses.total_loss(train_losses(net(d1), d1))
    with pytest.raises((_lib.GnrError, RuntimeError)):
        l0.backward()
    net.zero_grad(set_to_none=True)



@pytest.mark.gpu
def test_eval_after_optimizer_step_uses_the_updated_weights():
    """The reference trainer validates under eval()+no_grad between optimiser steps (train_valid.py:26): the packed copies of
    the hot-path weights and of the grasp head must follow optimizer.step() (in-place updates).  An eval forward after a
    training step equals the eval forward of a fresh model loaded from the trained state dict, bit for bit."""
    from graspnerf_amd.trainer import Trainer
    from graspnerf_amd.renderer import GraspNeRF
    net = build('cuda')
    data = scene_data('cuda')
    ev = dict(data, eval=True, full_vol=True)
    with torch.no_grad():
        net.eval()
        torch.manual_seed(5)
        before = net(ev)                                          # packs the hot path and the grasp head from the initial weights
    tr = Trainer(net, {'lr_init': 1e-2})
    torch.manual_seed(6)
    tr.step([data])
    with torch.no_grad():
        net.eval()
        torch.manual_seed(5)
        after = net(ev)
    fresh = GraspNeRF(CFG)
    fresh.load_state_dict(net.state_dict(), strict=True)
    fresh = fresh.cuda().eval()
    with torch.no_grad():
        torch.manual_seed(5)
        want = fresh(ev)
    torch.cuda.synchronize()
    # the packed copies are those of the trained parameters, bit for bit ...
    assert torch.equal(net.nr_net.hot().wc, fresh.nr_net.hot().wc) and torch.equal(net.nr_net.hot().wf, fresh.nr_net.hot().wf)
    assert torch.equal(net._head.w, fresh._head.w), 'grasp head packed from stale weights'
    # ... and the outputs follow (MIOpen's 2D backbones are not run-to-run deterministic, ~1e-5 on the feature maps, hence a
    # tolerance; the training step itself moved the outputs by orders of magnitude more)
    moved = (before['volume'] - after['volume']).abs().max().item()
    assert moved > 1e-3, 'the step should have moved the volume'
    for k in ('volume', 'sdf_values', 'alpha_values_fine', 'depth_mean', 'depth_mean_fine'):
        assert (after[k] - want[k]).abs().max().item() < 5e-5, k
    for a, b, c in zip(after['vgn_pred'], want['vgn_pred'], before['vgn_pred']):
        assert (a - b).abs().max().item() < 5e-5 < (a - c).abs().max().item(), 'grasp head ran on stale weights'


def _full_size_cfg():
    import copy
    cfg = copy.deepcopy(CFG)
    cfg.update(volume_resolution=40, depth_sample_num=40, fine_depth_sample_num=40, ray_batch_num=4096)
    cfg['agg_net_cfg']['sample_num'] = cfg['fine_agg_net_cfg']['sample_num'] = 40
    return cfg


def _full_size_scene(i, device='cuda'):
    ref, que = make_scene(i, 'cfg2')
    _, gt = synth_loss_case(seed=100 + i, rfn=6, h=288, w=512, rn=512, R=40)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    ri = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
    ri.update(true_depth=t(gt['true_depth']), sdf_gt=t(gt['sdf_gt']))
    qi = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
          'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    return {'step': 0, 'ref_imgs_info': ri, 'que_imgs_info': qi, 'src_imgs_info': dict(ri), 'grasp_info': tuple(t(x) for x in gt['grasp_info'])}


@pytest.mark.gpu
def test_full_size_train_step_matches_the_reference_statement():
    """BASELINE.json configs[4] at its real size (6 views 288x512, 40^3 volume, 512 rays x (40+40) samples): one scene through
    the product's HIP twin pairs against the differentiable PyTorch statement (pure autograd incl. the double backward) -- every
    loss term and the gradient of every parameter of the volumetric path and the grasp head; then two scenes batched
    (forward_scenes stacked + stacked losses, what the trainer and bench.py's train_step run) against the per-scene loop."""
    from graspnerf_amd.renderer import GraspNeRF
    from graspnerf_amd.trainer import train_losses, train_losses_stacked
    from graspnerf_amd import losses
    from reference_autograd import use_reference_statement
    cfg = _full_size_cfg()
    net = GraspNeRF(cfg)
    syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()}, strict=True)
    net = net.cuda().train()
    data = _full_size_scene(0)
    res = {}
    for hip in (False, True):
        use_reference_statement(net, on=not hip)
        for a in (net.nr_net.agg_net, net.nr_net.fine_agg_net):
            a.step = 0
        net.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        terms = train_losses(net(data), data)
        losses.total_loss(terms).backward()
        torch.cuda.synchronize()
        res[hip] = ({k: float(v.detach().mean()) for k, v in terms.items() if k.startswith('loss')},
                    {k: p.grad.detach().clone() for k, p in net.named_parameters()})
        del terms
        torch.cuda.empty_cache()
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) <= 2e-3 * abs(v) + 1e-7, (k, res[True][0][k], v)
    worst = (0.0, '')
    for k, g in res[False][1].items():
        if any(s in k for s in ('dist_decoder', 'agg_net', 'vgn_net')):
            if k.endswith('rgb_fc.4.bias'):               # bias in front of the softmax over views: the gradient is exactly zero,
                assert float(g.abs().max()) < 1e-4 and float(res[True][1][k].abs().max()) < 1e-4     # both sides hold rounding noise
                continue
            e = (res[True][1][k] - g).norm().item() / (g.norm().item() + 1e-9)
            worst = max(worst, (e, k))
    assert worst[0] < 5e-3, f'gradient mismatch at full size {worst}'
    # two scenes in one batched forward / backward against the per-scene loop
    datas = [data, _full_size_scene(1)]
    got = {}
    for batched in (False, True):
        for a in (net.nr_net.agg_net, net.nr_net.fine_agg_net):
            a.step = 0
        net.zero_grad(set_to_none=True)
        torch.manual_seed(9)
        if batched:
            st = net.forward_scenes(datas, stacked=True)
            assert st is not None
            losses.total_loss(train_losses_stacked(st, datas), scenes=2).backward()
        else:
            for d in datas:
                losses.total_loss(train_losses(net(d), d)).backward()
        torch.cuda.synchronize()
        got[batched] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if any(s in k for s in ('dist_decoder', 'agg_net', 'vgn_net'))}
    for k, g in got[False].items():
        if k.endswith('rgb_fc.4.bias'):
            continue
        e = (got[True][k] - g).norm().item() / (g.norm().item() + 1e-9)
        assert e < 3e-3, (k, e)


@pytest.mark.gpu
def test_benched_shape_eight_scenes_stacked_equals_the_per_scene_loop():
    """BASELINE.json configs[4] as bench.py's train_step runs it: EIGHT full-size scenes in one stacked forward / backward
    (forward_scenes(stacked=True) + the stacked losses) against the per-scene loop over the same eight scenes with the same RNG
    stream -- every loss term per scene and the gradient of every parameter of the volumetric path and the grasp head.  The two
    differ only in the order in which the weight-gradient atomics of the backward kernels and MIOpen's reductions add up.
    (Scene 0 of this batch against the pure-autograd statement: test_full_size_train_step_matches_the_reference_statement.)"""
    from graspnerf_amd.renderer import GraspNeRF
    from graspnerf_amd.trainer import train_losses, train_losses_stacked
    from graspnerf_amd import losses
    from reference_autograd import use_reference_statement
    cfg = _full_size_cfg()
    net = GraspNeRF(cfg)
    syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()}, strict=True)
    net = net.cuda().train()
    use_reference_statement(net, on=False)
    datas = [_full_size_scene(i) for i in range(8)]
    got, terms_got = {}, {}
    for batched in (False, True):
        for a in (net.nr_net.agg_net, net.nr_net.fine_agg_net):
            a.step = 0
        net.zero_grad(set_to_none=True)
        torch.manual_seed(9)
        if batched:
            st = net.forward_scenes(datas, stacked=True)
            assert st is not None
            terms = train_losses_stacked(st, datas)
            losses.total_loss(terms, scenes=8).backward()
            terms_got[True] = {k: v.detach().reshape(8, -1).mean(1).cpu().numpy() for k, v in terms.items() if k.startswith('loss')}
        else:
            per = []
            for d in datas:
                terms = train_losses(net(d), d)
                losses.total_loss(terms).backward()
                per.append({k: float(v.detach().mean()) for k, v in terms.items() if k.startswith('loss')})
            terms_got[False] = {k: np.array([p[k] for p in per]) for k in per[0]}
        torch.cuda.synchronize()
        got[batched] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if any(s in k for s in ('dist_decoder', 'agg_net', 'vgn_net'))}
        torch.cuda.empty_cache()
    for k, v in terms_got[False].items():
        np.testing.assert_allclose(terms_got[True][k], v, rtol=2e-4, atol=1e-7, err_msg=k)
    worst = (0.0, '')
    for k, g in got[False].items():
        if k.endswith('rgb_fc.4.bias'):
            continue
        worst = max(worst, ((got[True][k] - g).norm().item() / (g.norm().item() + 1e-9), k))
    assert worst[0] < 3e-3, f'stacked batch of 8 vs per-scene loop: {worst}'


@pytest.mark.gpu
def test_packed_weights_follow_writes_that_bypass_the_version_counters(capsys):
    """renderer.hot() keys its packed copies on (version counter, storage address) of the hot-path parameters.  `p.data = t`
    moves the storage and is seen; `p.data.mul_()` changes neither and needs invalidate_packed().  Also the reference's
    low-valid-ratio diagnostic (renderer.py:174-176), opt-in through cfg warn_low_valid_ratio."""
    net = build('cuda').eval()
    data = scene_data('cuda')
    nr = net.nr_net
    with torch.no_grad():
        ref = nr.image_encoder(data['ref_imgs_info']['imgs'])
        info = dict(data['ref_imgs_info'])
        info['img_feats'] = ref
        info['ray_feats'] = nr.vis_encoder(nr.init_net({'imgs': info['imgs']}, None, False), ref)
        v0 = nr.sample_volume(info).clone()
        p = dict(nr.named_parameters())['agg_net.agg_impl.geometry_fc.2.weight']
        p.data = p.data * 1.5                                     # new storage: picked up by the key
        v1 = nr.sample_volume(info).clone()
        assert (v1 - v0).abs().max() > 1e-4
        p.data.mul_(1.0 / 1.5)                                    # in place through .data: invisible to the key ...
        v2 = nr.sample_volume(info).clone()
        assert torch.equal(v2, v1)
        nr.invalidate_packed()                                    # ... until told
        v3 = nr.sample_volume(info).clone()
        assert (v3 - v0).abs().max() < 1e-5
        # diagnostic: cameras that see (almost) none of the workspace
        nr.cfg['warn_low_valid_ratio'] = True
        capsys.readouterr()
        assert torch.equal(nr.sample_volume(info), v3) and 'too low ratio' not in capsys.readouterr().out
        far = dict(info, bbox3d=info['bbox3d'] + 5.0)
        nr.sample_volume(far)
        assert '!! too low ratio' in capsys.readouterr().out
        nr.cfg['warn_low_valid_ratio'] = False
