"""Benchmark of the MI355X-native GraspNeRF volumetric hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one forward pass of the hot path over a batch of B synthetic scenes per GPU with all inputs already resident in
HBM: feature-map repack (gnr_prepare) + TSDF volume (sample_volume, 40^3) + ray rendering (512 rays, 40 coarse + 40 fine
samples, ground-truth pixel colours), 6 views of 288x512 (BASELINE.json configs[2]; configs[3] = the same on 8 GPUs).
Scenes are independent, so N GPUs shard scenes with no data-path collective (weak scaling: B scenes per GPU).

Before any timing is accepted rank 0 runs a PARITY GATE: scene 0 of its batch is the scene tests/golden/golden_cfg2.npz pins
(outputs of the imported reference), and the batched launch must reproduce it.

After the forward leg the same process measures, outside the headline's timed region and reported as sub-records of the ONE
JSON line rank 0 prints:  `train_step` = BASELINE.json configs[4] (8 full-size scenes per GPU: backbones + volumetric path +
grasp head + losses, backward, one flat gradient all-reduce over RCCL, Adam);  `with_backbones` = images -> volume + render +
grasps (N = 1 only);  `cpu_baseline` = the oracle on the host cores (N = 1 only).  See DESIGN.md §6.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from graspnerf_amd import weights, _lib                   # noqa: E402
from graspnerf_amd.synth import make_scene, CONFIGS       # noqa: E402
from graspnerf_amd.sharding import scene_shard, max_over_ranks   # noqa: E402

METRIC = 'scenes/sec TSDF+render fwd, 6-view 40^3 grid'
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak = the fp32 vector rate
PEAK_F16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense BF16/FP16 MFMA peak (v_mfma_f32_16x16x32_f16)
# MFMAs k_chain<6,*> issues per 16-point tile (SQ_INSTS_MFMA / tiles in profiles/r02_f_pmc_counters.json agrees):
# fp16-pair layers = 3 partial products per K32 block (per view 33 blocks, 34 with rgb_fc.0; per point 34: HOIST 5x4, GEO1 3x4,
# GEO2 2), the 1..4-k-step remainders of the per-view layers stay v_mfma_f32_16x16x4_f32 (19 per view, 25 with rgb_fc)
MFMA_PER_TILE = {False: (6 * 99 + 102, 6 * 19), True: (6 * 102 + 102, 6 * 25)}      # render? -> (16x16x32 f16, 16x16x4 f32)


def executed_mfma_flops(points, render):
    f16, f32 = MFMA_PER_TILE[render]
    return (points / 16.0) * (f16 * 2.0 * 16 * 16 * 32 + f32 * 2.0 * 16 * 16 * 4)
PEAK_HBM_TBPS = 8.0
# algorithmic (un-hoisted) MACs per (view, point) and per point executed by k_chain
MAC_VIEW_VOL, MAC_VIEW_RAY, MAC_POINT_CHAIN = 27736, 28464, 6528
# backward of the first view loop (k_view1_bwd): dX and dW of the mixture decoder (6 304 MAC forward, dist_decoder.py:64-88),
# prob_embed (2 112), ray_dir_fc (624) and the neuray gate (264) = 2 x 9 304 MAC per (view, point); the kernel's
# recomputation of that forward from the saved states is its own choice and not counted
MAC_VIEW1_BWD = 2 * (6304 + 2112 + 624 + 264)


def chain_flops(points, views, render):
    return 2.0 * points * (views * (MAC_VIEW_RAY if render else MAC_VIEW_VOL) + MAC_POINT_CHAIN)


def recorded_bwd_traffic(scenes):
    """HBM-side bytes per launch of k_view1_bwd (volume points, 8 scenes) from the same recorded PMC passes, or None."""
    try:
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_counters.json')))[-1]
        k = json.load(open(newest))['kernels']['k_view1_bwd']
        return (int(k['hbm_bytes_corrected']), os.path.relpath(newest, ROOT)) if scenes == 8 else (None, None)
    except (OSError, KeyError, ValueError, IndexError):
        return None, None


def recorded_pmc(batch):
    """Counters of the dominant kernel from the committed rocprofv3 PMC passes (newest profiles/r*_pmc_counters.json, made by
    tools/collect_profiles.sh: separate --pmc runs, gfx950 2x read correction applied to FETCH_SIZE).  PMC counters cannot be
    read from inside this process: the figures are the recorded ones of that build, labelled with their source, and only
    reported for the batch size they were measured at."""
    try:
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_counters.json')))[-1]
        k = next(v for n, v in json.load(open(newest))['kernels'].items() if n.startswith('k_chain<6, false'))
        if batch != 32:
            return None, None
        return int(k['hbm_bytes_corrected']), {
            'source': os.path.relpath(newest, ROOT),
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
            'mfma_pipe_busy': round(k['SQ_VALU_MFMA_BUSY_CYCLES'] / (k['GRBM_GUI_ACTIVE'] / 8 * 1024), 3),
            'l2_hit_rate': round(k['TCC_HIT_sum'] / (k['TCC_HIT_sum'] + k['TCC_MISS_sum']), 3),
            # SQ_ACTIVE_INST_VALU counts quad-cycles over all wavefronts: x4 / (cycles x SIMDs) = share of the SIMDs' time a
            # VALU (non-matrix) instruction is executing
            'valu_busy': round(4 * k['SQ_ACTIVE_INST_VALU'] / (k['GRBM_GUI_ACTIVE'] / 8 * 1024), 3) if 'SQ_ACTIVE_INST_VALU' in k else None,
            'mfma_per_launch': k['SQ_INSTS_MFMA'], 'valu_incl_mfma_per_launch': k['SQ_INSTS_VALU']}
    except (OSError, KeyError, ValueError, IndexError, ZeroDivisionError, StopIteration):
        return None, None


# ---- parity gate -----------------------------------------------------------------------------------------------------
RTOL, ATOL = 1e-3, {'volume': 2e-5, 'sdf_values': 2e-5, 'sdf_gradient_error': 2e-5, 'colors_nr': 3e-4}     # tests/test_gpu_parity.py
ATOL_DEFAULT = 6e-5


def parity_gate(hp, bref, bque, vol, co, fi_free, inds):
    """Scene 0 of the batched launch against the reference's outputs (tests/golden/golden_cfg2.npz, made by importing the
    reference: tools/make_goldens.py).  Index-valued outputs bit-exact (in-image view masks, ray masks; resampling indices
    equal except where the fixture records a cdf edge within 3e-5 of the sample), values within 1e-3 relative (+ the
    absolute floors of tests/test_gpu_parity.py).  The fine level depends on the resampled depths, so it is compared in a
    second single-scene launch teacher-forced on the reference's fine depths.  -> dict of measured errors; raises on mismatch."""
    G = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_cfg2.npz')))
    errs = {}

    def close(a, b, what, key):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64).reshape(np.shape(a))
        if not np.isfinite(a).all():
            raise SystemExit(f'parity gate: {what} is not finite')
        e = np.abs(a - b)
        tol = ATOL.get(key, ATOL_DEFAULT) + RTOL * np.abs(b)
        errs[what] = float(e.max())
        if (e > tol).any():
            raise SystemExit(f'parity gate FAILED: {what}: max |d| {e.max():.3e} exceeds {ATOL.get(key, ATOL_DEFAULT):.0e} + 1e-3 |ref|')
    close(vol[0].cpu().numpy(), G['volume'][0], 'volume', 'volume')
    keys = ['sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr', 'pixel_colors_gt', 'render_depth', 'sdf_gradient_error']
    for k in keys:
        close(co[k][0].cpu().numpy(), G['render.' + k][0], 'coarse ' + k, k)
    if not np.array_equal(co['ray_mask'][0].cpu().numpy(), G['render.ray_mask'][0]):
        raise SystemExit('parity gate FAILED: coarse ray_mask differs from the reference')
    bad = inds[0].cpu().numpy() != G['fine_inds']
    if (bad & (G['fine_inds_margin'] > 3e-5)).any():
        raise SystemExit('parity gate FAILED: resampling indices differ from the reference away from cdf edges')
    errs['fine_inds_differing'] = int(bad.sum())
    r1 = {k: v[:1].contiguous() for k, v in bref.items()}
    q1 = {k: v[:1].contiguous() for k, v in bque.items()}
    v1, vm = hp.sample_volume(r1, 40, want_mask=True)
    gm = np.unpackbits(G['volume_mask_bits']).reshape(6, 1600, 40).astype(bool)
    mine = vm[0].cpu().numpy()
    for v in range(6):
        if not np.array_equal(((mine >> v) & 1).astype(bool).reshape(1600, 40)[:, ::-1], gm[v]):
            raise SystemExit(f'parity gate FAILED: in-image mask of view {v} differs from the reference')
    if not torch.equal(v1[0], vol[0]):
        raise SystemExit('parity gate FAILED: batched volume of scene 0 differs from its single-scene launch')
    c1, f1 = hp.render(r1, q1, fine_depth_in=G['fine_depth_sorted'][None])
    for k in keys:
        close(f1[k][0].cpu().numpy(), G['render.' + k + '_fine'][0], 'fine ' + k, k)
    if not np.array_equal(f1['ray_mask'][0].cpu().numpy(), G['render.ray_mask_fine'][0]):
        raise SystemExit('parity gate FAILED: fine ray_mask differs from the reference')
    if not np.isfinite(fi_free['sdf_values'].cpu().numpy()).all():
        raise SystemExit('parity gate FAILED: free-running fine pass is not finite')
    return errs


# ---- CPU baseline (the oracle; reported at N = 1 only) ---------------------------------------------------------------
def cpu_baseline(weights_np):
    """SURVEY.md §8d protocol: the oracle (torch-CPU fp32 port of the reference path; the only place bench.py touches
    oracle/) on whole scenes of the same workload: 3 warm-ups, median of 10, plus a 1-thread figure."""
    from oracle import graspnerf_oracle as O
    # torch's intra-op pool stops scaling (and collapses from oversubscription) well below the hardware threads of the GPU box
    # on these op sizes; 16 threads is what we actually use and report
    cores = min(os.cpu_count() or 1, 16)
    W = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    ref, que = make_scene(0, 'cfg2')
    inp, q = O.to_torch(ref), O.to_torch(que)

    def one():
        t0 = time.perf_counter()
        O.sample_volume(W, inp, 40)
        O.render(W, inp, q)
        return time.perf_counter() - t0
    torch.set_num_threads(cores)
    for _ in range(3):
        one()
    ts = sorted(one() for _ in range(10))
    med = 0.5 * (ts[4] + ts[5])
    torch.set_num_threads(1)
    t1 = one()
    torch.set_num_threads(cores)
    return {'value': round(1.0 / med, 4), 'unit': 'scenes/s', 'cores': cores, 'kind': 'port',
            'value_1_thread': round(1.0 / t1, 4),
            'sample': f'whole scenes (6 views 288x512, 40^3 volume + 512 rays x (40+40) samples), oracle/graspnerf_oracle.py (torch '
                      f'{torch.__version__} CPU fp32): 3 warm-ups + median of 10 at {cores} threads ({med:.3f} s, min {ts[0]:.3f}, max '
                      f'{ts[-1]:.3f}); 1 thread: one scene ({t1:.2f} s)'}


# ---- BASELINE.json configs[4]: end-to-end train step -----------------------------------------------------------------
TRAIN_CFG = {
    'network': 'grasp_nerf', 'init_net_type': 'cost_volume', 'agg_net_type': 'neus', 'use_hierarchical_sampling': True,
    'use_depth_loss': True, 'dist_decoder_cfg': {'use_vis': False}, 'fine_dist_decoder_cfg': {'use_vis': False}, 'ray_batch_num': 4096,
    'sample_volume': True, 'render_rgb': True, 'volume_type': ['sdf'], 'volume_resolution': 40, 'depth_sample_num': 40,
    'fine_depth_sample_num': 40, 'agg_net_cfg': {'sample_num': 40, 'init_s': 0.3, 'fix_s': 0},
    'fine_agg_net_cfg': {'sample_num': 40, 'init_s': 0.3, 'fix_s': 0}, 'render_depth': True,
}                                                        # = configs/nrvgn_sdf.yaml of the reference, network part


def f32_mfma_build_leg(value):
    """The same step on the companion library (csrc/build.sh: -DGNR_SPLIT16=0, the chain on v_mfma_f32_16x16x4_f32 instead of fp16
    pairs on the f16 cores), in a child process (GNR_LIB selects the library at load time), with its own parity gate: the
    number an fp32-instruction-only implementation of the same kernels reaches on this box in this run."""
    import subprocess
    lib = os.path.join(ROOT, 'graspnerf_amd', 'csrc', 'libgnr_f32mfma.so')
    if not os.path.exists(lib):
        return {'skipped': 'libgnr_f32mfma.so not built'}
    env = dict(os.environ, GNR_LIB=lib)
    for k in [k for k in env if k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK', 'LOCAL_WORLD_SIZE') or k.startswith(('TORCHELASTIC', 'MASTER_'))]:
        env.pop(k)                                          # the child is a plain single-process run, also under torch.distributed.run
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--steps', '60', '--warmup', '10', '--no-train', '--no-backbones',
                            '--no-cpu-baseline', '--no-f32-build'], env=env, capture_output=True, text=True, timeout=300)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                  # noqa: BLE001  (a failed companion run must not lose the product's line)
        return {'skipped': f'{type(e).__name__}: {e}'[:200]}
    return {'library': 'graspnerf_amd/csrc/libgnr_f32mfma.so (-DGNR_SPLIT16=0)', 'value': d['value'], 'unit': d['unit'], 'steps': d['steps'],
            'ms_per_step': d['ms_per_step'], 'parity_checked': d['parity_checked'],
            'k_chain_volume_ms_per_launch': d['roofline']['ms_per_launch'], 'frac_of_fp32_mfma_peak': d['roofline']['frac'],
            'product_speedup': round(value / d['value'], 3)}


def build_model(dev):
    from graspnerf_amd.renderer import GraspNeRF
    from graspnerf_amd.synth import synth_state_dict
    net = GraspNeRF(dict(TRAIN_CFG))
    syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
    return net.to(dev)


def train_scenes(n, first, dev):
    from graspnerf_amd.synth import synth_loss_case
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    out = []
    for i in range(n):
        ref, que = make_scene(first + i, 'cfg2')
        _, gt = synth_loss_case(seed=100 + i, rfn=6, h=288, w=512, rn=512, R=40)
        ri = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
        ri.update(true_depth=t(gt['true_depth']), sdf_gt=t(gt['sdf_gt']))
        qi = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
              'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
        out.append({'ref_imgs_info': ri, 'que_imgs_info': qi, 'src_imgs_info': dict(ri), 'grasp_info': tuple(t(x) for x in gt['grasp_info'])})
    return out


def ev_ms(fn, iters=3):
    """Average milliseconds of fn() on the current stream (torch events; one untimed call first)."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def train_leg(args, world, rank, dev, dist, sync):
    """configs[4]: `--train-scenes` full-size scenes per GPU per step through graspnerf_amd.trainer.Trainer (forward in
    train mode, the configured losses, backward, ONE flat gradient all-reduce over RCCL, Adam)."""
    from graspnerf_amd.trainer import Trainer
    n = args.train_scenes
    net = build_model(dev)
    tr = Trainer(net)
    scenes = train_scenes(n, rank * n, dev)
    torch.cuda.reset_peak_memory_stats(dev)
    for _ in range(args.train_warmup):
        log = tr.step(scenes)
    sync()
    dom = 'k_view1_bwd@gnr_sample_volume_bwd'          # dominant backward kernel of the path: first view loop, volume points
    if rank == 0:
        _lib.timing_begin(only=dom)                    # the timed steps bracket this kernel only
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.train_steps + 1)]
    host = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.train_steps):
        h0 = time.perf_counter()
        log = tr.step(scenes)
        host.append((time.perf_counter() - h0) * 1e3)
        marks[i + 1].record()                              # per-step spans on the stream (no synchronisation added)
    sync()
    dt = max_over_ranks(time.perf_counter() - t0, dev)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.train_steps)]
    table = _lib.timing_end() if rank == 0 else {}
    K = args.train_steps
    # per-kernel table of the library: two further steps with every launch bracketed (outside the timed region: ~60 event
    # pairs per step would perturb it)
    if rank == 0:
        _lib.timing_begin()
    for _ in range(2):
        tr.step(scenes)
    sync()
    full = _lib.timing_end() if rank == 0 else {}
    rec = None
    if rank == 0:
        per_step = {k: round(v[1] / 2, 4) for k, v in sorted(full.items(), key=lambda kv: -kv[1][1])}
        head_ms = sum(v for k, v in per_step.items() if 'gnr_grasp_head' in k or 'conv3d' in k)
        path_ms = sum(per_step.values()) - head_ms
        cnt, tot = table.get(dom, (0, 0.0))
        ms = tot / max(cnt, 1)
        fl = 2.0 * MAC_VIEW1_BWD * n * 64000 * 6
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        rec = {
            'metric': 'train scenes/sec (fwd + losses + bwd + gradient all-reduce + Adam), 6-view 40^3 grid + 512 rays x (40+40)',
            'value': round(world * n * K / dt, 3), 'unit': 'scenes/s', 'ms_per_step': round(dt / K * 1e3, 3), 'steps': K,
            'warmup': args.train_warmup, 'scenes_per_gpu': n, 'global_batch': world * n, 'n_gpus': world, 'dtype': 'f32', 'data': 'synthetic',
            'config': 'BASELINE.json configs[4]: backbones + nr TSDF + render + depth-mean head + grasp head + losses (render, depth, sdf, vgn), '
                      'batch 8/GPU, one flat fp32 gradient all-reduce (4.66 M parameters), Adam',
            'ms_each_step': [round(x, 2) for x in step_ms], 'ms_per_step_median': round(float(np.median(step_ms)), 3),
            'host_ms_each_step': [round(x, 2) for x in host],
            'max_mem_GB': round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 3),
            'loss': {k: round(v, 6) for k, v in log.items() if k.startswith('loss')},
            'split_ms_per_step': {'hip_path_kernels': round(path_ms, 3), 'hip_grasp_head_kernels': round(head_ms, 3),
                                  'everything_else': round(dt / K * 1e3 - path_ms - head_ms, 3),
                                  'note': 'HIP events around every libgnr.so launch in two further steps (include/gnr.h gnr_timing_*); '
                                          'everything_else = 2D backbones, grasp head under autograd (MIOpen), losses, optimizer, '
                                          'all-reduce and host gaps'},
            'hip_kernels_ms_per_step': per_step,
            'roofline': {'bound': 'mfma', 'kernel': 'k_view1_bwd on the volume points (gnr_sample_volume_bwd)', 'achieved': round(ach, 3),
                         'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4), 'ms_per_launch': round(ms, 4),
                         'launches_timed': cnt, 'flops_per_launch': fl, 'traffic': recorded_bwd_traffic(n)[0],
                         'traffic_source': recorded_bwd_traffic(n)[1],
                         'note': 'algorithmic fp32 FLOPs 2*2*(6304+2112+624+264) per (view, point): dX + dW of decoder, prob_embed, '
                                 'ray_dir_fc, neuray gate; the recomputed forward is not counted.  Next to it the kernel scatters '
                                 '4 taps x 64 channels x 4 B = 1 KB of feature-map gradient per (view, point) through L2 atomics'},
        }
    # isolated fwd+bwd of the PyTorch parts on the same batch (torch events): what `everything_else` is made of
    if rank == 0:
        imgs = torch.cat([s['ref_imgs_info']['imgs'] for s in scenes])
        nr = net.nr_net

        def backbones():
            f = nr.image_encoder(imgs)
            r = nr.vis_encoder(nr.init_net({'imgs': imgs}, None, True), f)
            (f.sum() + r.sum()).backward()
        vol = torch.randn(n, 1, 40, 40, 40, device=dev, requires_grad=True)

        def head():
            q, r, w = net.vgn_net(vol)
            (q.sum() + r.sum() + w.sum()).backward()
        net.train()
        rec['isolated_ms'] = {'backbones_fwd_bwd': round(ev_ms(backbones), 3), 'grasp_head_fwd_bwd': round(ev_ms(head), 3)}
        net.zero_grad(set_to_none=True)
    if dist is not None and world > 1:
        # the step's only collective, alone: sum all-reduce of the flat fp32 gradient buffer (+1 scene counter)
        flat = torch.zeros(sum(p.numel() for p in net.parameters()) + 1, device=dev)
        ms = ev_ms(lambda: dist.all_reduce(flat), iters=10)
        if rank == 0:
            rec['allreduce'] = {'bytes': flat.numel() * 4, 'ms': round(ms, 4), 'GBps_bus': round(2 * (world - 1) / world * flat.numel() * 4 / (ms * 1e-3) / 1e9, 2)}
    del tr, net
    torch.cuda.empty_cache()
    return rec


def backbone_leg(hp, bref, bque, dev, B, step_ms):
    """SURVEY.md §8d second figure: images -> img_feats / ray_feats (PyTorch-ROCm 2D backbones) -> hot path -> HIP grasp head,
    B scenes per pass, forward only."""
    net = build_model(dev).eval()
    nr = net.nr_net
    imgs = bref['imgs'].reshape(-1, *bref['imgs'].shape[2:])
    vol = torch.zeros(B, 1, 40, 40, 40, device=dev)

    def backbones():
        with torch.no_grad():
            for i in range(0, imgs.shape[0], 48):                 # 48 images per call: MIOpen workspaces stay small
                f = nr.image_encoder(imgs[i:i + 48])
                nr.vis_encoder(nr.init_net({'imgs': imgs[i:i + 48]}, None, False), f)

    def head():
        with torch.no_grad():
            net.grasp_head(vol)
    bb, hd = ev_ms(backbones), ev_ms(head)
    tot = bb + step_ms + hd
    return {'value': round(B / tot * 1e3, 2), 'unit': 'scenes/s', 'ms': {'backbones': round(bb, 3), 'hot_path_step': round(step_ms, 3), 'grasp_head_hip': round(hd, 3)},
            'note': f'{B} scenes: 2D backbones under PyTorch-ROCm/MIOpen (kept in PyTorch by north_star) + the timed hot-path step + the HIP '
                    f'grasp head, run back to back; synthetic weights'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32, help='scenes per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-train', action='store_true', help='skip the configs[4] train-step sub-record')
    ap.add_argument('--no-backbones', action='store_true', help='skip the images -> grasps figure')
    ap.add_argument('--no-f32-build', action='store_true', help='skip timing the fp32-MFMA companion build next to the product')
    ap.add_argument('--train-scenes', type=int, default=8)
    ap.add_argument('--train-steps', type=int, default=8)
    ap.add_argument('--train-warmup', type=int, default=10, help='the caching allocators and MIOpen settle over ~10 steps')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f'--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})')
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a ROCm GPU; the hot path has no CPU fallback')
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ:      # under torch.distributed.run, also for one rank
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))

    from graspnerf_amd.hotpath import HotPath, batch_scenes
    wnp = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'weights_seed0.npz')))
    hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'), device=f'cuda:{local}')
    B = args.batch
    c = CONFIGS['cfg2']
    lo, hi = scene_shard(world * B, rank, world)          # contiguous block of the global scene list
    scenes = [make_scene(i, 'cfg2') for i in range(lo, hi)]
    bref, bque = batch_scenes(scenes)
    dev = hp.device
    bref = {k: torch.from_numpy(v).to(dev) for k, v in bref.items()}          # inputs resident in HBM
    bque = {k: torch.from_numpy(v).to(dev) for k, v in bque.items()}
    res, rn, dn = c['res'], c['rn'], 40

    def step():
        prep = hp.prepare(bref, res, rn, dn)
        vol = hp.sample_volume(bref, res, prepared=prep)
        co, fi = hp.render(bref, bque, prepared=prep)
        return vol, co, fi

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    rccl_ranks = None
    if dist is not None:                                  # an actual collective over RCCL: every rank contributes 1
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
        assert rccl_ranks == dist.get_world_size()

    for _ in range(max(args.warmup, 1)):
        step()
    # ---- parity gate in front of the timed region (rank 0 holds scene 0 = the scene the reference golden pins)
    parity = None
    if rank == 0 and B >= 1:
        prep = hp.prepare(bref, res, rn, dn)
        vol = hp.sample_volume(bref, res, prepared=prep)
        co, fi, inds = hp.render(bref, bque, prepared=prep, debug=True)
        torch.cuda.synchronize()
        try:
            parity = parity_gate(hp, bref, bque, vol, co, fi, inds)
        except SystemExit as e:                            # no timing is accepted: tell the other ranks, then stop
            print(e, file=sys.stderr, flush=True)
        del vol, co, fi, inds
    ok = max_over_ranks(0.0 if (rank != 0 or parity is not None) else 1.0, dev) == 0.0
    if not ok:
        if dist is not None:
            dist.destroy_process_group()
        sys.exit(3)
    sync()
    if rank == 0:
        _lib.timing_begin(only='k_chain.volume')      # HIP events around the dominant kernel's launches, on its launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, dev)
    table = _lib.timing_end() if rank == 0 else {}
    # per-kernel table: a few further steps with every launch bracketed (outside the timed region: 11 event pairs per step
    # cost ~2 % of it)
    if rank == 0:
        _lib.timing_begin()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        full = _lib.timing_end()
    per_rank = None
    if dist is not None:
        mine = torch.tensor([B * args.steps / dt_local], device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [round(float(v), 1) for v in allv]

    out = None
    if rank == 0:
        # dominant kernel: average of the HIP-event pairs recorded around each of its launches INSIDE the timed region; a
        # stand-alone re-timing is reported next to it
        n_vol, t_vol = table['k_chain.volume']
        n_ren, t_ren = full['k_chain.render']
        ms = t_vol / n_vol
        ms_ren = t_ren / n_ren
        ms_alone = hp.time_chain_kernel(bref, res, iters=10)
        fl = chain_flops(B * res ** 3, c['V'], render=False)
        fl_ren = chain_flops(B * rn * dn, c['V'], render=True)
        achieved = fl / (ms * 1e-3) / 1e12
        traffic, counters = recorded_pmc(B)
        K = args.steps
        out = {
            'metric': METRIC, 'value': round(world * B * args.steps / dt, 3), 'unit': 'scenes/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'dtype_note': 'fp32 values end to end (inputs, activations, accumulators, outputs); inside k_chain the products of the wide layers are '
                          'formed on the f16 matrix cores from fp32 operands carried as fp16 pairs (1 fp32 ulp; three exact partial products per MAC, '
                          'DESIGN.md 4.1b); f32_mfma_build is the same step with those products on fp32 instructions',
            'parity_checked': parity is not None, 'parity': parity,
            'config': {'workload': f'{B} scenes/GPU/step, 6 views 288x512 (feature maps 72x128x32 x2), 40^3 TSDF volume + '
                                   f'512 rays x (40 coarse + 40 fine) samples incl. pixel_colors_gt, forward only, eval-mode resampling, '
                                   f'inputs resident in HBM (BASELINE.json configs[2]; configs[3] = the same on 8 GPUs)',
                       'global_batch': world * B, 'parallelism': f'scene-sharded x{world}, no data-path collective'},
            'rccl_ranks': rccl_ranks, 'per_rank_scenes_per_s': per_rank,
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic,
                         'kernel': 'k_chain<6,false> on the volume points', 'ms_per_launch': round(ms, 4),
                         'launches_timed': n_vol, 'ms_per_launch_standalone': round(ms_alone, 4),
                         'flops_per_launch': fl,
                         'note': 'achieved = algorithmic (un-hoisted) fp32 FLOPs, 2*(6*27736+6528) per point (SURVEY.md §8d), over the launch '
                                 'time; peak = the fp32-instruction peak of the part (fp32-input MFMA = fp32 vector rate).  frac > 1 is not '
                                 'an accounting error: the wide layers run on the f16 matrix cores with every fp32 operand carried as '
                                 'an fp16 pair (h + m 2^-11, equal to the operand to 1 fp32 ulp) and three exact partial products per MAC '
                                 '(DESIGN.md §4.1b; error against fp64 below that of the fp32 MFMA chain, profiles/r02_f_split_mfma_ubench.txt).  '
                                 'What the kernel executes on the matrix pipe is in f16_mfma; what binds it now is VALU issue '
                                 '(operand splitting, activations, projection / bilinear), see counters',
                         'f16_mfma': {'executed_tflops': round(executed_mfma_flops(B * res ** 3, False) / (ms * 1e-3) / 1e12, 1),
                                      'peak': PEAK_F16_MFMA_TFLOPS,
                                      'frac': round(executed_mfma_flops(B * res ** 3, False) / (ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                                      'executed_flops_per_launch': executed_mfma_flops(B * res ** 3, False)},
                         'render_launch': {'kernel': 'k_chain<6,true> on the ray points (2 launches per step)', 'ms_per_launch': round(ms_ren, 4),
                                           'achieved': round(fl_ren / (ms_ren * 1e-3) / 1e12, 3),
                                           'frac': round(fl_ren / (ms_ren * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), 'flops_per_launch': fl_ren}},
            'kernels_ms_per_step': {k: round(v[1] / 5, 4) for k, v in sorted(full.items(), key=lambda kv: -kv[1][1])},
        }
        # SURVEY.md §8d extras: the whole step against both rooflines (38.7 GFLOP and 26.0 MB compulsory HBM bytes per scene,
        # TSDF + render) and the recorded counters of the dominant kernel
        sps = B * args.steps / dt
        out['roofline']['whole_step'] = {'fp32_fraction': round(38.7e9 * sps / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
                                         'hbm_fraction': round(26.0e6 * sps / (PEAK_HBM_TBPS * 1e12), 5),
                                         'traffic_GBps_dominant_kernel': None if traffic is None else round(traffic / (ms * 1e-3) / 1e9, 1)}
        out['roofline']['counters'] = counters
    step_ms = dt / args.steps * 1e3
    if not args.no_train:
        rec = train_leg(args, world, rank, dev, dist, sync)
        if rank == 0:
            out['train_step'] = rec
    if rank == 0 and world == 1 and not args.no_backbones:
        out['with_backbones'] = backbone_leg(hp, bref, bque, dev, B, step_ms)
    if rank == 0 and world == 1 and not args.no_f32_build:
        out['f32_mfma_build'] = f32_mfma_build_leg(out['value'])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:    # the CPU leg is reported at N=1 only (the other ranks would wait)
        out['cpu_baseline'] = cpu_baseline(wnp)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
