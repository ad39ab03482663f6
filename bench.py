"""Benchmark of the MI355X-native GraspNeRF volumetric hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one forward pass of the hot path over a batch of B synthetic scenes per GPU with all inputs already resident in
HBM: feature-map repack (gnr_prepare) + TSDF volume (sample_volume, 40^3) + ray rendering (512 rays, 40 coarse + 40 fine
samples, ground-truth pixel colours), 6 views of 288x512 (BASELINE.json configs[2]; configs[3] = the same on 8 GPUs).
Scenes are independent, so N GPUs shard scenes with no data-path collective (weak scaling: B scenes per GPU).

Before any timing is accepted rank 0 runs a PARITY GATE: scene 0 of its batch is the scene tests/golden/golden_cfg2.npz pins
(outputs of the imported reference), and the batched launch must reproduce it.

`--dist-backend gloo --stub-step-ms T` runs the SAME control flow without a GPU (the step is a sleep): the N > 1 branches --
RCCL rank count, barrier + MAX-over-ranks, per-rank gather, parity-gate failure propagation, the train leg's all-reduce
record -- are executed under gloo by tests/test_bench_dist.py, so the first multi-GPU lease does not debug them.

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


def source_sha16(*names):
    """sha256 of kernel source files (graspnerf_amd/csrc/<name>), first 16 hex digits: ties recorded counters to a build."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(ROOT, 'graspnerf_amd', 'csrc', n), 'rb').read())
    return h.hexdigest()[:16]


def newest_pmc(stamp_key, *sources):
    """The counters collected in this run (live_pmc) when there are any; else the newest committed profiles/r*_pmc_counters.json
    (tools/collect_profiles.sh), or None when its stamp `stamp_key` (written by tools/pmc_summary.py) is not the sha256 of the
    CURRENT kernel sources: counters of another build are not reported."""
    import glob
    if _LIVE_PMC['doc'] is not None:
        return _LIVE_PMC['doc'], 'measured in this run (rocprofv3 --pmc child passes, roofline.counters.live)'
    try:
        newest = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_counters.json')))[-1]
        d = json.load(open(newest))
        if d.get(stamp_key) != source_sha16(*sources):
            return None, None
        return d, os.path.relpath(newest, ROOT)
    except (OSError, ValueError, IndexError):
        return None, None


def recorded_bwd_traffic(scenes):
    """HBM-side bytes per launch of k_view1_bwd (volume points, 8 scenes) from the recorded PMC passes, or None."""
    d, src = newest_pmc('bwd_source_sha16', 'gnr_kernels.hip', 'gnr_bwd.inc', 'gnr_bwd_scatter.inc', 'gnr_bwd_view1_pw.inc', 'gnr_bwd_view2_pw.inc', 'gnr_bwd_geo_dual_mm.inc')
    if d is None:
        return None, None
    try:
        ks = d['kernels']                                  # (round 5: the partner-wavefront kernel, a template since the fixed-point mode; before: the single-wavefront kernel)
        # (round 6: the binned-scatter instantiation; its rows come back through k_scatter_place / k_scatter_gather, counted with it)
        k = ks.get('k_view1_bwd_pw<false, true>') or ks.get('k_view1_bwd_pw<false>') or ks.get('k_view1_bwd_pw') or ks.get('k_view1_bwd<false>') or ks['k_view1_bwd']
        extra = sum(int(ks[n]['hbm_bytes_corrected']) for n in ('k_scatter_place', 'k_scatter_gather') if n in ks and 'hbm_bytes_corrected' in ks[n])
        return (int(k['hbm_bytes_corrected']) + extra, src) if scenes == 8 else (None, None)
    except KeyError:
        return None, None


def recorded_pmc(batch):
    """Counters of the dominant kernel from the committed rocprofv3 PMC passes (newest profiles/r*_pmc_counters.json, made by
    tools/collect_profiles.sh: separate --pmc runs, gfx950 2x read correction applied to FETCH_SIZE).  PMC counters cannot be
    read from inside this process: the figures are the recorded ones, reported only when the file is stamped with the sha256
    of the kernel source this run was built from (`kernel_source_sha16`) and for the batch size they were measured at.
    -> (traffic bytes per launch, counters dict, valu dict) or (None, None, None)."""
    d, src = newest_pmc('kernel_source_sha16', 'gnr_kernels.hip')
    try:
        if d is None or batch != 32:
            return None, None, None
        k = d['kernels']['k_chain<6, false, false, false, true>']          # V = 6, volume points, inference, no vis branch, pair form
        cycles = k['GRBM_GUI_ACTIVE'] / 8                 # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        simds = 1024
        tiles = 32 * 64000 / 16
        valu_insts = k['SQ_INSTS_VALU'] - k['SQ_INSTS_MFMA']          # SQ_INSTS_VALU counts the MFMAs too
        counters = {
            'source': src, 'kernel_source_sha16': d['kernel_source_sha16'],
            'mfma_pipe_busy': round(k['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * simds), 3),   # summed over the 1024 SIMDs
            'l2_hit_rate': round(k['TCC_HIT_sum'] / (k['TCC_HIT_sum'] + k['TCC_MISS_sum']), 3),
            'mfma_per_launch': k['SQ_INSTS_MFMA'], 'valu_incl_mfma_per_launch': k['SQ_INSTS_VALU']}
        # the resource that binds k_chain: VALU issue.  SQ_ACTIVE_INST_VALU counts quad-cycles over all
        # wavefronts; a SIMD executes one VALU instruction at a time (16 lanes: 4 cycles per wave64 instruction)
        valu = {'instructions_per_tile': round(valu_insts / tiles, 1),
                'cycles_per_instruction': round(4 * k['SQ_ACTIVE_INST_VALU'] / k['SQ_INSTS_VALU'], 3),
                'issue_cycles_per_simd': round(4 * k['SQ_ACTIVE_INST_VALU'] / simds, 0),
                'available_cycles_per_simd': round(cycles, 0),
                'frac': round(4 * k['SQ_ACTIVE_INST_VALU'] / (cycles * simds), 3),
                'note': 'share of the SIMDs\' cycles in which a vector instruction (incl. the MFMA issue slot) executes; what is left '
                        'is the matrix pipe alone (mfma_pipe_busy overlaps partly) and stalls on LDS / memory counters'}
        # the other launches of a step / of a training step from the same stamped file: bytes at the L2's memory side (gfx950 read
        # correction applied by tools/pmc_summary.py), L2 hit rate, instruction mix -- per launch, recomputable from profiles/<source>
        def brief(name, scenes):
            q = d['kernels'].get(name)
            if not q:
                return None
            o = {'traffic': int(q['hbm_bytes_corrected']) if 'hbm_bytes_corrected' in q else None, 'scenes_per_launch': scenes}
            if 'TCC_HIT_sum' in q:
                o['l2_hit_rate'] = round(q['TCC_HIT_sum'] / max(q['TCC_HIT_sum'] + q['TCC_MISS_sum'], 1.0), 3)
            if 'SQ_INSTS_VALU' in q:
                o['mfma_per_launch'], o['valu_incl_mfma_per_launch'] = q['SQ_INSTS_MFMA'], q['SQ_INSTS_VALU']
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in q and 'GRBM_GUI_ACTIVE' in q:
                o['mfma_pipe_busy'] = round(q['SQ_VALU_MFMA_BUSY_CYCLES'] / (q['GRBM_GUI_ACTIVE'] / 8 * simds), 3)
            return o
        counters['other_kernels'] = {n: brief(n, sc) for n, sc in (('k_chain<6, true, false, false, true>', 32), ('k_ray<true>', 32), ('k_ray<false>', 32),
                                                                    ('k_repack_feats', 32))}
        if d.get('bwd_source_sha16') == source_sha16('gnr_kernels.hip', 'gnr_bwd.inc', 'gnr_bwd_scatter.inc', 'gnr_bwd_view1_pw.inc', 'gnr_bwd_view2_pw.inc', 'gnr_bwd_geo_dual_mm.inc'):
            counters['backward_kernels'] = {n: brief(n, 8) for n in d['kernels'] if '_bwd' in n or n.startswith('k_scatter_')}
        counters['pmc_age_commits'] = pmc_age_commits(d)
        return int(k['hbm_bytes_corrected']), counters, valu
    except (KeyError, ValueError, ZeroDivisionError, StopIteration):
        return None, None, None


_LIVE_PMC = {'doc': None, 'info': None}


def live_pmc(cap_s=240.0):
    """The PMC counters of this build measured IN THIS RUN.  rocprofv3 cannot attach to this process, so child runs of the same
    workloads are profiled -- tools/run_hot.py --iters 1 (the B = 32 forward step) and tools/time_volume_bwd.py --scenes 8 (the
    sample_volume backward) -- one counter set per run and nothing else enabled (`rocprofv3 --pmc <set> --output-format csv`: the
    separate passes MI355X_MICROARCH.md's HBM section prescribes; the same sets and the same aggregation as
    tools/collect_profiles.sh -> tools/pmc_summary.py, gfx950 read correction included).  Fills _LIVE_PMC: doc = a document of
    the form kept under profiles/*_pmc_counters.json (or None), info = what was done / why not."""
    import importlib.util, shutil, subprocess, tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        _LIVE_PMC['info'] = {'error': 'rocprofv3 not on PATH'}
        return
    if any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprofiler' in os.environ.get('LD_PRELOAD', ''):
        _LIVE_PMC['info'] = {'skipped': 'this process is itself running under a profiler: no nested counter passes'}
        return
    sp = importlib.util.spec_from_file_location('pmc_summary', os.path.join(ROOT, 'tools', 'pmc_summary.py'))
    pm = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(pm)
    t0 = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix='gnr_pmc_', dir='/tmp')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK', 'LOCAL_WORLD_SIZE')
           and not k.startswith(('TORCHELASTIC', 'MASTER_'))}
    env['TMPDIR'] = '/tmp'
    done, failed = [], []
    try:
        jobs = [('', st, ['tools/run_hot.py', '--iters', '1']) for st in pm.FWD_SETS] + \
               [('bwd_', st, ['tools/time_volume_bwd.py', '--scenes', '8']) for st in pm.BWD_SETS]
        for prefix, st, cmd in jobs:
            left = cap_s - (time.perf_counter() - t0)
            if left < 15:
                failed.append(prefix + st + ': time cap')
                continue
            try:
                subprocess.run([exe, '--pmc', *st.split(), '--output-format', 'csv', '-d', os.path.join(tmp, prefix + st.replace(' ', '_')), '-o', 'p', '--',
                                sys.executable, os.path.join(ROOT, cmd[0]), *cmd[1:]], cwd='/tmp', env=env, timeout=min(left, 90.0),
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
                done.append(prefix + st)
            except Exception as e:                           # a failed pass only drops its counters
                failed.append(prefix + st + ': ' + type(e).__name__)
        res = pm.aggregate(tmp)
        if res:
            doc = pm.document(res)
            doc['git_commit_count'] = None
            _LIVE_PMC['doc'] = doc
        _LIVE_PMC['info'] = {'passes': len(done), 'failed': failed or None, 'seconds': round(time.perf_counter() - t0, 1),
                             'command': 'rocprofv3 --pmc <set> --output-format csv -- python tools/run_hot.py --iters 1 | tools/time_volume_bwd.py --scenes 8 '
                                        '(one child run per counter set, after the timed forward steps of this run)'}
    except Exception as e:                                   # the counters are an extra: a box without a working profiler reports the recorded ones
        _LIVE_PMC['info'] = {'error': repr(e)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_age_commits(d):
    """Commits between the tree the PMC file was collected on and this one (0 = same commit), from the commit count the file was
    stamped with (`git_commit_count`, written here in the build container where the history exists); None where there is no .git
    (the GPU boxes get a snapshot without it -- there `kernel_source_sha16` is what ties the counters to the running kernels)."""
    import subprocess
    try:
        now = int(subprocess.check_output(['git', '-C', ROOT, 'rev-list', '--count', 'HEAD'], stderr=subprocess.DEVNULL, timeout=10).decode())
        return now - int(d['git_commit_count'])
    except Exception:
        return None


_EXTRAS = {'t0': None, 'budget': 300.0, 'skipped': []}


def extras_left(name):
    """True while the optional legs' wall-clock budget (--extras-budget-s) lasts; a leg refused here is named in the line."""
    if _EXTRAS['t0'] is None:
        _EXTRAS['t0'] = time.perf_counter()
    if time.perf_counter() - _EXTRAS['t0'] <= _EXTRAS['budget']:
        return True
    _EXTRAS['skipped'].append(name)
    return False


def recorded_bwd_arbiter():
    """The float64 arbiter of the BACKWARD at the benched training shape (tests/test_bwd_arbiter.py), from the newest committed
    profiles/*bwd_arbiter*.json: per parameter tensor |HIP gradient - float64| / |torch fp32 autograd - float64|."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*bwd_arbiter*.json')), reverse=True):
        try:
            doc = json.load(open(f))
        except (OSError, ValueError):
            continue
        out = {'source': os.path.relpath(f, ROOT)}
        try:                                            # (a file of another layout that matches the pattern -- the parity log of the same test -- is passed over)
            for tag, rows in doc.items():
                r = [x['rms_ratio'] for x in rows.values()]
                out[tag] = {'tensors': len(r), 'median_rms_ratio': round(float(np.median(r)), 3), 'within_1.5': int(sum(x <= 1.5 for x in r)),
                            'worst_rms_ratio': round(max(r), 2), 'd_ray_feats': rows.get('d_ray_feats'), 'd_img_feats': rows.get('d_img_feats')}
        except (AttributeError, KeyError, TypeError, ValueError):
            continue
        out['note'] = ('8 scenes, 6 views, 40^3 + 512 x (40 + 40): `arithmetic` = prob_embed.0 biased away from its ReLU kink (smooth path, the '
                       'ratios measure arithmetic), `as packed` = the test weights as they are (rows on the kink flip in every fp32 evaluation); '
                       'which tensors sit beyond 1.5 and why: tests/test_bwd_arbiter.py')
        return out
    return None


def recorded_arbiter():
    """The fp64 arbiter at the benched shape (tests/test_range_guard.py::test_fp64_arbiter_benched_shape), from the newest committed
    profiles/*parity_errors*.json that holds its rows: distance of the pair-form path from a float64 evaluation, relative to the fp32
    CPU oracle's distance from it.  (The arbiter is the oracle: bench.py may run the oracle in its cpu_baseline leg only, so the
    line quotes the recorded test run.)"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*parity_errors*.json')), reverse=True):
        try:
            rows = [r for r in json.load(open(f))['worst_per_check'] if 'benched shape' in r.get('what', '') and 'rms_ratio' in r and 'p99_ratio' in r]
        except (OSError, ValueError, KeyError, TypeError, AttributeError):
            continue
        if rows:
            return {'source': os.path.relpath(f, ROOT), 'checks': len(rows),
                    'rms_ratio_max': round(max(r['rms_ratio'] for r in rows), 3), 'p99_ratio_max': round(max(r['p99_ratio'] for r in rows), 3),
                    'rms_ratio_volume': round(max(r['rms_ratio'] for r in rows if r['what'].endswith('volume')), 3),
                    'note': '|HIP fp16-pair path - float64| / |fp32 CPU oracle - float64| over every voxel / sample of the benched shape '
                            '(6 views, 40^3, 512 x (40+40)); test bounds: rms <= 1.5, p99 <= 2'}
    return None


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
    errs['fine_inds_edge_samples'] = int((G['fine_inds_margin'] <= 3e-5).sum())     # samples the fixture records within 3e-5 of a cdf edge
    if errs['fine_inds_differing'] > errs['fine_inds_edge_samples']:
        raise SystemExit('parity gate FAILED: more resampling indices differ than the fixture has samples at cdf edges')
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
def cpu_baseline_all_cores(cap_s=90.0):
    """The same oracle scene at os.cpu_count() threads (SURVEY.md 8d's protocol), measured in THIS run in a child process with a
    wall-clock cap: on the GPU box's 256-thread host torch's intra-op pool collapses on these op sizes (one scene took 171.6 s
    in round 3), so the child usually hits the cap; then the record says so instead of quoting an old figure."""
    import subprocess
    n = os.cpu_count() or 1
    code = ("import sys, time, json, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from oracle import graspnerf_oracle as O; from graspnerf_amd.synth import make_scene\n"
            "torch.set_num_threads(%d)\n"
            "W = {k: torch.from_numpy(v) for k, v in np.load(%r).items()}\n"
            "ref, que = make_scene(0, 'cfg2'); inp, q = O.to_torch(ref), O.to_torch(que)\n"
            "ts = []\n"
            "for i in range(3):\n"
            "    t0 = time.perf_counter(); O.sample_volume(W, inp, 40); O.render(W, inp, q); ts.append(time.perf_counter() - t0)\n"
            "    print(json.dumps(ts), flush=True)\n") % (ROOT, n, os.path.join(ROOT, 'tests', 'golden', 'weights_seed0.npz'))
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=cap_s)
        out = r.stdout
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or '')
    wall = time.perf_counter() - t0
    lines = [l for l in out.strip().splitlines() if l.startswith('[')]
    ts = json.loads(lines[-1]) if lines else []
    rec = {'cores': n, 'measured_in_this_run': True, 'cap_s': cap_s, 'scenes_completed': len(ts), 'wall_s': round(wall, 1)}
    if ts:
        best = min(ts[1:]) if len(ts) > 1 else ts[0]              # the first call includes TorchScript / allocator warm-up
        rec.update(seconds_per_scene=round(best, 3), value=round(1.0 / best, 4))
    else:
        rec.update(capped=True, value_upper_bound=round(1.0 / cap_s, 4), note=f'not one scene finished within {cap_s:.0f} s at {n} threads')
    return rec


def cpu_baseline(weights_np):
    """SURVEY.md §8d protocol: the oracle (torch-CPU fp32 port of the reference path; the only place bench.py touches
    oracle/) on whole scenes of the same workload: 3 warm-ups + median of 10 at 16 threads, one scene on one thread, and the
    same scene at os.cpu_count() threads in a child process under a 90 s cap (`all_cores`: measured in this run, or reported as
    capped -- on the GPU box's 256-thread host torch's intra-op pool collapses from oversubscription on these op sizes)."""
    from oracle import graspnerf_oracle as O
    from graspnerf_amd.hostenv import cpu_budget
    budget = cpu_budget()                                  # CPUs the container may use (cgroup quota / affinity), not the 256 it sees
    cores = min(budget, 16)
    prev_threads = torch.get_num_threads()
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
    torch.set_num_threads(prev_threads)
    return {'value': round(1.0 / med, 4), 'unit': 'scenes/s', 'cores': cores, 'kind': 'port',
            'value_1_thread': round(1.0 / t1, 4),
            'host': {'cpus_visible': os.cpu_count(), 'cpu_budget': budget,
                     'note': 'the container sees every hardware thread of the node but its cgroup grants cpu_budget of them: the `cores` '
                             'figure runs inside the budget, all_cores (os.cpu_count() threads, SURVEY 8d) oversubscribes it and is throttled'},
            'all_cores': cpu_baseline_all_cores() if (os.cpu_count() or 1) > cores else {'cores': cores, 'same_as': 'value'},
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
                            '--no-cpu-baseline', '--no-f32-build', '--no-live-pmc'], env=env, capture_output=True, text=True, timeout=300)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                  # noqa: BLE001  (a failed companion run must not lose the product's line)
        return {'skipped': f'{type(e).__name__}: {e}'[:200]}
    return {'library': 'graspnerf_amd/csrc/libgnr_f32mfma.so (-DGNR_SPLIT16=0)', 'value': d['value'], 'unit': d['unit'], 'steps': d['steps'],
            'ms_per_step': d['ms_per_step'], 'parity_checked': d['parity_checked'],
            'k_chain_volume_ms_per_launch': d['roofline']['ms_per_launch'],
            'frac_of_fp32_mfma_peak': round(d['roofline']['algorithmic_fp32_equiv']['tflops'] / PEAK_F32_MFMA_TFLOPS, 4),
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
    """Average milliseconds of fn() on the current stream (torch events; one untimed call first); wall clock without a GPU."""
    fn()
    if not torch.cuda.is_available():
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        return (time.perf_counter() - t0) / iters * 1e3
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
    tr = Trainer(net, log_every=args.train_log_step)
    scenes = train_scenes(n, rank * n, dev)
    torch.cuda.reset_peak_memory_stats(dev)
    for _ in range(args.train_warmup):
        log = tr.step(scenes)
    sync()
    # the model, the optimizer state and the autograd Function classes are long-lived: move them out of the cyclic collector's
    # young generations (a generation-1 / -2 pass over them cost one step in ~25 its 6-7 ms: tools/dbg/train_step_series.py)
    import gc
    gc.collect()
    gc.freeze()
    dom = 'k_view1_bwd@gnr_sample_volume_bwd'          # dominant backward kernel of the path: first view loop, volume points
    if rank == 0:
        _lib.timing_begin(only=dom)                    # the timed steps bracket this kernel only
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.train_steps + 1)]
    host, host_cpu = [], []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.train_steps):
        h0, c0 = time.perf_counter(), time.process_time()
        log = tr.step(scenes)
        host.append((time.perf_counter() - h0) * 1e3)
        host_cpu.append((time.process_time() - c0) * 1e3)   # CPU time of the whole process (all threads) spent on the step
        marks[i + 1].record()                              # per-step spans on the stream (no synchronisation added)
    sync()
    log = tr.last_log()                                    # the terms of the last timed step (on a logging step step() returned them itself)
    dt = max_over_ranks(time.perf_counter() - t0, dev)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.train_steps)]
    table = _lib.timing_end() if rank == 0 else {}
    K = args.train_steps
    # the same steps with the loss terms read back at the end of EVERY step (log_every = 1: what a per-step progress bar costs,
    # trainer.py:190 of the reference): the same number of steps, reported with the same prominence
    every = tr.log_every
    tr.log_every = 1
    sync()
    t1 = time.perf_counter()
    for _ in range(K):                                     # the same number of steps as the lazy-log figure
        tr.step(scenes)
    sync()
    ms_log1 = max_over_ranks(time.perf_counter() - t1, dev) / K * 1e3
    tr.log_every = every
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
            'value_with_per_step_readback': round(world * n / (ms_log1 * 1e-3), 3),
            'warmup': args.train_warmup, 'log_every': args.train_log_step, 'ms_per_step_with_log_every_1': round(ms_log1, 3),
            'which_is_which': "value = the reference's LOGGING cadence (its log is written every train_log_step = 20 steps, trainer.py:31,159); "
                              "value_with_per_step_readback = the reference's PROGRESS-BAR semantics (the loss is read back after every step, "
                              "trainer.py:190), same number of steps; a comparison with a reference run that shows its progress bar should use the latter",
            'log_note': "every step runs forward, losses, backward, gradient all-reduce and Adam; the loss TERMS are copied to the host every log_every-th step (the reference's train_log_step = 20, trainer.py:31,159) instead of at the end of every step, where the host would wait for the copy and start the next step's ~2 000 launches from an empty queue; ms_per_step_with_log_every_1 = eight further steps with the per-step read-back (the reference's progress bar, trainer.py:190)", 'scenes_per_gpu': n, 'global_batch': world * n, 'n_gpus': world, 'dtype': 'f32', 'data': 'synthetic',
            'config': 'BASELINE.json configs[4]: backbones + nr TSDF + render + depth-mean head + grasp head + losses (render, depth, sdf, vgn), '
                      'batch 8/GPU, one flat fp32 gradient all-reduce (4.66 M parameters), Adam',
            'ms_each_step': [round(x, 2) for x in step_ms], 'ms_per_step_median': round(float(np.median(step_ms)), 3),
            'ms_per_step_max': round(float(np.max(step_ms)), 3), 'max_over_median': round(float(np.max(step_ms) / np.median(step_ms)), 3),
            'stalled_steps': int(sum(x > 1.05 * float(np.median(step_ms)) for x in step_ms)),       # steps more than 5 % over the median, inside the timed region
            'host_ms_each_step': [round(x, 2) for x in host],
            'host_ms_unblocked': round(float(np.min(host)), 2),
            'host_cpu_ms_each_step': [round(x, 2) for x in host_cpu], 'host_cpu_ms_median': round(float(np.median(host_cpu)), 2),
            'host_note': 'host_ms_each_step = wall time the host spends inside step(): it runs one to two steps ahead of the GPU and then blocks '
                         'on a full queue, so the values alternate between its own queueing time (host_ms_unblocked) and waits for the GPU; '
                         'host_cpu_ms_each_step = CPU time of the process (all threads) per step -- what the step needs from the host',
            'max_mem_GB': round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 3),
            'loss': {k: round(v, 6) for k, v in log.items() if k.startswith('loss')},
            'optimizer': 'torch.optim.Adam(fused=True); a step whose backward reports GNR_STATUS_LOST_PARTNER is skipped on the device (found_inf), no host wait',
            'skipped_steps': tr.skipped_steps(),
            'bwd_fp64_arbiter_at_benched_shape': recorded_bwd_arbiter(),
            'split_ms_per_step': {'hip_path_kernels': round(path_ms, 3), 'hip_grasp_head_kernels': round(head_ms, 3),
                                  'everything_else': round(dt / K * 1e3 - path_ms - head_ms, 3),
                                  'note': 'HIP events around every libgnr.so launch in two further steps (include/gnr.h gnr_timing_*); '
                                          'everything_else = 2D backbones, grasp head under autograd (MIOpen), losses, optimizer, '
                                          'all-reduce and host gaps'},
            'hip_kernels_ms_per_step': per_step,
            'roofline': {'bound': 'mfma', 'kernel': 'k_view1_bwd_pw (first view loop backward, compute + partner wavefronts) on the volume points (gnr_sample_volume_bwd)', 'achieved': round(ach, 3),
                         'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4), 'ms_per_launch': round(ms, 4),
                         'launches_timed': cnt, 'flops_per_launch': fl, 'traffic': recorded_bwd_traffic(n)[0],
                         'traffic_source': recorded_bwd_traffic(n)[1],
                         'note': 'algorithmic fp32 FLOPs 2*2*(6304+2112+624+264) per (view, point): dX + dW of decoder, prob_embed, '
                                 'ray_dir_fc, neuray gate; the recomputed forward is not counted.  The feature-map gradient leaves the kernel as one parked '
                                 '256-byte row per (view, point) (binned scatter, csrc/gnr_bwd_scatter.inc: k_scatter_place / k_scatter_gather sum the rows '
                                 'per feature-map pixel -- their times are in hip_kernels_ms_per_step); round 5 scattered 4 taps x 64 channels per (view, point) '
                                 'through L2 atomics from this kernel'},
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
        ar = allreduce_record(dist, world, sum(p.numel() for p in net.parameters()), dev)
        if rank == 0:
            rec['allreduce'] = ar
    del tr, net
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_train_2cpu and extras_left('train_step.at_2_cpus'):
        rec['at_2_cpus'] = train_at_2_cpus(args, rec['ms_per_step'])
    return rec


def train_at_2_cpus(args, ms_here):
    """The same train step in a child process pinned to 2 CPUs (tools/train_step_bench.py --cpus 2: sched_setaffinity before torch
    starts, pools sized for that share) = the host budget of one rank of an 8-rank node under the boxes' 16-CPU cgroup, followed by the
    same steps with the N > 1 gradient exchange (persistent flat buffer, RCCL all-reduce on a one-rank group, copy back).  What a
    1-GPU box can say about the 8-rank step: whether the host keeps up, and what the exchange costs next to the step."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'train_step_bench.py'), '--scenes', str(args.train_scenes), '--steps', str(args.train_steps),
           '--warmup', str(args.train_warmup), '--cpus', '2', '--flat-exchange-steps', '8']
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=dict(os.environ, LOG_EVERY=str(args.train_log_step)))
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1])
        return {'ms_per_step_at_2_cpus': round(d['ms_per_step'], 3), 'value': round(d['value'], 3), 'vs_this_process': round(d['ms_per_step'] / ms_here, 4),
                'cpus_pinned': d['cpus_pinned'], 'torch_intra_op_threads': d['torch_intra_op_threads'], 'host_ms_each_step': d['host_ms_each_step'],
                'with_flat_gradient_exchange': d['with_flat_gradient_exchange'], 'command': ' '.join(cmd[1:]).replace(ROOT + '/', '')}
    except Exception as e:
        return {'error': repr(e)[:300]}


def allreduce_record(dist, world, n_params, dev):
    """The train step's only collective, alone: sum all-reduce of the flat fp32 gradient buffer (+1 scene counter)."""
    flat = torch.zeros(n_params + 1, device=dev)
    ms = ev_ms(lambda: dist.all_reduce(flat), iters=10)
    return {'bytes': flat.numel() * 4, 'ms': round(ms, 4), 'GBps_bus': round(2 * (world - 1) / world * flat.numel() * 4 / (ms * 1e-3) / 1e9, 2)}


def train_leg_stub(args, world, rank, dev, dist, sync):
    """--stub-step-ms: the control flow of train_leg without a GPU (a step = a sleep four times the forward stub): barrier +
    MAX-over-ranks timing and the all-reduce record (the model has 4 659 307 parameters)."""
    n, K = args.train_scenes, args.train_steps
    for _ in range(min(args.train_warmup, 2)):
        time.sleep(4e-3 * args.stub_step_ms)
    sync()
    t0 = time.perf_counter()
    for _ in range(K):
        time.sleep(4e-3 * args.stub_step_ms)
    sync()
    dt = max_over_ranks(time.perf_counter() - t0, dev)
    rec = None
    if rank == 0:
        rec = {'metric': 'train scenes/sec (stub)', 'value': round(world * n * K / dt, 3), 'unit': 'scenes/s', 'ms_per_step': round(dt / K * 1e3, 3),
               'steps': K, 'scenes_per_gpu': n, 'global_batch': world * n, 'n_gpus': world, 'data': 'stub (no GPU work)'}
    if dist is not None and world > 1:
        ar = allreduce_record(dist, world, 4659307, dev)
        if rank == 0:
            rec['allreduce'] = ar
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


class StubPath:
    """--stub-step-ms: stands in for HotPath + the library's timing hooks when there is no GPU (tests/test_bench_dist.py runs the
    N > 1 control flow of this file under gloo).  A step is a sleep; the kernel tables are made up from it."""

    def __init__(self, ms):
        self.ms = ms
        self.device = torch.device('cpu')

    def step(self):
        time.sleep(self.ms * 1e-3)

    def table(self, steps):
        return {'k_chain.volume': (steps, 0.5 * self.ms * steps), 'k_chain.render': (2 * steps, 0.3 * self.ms * steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32, help='scenes per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-train', action='store_true', help='skip the configs[4] train-step sub-record')
    ap.add_argument('--no-backbones', action='store_true', help='skip the images -> grasps figure')
    ap.add_argument('--no-live-pmc', action='store_true', help='skip the two rocprofv3 --pmc child passes that measure roofline.traffic in this run (N = 1)')
    ap.add_argument('--extras-budget-s', type=float, default=300.0, help='wall-clock budget of the OPTIONAL legs behind the timed steps (live PMC child passes, the 2-CPU train child, images -> grasps, the fp32-MFMA companion build): a leg that would start beyond it is skipped and the line says so; the timed forward steps, the train_step record and cpu_baseline always run')
    ap.add_argument('--no-f32-build', action='store_true', help='skip timing the fp32-MFMA companion build next to the product')
    ap.add_argument('--train-scenes', type=int, default=8)
    ap.add_argument('--no-train-2cpu', action='store_true', help='skip the train step of a child process pinned to 2 CPUs (the host budget of one rank of an 8-rank node)')
    ap.add_argument('--train-steps', type=int, default=16, help='timed steps of the train_step record (16: one stalled step moves the mean by 5 %%, not 12)')
    ap.add_argument('--train-log-step', type=int, default=20, help="the loss terms leave the device every N-th step, the reference's train_log_step (trainer.py:31,159: 20); 1 = a device-to-host copy the host waits for at the end of EVERY step (what the reference's progress bar does, trainer.py:190)")
    ap.add_argument('--train-warmup', type=int, default=24, help='the caching allocators and MIOpen (solvers compiled on their first uses) settle over ~20 steps: profiles/r03_d, r03_f show stalled steps up to the 13th')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'], help='nccl = RCCL (the product); gloo only with --stub-step-ms')
    ap.add_argument('--stub-step-ms', type=float, default=0.0, help='> 0: no GPU, a step is a sleep of this length (control-flow test of the N > 1 branches)')
    ap.add_argument('--stub-parity-fail', action='store_true', help='with --stub-step-ms: rank 0 fails its parity gate (every rank must exit 3)')
    args = ap.parse_args()
    # host threads: the node shows 256 hardware threads, the container's cgroup grants 16 of them; torch's default 128-thread pool
    # exhausts that quota on one small CPU op and the kernel throttles the whole process for the rest of the 100 ms period
    # (graspnerf_amd/hostenv.py): every rank takes its share
    from graspnerf_amd.hostenv import limit_host_threads, cpu_budget
    host_threads = limit_host_threads(int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))
    stub = args.stub_step_ms > 0

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f'--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})')
    if args.dist_backend == 'gloo' and not stub:
        sys.exit('--dist-backend gloo is for --stub-step-ms runs: the product communicates over RCCL')
    if not stub:
        if not torch.cuda.is_available():
            sys.exit('bench.py needs a ROCm GPU; the hot path has no CPU fallback')
        torch.cuda.set_device(local)
    dist = None
    if world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ:      # under torch.distributed.run, also for one rank
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if stub:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))

    B = args.batch
    c = CONFIGS['cfg2']
    lo, hi = scene_shard(world * B, rank, world)          # contiguous block of the global scene list
    res, rn, dn = c['res'], c['rn'], 40
    wnp = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'weights_seed0.npz')))
    if stub:
        hp = StubPath(args.stub_step_ms)
        dev = hp.device
        step = hp.step
    else:
        from graspnerf_amd.hotpath import HotPath, batch_scenes
        hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'), device=f'cuda:{local}')
        scenes = [make_scene(i, 'cfg2') for i in range(lo, hi)]
        bref, bque = batch_scenes(scenes)
        dev = hp.device
        bref = {k: torch.from_numpy(v).to(dev) for k, v in bref.items()}          # inputs resident in HBM
        bque = {k: torch.from_numpy(v).to(dev) for k, v in bque.items()}

        def step():
            prep = hp.prepare(bref, res, rn, dn)
            vol = hp.sample_volume(bref, res, prepared=prep)
            co, fi = hp.render(bref, bque, prepared=prep)
            return vol, co, fi

    def sync():
        if dist is not None:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    rccl_ranks = None
    if dist is not None:                                  # an actual collective over RCCL: every rank contributes 1
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
        assert rccl_ranks == dist.get_world_size()

    # ---- parity gate in front of the warm-up and the timed region (rank 0 holds scene 0 = the scene the reference golden pins).  It
    # stands BEFORE the warm-up steps since round 6: the gate compares on the host for a few hundred milliseconds while the GPU idles
    # and drops its clocks, and a timed region that started right behind it ran its first ~20 steps 2.5 % slow (20 timed steps:
    # 6.47 ms per step with 5 or 50 warm-up steps alike, 200 timed steps: 6.32) -- the warm-up exists to absorb exactly that.
    step()                                                 # (first use: allocations, attribute calls)
    parity = None
    range_flags = None
    if rank == 0 and B >= 1:
        try:
            if stub:
                if args.stub_parity_fail:
                    raise SystemExit('parity gate FAILED: (stub) forced failure')
                parity = {'stub': 0.0}
            else:
                prep = hp.prepare(bref, res, rn, dn)
                vol = hp.sample_volume(bref, res, prepared=prep)
                co, fi, inds = hp.render(bref, bque, prepared=prep, debug=True)
                torch.cuda.synchronize()
                range_flags = hp.range_status(prep)          # 0: every chain launch of the step ran in the fp16-pair form
                parity = parity_gate(hp, bref, bque, vol, co, fi, inds)
                del vol, co, fi, inds
        except SystemExit as e:                            # no timing is accepted: tell the other ranks, then stop
            print(e, file=sys.stderr, flush=True)
    ok = max_over_ranks(0.0 if (rank != 0 or parity is not None) else 1.0, dev) == 0.0
    if not ok:
        if dist is not None:
            dist.destroy_process_group()
        sys.exit(3)
    for _ in range(max(args.warmup, 1)):
        step()
    sync()
    if rank == 0 and not stub:
        _lib.timing_begin(only='k_chain.volume')      # HIP events around the dominant kernel's launches, on its launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, dev)
    table = {}
    full = {}
    if rank == 0:
        if stub:
            table, full = hp.table(args.steps), hp.table(5)
        else:
            table = _lib.timing_end()
            # per-kernel table: a few further steps with every launch bracketed (outside the timed region: 14 event pairs per
            # step cost ~2 % of it)
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
        ms_alone = args.stub_step_ms * 0.5 if stub else hp.time_chain_kernel(bref, res, iters=10)
        fl_alg = chain_flops(B * res ** 3, c['V'], render=False)
        fl_exe = executed_mfma_flops(B * res ** 3, False)
        fl_ren = executed_mfma_flops(B * rn * dn, True)
        fl_ren_alg = chain_flops(B * rn * dn, c['V'], render=True)
        achieved = fl_exe / (ms * 1e-3) / 1e12
        _EXTRAS['budget'] = args.extras_budget_s
        if world == 1 and not args.no_live_pmc and not stub and extras_left('live_pmc'):
            live_pmc()                                   # counters of THIS run when the box has the profiler (else the recorded, sha-stamped ones)
        traffic, counters, valu = (None, None, None) if stub else recorded_pmc(B)
        if counters is not None:
            counters['live'] = _LIVE_PMC['info']
        out = {
            'metric': METRIC, 'value': round(world * B * args.steps / dt, 3), 'unit': 'scenes/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic' if not stub else 'stub (no GPU work)',
            'dtype_note': 'fp32 values end to end (inputs, activations, accumulators, outputs); inside k_chain the products of the wide layers are '
                          'formed on the f16 matrix cores from fp32 operands carried as fp16 pairs (1 fp32 ulp; three exact partial products per MAC, '
                          'DESIGN.md 4.1b; as close to a float64 evaluation as the fp32 CPU oracle, tests/test_range_guard.py); operands beyond the '
                          'fp16 range make the launch fall back to the fp32-input MFMA (range_flags); f32_mfma_build is the same step with '
                          'every product on fp32 instructions',
            'fp64_arbiter_at_benched_shape': None if stub else recorded_arbiter(),
            'parity_checked': parity is not None, 'parity': parity, 'range_flags': range_flags,
            'host_threads': {'torch_intra_op': host_threads, 'cpu_budget': cpu_budget(), 'cpus_visible': os.cpu_count()},
            'config': {'workload': f'{B} scenes/GPU/step, 6 views 288x512 (feature maps 72x128x32 x2), 40^3 TSDF volume + '
                                   f'512 rays x (40 coarse + 40 fine) samples incl. pixel_colors_gt, forward only, eval-mode resampling, '
                                   f'inputs resident in HBM (BASELINE.json configs[2]; configs[3] = the same on 8 GPUs)',
                       'global_batch': world * B, 'parallelism': f'scene-sharded x{world}, no data-path collective'},
            'rccl_ranks': rccl_ranks, 'per_rank_scenes_per_s': per_rank, 'dist_backend': args.dist_backend if dist is not None else None,
            # The roof the kernel sits under: the wide layers run on the f16 matrix cores (three exact partial products per fp32
            # MAC), the 1..4-k-step remainders on the fp32-input MFMA: achieved = the MFMA FLOPs the kernel EXECUTES per launch
            # (MFMA_PER_TILE x tiles; PMC SQ_INSTS_MFMA agrees) / its HIP-event duration, peak = the dense f16 MFMA peak.
            # What binds it is VALU issue (`valu`): 12 vector instructions per MFMA (activations, operand splitting, queue moves).
            # achieved = ALGORITHMIC FLOPs per launch (un-hoisted: 2 x (6 x 27 736 + 6 528) = 345 888 per point x
            # B x 40^3 points) / the HIP-event launch time, against the dense f16 MFMA peak the wide layers run under.
            # frac_executed prices the matrix-core FLOPs the kernel actually EXECUTES (three f16 partial products per fp32 MAC,
            # zero-padded K32 tails: 696 x 16 384 + 114 x 2 048 FLOP per 16-point tile) over the same time and peak.
            'roofline': {'bound': 'mfma', 'binding_resource': 'valu_issue', 'achieved': round(fl_alg / (ms * 1e-3) / 1e12, 2), 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(fl_alg / (ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4), 'traffic': traffic,
                         'flops_per_launch': fl_alg, 'flops_per_point': 2 * (c['V'] * MAC_VIEW_VOL + MAC_POINT_CHAIN), 'points_per_launch': B * res ** 3,
                         'achieved_executed': round(achieved, 2), 'frac_executed': round(achieved / PEAK_F16_MFMA_TFLOPS, 4),
                         'flops_per_launch_executed': fl_exe,
                         'kernel': 'k_chain<6,false> on the volume points', 'ms_per_launch': round(ms, 4),
                         'launches_timed': n_vol, 'ms_per_launch_standalone': round(ms_alone, 4),
                         'note': 'frac = achieved / peak with achieved = flops_per_launch / ms_per_launch (algorithmic fp32 FLOPs, SURVEY 8d); '
                                 'frac_executed = flops_per_launch_executed / ms_per_launch / peak (696 v_mfma_f32_16x16x32_f16 + 114 '
                                 'v_mfma_f32_16x16x4_f32 per 16-point tile; counters.mfma_per_launch / (points_per_launch / 16) = 810 agrees).  '
                                 'fp32 operands are carried as fp16 pairs and every fp32 MAC costs three f16 MACs plus a split on the VALU: the '
                                 'kernel is bound by vector issue (valu), not by the matrix pipe (counters.mfma_pipe_busy); the fp32-instruction '
                                 'peak of the part is 157.3 TFLOP/s (f32_mfma_build prices the all-fp32-instruction build against it)',
                         'valu': valu,
                         'algorithmic_fp32_equiv': {'tflops': round(fl_alg / (ms * 1e-3) / 1e12, 2), 'flops_per_launch': fl_alg,
                                                    'note': 'un-hoisted fp32 FLOPs 2*(6*27736+6528) per point (SURVEY.md 8d); the fp32-instruction peak of '
                                                            'the part is 157.3 TFLOP/s (f32_mfma_build.frac_of_fp32_mfma_peak is measured against it)'},
                         'render_launch': {'kernel': 'k_chain<6,true> on the ray points (2 launches per step)', 'ms_per_launch': round(ms_ren, 4),
                                           'achieved': round(fl_ren_alg / (ms_ren * 1e-3) / 1e12, 2),
                                           'frac': round(fl_ren_alg / (ms_ren * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4), 'flops_per_launch': fl_ren_alg,
                                           'flops_per_point': 2 * (c['V'] * MAC_VIEW_RAY + MAC_POINT_CHAIN), 'points_per_launch': B * rn * dn,
                                           'frac_executed': round(fl_ren / (ms_ren * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4), 'flops_per_launch_executed': fl_ren,
                                           'traffic': ((counters or {}).get('other_kernels', {}).get('k_chain<6, true, false, false, true>') or {}).get('traffic'),
                                           'l2_hit_rate': ((counters or {}).get('other_kernels', {}).get('k_chain<6, true, false, false, true>') or {}).get('l2_hit_rate')}},
            'kernels_ms_per_step': {k: round(v[1] / 5, 4) for k, v in sorted(full.items(), key=lambda kv: -kv[1][1])},
        }
        # SURVEY.md §8d extras: the whole step against the HBM roof (26.0 MB compulsory bytes per scene, TSDF + render) and the
        # recorded counters of the dominant kernel
        sps = B * args.steps / dt
        out['roofline']['whole_step'] = {'hbm_fraction': round(26.0e6 * sps / (PEAK_HBM_TBPS * 1e12), 5),
                                         'fp32_equiv_tflops': round(38.7e9 * sps / 1e12, 1),
                                         'traffic_GBps_dominant_kernel': None if traffic is None else round(traffic / (ms * 1e-3) / 1e9, 1)}
        out['roofline']['counters'] = counters
    step_ms = dt / args.steps * 1e3
    if not args.no_train:
        rec = train_leg_stub(args, world, rank, dev, dist, sync) if stub else train_leg(args, world, rank, dev, dist, sync)
        if rank == 0:
            out['train_step'] = rec
    if rank == 0 and world == 1 and not args.no_backbones and not stub and extras_left('with_backbones'):
        out['with_backbones'] = backbone_leg(hp, bref, bque, dev, B, step_ms)
    if rank == 0 and world == 1 and not args.no_f32_build and not stub and extras_left('f32_mfma_build'):
        out['f32_mfma_build'] = f32_mfma_build_leg(out['value'])
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not stub:    # the CPU leg is reported at N=1 only (the other ranks would wait)
        out['cpu_baseline'] = cpu_baseline(wnp)
    if rank == 0:
        out['optional_legs_skipped_for_time'] = _EXTRAS['skipped']
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
