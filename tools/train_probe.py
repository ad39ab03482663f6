"""Does the fp16-pair form hold on weights that an optimiser has moved?  (The round-5 review: every weight in every test is seeded-random,
the range guard's trip rate on other dynamic ranges is unknown; there is no trained model_best.pth in this environment.)
Trains the model mirror with this repository's own trainer -- HIP path in both directions, the reference's losses, synthetic scenes and
targets -- for --steps Adam steps at --lr (ten times the reference's 1e-4 by default, so that the weights move by O(1) in a minute),
8 full-size scenes per step, and records along the way: the loss terms, the range-guard status of every --every-th step's
forward / backward (bits 0-2 = a launch fell back to its fp32 twin, 4 = a lost partner), skipped optimiser steps, and how far each
hot-path tensor moved.  The hot-path weights after training are written as an .npz in the layout of tests/golden/weights_seed0.npz:
    python tools/train_probe.py --steps 1000 --out gpurun_out/weights_trained_probe.npz --log gpurun_out/train_probe.json
tests/test_range_guard.py runs the float64 arbiter and the range status on the committed copy (tests/golden/weights_trained_probe.npz)."""
import argparse, json, os, sys, time
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=1000)
ap.add_argument('--lr', type=float, default=1e-3)
ap.add_argument('--scenes', type=int, default=8)
ap.add_argument('--every', type=int, default=50)
ap.add_argument('--groups', type=int, default=3, help='the steps rotate through this many groups of --scenes synthetic scenes')
ap.add_argument('--out', default='gpurun_out/weights_trained_probe.npz')
ap.add_argument('--log', default='gpurun_out/train_probe.json')
a = ap.parse_args()
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd.hostenv import limit_host_threads
limit_host_threads(1)
from graspnerf_amd.renderer import GraspNeRF
from graspnerf_amd.synth import make_scene, synth_state_dict, synth_loss_case
from graspnerf_amd.trainer import Trainer
import importlib.util
spec = importlib.util.spec_from_file_location('tsb_cfg', os.path.join(ROOT, 'tools', 'train_step_bench.py'))
src = open(os.path.join(ROOT, 'tools', 'train_step_bench.py')).read()
CFG = yaml.safe_load(src[src.index('CFG = yaml.safe_load("""') + len('CFG = yaml.safe_load("""'):src.index('""")', src.index('CFG = yaml.safe_load("""'))])
CFG['depth_coords_rng'] = 'device'
dev = torch.device('cuda:0')
net = GraspNeRF(CFG)
syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
net = net.to(dev)
seed_keys = list(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')).keys())
hot0 = {k: net.state_dict()['nr_net.' + k].detach().cpu().numpy().copy() for k in seed_keys}
tr = Trainer(net, lr_cfg={'lr_init': a.lr, 'lr_min': a.lr}, log_every=a.every)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
scenes = []
for i in range(a.scenes * a.groups):
    ref, que = make_scene(i, 'cfg2')
    _, gt = synth_loss_case(seed=100 + i, rfn=6, h=288, w=512, rn=512, R=40)
    ri = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
    ri.update(true_depth=t(gt['true_depth']), sdf_gt=t(gt['sdf_gt']))
    qi = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None], 'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    scenes.append({'ref_imgs_info': ri, 'que_imgs_info': qi, 'src_imgs_info': dict(ri), 'grasp_info': tuple(t(x) for x in gt['grasp_info'])})
rec = {'steps': a.steps, 'lr': a.lr, 'scenes_per_step': a.scenes, 'scene_groups': a.groups, 'log': []}
status_or = 0
t0 = time.perf_counter()
for s in range(a.steps):
    g0 = (s % a.groups) * a.scenes
    log = tr.step(scenes[g0:g0 + a.scenes])
    if (s + 1) % a.every == 0:
        hot = net.nr_net._hot
        words = hot.status_words().cpu().numpy() if hot is not None else np.zeros(1, np.int32)
        st = int(np.bitwise_or.reduce(words.astype(np.int64)))
        status_or |= st
        rec['log'].append({'step': s + 1, 'status_bits_of_this_step': st, **{k: float(v) for k, v in tr.last_log().items() if k.startswith('loss') or k == 'lr'}})
        print(rec['log'][-1], flush=True)
rec['seconds'] = time.perf_counter() - t0
rec['status_bits_or_over_sampled_steps'] = status_or
rec['skipped_optimizer_steps'] = tr.skipped_steps()
sd = net.state_dict()
hot1 = {k: sd['nr_net.' + k].detach().cpu().numpy().astype(np.float32) for k in seed_keys}
assert all(np.isfinite(v).all() for v in hot1.values()), 'a hot-path tensor is not finite after training'
moved = {k: {'rel_change': float(np.abs(hot1[k] - hot0[k]).max() / (np.abs(hot0[k]).max() + 1e-12)), 'absmax_before': float(np.abs(hot0[k]).max()),
             'absmax_after': float(np.abs(hot1[k]).max())} for k in seed_keys}
rec['largest_weight_after'] = max(m['absmax_after'] for m in moved.values())
rec['median_rel_change'] = float(np.median([m['rel_change'] for m in moved.values()]))
rec['max_rel_change'] = max(m['rel_change'] for m in moved.values())
rec['tensors'] = moved
os.makedirs(os.path.dirname(os.path.join(ROOT, a.out)) or '.', exist_ok=True)
np.savez_compressed(os.path.join(ROOT, a.out), **hot1)
json.dump(rec, open(os.path.join(ROOT, a.log), 'w'), indent=1)
print(json.dumps({k: v for k, v in rec.items() if k not in ('tensors', 'log')}))
