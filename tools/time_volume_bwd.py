"""Time the HIP training forward + backward of sample_volume (gnr_sample_volume_fwd_train / gnr_sample_volume_bwd) on a
batch of full-size scenes, stage by stage.  python tools/time_volume_bwd.py [--scenes 8]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene

ap = argparse.ArgumentParser()
ap.add_argument('--scenes', type=int, default=8)
ap.add_argument('--fixed-point-feature-grads', action='store_true', help='GNR_OPT_FEATURE_GRAD_FIXED: 64-bit fixed-point scatter')
ap.add_argument('--direct-scatter', action='store_true', help='feature-map gradients as direct float atomics (GNR_OPT_DIRECT_SCATTER)')
ap.add_argument('--distinct-scenes', action='store_true', help='scenes 0..n-1 of the synthetic generator instead of n copies of scene 0')
a = ap.parse_args()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
can = weights.canonical_blob(wnp, 'coarse')
hp.set_bwd_weights(weights.pack_bwd(can))
can_dev = torch.from_numpy(can).cuda()
hp.feature_grad_mode(a.fixed_point_feature_grads)
hp.set_option('direct_scatter', a.direct_scatter)
one = make_scene(0, 'cfg2', with_query_image=False)
bref, _ = batch_scenes([make_scene(i, 'cfg2', with_query_image=False) for i in range(a.scenes)] if a.distinct_scenes else [one] * a.scenes)
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
dvol = torch.randn(a.scenes, 1, 40, 40, 40, device='cuda')

def timed(f, n=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
prep = hp.prepare(bref, 40)
t_inf = timed(lambda: hp.sample_volume(bref, 40, prepared=prep))
t_fwd = timed(lambda: hp.sample_volume_train(bref, 40, prepared=prep))
print(f'{a.scenes} scenes: inference forward {t_inf:.2f} ms, training forward (saves states) {t_fwd:.2f} ms')
prev = 0.0
for name, stages in (('tail', 16), ('+ geometry_fc / reduction 2', 24), ('+ view loop 2', 28), ('+ hoist / reduction 1', 30), ('+ view loop 1 / scatter', 31)):
    t = timed(lambda: hp.sample_volume_bwd(dvol, can_dev, stages=stages))
    print(f'  backward {name:32s} {t:8.2f} ms  (stage {t - prev:7.2f} ms)')
    prev = t
print(f'fwd+bwd per scene: {(t_fwd + prev) / a.scenes:.2f} ms')
