"""Agreement of the two per-point kernels of gnr_geo_dual_bwd where statistics of 3e4 put pre-activations of 1e5 on both sides of the ELU's
kink: the numbers behind the bounds of tests/test_ray_tail.py::test_geo_dual_bwd_on_the_matrix_cores_agrees_with_the_fp32_kernel."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights, _lib
from graspnerf_amd.hotpath import HotPath
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
canon = torch.from_numpy(weights.canonical_blob(wnp, 'fine')).cuda()
L = _lib.lib()
for seed in (11, 12):
    rng = np.random.default_rng(seed)
    Pn = 4 * 512 * 40 + 13
    stats = (rng.standard_normal((Pn, 66)) * 3e4).astype(np.float32)
    stats[:, 32:64] = np.abs(stats[:, 32:64]); stats[:, 64] = rng.uniform(0, 1, Pn); stats[:, 65] = 6
    pts = rng.uniform(-0.5, 0.5, (Pn, 3)).astype(np.float32)
    gamma = (rng.standard_normal((Pn, 3)) * 1e3).astype(np.float32)
    gbar = (rng.standard_normal((Pn, 16)) * 1e3).astype(np.float32)
    gdbar = (rng.standard_normal((Pn, 16)) * 1e2).astype(np.float32)
    hp.set_option('geo_dual_fp32', False)
    got = [x.double() for x in hp.geo_dual_bwd(canon, stats, pts, gamma, gbar, gdbar)]
    hp.set_option('geo_dual_fp32', True)
    want = [x.double() for x in hp.geo_dual_bwd(canon, stats, pts, gamma, gbar, gdbar)]
    hp.set_option('geo_dual_fp32', False)
    for g, w, name in zip(got, want, ('d stats', 'd geometry_fc')):
        d = (g - w).abs()
        print(seed, name, 'max', float(d.max() / w.abs().max()), 'rms rel', float(d.pow(2).mean().sqrt() / w.pow(2).mean().sqrt()),
              'frac > 1e-3 max', float((d > 1e-3 * w.abs().max()).double().mean()), 'frac > 1e-4 max', float((d > 1e-4 * w.abs().max()).double().mean()),
              'rows touched', int((d.reshape(d.shape[0], -1).max(1)[0] > 1e-4 * w.abs().max()).sum()) if d.dim() == 2 else -1, flush=True)
