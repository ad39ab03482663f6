"""gnr_geo_dual_fwd: the matrix-core kernel against the fp32 FMA kernel at a training pass's point count (4 scenes x 512 rays x 80 samples);
HIP events of the library's own timing hooks.  python tools/time_geo_dual_fwd.py"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights, _lib
from graspnerf_amd.hotpath import HotPath
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
canon = torch.from_numpy(weights.canonical_blob(wnp, 'fine')).cuda()
rng = np.random.default_rng(0)
Pn = 4 * 512 * 80
stats = rng.standard_normal((Pn, 66)).astype(np.float32); stats[:, 32:64] = np.abs(stats[:, 32:64]); stats[:, 65] = 6
stats, pts, gamma = (torch.from_numpy(x).cuda() for x in (stats, rng.uniform(-0.5, 0.5, (Pn, 3)).astype(np.float32), rng.standard_normal((Pn, 3)).astype(np.float32)))
out = {}
for fp32 in (False, True):
    hp.set_option('geo_dual_fp32', fp32)
    for _ in range(3): hp.geo_dual_fwd(canon, stats, pts, gamma)
    torch.cuda.synchronize()
    _lib.timing_begin()
    for _ in range(20): hp.geo_dual_fwd(canon, stats, pts, gamma)
    torch.cuda.synchronize()
    out['fp32 FMA kernel' if fp32 else 'matrix cores'] = {k: round(v[1] / v[0] * 1e3, 2) for k, v in _lib.timing_end().items()}
print(json.dumps({'points': Pn, 'us_per_launch': out}, indent=1))
