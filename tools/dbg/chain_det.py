"""Is the chain bitwise reproducible?  Runs sample_volume / render several times on the same inputs (B=1 and B=3)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene

wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
for name, res, dn in (('cfg1', 16, 16), ('cfg2', 40, 40)):
    scenes = [make_scene(i, name) for i in range(3)]
    for B in (1, 3):
        bref, bque = batch_scenes(scenes[:B])
        bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
        bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
        vols = [hp.sample_volume(bref, res).cpu().numpy().copy() for _ in range(4)]
        print(name, 'B', B, 'volume runs differ from run 0 at', [int((v != vols[0]).sum()) for v in vols[1:]], 'max', [float(np.abs(v - vols[0]).max()) for v in vols[1:]])
        cfg = {'depth_sample_num': dn, 'fine_depth_sample_num': dn}
        outs = []
        for _ in range(3):
            co, fi = hp.render(bref, bque, cfg)
            outs.append({k: v.cpu().numpy().copy() for k, v in co.items() if v is not None and hasattr(v, 'cpu')})
        print(name, 'B', B, 'coarse hit_prob runs differ at', [int((o['hit_prob_nr'] != outs[0]['hit_prob_nr']).sum()) for o in outs[1:]])
        if B == 1:
            v1 = vols[0]
        else:
            print(name, 'batched scene 0 vs single: differ at', int((vols[0][0] != v1[0]).sum()), 'max', float(np.abs(vols[0][0] - v1[0]).max()))
