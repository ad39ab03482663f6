"""Run the HIP grasp head a few times (for profilers).  python tools/run_head.py [B]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd.backbone import ConvNet
from graspnerf_amd.grasp_head import GraspHead
from graspnerf_amd.synth import synth_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = ConvNet()
syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=11)
hh = GraspHead({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
vol = torch.rand(B, 1, 40, 40, 40, device='cuda') * 2 - 1
for _ in range(5):
    hh(vol)
torch.cuda.synchronize()
