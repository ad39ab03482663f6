"""Grasp head (gd/networks.py ConvNet) forward + backward of 8 volumes, the product's routes, for a per-kernel profile:
rocprofv3 --kernel-trace --stats -- python tools/dbg/head_train_prof.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graspnerf_amd.backbone import ConvNet
from graspnerf_amd import backbone as BB
BB.FOLD_UPSAMPLED_K5 = os.environ.get('FOLD', '1') == '1'
BB.STRIDE2_AS_S2D = os.environ.get('S2D', '1') == '1'

torch.manual_seed(0)
net = ConvNet().cuda()
x = torch.randn(8, 1, 40, 40, 40, device='cuda', requires_grad=True)


def step():
    q, r, w = net(x)
    (q.sum() + r.sum() + w.sum()).backward()


for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
n = int(os.environ.get('N', 20))
for _ in range(n): step()
torch.cuda.synchronize()
print('fold' if BB.FOLD_UPSAMPLED_K5 else 'plain', 's2d' if BB.STRIDE2_AS_S2D else 'miopen-encoder', 'head fwd+bwd, 8 volumes:', round((time.perf_counter() - t) / n * 1e3, 3), 'ms')
