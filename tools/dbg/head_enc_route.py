"""Grasp head encoder (gd/networks.py: three stride-2 convolutions) forward + backward of 8 volumes: MIOpen's strided conv3d against the
stride-1 HIP path followed by a [::2, ::2, ::2] subsample (same values: the strided convolution IS the subsampled same-padding one).
Measurement only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from graspnerf_amd import backbone as BB
BB.FOLD_UPSAMPLED_K5 = BB.STRIDE2_AS_S2D = False      # this tool measures the routes of the UNFOLDED module (round-4 history; the product folds, tools/dbg/head_train_prof.py)
from graspnerf_amd.backbone import ConvNet

ORIG = BB._Encoder.forward


def hip_enc(self, x):
    sub = lambda t: t[..., ::2, ::2, ::2]
    x = F.relu(sub(BB.conv3d_same(x, self.conv1.weight, self.conv1.bias)))
    x = F.relu(sub(BB.conv3d_same(x, self.conv2.weight, self.conv2.bias)))
    return F.relu(sub(BB.conv3d_same(x, self.conv3.weight, self.conv3.bias)))


torch.manual_seed(0)
net = ConvNet().cuda()
x = torch.randn(8, 1, 40, 40, 40, device='cuda', requires_grad=True)


def step():
    q, r, w = net(x)
    (q.sum() + r.sum() + w.sum()).backward()


for mode in ('miopen', 'hip', 'miopen', 'hip'):
    BB._Encoder.forward = ORIG if mode == 'miopen' else hip_enc
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(mode, round((time.perf_counter() - t) / 20 * 1e3, 3), 'ms fwd+bwd', flush=True)
    net.zero_grad()
gs = []
for mode in ('miopen', 'hip'):
    BB._Encoder.forward = ORIG if mode == 'miopen' else hip_enc
    net.zero_grad(); x.grad = None; step()
    gs.append([p.grad.clone() for p in net.parameters()] + [x.grad.clone()])
print('max rel grad diff', max(float((a - b).abs().max() / (a.abs().max() + 1e-12)) for a, b in zip(*gs)))
