"""Grasp head (gd/networks.py ConvNet) forward + backward of 8 volumes with the decoder's two k3 layers through MIOpen (F.conv3d) or
through the HIP path (conv3d_same): round-4 measurement, 4.69 vs 4.38 ms, gradients equal to 1e-6.  The product routes them through HIP."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graspnerf_amd.backbone import ConvNet
import torch.nn.functional as F
from graspnerf_amd import backbone as BB
BB.FOLD_UPSAMPLED_K5 = BB.STRIDE2_AS_S2D = False      # this tool measures the routes of the UNFOLDED module (round-4 history; the product folds, tools/dbg/head_train_prof.py)


def _route(mode):
    """'1': the product's decoder (all three layers via conv3d_same); '0': the two k3 layers through F.conv3d (MIOpen)."""
    def fwd_miopen(self, x):
        x = F.interpolate(F.relu(self.conv1(x)), 10)
        x = F.interpolate(F.relu(self.conv2(x)), 20)
        return F.interpolate(F.relu(BB.conv3d_same(x, self.conv3.weight, self.conv3.bias)), 40)
    BB._Decoder.forward = _ORIG if mode == '1' else fwd_miopen


_ORIG = BB._Decoder.forward
torch.manual_seed(0)
net = ConvNet().cuda()
x = torch.randn(8, 1, 40, 40, 40, device='cuda', requires_grad=True)
def step():
    q, r, w = net(x)
    (q.sum() + r.sum() + w.sum()).backward()
for mode in ('0', '1', '0', '1'):
    _route(mode)
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    g = [p.grad.clone() for p in net.parameters()]
    print('k3 via HIP' if mode == '1' else 'k3 via MIOpen', round((time.perf_counter() - t) / 20 * 1e3, 3), 'ms fwd+bwd')
    if mode == '0': ref = None
    net.zero_grad()
# parity of the two routes
_route('0'); net.zero_grad(); x.grad = None; step(); g0 = [p.grad.clone() for p in net.parameters()]; gx0 = x.grad.clone()
_route('1'); net.zero_grad(); x.grad = None; step(); g1 = [p.grad.clone() for p in net.parameters()]; gx1 = x.grad.clone()
print('max rel grad diff', max(float((a - b).abs().max() / (a.abs().max() + 1e-12)) for a, b in zip(g0, g1)), float((gx0 - gx1).abs().max() / gx0.abs().max()))
