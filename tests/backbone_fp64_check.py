"""Helper of test_backbone_ops.py (run as a script in its own process, GPU): the init-net feature extractor (ResUNetLight + head)
with the HIP glue ops and with ATen's ops, forward and every parameter gradient, against the float64 evaluation on the CPU.
Its own process because MIOpen's Winograd convolutions are switched off for it (MIOPEN_DEBUG_CONV_WINOGRAD=0 must be in the
environment before the first convolution): which solver MIOpen picks for a layer depends on the process' allocation history, and
the fp32 Winograd ones move gradients by up to 5e-3 -- more than the difference under test (measured: worst gradient 1.2e-5 from
float64 with the HIP glue, 4.2e-4 with ATen's instance norm, either way 5e-3 once a Winograd solver is picked).  Prints one JSON line."""
import copy, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspnerf_amd import backbone

torch.manual_seed(5)
net = backbone.CostVolumeInitNet().cuda()
imgs = torch.rand(3, 3, 96, 128, device='cuda')
ref = copy.deepcopy(net).cpu().double()
y64 = ref({'imgs': imgs.cpu().double()})
dy64 = torch.randn(y64.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(6))
(y64 * dy64).sum().backward()
g64 = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
dy = dy64.float().cuda()
out = {}
for name, on in (('hip', True), ('aten', False)):
    backbone.HIP_GLUE.update(norm=on, pad=on, upsample=on)
    net.zero_grad()
    y = net({'imgs': imgs})
    (y * dy).sum().backward()
    rel = {}
    for k, p in net.named_parameters():
        n = float(g64[k].norm())
        if n > 1e-6:             # biases in front of an instance norm: exactly-zero gradient, rounding noise on all sides
            rel[k] = float((p.grad.double().cpu() - g64[k]).norm()) / n
    out[name] = {'features': float((y.detach().double().cpu() - y64).abs().max() / y64.abs().max()), 'grads': rel}
print(json.dumps(out))
