"""Which of the HIP glue ops moves the feature extractor's gradients away from float64?  (GPU)  python tools/dbg/backbone_glue_diag.py"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
for name, sw in [('all first', dict(norm=True, pad=True, upsample=True)), ('all second', dict(norm=True, pad=True, upsample=True)), ('stock', dict(norm=False, pad=False, upsample=False)), ('stock again', dict(norm=False, pad=False, upsample=False)),
                 ('norm', dict(norm=True, pad=False, upsample=False)), ('pad', dict(norm=False, pad=True, upsample=False)),
                 ('norm again', dict(norm=True, pad=False, upsample=False)), ('norm+pad', dict(norm=True, pad=True, upsample=False)), ('norm+up', dict(norm=True, pad=False, upsample=True)),
                 ('upsample', dict(norm=False, pad=False, upsample=True)), ('all', dict(norm=True, pad=True, upsample=True))]:
    backbone.HIP_GLUE.update(sw)
    net.zero_grad()
    y = net({'imgs': imgs})
    (y * dy).sum().backward()
    ey = float((y.detach().double().cpu() - y64).abs().max() / y64.abs().max())
    rs = []
    for k, p in net.named_parameters():
        n = float(g64[k].norm())
        if n > 1e-6:
            rs.append((float((p.grad.double().cpu() - g64[k]).norm()) / n, k))
    rs.sort(reverse=True)
    late = {k: r for r, k in rs}
    print('   ', ' '.join(f'{k}={late[k]:.1e}' for k in ['out_conv.2.weight', 'out_conv.1.conv.5.weight', 'out_conv.1.conv.3.weight', 'out_conv.1.conv.2.weight', 'out_conv.1.conv.0.weight', 'out_conv.0.weight', 'res_net.out_conv.weight', 'res_net.iconv2.bn.weight', 'res_net.iconv2.conv.weight', 'res_net.upconv2.conv.bn.weight', 'res_net.iconv3.bn.weight', 'res_net.layer3.2.bn2.weight', 'res_net.conv1.weight'] if k in late))
    print(f'{name:12s} y err {ey:.2e}  worst grads: ' + ', '.join(f'{k} {r:.1e}' for r, k in rs[:4]), ' median %.1e' % rs[len(rs) // 2][0], flush=True)
