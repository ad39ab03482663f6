import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from test_train_step import build, scene_data, CFG
from graspnerf_amd.trainer import Trainer, train_losses
from graspnerf_amd.renderer import GraspNeRF
from graspnerf_amd import losses
net = build('cuda'); data = scene_data('cuda'); ev = dict(data, eval=True, full_vol=True)
def run(n):
    with torch.no_grad():
        n.eval(); torch.manual_seed(5); return n(ev)
a = run(net); b = run(net)
for k in ('volume', 'depth_mean', 'depth_mean_fine', 'depth_coords'):
    print('repeat', k, torch.equal(a[k], b[k]))
tr = Trainer(net, {'lr_init': 1e-2}); torch.manual_seed(6); tr.step([data])
c = run(net); d = run(net)
fresh = GraspNeRF(CFG); fresh.load_state_dict(net.state_dict(), strict=True); fresh = fresh.cuda()
e = run(fresh); f = run(fresh)
for k in ('volume', 'depth_mean', 'depth_mean_2', 'depth_mean_fine', 'depth_coords', 'sdf_values'):
    print(k, 'net twice', torch.equal(c[k], d[k]), 'fresh twice', torch.equal(e[k], f[k]), 'net vs fresh', torch.equal(c[k], e[k]),
          (c[k].float() - e[k].float()).abs().max().item())
# gradient check: which parameter is worst
G = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_step.npz')))
net = build('cuda').train(); torch.manual_seed(321)
terms = train_losses(net(data), data); losses.total_loss(terms).backward(); torch.cuda.synchronize()
norms = dict(zip(G['param_names'].tolist(), G['grad_norms'].tolist()))
rows = []
for k, p in net.named_parameters():
    n = float(p.grad.double().norm())
    rows.append(((abs(n - norms[k]) - 1e-7) / (norms[k] + 1e-12), k, norms[k]))
rows.sort(reverse=True)
for r in rows[:12]: print('%.4e %s %.3e' % r)
hot = [r for r in rows if any(s in r[1] for s in ('dist_decoder', 'agg_net'))]
print('worst hot-path', hot[0]); print('worst vgn', [r for r in rows if 'vgn_net' in r[1]][0])
