import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
ref, que = make_scene(0, 'cfg1')
bref, bque = batch_scenes([(ref, que)])
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
g = torch.Generator().manual_seed(0)
idx = torch.randperm(96 * 128, generator=g)[:8192]
xy = torch.stack([idx // 128, idx % 128], -1).float()[None].cuda()
prep = hp.prepare(bref, 16, 64, 16)
outs = [hp.depth_mean(bref, xy, 'coarse', prepared=prep).clone() for _ in range(4)]
torch.cuda.synchronize()
print('same prep:', [torch.equal(outs[0], o) for o in outs[1:]], [(outs[0] - o).abs().max().item() for o in outs[1:]])
outs2 = []
for _ in range(3):
    prep = hp.prepare(bref, 16, 64, 16)
    outs2.append(hp.depth_mean(bref, xy, 'coarse', prepared=prep).clone())
print('re-prepared:', [torch.equal(outs[0], o) for o in outs2])
d = (outs[0] - outs[1]).abs()
nz = d.nonzero()
print('n differing', len(nz), nz[:10].tolist())
# model context
from test_train_step import build, scene_data
net = build('cuda').eval(); data = scene_data('cuda'); ev = dict(data, eval=True, full_vol=True)
with torch.no_grad():
    rs = []
    for i in range(4):
        torch.manual_seed(5); rs.append(net(ev))
    for k in ('volume', 'depth_mean', 'depth_mean_fine'):
        print(k, [torch.equal(rs[1][k], r[k]) for r in rs[2:]], [(rs[1][k] - r[k]).abs().max().item() for r in rs[2:]])
    nr = net.nr_net
    ref = dict(data['ref_imgs_info'])
    f1 = nr.image_encoder(ref['imgs']); f2 = nr.image_encoder(ref['imgs'])
    print('image_encoder deterministic', torch.equal(f1, f2))
    ref['img_feats'] = f1
    r1 = nr.vis_encoder(nr.init_net(ref, None, False), f1); r2 = nr.vis_encoder(nr.init_net(ref, None, False), f1)
    print('ray_feats deterministic', torch.equal(r1, r2), (r1 - r2).abs().max().item())
