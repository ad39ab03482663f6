"""Wall-clock split of one full GraspNeRF.forward (reference shapes: 6 views 288x512, 40^3, 512 rays) on the GPU:
PyTorch 2D backbones | HIP hot path (sample_volume, render, depth-mean head) | PyTorch grasp head."""
import os, sys, time
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd.renderer import GraspNeRF
from graspnerf_amd.synth import make_scene, synth_state_dict

CFG = yaml.safe_load("""
init_net_type: cost_volume
agg_net_type: neus
use_hierarchical_sampling: true
dist_decoder_cfg: {use_vis: false}
fine_dist_decoder_cfg: {use_vis: false}
sample_volume: true
render_rgb: true
volume_type: [sdf]
volume_resolution: 40
depth_sample_num: 40
fine_depth_sample_num: 40
agg_net_cfg: {sample_num: 40}
fine_agg_net_cfg: {sample_num: 40}
render_depth: true
""")
net = GraspNeRF(CFG).eval()
syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
net = net.cuda()
ref, que = make_scene(0, 'cfg2')
t = lambda a: torch.from_numpy(a).cuda()
ref_info = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
            'depth_range': t(que['depth_range'])[None]}
nr = net.nr_net


def timed(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


with torch.no_grad():
    def backbones():
        r = dict(ref_info)
        r['img_feats'] = nr.image_encoder(r['imgs'])
        r['ray_feats'] = nr.vis_encoder(nr.init_net(r), r['img_feats'])
        return r
    tb, r = timed(backbones)
    tv, vol = timed(lambda: nr.sample_volume(r))
    tr, _ = timed(lambda: nr.render(que_info, r, False))
    td, _ = timed(lambda: nr.predict_mean_for_depth_loss(r))
    th, _ = timed(lambda: net.vgn_net(vol))
    vol32 = vol.repeat(32, 1, 1, 1, 1)
    th32, _ = timed(lambda: net.vgn_net(vol32))
    data = {'step': 0, 'eval': True, 'full_vol': True, 'ref_imgs_info': ref_info, 'que_imgs_info': que_info}
    tf, _ = timed(lambda: net(data))
print(f'single scene (B=1) ms: backbones {tb:.2f} | sample_volume {tv:.3f} | render 512 rays {tr:.3f} | depth-mean {td:.3f} | '
      f'grasp head {th:.3f} | full forward {tf:.2f}')
print(f'grasp head (PyTorch/MIOpen) on 32 volumes: {th32:.2f} ms')
from graspnerf_amd.grasp_head import GraspHead
hh = GraspHead(net.vgn_net.state_dict())
t1, _ = timed(lambda: hh(vol))
t32, _ = timed(lambda: hh(vol32))
print(f'grasp head (HIP MFMA) : {t1:.3f} ms single, {t32:.3f} ms on 32 volumes')

if '--profile' in sys.argv:
    import cProfile, pstats
    with torch.no_grad():
        for _ in range(3): net(data)
        torch.cuda.synchronize()
        pr = cProfile.Profile(); pr.enable()
        for _ in range(5): net(data)
        torch.cuda.synchronize()
        pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)

if '--graph' in sys.argv:
    from graspnerf_amd.graph import GraphedForward
    net.nr_net.cfg['depth_coords_rng'] = 'device'
    with torch.no_grad():
        te, oe = timed(lambda: net(data))
    gf = GraphedForward(net, data)
    tg, og = timed(lambda: gf(data))
    print(f'full forward: eager {te:.2f} ms, hipGraph replay {tg:.2f} ms; volume max|diff| '
          f'{(og["volume"] - oe["volume"]).abs().max().item():.2e}')
