"""End-to-end train step (BASELINE.json configs[4]): backbones + volumetric path + grasp head + losses, backward, one flat
gradient all-reduce over RCCL, Adam.  `--scenes` scenes per GPU per step (default 8), full-size scenes (6 views 288x512,
40^3 volume, 512 rays x (40+40) samples).  The volumetric path runs in HIP in both directions; --coords-rng cpu (default) draws the depth-loss pixels with the reference's CPU
randperm stream (through gnr_host_randperm_prefix), device on the GPU generator.
    python tools/train_step_bench.py [--scenes 8] [--steps 3] [--warmup 1]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_step_bench.py ..."""
import argparse, json, os, sys, time
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd.renderer import GraspNeRF
from graspnerf_amd.synth import make_scene, synth_state_dict, synth_loss_case
from graspnerf_amd.trainer import Trainer

CFG = yaml.safe_load("""
network: grasp_nerf
init_net_type: cost_volume
agg_net_type: neus
use_hierarchical_sampling: true
use_depth_loss: true
dist_decoder_cfg: {use_vis: false}
fine_dist_decoder_cfg: {use_vis: false}
ray_batch_num: 4096
sample_volume: true
render_rgb: true
volume_type: [sdf]
volume_resolution: 40
depth_sample_num: 40
fine_depth_sample_num: 40
agg_net_cfg: {sample_num: 40, init_s: 0.3, fix_s: 0}
fine_agg_net_cfg: {sample_num: 40, init_s: 0.3, fix_s: 0}
render_depth: true
""")

ap = argparse.ArgumentParser()
ap.add_argument('--scenes', type=int, default=8)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--warmup', type=int, default=1)
ap.add_argument('--coords-rng', default='cpu', choices=['cpu', 'device'], help="depth-loss pixel draw: the reference's CPU randperm stream, or the GPU generator")
ap.add_argument('--sync-debug', action='store_true')
ap.add_argument('--miopen-find', action='store_true', help='torch.backends.cudnn.benchmark: let MIOpen search its convolution solvers')
ap.add_argument('--profile', default=None, help='write torch.profiler tables of one extra step to this file')
a = ap.parse_args()
world, rank, local = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
torch.cuda.set_device(local)
torch.backends.cudnn.benchmark = bool(a.miopen_find)
dev = torch.device('cuda', local)
dist = None
if 'TORCHELASTIC_RUN_ID' in os.environ or world > 1:
    import torch.distributed as dist
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
CFG['depth_coords_rng'] = a.coords_rng
net = GraspNeRF(CFG)
syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
net = net.to(dev)
tr = Trainer(net, log_every=int(os.environ.get('LOG_EVERY', 20)))   # the reference's train_log_step (trainer.py:31)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
scenes = []
for i in range(a.scenes):
    ref, que = make_scene(rank * a.scenes + i, 'cfg2')
    _, gt = synth_loss_case(seed=100 + i, rfn=6, h=288, w=512, rn=512, R=40)
    ri = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
    ri.update(true_depth=t(gt['true_depth']), sdf_gt=t(gt['sdf_gt']))
    qi = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None],
          'depth_range': t(que['depth_range'])[None], 'imgs': t(que['imgs'])}
    scenes.append({'ref_imgs_info': ri, 'que_imgs_info': qi, 'src_imgs_info': dict(ri), 'grasp_info': tuple(t(x) for x in gt['grasp_info'])})

def sync():
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
for _ in range(a.warmup):
    log = tr.step(scenes)
sync(); t0 = time.perf_counter()
for _ in range(a.steps):
    log = tr.step(scenes)
sync(); dt = time.perf_counter() - t0
tm = torch.tensor([dt], device=dev)
if dist is not None:
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
if rank == 0:
    dt = float(tm)
    print(json.dumps({'metric': 'train scenes/sec (fwd+loss+bwd+allreduce+Adam), 6-view 40^3 grid + 512 rays', 'value': world * a.scenes * a.steps / dt,
                      'unit': 'scenes/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
                      'scenes_per_gpu': a.scenes, 'depth_coords_rng': a.coords_rng, 'miopen_find': bool(a.miopen_find),
                      'max_mem_GB': torch.cuda.max_memory_allocated() / 2 ** 30,
                      'loss': {k: round(v, 6) for k, v in log.items() if k.startswith('loss')}}))
if a.sync_debug and rank == 0:                      # list every host<->device synchronisation of one step (stderr)
    import warnings
    warnings.simplefilter('always')
    torch.cuda.set_sync_debug_mode(1)
    tr.step(scenes)
    torch.cuda.set_sync_debug_mode(0)
if a.profile and rank == 0:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        tr.step(scenes)
        torch.cuda.synchronize()
    ka = prof.key_averages()
    with open(a.profile, 'w') as f:
        f.write(ka.table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=70))
        f.write('\n')
        f.write(ka.table(sort_by='self_cpu_time_total', row_limit=35, max_name_column_width=70))
if dist is not None:
    dist.destroy_process_group()
