"""End-to-end train step (BASELINE.json configs[4]): backbones + volumetric path + grasp head + losses, backward, one flat
gradient all-reduce over RCCL, Adam.  `--scenes` scenes per GPU per step (default 8), full-size scenes (6 views 288x512,
40^3 volume, 512 rays x (40+40) samples).  The volumetric path runs in HIP in both directions; --coords-rng cpu (default) draws the depth-loss pixels with the reference's CPU
randperm stream (through gnr_host_randperm_prefix), device on the GPU generator.
    python tools/train_step_bench.py [--scenes 8] [--steps 3] [--warmup 1]
    python tools/train_step_bench.py --cpus 2 --steps 16 --warmup 24 --flat-exchange-steps 8
        the step under the host budget of ONE RANK OF AN 8-RANK NODE (the boxes grant a container 16 CPUs: 2 per rank): the process is
        pinned to 2 CPUs before torch starts (sched_setaffinity = taskset -c), the pools are sized for that share, and after the
        timed steps the same steps run with the gradient exchange of the N > 1 path (persistent flat buffer, RCCL all-reduce on a
        one-rank group, copy back) -- the two things a 1-GPU box can tell about the 8-rank step
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_step_bench.py ..."""
import argparse, json, os, sys, time
ap = argparse.ArgumentParser()
ap.add_argument('--scenes', type=int, default=8)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--warmup', type=int, default=1)
ap.add_argument('--coords-rng', default='cpu', choices=['cpu', 'device'], help="depth-loss pixel draw: the reference's CPU randperm stream, or the GPU generator")
ap.add_argument('--sync-debug', action='store_true')
ap.add_argument('--miopen-find', action='store_true', help='torch.backends.cudnn.benchmark: let MIOpen search its convolution solvers')
ap.add_argument('--profile', default=None, help='write torch.profiler tables of one extra step to this file')
ap.add_argument('--reproducible-feature-grads', action='store_true', help='feature-map gradients through 64-bit fixed-point adds (GNR_OPT_FEATURE_GRAD_FIXED)')
ap.add_argument('--cpus', type=int, default=0, help='pin the process to its first N allowed CPUs before torch is imported (0: leave the affinity alone)')
ap.add_argument('--flat-exchange-steps', type=int, default=0, help='after the timed steps: this many further steps with the N > 1 gradient exchange (flat buffer + RCCL all-reduce on a one-rank group)')
a = ap.parse_args()
if a.cpus > 0:
    os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:a.cpus]))
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd.hostenv import limit_host_threads, cpu_budget
host_threads = limit_host_threads(int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))
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
tr = Trainer(net, log_every=int(os.environ.get('LOG_EVERY', 20)), reproducible_feature_grads=a.reproducible_feature_grads)   # the reference's train_log_step (trainer.py:31)
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
import gc
gc.collect(); gc.freeze()
sync(); t0 = time.perf_counter()
host = []
for _ in range(a.steps):
    h0 = time.perf_counter()
    log = tr.step(scenes)
    host.append((time.perf_counter() - h0) * 1e3)
sync(); dt = time.perf_counter() - t0
log = tr.last_log()
tm = torch.tensor([dt], device=dev)
if dist is not None:
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
flat = None
if a.flat_exchange_steps > 0 and world == 1:
    try:
        import torch.distributed as tdist
        if not tdist.is_initialized():
            tdist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % (29000 + os.getpid() % 2000), rank=0, world_size=1, device_id=dev)
        tr.flat_exchange = True
        for _ in range(3):
            tr.step(scenes)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(a.flat_exchange_steps):
            tr.step(scenes)
        torch.cuda.synchronize()
        ms_flat = (time.perf_counter() - t1) / a.flat_exchange_steps * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for p in tr.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        h0 = time.perf_counter(); e0.record()
        for _ in range(10):
            tr._allreduce_grads(a.scenes)
        e1.record(); h1 = time.perf_counter(); torch.cuda.synchronize()
        flat = {'ms_per_step': round(ms_flat, 3), 'exchange_alone_gpu_ms': round(e0.elapsed_time(e1) / 10, 4), 'exchange_alone_host_ms': round((h1 - h0) * 100, 4),
                'bytes': int(tr._flat.numel() * 4), 'note': 'multi-tensor copy of the 346 gradients into the persistent flat buffer, RCCL sum all-reduce (one-rank group), division by the global scene count, multi-tensor copy back'}
        tr.flat_exchange = 'auto'
        tdist.destroy_process_group()
    except Exception as e:                           # the measurement is optional: a box whose RCCL refuses a one-rank group still reports its step
        flat = {'error': repr(e)[:300]}
if rank == 0:
    dt = float(tm)
    print(json.dumps({'metric': 'train scenes/sec (fwd+loss+bwd+allreduce+Adam), 6-view 40^3 grid + 512 rays', 'value': world * a.scenes * a.steps / dt,
                      'unit': 'scenes/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
                      'scenes_per_gpu': a.scenes, 'depth_coords_rng': a.coords_rng, 'miopen_find': bool(a.miopen_find),
                      'reproducible_feature_grads': bool(a.reproducible_feature_grads), 'cpus_pinned': a.cpus or None, 'cpu_budget': cpu_budget(), 'torch_intra_op_threads': host_threads, 'log_every': tr.log_every,
                      'host_ms_each_step': [round(x, 2) for x in host], 'host_ms_median': round(float(np.median(host)), 3),
                      'with_flat_gradient_exchange': flat,
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
