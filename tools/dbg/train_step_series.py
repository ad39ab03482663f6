"""Per-step durations of N consecutive train steps (BASELINE configs[4], bench.py's train leg) with and without Python's cyclic
garbage collector, to locate the periodic slow step the driver's run showed (one 101 ms step among 16).
python tools/dbg/train_step_series.py [--steps 96] [--warmup 24]"""
import argparse, gc, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from graspnerf_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=96)
ap.add_argument('--warmup', type=int, default=24)
ap.add_argument('--scenes', type=int, default=8)
ap.add_argument('--threads', type=int, default=0, help='0: graspnerf_amd.hostenv.limit_host_threads(); -1: leave torch default (128 on the GPU boxes)')
a = ap.parse_args()
from graspnerf_amd import hostenv
if a.threads == 0:
    print('threads', hostenv.limit_host_threads(), 'budget', hostenv.cpu_budget())
elif a.threads > 0:
    torch.set_num_threads(a.threads)
print('torch threads', torch.get_num_threads(), flush=True)
dev = torch.device('cuda:0')
net = bench.build_model(dev)
tr = Trainer(net)
scenes = bench.train_scenes(a.scenes, 0, dev)
for _ in range(a.warmup):
    tr.step(scenes)
torch.cuda.synchronize()
out = {}
for mode in ('gc_on', 'gc_off'):
    if mode == 'gc_off':
        gc.collect(); gc.freeze(); gc.disable()
    elif mode == 'gc_on_again':
        gc.enable()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    host, counts = [], []
    marks[0].record()
    for i in range(a.steps):
        h0 = time.perf_counter()
        c0 = gc.get_stats()[2]['collections']
        tr.step(scenes)
        host.append((time.perf_counter() - h0) * 1e3)
        counts.append(gc.get_stats()[2]['collections'] - c0)
        marks[i + 1].record()
    torch.cuda.synchronize()
    ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    med = float(np.median(ms))
    slow = [(i, round(ms[i], 2), round(host[i], 2), counts[i]) for i in range(a.steps) if ms[i] > 1.04 * med]
    out[mode] = {'median_ms': round(med, 3), 'max_ms': round(max(ms), 3), 'slow_steps(index, device_ms, host_ms, gen2_collections)': slow,
                 'alloc_retries': torch.cuda.memory_stats(dev).get('num_alloc_retries', 0), 'reserved_GB': round(torch.cuda.memory_reserved(dev) / 2**30, 2)}
    print(mode, json.dumps(out[mode]), flush=True)
