"""Are the occasional slow train steps the Python cyclic collector?  40 steps; per-step host time and the collections that ran."""
import gc, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from graspnerf_amd.trainer import Trainer

dev = torch.device('cuda', 0)
tr = Trainer(bench.build_model(dev))
scenes = bench.train_scenes(8, 0, dev)
for _ in range(10):
    tr.step(scenes)
torch.cuda.synchronize()
events = []
gc.callbacks.append(lambda phase, info: events.append((phase, info['generation'], time.perf_counter())))
for mode in ("default", "again"):
    if mode == "freeze":
        gc.collect(); gc.freeze()
    rows = []
    for i in range(20):
        events.clear()
        t0 = time.perf_counter()
        tr.step(scenes)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        gens = [g for ph, g, _ in events if ph == 'start']
        dur = sum((b[2] - a[2]) for a, b in zip(events[::2], events[1::2])) * 1e3
        rows.append((round(dt, 1), gens.count(0), gens.count(1), gens.count(2), round(dur, 1)))
    print(mode, 'per step: (ms, gen0, gen1, gen2 collections, ms inside the collector)')
    print(rows)
    print(mode, 'tracked objects', len(gc.get_objects()))
