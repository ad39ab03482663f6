"""Throughput of the PyTorch-ROCm 2D backbones (out of the HIP path's scope, north_star) on a batch of scenes, for the
"second figure including the backbone" of SURVEY.md §8d.  python tools/time_backbones.py [--scenes 32]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd.backbone import ResUNetLight, CostVolumeInitNet, DefaultVisEncoder

ap = argparse.ArgumentParser()
ap.add_argument('--scenes', type=int, default=32)
ap.add_argument('--channels-last', action='store_true')
a = ap.parse_args()
dev = 'cuda'
enc, init, vis = ResUNetLight(out_dim=32).to(dev).eval(), CostVolumeInitNet().to(dev).eval(), DefaultVisEncoder().to(dev).eval()
x = torch.rand(a.scenes * 6, 3, 288, 512, device=dev)
if a.channels_last:
    x = x.contiguous(memory_format=torch.channels_last)
    enc, init, vis = (m.to(memory_format=torch.channels_last) for m in (enc, init, vis))

def run(chunk):
    outs = []
    with torch.no_grad():
        for i in range(0, x.shape[0], chunk):
            xi = x[i:i + chunk]
            f = enc(xi)
            r = vis(init({'imgs': xi}), f)
            outs.append((f, r))
    return outs
for chunk in (6, 48, 192):
    if chunk > x.shape[0]:
        continue
    run(chunk); torch.cuda.synchronize()
    t = time.perf_counter(); run(chunk); run(chunk); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 2
    print(f'{a.scenes} scenes x 6 views, {chunk} images per call: {dt * 1e3:.1f} ms  ({a.scenes / dt:.0f} scenes/s, {dt / a.scenes * 1e3:.2f} ms/scene)'
          f'{" channels_last" if a.channels_last else ""}', flush=True)
