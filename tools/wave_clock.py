"""Where the wave slots of a k_chain launch spend the launch (measurement build: tools/build_variant.sh wclk -DGNR_WAVE_CLOCK=1).
GNR_LIB=libgnr_wclk.so python tools/wave_clock.py [--batch 32] [--out FILE.json]
Every wavefront of the inference launches stamps the 100 MHz clock at entry, after the weight image is staged, and at exit, and
every tile its own duration; this prints, per launch kind (volume / render): the launch span, the mean wavefront life as a fraction of it
(1 - that = wave slots standing empty: ramp + tail), the staging time, tiles per wavefront and the spread of the tile durations, per XCD."""
import argparse, ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights
from graspnerf_amd.hotpath import HotPath, batch_scenes
from graspnerf_amd.synth import make_scene

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--out', default=None)
ap.add_argument('--ray-order', action='store_true')
a = ap.parse_args()
wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
if a.ray_order:
    hp.set_option('ray_order_morton', True)
scenes = [make_scene(i, 'cfg2', with_query_image=False) for i in range(a.batch)]
bref, bque = batch_scenes(scenes)
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
bque = {k: torch.from_numpy(v).cuda() for k, v in bque.items()}
for _ in range(3):
    prep = hp.prepare(bref, 40, 512, 40)
    hp.sample_volume(bref, 40, prepared=prep)
    hp.render(bref, bque, prepared=prep)
torch.cuda.synchronize()
L = hp.L
L.gnr_dbg_wave_clock.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
L.gnr_dbg_wave_clock.restype = C.c_int
res = {}
for which, name in ((0, 'volume'), (1, 'render (the step\'s last launch: the fine pass)')):
    w = np.zeros((4096, 8), np.uint64)
    t = np.zeros(1 << 17, np.uint32)
    assert L.gnr_dbg_wave_clock(which, w.ctypes.data, t.ctypes.data) == 0
    keep = w[:, 3] > 0
    wg = (np.arange(4096) // 8)[keep]
    w = w[keep].astype(np.int64)
    t0, t1, t2, n = w[:, 0], w[:, 1], w[:, 2], w[:, 3]
    span = (t2.max() - t0.min()) * 0.01              # us
    life = (t2 - t0) * 0.01
    ntile = int(n.sum())
    tt = t[:ntile].astype(np.float64) * 0.01
    r = dict(wavefronts=int(len(w)), span_us=float(span), mean_life_frac=float(life.mean() / span), min_life_frac=float(life.min() / span),
             start_spread_us=float((t0.max() - t0.min()) * 0.01), staging_us_mean=float(((t1 - t0) * 0.01).mean()),
             end_spread_us=float((t2.max() - t2.min()) * 0.01), end_p10_to_max_us=float((t2.max() - np.percentile(t2, 10)) * 0.01),
             tiles=ntile, tiles_per_wave_min=int(n.min()), tiles_per_wave_max=int(n.max()),
             tile_us_mean=float(tt.mean()), tile_us_p05=float(np.percentile(tt, 5)), tile_us_p50=float(np.percentile(tt, 50)),
             tile_us_p95=float(np.percentile(tt, 95)), tile_us_max=float(tt.max()),
             sum_tile_over_life=float(tt.sum() / life.sum()))
    r['phase_us_per_tile'] = {k: float(w[:, 4 + j].sum() * 0.01 / ntile) for j, k in enumerate(('head + view loop 1', 'reduction 1 + hoist', 'view loop 2', 'reduction 2 + geometry_fc + record'))}
    widx = (np.arange(4096) % 8)[keep]
    r['tiles_per_wave_by_index_in_workgroup'] = [float(n[widx == k].mean()) for k in range(8)]
    r['end_us_after_launch_start_by_index'] = [float((t2[widx == k].mean() - t0.min()) * 0.01) for k in range(8)]
    # per XCD (workgroup b -> XCD b % 8): when its last wavefront ends, relative to the launch start
    r['xcd_end_us'] = [float((t2[(wg % 8) == x].max() - t0.min()) * 0.01) for x in range(8)]
    r['xcd_mean_life_us'] = [float(life[(wg % 8) == x].mean()) for x in range(8)]
    res[name] = r
    print(name)
    for k, v in r.items():
        print(f'  {k}: {v}')
if a.out:
    json.dump(res, open(a.out, 'w'), indent=1)
