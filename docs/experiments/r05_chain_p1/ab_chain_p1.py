"""Phase-1 prototype of the 8-point x (view pair) tile (graspnerf_amd/csrc/gnr_chain_p1.inc), measured on the GPU:
    tools/build_variant.sh p1 -DGNR_PROTO_P1=1        (here, cross-compiled)
    GNR_LIB=libgnr_p1.so python tools/ab_chain_p1.py [--batch 32] [--iters 10] [--out gpurun_out/p1.json]
For each (mapping, threads per workgroup) it prints the ms per launch of phase 1 + first cross-view reduction over the B x 40^3
volume points, and compares the two mappings' statistics (scene 0) so that the 8-point mapping is known to compute the same thing."""
import argparse, ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graspnerf_amd import weights, _lib
from graspnerf_amd.hotpath import HotPath, batch_scenes, _f32
from graspnerf_amd.synth import make_scene

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--res', type=int, default=40)
ap.add_argument('--repeat', type=int, default=3)
ap.add_argument('--out', default='')
a = ap.parse_args()

wnp = dict(np.load(os.path.join(ROOT, 'tests/golden/weights_seed0.npz')))
hp = HotPath(weights.pack_state_dict(wnp, 'coarse'), weights.pack_state_dict(wnp, 'fine'))
fn = hp.L.gnr_proto_chain_p1
fn.restype = C.c_int
scenes = [make_scene(i, 'cfg2', with_query_image=False) for i in range(a.batch)]
bref, _ = batch_scenes(scenes)
bref = {k: torch.from_numpy(v).cuda() for k, v in bref.items()}
R = a.res
scene, keep, ws = hp.prepare(bref, R)
bbox_min = _f32(bref['bbox3d'], hp.device)[:, 0].contiguous()
P = R ** 3


def run(mode, threads, want_sv):
    pt = 16 if mode == 0 else 8
    out_sum = torch.zeros(a.batch * ((P + pt - 1) // pt) * 64, dtype=torch.float32, device='cuda')
    out_sv = torch.zeros(a.batch * P * 4 * 36, dtype=torch.float32, device='cuda') if want_sv else None
    ms = C.c_float(0)
    rc = fn(C.byref(scene), C.c_void_p(bbox_min.data_ptr()), C.c_int(R), C.c_void_p(hp.wc.data_ptr()), C.c_int(mode), C.c_int(threads),
            C.c_void_p(out_sum.data_ptr()), C.c_void_p(out_sv.data_ptr() if want_sv else None), C.c_void_p(ws.data_ptr()),
            C.c_size_t(ws.numel()), C.c_int(a.iters), C.byref(ms), hp._stream())
    _lib.check(rc, 'gnr_proto_chain_p1')
    torch.cuda.synchronize()
    return ms.value, out_sum, out_sv


res = {'batch': a.batch, 'res': R, 'points': a.batch * P, 'runs': []}
_, s0, sv0 = run(0, 512, True)
_, s1, sv1 = run(1, 768, True)
_, s2, sv2 = run(1, 512, True)
sv0, sv1, sv2 = sv0.double(), sv1.double(), sv2.double()
scale = sv0.abs().max().item()
res['statistics_scale'] = scale
res['max_abs_diff_8pt768_vs_16pt'] = (sv1 - sv0).abs().max().item()
res['max_abs_diff_8pt512_vs_16pt'] = (sv2 - sv0).abs().max().item()
res['rms_diff_8pt768_vs_16pt'] = (sv1 - sv0).pow(2).mean().sqrt().item()
res['8pt768_equals_8pt512'] = bool(torch.equal(sv1, sv2))
del sv0, sv1, sv2
full = hp.time_chain_kernel(bref, R, iters=a.iters)
res['k_chain_full_ms'] = full
for rep in range(a.repeat):
    for mode, threads in ((0, 512), (1, 512), (1, 768)):
        ms, _, _ = run(mode, threads, False)
        res['runs'].append({'mapping': '16 points x 6 views' if mode == 0 else '8 points x 2 parities x 3 trips', 'threads': threads,
                            'waves_per_simd': threads // 256, 'ms_per_launch': round(ms, 4), 'ns_per_point_view': round(ms * 1e6 / (a.batch * P * 6), 4)})
print(json.dumps(res, indent=1))
if a.out:
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, 'w'), indent=1)
