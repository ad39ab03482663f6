"""GPU bring-up report: run the HIP path on synthetic scenes and print per-output / per-stage
errors against the CPU oracle and the golden fixtures.  Usage (on the GPU box):
    python tools/gpu_check.py [cfg1|cfg2] [--no-render]
Everything printed here is also asserted (with tolerances) by tests/test_gpu_parity.py."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from graspnerf_amd import weights                                   # noqa: E402
from graspnerf_amd.hotpath import HotPath, batch_scenes              # noqa: E402
from graspnerf_amd.synth import make_scene, CONFIGS                  # noqa: E402
from oracle import graspnerf_oracle as O                             # noqa: E402


def stats(name, a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    rel = d / (np.abs(b) + 1e-3)
    print(f'  {name:28s} max|d|={d.max():.3e}  mean|d|={d.mean():.3e}  maxrel={rel.max():.3e}  ref|max|={np.abs(b).max():.3e}'
          f'  nan={int(np.isnan(a).sum())}')


def main():
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'cfg1'
    do_render = '--no-render' not in sys.argv
    c = CONFIGS[name]
    res, dn = c['res'], (16 if name == 'cfg1' else 40)
    Wnp = dict(np.load('tests/golden/weights_seed0.npz'))
    G = dict(np.load(f'tests/golden/golden_{name}.npz'))
    W = {k: torch.from_numpy(v) for k, v in Wnp.items()}
    hp = HotPath(weights.pack_state_dict(Wnp, 'coarse'), weights.pack_state_dict(Wnp, 'fine'))
    ref, que = make_scene(0, name)
    bref, bque = batch_scenes([(ref, que)])
    inp, q = O.to_torch(ref), O.to_torch(que)

    print(f'== {name}: volume')
    t = time.time()
    dbg_o = {}
    vol_o = O.sample_volume(W, inp, res, debug=dbg_o).numpy()
    print(f'  oracle {time.time() - t:.2f}s')
    prep = hp.prepare(bref, res, c['rn'], dn)
    dbg = hp.debug_volume_chain(bref, res, prepared=prep).cpu().numpy()[0]
    V = c['V']
    stats('hit', dbg[:, 0:V].T, dbg_o['hit'].numpy())
    stats('vis', dbg[:, 8:8 + V].T, dbg_o['vis'].numpy())
    stats('v2', dbg[:, 16:16 + V].T, dbg_o['v2'].numpy())
    stats('n_valid', dbg[:, 24], dbg_o['msum'].numpy())
    stats('mean0[rgb r]', dbg[:, 28], dbg_o['mean0_rgb0'].numpy())
    stats('hoisted pre f0', dbg[:, 29], dbg_o['pre_f0'].numpy())
    stats('sum v2', dbg[:, 31], dbg_o['vsum'].numpy())
    stats('wbar', dbg[:, 25], dbg_o['wbar'].numpy())
    stats('mean f0', dbg[:, 26], dbg_o['mean_f0'].numpy())
    stats('var f0', dbg[:, 27], dbg_o['var_f0'].numpy())
    stats('geometry f0', dbg[:, 30], dbg_o['g_f0'].numpy())
    vol, vm = hp.sample_volume(bref, res, want_mask=True, prepared=prep)
    torch.cuda.synchronize()
    vol = vol.cpu().numpy()[0]
    stats('volume vs oracle', vol, vol_o[0])
    stats('volume vs golden', vol, G['volume'][0])
    gm = np.unpackbits(G['volume_mask_bits']).reshape(V, res * res, res).astype(bool)
    mine = vm.cpu().numpy()[0]
    ok = True
    for v in range(V):
        mv = ((mine >> v) & 1).astype(bool).reshape(res * res, res)[:, ::-1]
        ok &= np.array_equal(mv, gm[v])
    print('  view masks bit-exact vs reference:', ok)

    if not do_render:
        return
    print(f'== {name}: render')
    cfg = {'depth_sample_num': dn, 'fine_depth_sample_num': dn}
    dbo = {}
    out_o = O.render(W, inp, q, cfg, debug=dbo, fine_depth_override=torch.from_numpy(G['fine_depth_sorted']))
    co, fi, inds = hp.render(bref, bque, cfg, fine_depth_in=G['fine_depth_sorted'][None], debug=True, prepared=prep)
    torch.cuda.synchronize()
    stats('coarse depth', co['depth'].cpu().numpy()[0], dbo['coarse_depth'].numpy())
    stats('coarse grad', co['sdf_gradient'].cpu().numpy()[0], dbo['coarse']['grad'].numpy())
    for k in ['sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr', 'pixel_colors_gt', 'render_depth',
              'sdf_gradient_error']:
        stats('coarse ' + k, co[k].cpu().numpy(), G['render.' + k].reshape(co[k].shape))
    print('  coarse ray_mask equal:', np.array_equal(co['ray_mask'].cpu().numpy(), G['render.ray_mask']))
    ii = inds.cpu().numpy()[0]
    print('  fine inds mismatches:', int((ii != G['fine_inds']).sum()), 'of', ii.size)
    stats('fine grad', fi['sdf_gradient'].cpu().numpy()[0], dbo['fine']['grad'].numpy())
    for k in ['sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr', 'render_depth', 'sdf_gradient_error']:
        stats('fine ' + k, fi[k].cpu().numpy(), G['render.' + k + '_fine'].reshape(fi[k].shape))
    print('  fine ray_mask equal:', np.array_equal(fi['ray_mask'].cpu().numpy(), G['render.ray_mask_fine']))
    # free-running fine depths
    co2, fi2 = hp.render(bref, bque, cfg, prepared=prep)
    torch.cuda.synchronize()
    stats('fine depth (free) vs ref', fi2['depth'].cpu().numpy()[0], G['fine_depth_sorted'])


if __name__ == '__main__':
    main()
