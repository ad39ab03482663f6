"""Reference state-dict keys -> canonical blob -> packed MFMA-fragment blob (via libgnr.so).

Key names are the reference's (checkpoint compatible): a level is
(`dist_decoder.`, `agg_net.`) or (`fine_dist_decoder.`, `fine_agg_net.`), optionally under a
module prefix such as `nr_net.` (ref: renderer.py:293-303 wraps the renderer as `nr_net`).
"""
import ctypes as C

import numpy as np

from . import _lib

# (suffix, shape) in reference state-dict order  (ref: dist_decoder.py:64-88,
# aggregate_net.py:29-33, ibrnet.py:382-423, neus.py:9)
DEC_KEYS = []
for br, nout in (('mean_decoder', 2), ('var_decoder', 2), ('aw_decoder', 1)):
    DEC_KEYS += [(f'{br}.0.weight', (32, 32)), (f'{br}.0.bias', (32,)),
                 (f'{br}.2.weight', (32, 32)), (f'{br}.2.bias', (32,)),
                 (f'{br}.4.weight', (nout, 32)), (f'{br}.4.bias', (nout,))]
AGG_KEYS = [
    ('prob_embed.0.weight', (32, 34)), ('prob_embed.0.bias', (32,)),
    ('prob_embed.2.weight', (32, 32)), ('prob_embed.2.bias', (32,)),
    ('agg_impl.ray_dir_fc.0.weight', (16, 4)), ('agg_impl.ray_dir_fc.0.bias', (16,)),
    ('agg_impl.ray_dir_fc.2.weight', (35, 16)), ('agg_impl.ray_dir_fc.2.bias', (35,)),
    ('agg_impl.base_fc.0.weight', (64, 207)), ('agg_impl.base_fc.0.bias', (64,)),
    ('agg_impl.base_fc.2.weight', (32, 64)), ('agg_impl.base_fc.2.bias', (32,)),
    ('agg_impl.vis_fc.0.weight', (32, 32)), ('agg_impl.vis_fc.0.bias', (32,)),
    ('agg_impl.vis_fc.2.weight', (33, 32)), ('agg_impl.vis_fc.2.bias', (33,)),
    ('agg_impl.vis_fc2.0.weight', (32, 32)), ('agg_impl.vis_fc2.0.bias', (32,)),
    ('agg_impl.vis_fc2.2.weight', (1, 32)), ('agg_impl.vis_fc2.2.bias', (1,)),
    ('agg_impl.geometry_fc.0.weight', (64, 86)), ('agg_impl.geometry_fc.0.bias', (64,)),
    ('agg_impl.geometry_fc.2.weight', (16, 64)), ('agg_impl.geometry_fc.2.bias', (16,)),
    ('agg_impl.ray_attention.w_qs.weight', (16, 16)), ('agg_impl.ray_attention.w_ks.weight', (16, 16)),
    ('agg_impl.ray_attention.w_vs.weight', (16, 16)), ('agg_impl.ray_attention.fc.weight', (16, 16)),
    ('agg_impl.ray_attention.layer_norm.weight', (16,)), ('agg_impl.ray_attention.layer_norm.bias', (16,)),
    ('agg_impl.out_geometry_fc.0.weight', (16, 16)), ('agg_impl.out_geometry_fc.0.bias', (16,)),
    ('agg_impl.out_geometry_fc.1.weight', (1, 16)), ('agg_impl.out_geometry_fc.1.bias', (1,)),
    ('agg_impl.rgb_fc.0.weight', (16, 37)), ('agg_impl.rgb_fc.0.bias', (16,)),
    ('agg_impl.rgb_fc.2.weight', (8, 16)), ('agg_impl.rgb_fc.2.bias', (8,)),
    ('agg_impl.rgb_fc.4.weight', (1, 8)), ('agg_impl.rgb_fc.4.bias', (1,)),
    ('agg_impl.neuray_fc.0.weight', (8, 32)), ('agg_impl.neuray_fc.0.bias', (8,)),
    ('agg_impl.neuray_fc.2.weight', (1, 8)), ('agg_impl.neuray_fc.2.bias', (1,)),
    ('deviation_network.variance', ()),
]
VIS_KEYS = [('vis_decoder.0.weight', (32, 32)), ('vis_decoder.0.bias', (32,)), ('vis_decoder.2.weight', (32, 32)),
            ('vis_decoder.2.bias', (32,)), ('vis_decoder.4.weight', (1, 32)), ('vis_decoder.4.bias', (1,))]
LEVELS = {'coarse': ('dist_decoder.', 'agg_net.'), 'fine': ('fine_dist_decoder.', 'fine_agg_net.')}


def level_keys(level, prefix='', use_vis=False):
    """State-dict keys of a level in canonical order; use_vis: the vis_decoder's six tensors BEHIND them (the order of the
    gradient blobs of a use_vis level, include/gnr.h)."""
    dec, agg = LEVELS[level]
    keys = [(prefix + dec + k, s) for k, s in DEC_KEYS] + [(prefix + agg + k, s) for k, s in AGG_KEYS]
    return keys + ([(prefix + dec + k, s) for k, s in VIS_KEYS] if use_vis else [])


def canonical_blob(state_dict, level, prefix=''):
    """Concatenate one level's parameters in reference order -> float32 numpy [36958]."""
    parts = []
    for k, shape in level_keys(level, prefix):
        if k not in state_dict:
            raise KeyError(f'missing parameter {k!r}')
        v = state_dict[k]
        v = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f'{k}: expected shape {shape}, got {tuple(v.shape)}')
        parts.append(np.asarray(v, np.float32).reshape(-1))
    blob = np.concatenate(parts)
    assert blob.size == _lib.lib().gnr_canonical_weights_floats()
    return blob


def pack(canonical):
    """canonical float32 [36958] -> packed float32 blob (host numpy)."""
    L = _lib.lib()
    canonical = np.ascontiguousarray(canonical, np.float32)
    if canonical.size != L.gnr_canonical_weights_floats():
        raise ValueError('canonical blob has the wrong size')
    out = np.zeros(L.gnr_packed_weights_floats(), np.float32)
    _lib.check(L.gnr_pack_weights(canonical.ctypes.data_as(_lib.c_float_p), out.ctypes.data_as(_lib.c_float_p)),
               'gnr_pack_weights')
    return out


def has_vis_decoder(state_dict, level, prefix=''):
    """cfg `use_vis: true` (dist_decoder.py:89-97): the level's decoder carries a fourth branch."""
    return prefix + LEVELS[level][0] + VIS_KEYS[0][0] in state_dict


def vis_blob(state_dict, level, prefix=''):
    """The vis_decoder's six tensors of a level in state-dict order -> float32 numpy [2145]."""
    parts = []
    for k, shape in VIS_KEYS:
        v = state_dict[prefix + LEVELS[level][0] + k]
        v = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f'{k}: expected shape {shape}, got {tuple(v.shape)}')
        parts.append(np.asarray(v, np.float32).reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


def pack_state_dict(state_dict, level, prefix=''):
    """One level's packed blob from a state dict; a `vis_decoder` (use_vis) is packed behind it when the state dict has one."""
    out = pack(canonical_blob(state_dict, level, prefix))
    if has_vis_decoder(state_dict, level, prefix):
        vis = vis_blob(state_dict, level, prefix)
        L = _lib.lib()
        _lib.check(L.gnr_pack_vis_decoder(vis.ctypes.data_as(_lib.c_float_p), out.ctypes.data_as(_lib.c_float_p)), 'gnr_pack_vis_decoder')
    return out


def pack_bwd(canonical, vis=None):
    """canonical float32 [36958] (+ the level's vis_blob when use_vis) -> transposed-fragment blob for the backward twins (host numpy)."""
    L = _lib.lib()
    canonical = np.ascontiguousarray(canonical, np.float32)
    out = np.zeros(L.gnr_packed_bwd_floats(), np.float32)
    _lib.check(L.gnr_pack_weights_bwd(canonical.ctypes.data_as(_lib.c_float_p), out.ctypes.data_as(_lib.c_float_p)),
               'gnr_pack_weights_bwd')
    if vis is not None:
        vis = np.ascontiguousarray(vis, np.float32)
        _lib.check(L.gnr_pack_vis_decoder_bwd(vis.ctypes.data_as(_lib.c_float_p), out.ctypes.data_as(_lib.c_float_p)), 'gnr_pack_vis_decoder_bwd')
    return out


def split_canonical(flat, level, prefix='', use_vis=False):
    """A canonical-layout array (e.g. a gradient blob from a *_bwd entry point) -> {state-dict key: view}."""
    out, off = {}, 0
    for k, shape in level_keys(level, prefix, use_vis):
        n = int(np.prod(shape)) if len(shape) else 1
        out[k] = flat[off:off + n].reshape(shape)
        off += n
    return out


def canonical_blob_device(params, level, prefix='', as_tensor=False):
    """Same blob from live torch tensors with ONE device->host copy (the per-key path costs a sync per tensor);
    as_tensor=True returns the flat device tensor instead (the backward twins read the canonical weights on the device)."""
    import torch
    flat = torch.cat([params[k].detach().reshape(-1).to(torch.float32) for k, _ in level_keys(level, prefix)])
    assert flat.numel() == _lib.lib().gnr_canonical_weights_floats()
    return flat if as_tensor else flat.cpu().numpy()
