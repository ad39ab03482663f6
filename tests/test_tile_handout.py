"""k_chain hands its tiles out through per-XCD counters in the workspace (round 6; csrc/gnr_kernels.hip `tile_ctr`): which wavefront
computes a tile must not matter.  The outputs of the counter path are BITWISE those of static round-robin shares (GNR_OPT_STATIC_TILES), on
launches smaller and larger than the machine, call after call on one prepared workspace (the last wavefront of every launch re-arms the
counters: the 16 words behind the status words read zero after every call), and in the training-forward instantiation."""
import numpy as np
import pytest
import torch

from graspnerf_amd import weights, _lib
from graspnerf_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _hot(weights_np):
    from graspnerf_amd.hotpath import HotPath, batch_scenes
    return HotPath(weights.pack_state_dict(weights_np, 'coarse'), weights.pack_state_dict(weights_np, 'fine')), batch_scenes


def _counters(hp, prep):
    scene, keep, ws = prep
    import ctypes as C
    off = hp.L.gnr_status_words_offset(C.byref(scene))
    return ws[off + 256:off + 256 + 64].view(torch.int32)


def _flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, dict):
        return [t for k in sorted(o) for t in _flat(o[k])]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in _flat(x)]
    return []


@pytest.mark.parametrize('cfg,B', [('cfg1', 1), ('cfg1', 3), ('cfg2', 2), ('cfg2', 9)])
def test_counter_hand_out_equals_static_shares_bitwise_and_rearms(cfg, B, weights_np):
    hp, batch_scenes = _hot(weights_np)
    from graspnerf_amd.synth import CONFIGS
    c = CONFIGS[cfg]
    ref, que = batch_scenes([make_scene(i, cfg, with_query_image=False) for i in range(B)])
    ref = {k: torch.from_numpy(v).cuda() for k, v in ref.items()}
    que = {k: torch.from_numpy(v).cuda() for k, v in que.items()}

    def step(n_calls):
        prep = hp.prepare(ref, c['res'], c['rn'], 40)
        outs = []
        for _ in range(n_calls):
            vol = hp.sample_volume(ref, c['res'], prepared=prep)
            ren = hp.render(ref, que, prepared=prep)
            outs.append([t.clone() for t in _flat(vol) + _flat(ren)])
            torch.cuda.synchronize()
            assert int(_counters(hp, prep)[:16].abs().sum()) == 0, 'tile counters not re-armed after a call'
        return outs

    dyn = step(3)                                   # three calls on one prepared workspace
    hp.set_option('static_tiles', True)
    try:
        sta = step(1)
    finally:
        hp.set_option('static_tiles', False)
    assert len(dyn[0]) == len(sta[0]) and len(dyn[0]) >= 10
    for call in dyn:
        for a, b in zip(call, sta[0]):
            assert a.dtype == b.dtype and a.shape == b.shape
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)) if a.dtype == torch.float32 else torch.equal(a, b)


def test_training_forward_is_unchanged_by_the_hand_out(weights_np):
    """The training-forward volume instantiation takes its tiles from the counters too: same volume, same saved states."""
    hp, batch_scenes = _hot(weights_np)
    ref, _ = batch_scenes([make_scene(i, 'cfg2', with_query_image=False) for i in range(2)])
    ref = {k: torch.from_numpy(v).cuda() for k, v in ref.items()}
    prep = hp.prepare(ref, 40)
    vol_d = hp.sample_volume_train(ref, 40, prepared=prep).clone()
    scene = prep[0]
    saves_d = [hp.train_ws_section(n, scene, 40).clone() for n in ('save1', 'save2', 'saveG')]
    torch.cuda.synchronize()
    assert int(_counters(hp, prep)[:16].abs().sum()) == 0
    hp.set_option('static_tiles', True)
    try:
        vol_s = hp.sample_volume_train(ref, 40, prepared=prep).clone()
        saves_s = [hp.train_ws_section(n, scene, 40).clone() for n in ('save1', 'save2', 'saveG')]
    finally:
        hp.set_option('static_tiles', False)
    assert torch.equal(vol_d, vol_s)
    for a, b in zip(saves_d, saves_s):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize('B', [2, 5])
def test_split_launch_option_is_bitwise_the_single_stream_call(B, weights_np):
    """GNR_OPT_SPLIT_LAUNCH (experiment, measured +1 %: kept as an option): the volume call's two half-batches on the caller's stream and the
    library's side stream, joined before the call hands the stream back -- same volume and masks bit for bit, odd batch sizes included, and the
    next call on the caller's stream sees the finished result without any synchronisation of its own."""
    hp, batch_scenes = _hot(weights_np)
    ref, _ = batch_scenes([make_scene(i, 'cfg2', with_query_image=False) for i in range(B)])
    ref = {k: torch.from_numpy(v).cuda() for k, v in ref.items()}
    prep = hp.prepare(ref, 40)
    v0, m0 = hp.sample_volume(ref, 40, want_mask=True, prepared=prep)
    v0, m0 = v0.clone(), m0.clone()
    hp.set_option('split_launch', True)
    try:
        for _ in range(3):
            v1, m1 = hp.sample_volume(ref, 40, want_mask=True, prepared=prep)
            s = v1.sum()                                  # a consumer on the caller's stream right behind the call
        torch.cuda.synchronize()
    finally:
        hp.set_option('split_launch', False)
    assert torch.equal(v0.view(torch.int32), v1.view(torch.int32)) and torch.equal(m0, m1)
    assert float(s) == float(v0.sum())
    assert int(_counters(hp, prep)[:32].abs().sum()) == 0
