"""Device-side weight packer (csrc/gnr_pack_dev.hip, include/gnr.h gnr_pack_*_device) against the host packer (pytest -m gpu).

The training loop re-packs the hot path's parameters after every optimiser step.  Rounds 1-4 did that on the host (parameters ->
pinned host memory -> gnr_pack_weights -> upload, with one wait per step); the reference keeps its parameters on the device
(train/trainer.py:146-158).  The device packer runs the host packer's own source (csrc/gnr_pack_body.h) with grid-strided loops
and must produce the SAME BITS: forward blob (fp32 fragments, tables, the fp16-pair image), backward blob, the use_vis branch, and
the "a weight has no fp16 pair" flag -- set and cleared again."""
import ctypes as C

import numpy as np
import pytest
import torch

from graspnerf_amd import weights, _lib

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bits(t):
    return t.cpu().numpy().view(np.uint32)


def _random_canonical(seed, scale=0.3):
    rng = np.random.default_rng(seed)
    n = _lib.lib().gnr_canonical_weights_floats()
    # magnitudes over ten decades (fp16 subnormal residuals, large weights) on top of a normal body
    c = rng.standard_normal(n) * scale * np.exp(rng.uniform(-6, 3, n) * (rng.uniform(size=n) < 0.1))
    return c.astype(np.float32)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_device_packs_equal_host_packs_bitwise(seed, weights_np):
    L = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    can_a = weights.canonical_blob(weights_np, 'coarse' if seed % 2 == 0 else 'fine')
    can_b = _random_canonical(seed)
    fwd = _dev(weights.pack(can_a))                    # the blob exists (host pack, weights A) ...
    bwd = _dev(weights.pack_bwd(can_a))
    cb = _dev(can_b)
    _lib.check(L.gnr_pack_weights_device(cb.data_ptr(), fwd.data_ptr(), st), 'gnr_pack_weights_device')      # ... and is re-packed with weights B
    _lib.check(L.gnr_pack_weights_bwd_device(cb.data_ptr(), bwd.data_ptr(), st), 'gnr_pack_weights_bwd_device')
    torch.cuda.synchronize()
    want_f, want_b = weights.pack(can_b), weights.pack_bwd(can_b)
    diff = np.nonzero(_bits(fwd) != want_f.view(np.uint32))[0]
    assert diff.size == 0, f'forward blob: {diff.size} words differ, first at {diff[:8]}'
    diff = np.nonzero(_bits(bwd) != want_b.view(np.uint32))[0]
    assert diff.size == 0, f'backward blob: {diff.size} words differ, first at {diff[:8]}'


def test_device_pack_of_the_use_vis_branch(weights_np):
    L = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(11)
    can_a, can_b = weights.canonical_blob(weights_np, 'coarse'), _random_canonical(7)
    vis_a, vis_b = (rng.standard_normal(2145) * 0.3).astype(np.float32), (rng.standard_normal(2145) * 0.3).astype(np.float32)

    def host(can, vis):
        f = weights.pack(can)
        _lib.check(L.gnr_pack_vis_decoder(vis.ctypes.data_as(_lib.c_float_p), f.ctypes.data_as(_lib.c_float_p)), 'gnr_pack_vis_decoder')
        return f, weights.pack_bwd(can, vis)
    fa, ba = host(can_a, vis_a)
    fwd, bwd, cb, vb = _dev(fa), _dev(ba), _dev(can_b), _dev(vis_b)
    _lib.check(L.gnr_pack_weights_device(cb.data_ptr(), fwd.data_ptr(), st), 'gnr_pack_weights_device')
    _lib.check(L.gnr_pack_vis_decoder_device(vb.data_ptr(), fwd.data_ptr(), st), 'gnr_pack_vis_decoder_device')
    _lib.check(L.gnr_pack_weights_bwd_device(cb.data_ptr(), bwd.data_ptr(), st), 'gnr_pack_weights_bwd_device')
    _lib.check(L.gnr_pack_vis_decoder_bwd_device(vb.data_ptr(), bwd.data_ptr(), st), 'gnr_pack_vis_decoder_bwd_device')
    torch.cuda.synchronize()
    want_f, want_b = host(can_b, vis_b)
    assert np.array_equal(_bits(fwd), want_f.view(np.uint32))
    assert np.array_equal(_bits(bwd), want_b.view(np.uint32))


def test_a_weight_without_an_fp16_pair_is_flagged_and_the_flag_clears(weights_np):
    L = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    can = weights.canonical_blob(weights_np, 'coarse')
    big = can.copy()
    big[5] = 1e5                                      # mean_decoder.0.weight[0, 5]: beyond the fp16 range
    fwd = _dev(weights.pack(can))
    flag, flag16 = L.gnr_layout_offset(b'T_VIS') + 2, L.gnr_layout_offset(b'C16.T_VIS') + 2
    _lib.check(L.gnr_pack_weights_device(_dev(big).data_ptr(), fwd.data_ptr(), st), 'gnr_pack_weights_device')
    torch.cuda.synchronize()
    want = weights.pack(big)
    assert want[flag] == 1.0 and want[flag16] == 1.0
    assert np.array_equal(_bits(fwd), want.view(np.uint32))
    _lib.check(L.gnr_pack_weights_device(_dev(can).data_ptr(), fwd.data_ptr(), st), 'gnr_pack_weights_device')
    torch.cuda.synchronize()
    assert np.array_equal(_bits(fwd), weights.pack(can).view(np.uint32))


def test_null_pointers_are_refused():
    L = _lib.lib()
    for fn in (L.gnr_pack_weights_device, L.gnr_pack_weights_bwd_device, L.gnr_pack_vis_decoder_device, L.gnr_pack_vis_decoder_bwd_device):
        assert fn(None, None, None) == _lib.GNR_ERR_ARG
