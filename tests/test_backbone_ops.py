"""Image-side streaming ops of the 2D feature extractors (csrc/gnr_img.hip, include/gnr.h) against the PyTorch ops the reference
runs in their place: nn.InstanceNorm2d + residual add + ReLU / ELU (src/nr/network/ops.py:101-121,135-138,215),
nn.Conv2d(padding_mode='reflect')'s F.pad (ops.py:8,134,163) and F.interpolate(scale_factor=2, bilinear, align_corners=True)
(ops.py:147).  Floating point: the norm's statistics are summed in another order than ATen's (tolerance 2e-6 relative to the
tensor's scale on values, 2e-5 on gradients); the padding is a copy (bit-exact, its backward sums <= 9 values); the upsampling
repeats ATen's arithmetic, in which the interpolation weight `scale * dst - floor(.)` itself carries eps * dst of rounding
(contracted into one fma or not): 2e-5."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from graspnerf_amd import _lib, backbone

# plane sizes of the benchmark's feature extractors (288 x 512 images) + shapes that take the other kernels
PLANES = [(2, 32, 144, 256), (2, 32, 72, 128), (2, 64, 36, 64), (2, 128, 18, 32),      # <1024,9>, <256,9>, <256,3>, <64,3>
          (3, 5, 7, 5), (2, 3, 200, 204), (1, 4, 6, 6)]                                # odd size, larger than the register form, tiny


def _ref_norm(x, w, b, act, res):
    y = F.instance_norm(x, weight=w, bias=b, eps=1e-5)
    if res is not None:
        y = y + res
    return F.relu(y) if act == backbone.ACT_RELU else (F.elu(y) if act == backbone.ACT_ELU else y)


def _close(got, want, tol, what):
    scale = float(want.abs().max()) + 1e-12
    err = float((got - want).abs().max()) / scale
    assert err <= tol, f'{what}: max error {err:.3e} of the tensor scale (tolerance {tol:.1e})'


@pytest.mark.gpu
@pytest.mark.parametrize('shape', PLANES)
@pytest.mark.parametrize('act', [backbone.ACT_NONE, backbone.ACT_RELU, backbone.ACT_ELU])
@pytest.mark.parametrize('with_res', [False, True])
def test_instnorm_act_forward_and_backward(shape, act, with_res):
    g = torch.Generator().manual_seed(hash((shape, act, with_res)) % 1000)
    x = (torch.randn(shape, generator=g) * 2.0 + 0.5).cuda().requires_grad_(True)
    w = (torch.rand(shape[1], generator=g) + 0.5).cuda().requires_grad_(True)
    b = torch.randn(shape[1], generator=g).cuda().requires_grad_(True)
    res = torch.randn(shape, generator=g).cuda().requires_grad_(True) if with_res else None
    dy = torch.randn(shape, generator=g).cuda()
    y = backbone._InstNormActFn.apply(x, w, b, 1e-5, act, res)
    ins = [x, w, b] + ([res] if with_res else [])
    got = torch.autograd.grad(y, ins, dy)
    want_y = _ref_norm(x, w, b, act, res)
    want = torch.autograd.grad(want_y, ins, dy)
    _close(y.detach(), want_y.detach(), 2e-6, 'y')
    for name, u, v in zip(['dx', 'dweight', 'dbias', 'dres'], got, want):
        _close(u, v, 2e-5, name)


@pytest.mark.gpu
def test_instnorm_constant_plane_and_unaligned_views():
    """var = 0 (rstd = 1/sqrt(eps)) and inputs that are not 16-byte aligned (the scalar kernels)."""
    x = torch.full((1, 2, 8, 8), 3.0, device='cuda')
    w, b = torch.ones(2, device='cuda'), torch.tensor([0.5, -0.5], device='cuda')
    y = backbone._InstNormActFn.apply(x, w, b, 1e-5, backbone.ACT_NONE, None)
    assert torch.equal(y, b[None, :, None, None].expand_as(y).contiguous())
    buf = torch.randn(1 + 2 * 3 * 8 * 8, device='cuda')
    xv = buf[1:].view(2, 3, 8, 8)                                                      # contiguous, 4-byte aligned only
    w3, b3 = torch.rand(3, device='cuda') + 0.5, torch.randn(3, device='cuda')
    _close(backbone._InstNormActFn.apply(xv, w3, b3, 1e-5, backbone.ACT_RELU, None), _ref_norm(xv, w3, b3, backbone.ACT_RELU, None), 2e-6, 'unaligned')


@pytest.mark.gpu
@pytest.mark.parametrize('shape,pad', [((2, 3, 144, 256), 1), ((2, 3, 288, 512), 3), ((1, 2, 5, 7), 1), ((1, 2, 4, 4), 3), ((3, 1, 2, 2), 1),
                                       ((1, 1, 9, 300), 2)])
def test_reflect_pad_forward_exact_and_backward(shape, pad):
    g = torch.Generator().manual_seed(pad * 100 + shape[2])
    x = torch.randn(shape, generator=g).cuda().requires_grad_(True)
    y = backbone._ReflectPadFn.apply(x, pad)
    want = F.pad(x, (pad,) * 4, mode='reflect')
    assert torch.equal(y, want)
    dy = torch.randn(want.shape, generator=g).cuda()
    _close(torch.autograd.grad(y, x, dy)[0], torch.autograd.grad(want, x, dy)[0], 1e-6, 'dx')


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 128, 18, 32), (2, 64, 36, 64), (1, 3, 5, 7), (1, 2, 1, 1), (1, 2, 3, 1)])
def test_upsample2x_matches_aten(shape):
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randn(shape, generator=g).cuda().requires_grad_(True)
    y = backbone.upsample2x(x)
    want = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    _close(y.detach(), want.detach(), 2e-5, 'y')
    dy = torch.randn(want.shape, generator=g).cuda()
    _close(torch.autograd.grad(y, x, dy)[0], torch.autograd.grad(want, x, dy)[0], 1e-6, 'dx')


@pytest.mark.gpu
def test_feature_extractor_with_the_hip_glue_is_no_further_from_float64_than_with_atens_ops():
    """tests/backbone_fp64_check.py in its own process (MIOpen's Winograd solvers off, see there): ResUNetLight + the init-net head,
    features and all parameter gradients, HIP glue vs ATen's ops vs float64."""
    import json, os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MIOPEN_DEBUG_CONV_WINOGRAD='0')
    r = subprocess.run([sys.executable, os.path.join(here, 'backbone_fp64_check.py')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    hip, aten = res['hip'], res['aten']
    assert hip['features'] <= 3 * aten['features'] + 1e-6, (hip['features'], aten['features'])
    assert hip['grads'].keys() == aten['grads'].keys() and len(hip['grads']) > 60
    for k, r1 in hip['grads'].items():
        assert r1 <= 3 * aten['grads'][k] + 1e-5, (k, r1, aten['grads'][k])
    assert max(hip['grads'].values()) < 1e-4             # measured 1.2e-5 (ATen's ops: 4.2e-4)


def test_cpu_path_of_the_fused_modules_is_the_stock_composition():
    torch.manual_seed(0)
    x, res = torch.randn(2, 4, 6, 10), torch.randn(2, 4, 6, 10)
    m = backbone._inorm(4)
    with torch.no_grad():
        m.weight.uniform_(0.5, 1.5); m.bias.normal_()
    stock = torch.nn.InstanceNorm2d(4, affine=True, track_running_stats=False)
    stock.load_state_dict(m.state_dict())
    assert torch.equal(m(x, backbone.ACT_RELU, res), F.relu(stock(x) + res))
    assert torch.equal(m(x, backbone.ACT_ELU), F.elu(stock(x)))
    assert torch.equal(m(x), stock(x))
    conv = backbone._c3(4, 5)
    ref = torch.nn.Conv2d(4, 5, 3, 1, 1, bias=False, padding_mode='reflect')
    ref.load_state_dict(conv.state_dict())
    assert torch.equal(conv(x), ref(x))
    assert torch.equal(backbone.upsample2x(x), F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True))


def test_image_entry_points_check_their_arguments_without_a_device():
    L = _lib.lib()
    one = C.c_void_p(16)                                  # never dereferenced: the argument checks come first
    assert L.gnr_instnorm_act(None, None, one, one, one, one, one, 4, 2, 16, 1e-5, 0, None) == _lib.GNR_ERR_ARG
    assert L.gnr_instnorm_act(one, None, one, one, one, one, one, 5, 2, 16, 1e-5, 0, None) == _lib.GNR_ERR_SHAPE      # planes % C
    assert L.gnr_instnorm_act(one, None, one, one, one, one, one, 4, 2, 16, 1e-5, 3, None) == _lib.GNR_ERR_SHAPE      # act
    assert b'act' in L.gnr_img_last_error()
    assert L.gnr_instnorm_act(one, None, one, one, one, one, one, 0, 2, 16, 1e-5, 0, None) == _lib.GNR_OK             # empty batch
    assert L.gnr_instnorm_act_bwd(one, None, one, one, one, one, one, None, one, one, one, one, 4, 2, 16, 1, None) == _lib.GNR_ERR_ARG   # act needs out
    assert L.gnr_instnorm_act_bwd(one, one, one, one, one, one, one, None, one, one, one, one, 4, 0, 16, 1, None) == _lib.GNR_ERR_SHAPE
    assert L.gnr_reflect_pad2d(one, one, 2, 4, 4, 4, None) == _lib.GNR_ERR_SHAPE                                       # pad < size
    assert L.gnr_reflect_pad2d_bwd(one, None, 2, 4, 4, 1, None) == _lib.GNR_ERR_ARG
    assert L.gnr_reflect_pad2d(one, one, 0, 4, 4, 1, None) == _lib.GNR_OK
    assert L.gnr_upsample2x_bilinear(one, one, 2, 0, 4, None) == _lib.GNR_ERR_SHAPE
    assert L.gnr_upsample2x_bilinear(None, one, 2, 4, 4, None) == _lib.GNR_ERR_ARG
    assert L.gnr_upsample2x_bilinear(one, one, 0, 4, 4, None) == _lib.GNR_OK
