"""HIP grasp head (csrc/gnr_head.hip) vs the PyTorch fp32 ConvNet (backbone.ConvNet, itself bit-identical to the
reference's gd.networks.ConvNet on CPU: tests/test_model_mirror.py)."""
import numpy as np
import pytest
import torch

from graspnerf_amd import _lib, grasp_head
from graspnerf_amd.backbone import ConvNet
from graspnerf_amd.synth import synth_state_dict


def _net():
    net = ConvNet().eval()
    syn = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=11)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.items()})
    return net


def test_pack_sizes_and_key_order():
    net = _net()
    blob = grasp_head.canonical_blob(net.state_dict())
    assert blob.size == sum(v.numel() for v in net.state_dict().values()) == _lib.lib().gnr_head_canonical_floats()
    packed = grasp_head.pack(blob)
    assert packed.size == _lib.lib().gnr_head_packed_floats() and np.isfinite(packed).all()
    # first layer keeps canonical order; fused heads: bias block order rot0..3, qual, width
    np.testing.assert_array_equal(packed[:2000], net.state_dict()['encoder.conv1.weight'].numpy().reshape(-1))
    sd = net.state_dict()
    hb = packed[-16:-10]
    np.testing.assert_array_equal(hb, np.concatenate([sd['conv_rot.bias'].numpy(), sd['conv_qual.bias'].numpy(), sd['conv_width.bias'].numpy()]))
    assert _lib.lib().gnr_pack_grasp_head(None, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize('R,B', [(40, 2), (16, 1)])
def test_head_matches_pytorch(R, B):
    net = _net()
    g = torch.Generator().manual_seed(R)
    vol = (torch.rand(B, 1, R, R, R, generator=g) * 2 - 1)
    with torch.no_grad():
        q0, r0, w0 = net(vol)
    head = grasp_head.GraspHead(net.state_dict())
    q, r, w = head(vol.cuda())
    torch.cuda.synchronize()
    assert q.shape == (B, 1, 40, 40, 40) and r.shape == (B, 4, 40, 40, 40) and w.shape == (B, 1, 40, 40, 40)
    np.testing.assert_allclose(q.cpu().numpy(), q0.numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(w.cpu().numpy(), w0.numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(r.cpu().numpy(), r0.numpy(), rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(np.linalg.norm(r.cpu().numpy(), axis=1), 1.0, atol=1e-5)     # unit quaternions


@pytest.mark.gpu
@pytest.mark.parametrize('B,Cin,Cout,D,H,W,K', [(2, 16, 6, 9, 7, 10, 5), (1, 32, 16, 6, 6, 6, 5), (3, 20, 33, 5, 4, 7, 3), (1, 16, 6, 40, 40, 40, 5)])
def test_conv3d_same_all_three_directions(B, Cin, Cout, D, H, W, K):
    """conv3d_same under autograd = gnr_conv3d_same (forward, backward data) + gnr_conv3d_same_bwd_weight against PyTorch's own
    conv3d forward / backward (ragged channel counts incl. several 16-channel chunks, non-cubic volumes that are not
    multiples of the 8x8x4 brick)."""
    from graspnerf_amd.backbone import conv3d_same
    g = torch.Generator().manual_seed(B * 100 + Cin)
    x = torch.randn(B, Cin, D, H, W, generator=g).cuda().requires_grad_(True)
    w = (0.1 * torch.randn(Cout, Cin, K, K, K, generator=g)).cuda().requires_grad_(True)
    b = torch.randn(Cout, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(B, Cout, D, H, W, generator=g).cuda()
    y = conv3d_same(x, w, b)
    (y * dy).sum().backward()
    got = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    y0 = torch.nn.functional.conv3d(x, w, b, padding=K // 2)
    (y0 * dy).sum().backward()
    for a, r, name in zip(got, (y0.detach(), x.grad, w.grad, b.grad), ('y', 'dx', 'dw', 'db')):
        assert a.shape == r.shape and (a - r).abs().max() <= 2e-4 * r.abs().max() + 1e-5, name


def test_folded_upsample_k5_is_the_plain_convolution():
    """backbone.upconv5_x2 = F.conv3d(F.interpolate(u, x2 nearest), w, b, padding=2) in values and in all three gradients (float64 on
    the CPU, ragged sizes: the fold is exact algebra, the zero padding of the upsampled grid included)."""
    from graspnerf_amd import backbone
    import torch.nn.functional as F
    torch.manual_seed(3)
    u = torch.randn(2, 5, 6, 4, 7, dtype=torch.float64, requires_grad=True)
    w = torch.randn(3, 5, 5, 5, 5, dtype=torch.float64, requires_grad=True)
    b = torch.randn(3, dtype=torch.float64, requires_grad=True)
    saved = dict(backbone._FOLD_MATS)
    try:
        backbone._FOLD_MATS.clear()
        backbone._FOLD_MATS[('up', 'cpu')] = backbone._upfold_matrix('cpu').double()
        y = backbone.upconv5_x2(u, w, b)
    finally:
        backbone._FOLD_MATS.clear(); backbone._FOLD_MATS.update(saved)
    r = F.conv3d(F.interpolate(u, scale_factor=2, mode='nearest'), w, b, padding=2)
    assert y.shape == r.shape and float((y - r).abs().max()) < 1e-12
    g = torch.randn_like(r)
    for a, c in zip(torch.autograd.grad((y * g).sum(), (u, w, b)), torch.autograd.grad((r * g).sum(), (u, w, b))):
        assert float((a - c).abs().max()) < 1e-11


@pytest.mark.parametrize('k,ci,co,dims', [(3, 4, 5, (6, 4, 8)), (5, 1, 3, (8, 6, 4)), (5, 2, 3, (4, 4, 4)), (3, 3, 2, (2, 2, 2))])
def test_stride2_as_space_to_depth_is_the_plain_convolution(k, ci, co, dims):
    """backbone.conv3d_stride2 = F.conv3d(x, w, b, stride=2, padding=k // 2) in values and all three gradients (float64, CPU)."""
    from graspnerf_amd import backbone
    import torch.nn.functional as F
    torch.manual_seed(k + ci)
    x = torch.randn(2, ci, *dims, dtype=torch.float64, requires_grad=True)
    w = torch.randn(co, ci, k, k, k, dtype=torch.float64, requires_grad=True)
    b = torch.randn(co, dtype=torch.float64, requires_grad=True)
    saved = dict(backbone._FOLD_MATS)
    try:
        backbone._FOLD_MATS.clear()
        backbone._FOLD_MATS[(k, 'cpu')] = backbone._stride2_matrix(k, 'cpu').double()
        y = backbone.conv3d_stride2(x, w, b)
    finally:
        backbone._FOLD_MATS.clear(); backbone._FOLD_MATS.update(saved)
    r = F.conv3d(x, w, b, stride=2, padding=k // 2)
    assert y.shape == r.shape and float((y - r).abs().max()) < 1e-12
    g = torch.randn_like(r)
    for a, c in zip(torch.autograd.grad((y * g).sum(), (x, w, b)), torch.autograd.grad((r * g).sum(), (x, w, b))):
        assert float((a - c).abs().max()) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize('k,ci,co,n,B', [(5, 1, 16, 40, 2), (3, 16, 32, 20, 2), (3, 32, 64, 10, 3), (3, 20, 24, 6, 1), (5, 3, 5, 8, 2)])
def test_stride2_layers_on_the_hip_path(k, ci, co, n, B):
    """The encoder's stride-2 layers (gd/networks.py:33-37) as space-to-depth + masked k3 convolution (gnr_conv3d_same_masked,
    gnr_conv3d_same_bwd_weight_masked: taps without a weight skipped per 16 x 16 channel block) against F.conv3d(stride=2): outputs
    and all three gradients.  Some weights are set to exactly 0.0: the mask is structure, their gradients must still come out."""
    from graspnerf_amd import backbone
    import torch.nn.functional as F
    torch.manual_seed(k * 100 + ci)
    x = torch.randn(B, ci, n, n, n, device='cuda', requires_grad=True)
    w = (torch.randn(co, ci, k, k, k, device='cuda') * 0.2)
    w[:, :, 0] = 0.0
    w[: co // 2, :, :, 1] = 0.0
    w.requires_grad_(True)
    b = torch.randn(co, device='cuda', requires_grad=True)
    y = backbone.conv3d_stride2(x, w, b)
    r = F.conv3d(x, w, b, stride=2, padding=k // 2)
    assert y.shape == r.shape and float((y - r).detach().abs().max()) <= 2e-5 * float(r.detach().abs().max())
    g = torch.randn_like(r)
    ga = torch.autograd.grad((y * g).sum(), (x, w, b))
    gr = torch.autograd.grad((r * g).sum(), (x, w, b))
    for name, a, c in zip(('dx', 'dw', 'db'), ga, gr):
        assert float((a - c).abs().max()) <= 1e-4 * float(c.abs().max()) + 1e-6, name
    assert float(ga[1][:, :, 0].abs().max()) > 0                      # a gradient where the weight's VALUE is zero


@pytest.mark.gpu
@pytest.mark.parametrize('masked', [False, True])
def test_k3_kernel_generations_agree(masked):
    """The pipelined K = 3 kernels (k_conv3d_s1_k3 / k_conv3d_wgrad_k3: buffer loads, register prefetch) against the first-generation
    ones they replace (kept for volumes beyond 32-bit byte offsets; GNR_CONV3D_FIRST_GEN per call): same products in the same order per
    output, so forward and backward data agree to rounding of the MFMA chain and the weight gradient to 1e-5 of its scale."""
    from graspnerf_amd import backbone, _lib
    L = _lib.lib()
    torch.manual_seed(5)
    if masked:
        x = torch.randn(2, 16, 20, 20, 20, device='cuda', requires_grad=True)
        w = (torch.randn(32, 16, 3, 3, 3, device='cuda') * 0.1).requires_grad_(True)
        f = lambda: backbone.conv3d_stride2(x, w, b)
    else:
        x = torch.randn(2, 40, 9, 10, 11, device='cuda', requires_grad=True)
        w = (torch.randn(24, 40, 3, 3, 3, device='cuda') * 0.1).requires_grad_(True)
        f = lambda: backbone.conv3d_same(x, w, b)
    b = torch.randn(w.shape[0], device='cuda', requires_grad=True)
    res = []
    for gen in (0, 1):
        prev, backbone.CONV3D_FIRST_GEN = backbone.CONV3D_FIRST_GEN, bool(gen)
        try:
            y = f()
            g = torch.autograd.grad((y * torch.cos(y.detach())).sum(), (x, w))
            torch.cuda.synchronize()
        finally:
            backbone.CONV3D_FIRST_GEN = prev
        res.append((y.detach(), g[0], g[1]))
    for a, c in zip(*res):
        assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max()) + 1e-7


@pytest.mark.gpu
def test_convnet_under_autograd_matches_pytorch():
    """gd.networks.ConvNet mirror in training mode (HIP convolutions; decoder.conv3 and the fused heads with their x2 upsampling
    folded into pre-summed k3 weights, backbone.upconv5_x2; the encoder's stride-2 layers as space-to-depth + k3, backbone.conv3d_stride2) against the same module stated plainly in PyTorch (F.interpolate +
    F.conv3d, no folding): outputs and all 18 parameter gradients."""
    from graspnerf_amd import backbone
    torch.manual_seed(0)
    net = backbone.ConvNet().cuda()
    vol = torch.randn(2, 1, 40, 40, 40, device='cuda').clamp(-1, 1)
    ups = [torch.randn(2, c, 40, 40, 40, device='cuda') for c in (1, 4, 1)]
    res = {}
    for hip in (True, False):
        net.zero_grad(set_to_none=True)
        orig, fold, s2d = backbone.conv3d_same, backbone.FOLD_UPSAMPLED_K5, backbone.STRIDE2_AS_S2D
        if not hip:
            backbone.conv3d_same = lambda x, w, b: torch.nn.functional.conv3d(x, w, b, padding=w.shape[-1] // 2)
            backbone.FOLD_UPSAMPLED_K5 = backbone.STRIDE2_AS_S2D = False
        try:
            out = net(vol)
            sum((o * u).sum() for o, u in zip(out, ups)).backward()
        finally:
            backbone.conv3d_same, backbone.FOLD_UPSAMPLED_K5, backbone.STRIDE2_AS_S2D = orig, fold, s2d
        torch.cuda.synchronize()
        res[hip] = ([o.detach().clone() for o in out], {k: p.grad.clone() for k, p in net.named_parameters()})
    for a, r in zip(res[True][0], res[False][0]):
        assert (a - r).abs().max() <= 1e-4 * r.abs().max() + 1e-6
    for k, r in res[False][1].items():
        assert (res[True][1][k] - r).abs().max() <= 5e-4 * r.abs().max() + 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize('B,Cin,Cout,D,H,W,K', [(2, 16, 6, 9, 7, 10, 5), (1, 40, 20, 6, 5, 9, 3)])
def test_first_generation_weight_gradient_kernel(B, Cin, Cout, D, H, W, K):
    """gnr_conv3d_bwd_weight (any odd K; kept in the ABI) against PyTorch."""
    import ctypes as C
    from graspnerf_amd import _lib
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, D, H, W, generator=g).cuda()
    dy = torch.randn(B, Cout, D, H, W, generator=g).cuda()
    dw = torch.zeros(Cout, Cin, K, K, K, device='cuda')
    assert _lib.lib().gnr_conv3d_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, Cin, Cout, D, H, W, K, None) == 0
    want = torch.nn.grad.conv3d_weight(x, dw.shape, dy, padding=K // 2)
    torch.cuda.synchronize()
    assert (dw - want).abs().max() <= 2e-4 * want.abs().max() + 1e-5
