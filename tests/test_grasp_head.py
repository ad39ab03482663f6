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


@pytest.mark.gpu
def test_convnet_under_autograd_matches_pytorch():
    """gd.networks.ConvNet mirror in training mode (HIP convolutions for decoder.conv3 and the fused heads) against the same
    module with every convolution in PyTorch: outputs and all 18 parameter gradients."""
    from graspnerf_amd import backbone
    torch.manual_seed(0)
    net = backbone.ConvNet().cuda()
    vol = torch.randn(2, 1, 40, 40, 40, device='cuda').clamp(-1, 1)
    ups = [torch.randn(2, c, 40, 40, 40, device='cuda') for c in (1, 4, 1)]
    res = {}
    for hip in (True, False):
        net.zero_grad(set_to_none=True)
        orig = backbone.conv3d_same
        if not hip:
            backbone.conv3d_same = lambda x, w, b: torch.nn.functional.conv3d(x, w, b, padding=w.shape[-1] // 2)
        try:
            out = net(vol)
            sum((o * u).sum() for o, u in zip(out, ups)).backward()
        finally:
            backbone.conv3d_same = orig
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
