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
def test_conv3d_weight_gradient_kernel(B, Cin, Cout, D, H, W, K):
    """gnr_conv3d_bwd_weight against PyTorch's own conv3d backward (ragged channel counts, non-cubic volumes)."""
    from graspnerf_amd.backbone import conv3d_same
    g = torch.Generator().manual_seed(B * 100 + Cin)
    x = torch.randn(B, Cin, D, H, W, generator=g).cuda().requires_grad_(True)
    w = (0.1 * torch.randn(Cout, Cin, K, K, K, generator=g)).cuda().requires_grad_(True)
    b = torch.randn(Cout, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(B, Cout, D, H, W, generator=g).cuda()
    (conv3d_same(x, w, b) * dy).sum().backward()
    got = (x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    (torch.nn.functional.conv3d(x, w, b, padding=K // 2) * dy).sum().backward()
    for a, r, name in zip(got, (x.grad, w.grad, b.grad), ('dx', 'dw', 'db')):
        assert (a - r).abs().max() <= 2e-4 * r.abs().max() + 1e-5, name
