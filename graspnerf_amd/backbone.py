"""PyTorch-ROCm side of the model that stays outside the HIP hot path (north_star: "host code stays
Python on PyTorch-ROCm for the 2D backbone and grasp head"): the residual U-Net feature extractors,
the visibility encoder and the VGN 3D-conv grasp head.  Written from the architecture description
of the reference with IDENTICAL module / parameter names, so a reference checkpoint
(`network_state_dict`) loads with strict=True.

ref: src/nr/network/ops.py:43-230 (blocks, ResUNetLight), init_net.py:8-35, vis_encoder.py:6-22,
     src/gd/networks.py:39-97 (ConvNet).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _InstanceNormFn(torch.autograd.Function):
    """Instance norm (biased variance, eps inside the rsqrt, affine) with a hand-written backward: the same formula as
    autograd's, evaluated in 7 reads + 3 writes of the activation instead of the ~13 + 8 the op-by-op graph of
    var_mean / rsqrt / addcmul costs (the norms of the 2D backbones were ~15 % of a train step's GPU time).
        dx = A dy + B x + C  per (image, channel), with  A = w rstd,  B = -A rstd <dy, xhat> / HW,  C = -A <dy> / HW - B mean."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        var, mean = torch.var_mean(x, dim=(2, 3), unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + eps)
        scale = rstd * weight[None, :, None, None]
        ctx.save_for_backward(x, mean, rstd, weight)
        return torch.addcmul(bias[None, :, None, None] - mean * scale, x, scale)

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, weight = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = dy.contiguous()
        s1 = dy.sum((2, 3), keepdim=True)                                              # <dy>
        s2 = (dy * x).sum((2, 3), keepdim=True)          # <dy, x>  (a batched dot through rocBLAS' bmm took 0.84 ms per layer)
        sxh = rstd * (s2 - mean * s1)                                                  # <dy, xhat>
        a = weight[None, :, None, None] * rstd
        b = -a * rstd * sxh / (h * w)
        dx = torch.addcmul(-a * s1 / (h * w) - b * mean, dy, a)
        dx.addcmul_(x, b)
        return dx, sxh.sum(0).reshape(c), s1.sum(0).reshape(c), None


ACT_NONE, ACT_RELU, ACT_ELU = 0, 1, 2          # include/gnr.h GNR_ACT_*


def _img_lib():
    from . import _lib
    return _lib, _lib.lib()


def _img_check(rc, what):
    if rc:
        _lib, L = _img_lib()
        raise _lib.GnrError(f'{what} failed: {_lib.ERRORS.get(rc, rc)} ({L.gnr_img_last_error().decode(errors="replace")})')


def _stream(t):
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _InstNormActFn(torch.autograd.Function):
    """act(InstanceNorm2d(x) + res) in one pass per direction on the device (csrc/gnr_img.hip: the plane stays in registers
    between the statistics and the apply pass).  The op-by-op statement of the same arithmetic is `_InstanceNormFn` above plus
    the residual add and F.relu / F.elu (kept as the readable form of the backward formula; the product no longer calls it);
    tests/test_backbone_ops.py holds this kernel against ATen's instance norm + add + activation and against float64."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, act, res):
        _, L = _img_lib()
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(n * c, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        weight, bias = weight.contiguous(), bias.contiguous()
        _img_check(L.gnr_instnorm_act(x.data_ptr(), res.data_ptr() if res is not None else None, weight.data_ptr(), bias.data_ptr(),
                                      y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), n * c, c, h * w, eps, act, _stream(x)), 'gnr_instnorm_act')
        ctx.save_for_backward(x, y, mean, rstd, weight)
        ctx.act, ctx.has_res = act, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        _, L = _img_lib()
        x, y, mean, rstd, weight = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[5]) else None
        scratch = torch.empty(2 * n * c, dtype=torch.float32, device=x.device)            # per-plane sums
        wb = torch.empty(2 * c, dtype=torch.float32, device=x.device)                     # d weight, d bias: their own small tensor
        s1, s2, dw, db = scratch[:n * c], scratch[n * c:], wb[:c], wb[c:]                 # (AccumulateGrad may keep the views as .grad)
        _img_check(L.gnr_instnorm_act_bwd(dy.data_ptr(), y.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), weight.data_ptr(),
                                          dx.data_ptr(), dres.data_ptr() if dres is not None else None, s1.data_ptr(), s2.data_ptr(),
                                          dw.data_ptr(), db.data_ptr(), n * c, c, h * w, ctx.act, _stream(x)), 'gnr_instnorm_act_bwd')
        return dx, dw, db, None, None, dres


class _ReflectPadFn(torch.autograd.Function):
    """F.pad(x, (p, p, p, p), mode='reflect'); the backward gathers (<= 9 taps per pixel) instead of scattering atomics."""

    @staticmethod
    def forward(ctx, x, p):
        _, L = _img_lib()
        x = x.contiguous()
        n, c, h, w = x.shape
        y = torch.empty(n, c, h + 2 * p, w + 2 * p, dtype=torch.float32, device=x.device)
        _img_check(L.gnr_reflect_pad2d(x.data_ptr(), y.data_ptr(), n * c, h, w, p, _stream(x)), 'gnr_reflect_pad2d')
        ctx.p, ctx.shape = p, (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        _, L = _img_lib()
        n, c, h, w = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(n, c, h, w, dtype=torch.float32, device=dy.device)
        _img_check(L.gnr_reflect_pad2d_bwd(dy.data_ptr(), dx.data_ptr(), n * c, h, w, ctx.p, _stream(dy)), 'gnr_reflect_pad2d_bwd')
        return dx, None


class _Upsample2xFn(torch.autograd.Function):
    """F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True): ATen's forward kernel runs at ~0.15 TB/s on these
    shapes (1.0 - 2.1 ms per call, 6.3 ms of a training step); its backward is fine and stays."""

    @staticmethod
    def forward(ctx, x):
        _, L = _img_lib()
        x = x.contiguous()
        n, c, h, w = x.shape
        y = torch.empty(n, c, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
        _img_check(L.gnr_upsample2x_bilinear(x.data_ptr(), y.data_ptr(), n * c, h, w, _stream(x)), 'gnr_upsample2x_bilinear')
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        return torch.ops.aten.upsample_bilinear2d_backward(dy.contiguous(), [2 * h, 2 * w], [n, c, h, w], True, None, None)


HIP_GLUE = {'norm': True, 'pad': True, 'upsample': True}      # switches for A/B runs and tests; all on in the product


def _on_device(x, what=None):
    """The HIP glue kernels launch on the CURRENT device with the stream of x's device: take them only when the two agree
    (a model on cuda:1 while cuda:0 is current, or nn.DataParallel replicas, keep the ATen path)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.device.index == torch.cuda.current_device()
            and (what is None or HIP_GLUE[what]))


def upsample2x(x):
    if _on_device(x, 'upsample'):
        return _Upsample2xFn.apply(x)
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)


class _InstanceNorm(nn.InstanceNorm2d):
    """nn.InstanceNorm2d(affine=True, track_running_stats=False) with the same parameters/keys, optionally fused with the
    residual add and the activation that follow it: forward(x, act, res) = act(norm(x) + res).  On the GPU one HIP kernel per
    direction (`_InstNormActFn`; `torch.instance_norm` on ROCm goes through a batch-norm path that blocks the host for
    ~0.23 ms per call, and the op-by-op graph reads the activation 5 times forward and 11 times backward); on the CPU the
    stock module followed by the stock ops."""

    def forward(self, x, act=ACT_NONE, res=None):
        if _on_device(x, 'norm') and x.shape[-1] * x.shape[-2] > 1 and self.weight.dtype == torch.float32 and self.weight.device == x.device:
            return _InstNormActFn.apply(x, self.weight, self.bias, self.eps, act, res)
        y = super().forward(x)
        if res is not None:
            y = y + res
        return F.relu(y) if act == ACT_RELU else (F.elu(y) if act == ACT_ELU else y)


class _ReflectConv2d(nn.Conv2d):
    """nn.Conv2d(padding_mode='reflect') with the padding as one HIP gather in each direction on the GPU (ATen's reflection
    pad backward scatters atomics: 5.1 ms of a training step) and no padding pass at all for the 1x1 convolutions."""

    def forward(self, x):
        p = self.padding[0]
        if _on_device(x, 'pad') and self.padding_mode == 'reflect' and self.padding[1] == p:
            return F.conv2d(_ReflectPadFn.apply(x, p) if p else x, self.weight, self.bias, self.stride, 0, self.dilation, self.groups)
        return super().forward(x)


def _inorm(ch):
    return _InstanceNorm(ch, track_running_stats=False, affine=True)


def _c3(cin, cout, stride=1):
    return _ReflectConv2d(cin, cout, 3, stride, 1, bias=False, padding_mode='reflect')


def _c1(cin, cout, stride=1):
    return _ReflectConv2d(cin, cout, 1, stride, bias=False, padding_mode='reflect')


class BasicBlock(nn.Module):
    """3x3 -> IN -> ReLU -> 3x3 -> IN, + identity (1x1/IN downsample when the shape changes)."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1, self.bn1 = _c3(cin, cout, stride), _inorm(cout)
        self.conv2, self.bn2 = _c3(cout, cout), _inorm(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_c1(cin, cout, stride), _inorm(cout))

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        return self.bn2(self.conv2(self.bn1(self.conv1(x), ACT_RELU)), ACT_RELU, identity)


class conv(nn.Module):       # lower-case name kept: it is part of the checkpoint key path
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.conv = _ReflectConv2d(cin, cout, k, stride, (k - 1) // 2, padding_mode='reflect')
        self.bn = _inorm(cout)

    def forward(self, x):
        return self.bn(self.conv(x), ACT_ELU)


class upconv(nn.Module):
    def __init__(self, cin, cout, k, scale):
        super().__init__()
        self.scale = scale
        self.conv = conv(cin, cout, k, 1)

    def forward(self, x):
        if self.scale == 2:
            return self.conv(upsample2x(x))
        return self.conv(F.interpolate(x, scale_factor=self.scale, mode='bilinear', align_corners=True))


class ResUNetLight(nn.Module):
    """stride-2 stem, three residual stages (1/4, 1/8, 1/16), two up-convolutions with skip
    connections back to 1/4 resolution, 1x1 head.   ref: ops.py:150-230"""

    def __init__(self, in_dim=3, layers=(2, 3, 6, 3), out_dim=32, inplanes=32):
        super().__init__()
        self.conv1 = _ReflectConv2d(in_dim, inplanes, 7, 2, 3, bias=False, padding_mode='reflect')
        self.bn1 = _inorm(inplanes)
        chans, c = (32, 64, 128), inplanes
        stages = []
        for planes, n in zip(chans, layers[:3]):
            blocks = [BasicBlock(c, planes, 2)] + [BasicBlock(planes, planes) for _ in range(n - 1)]
            stages.append(nn.Sequential(*blocks))
            c = planes
        self.layer1, self.layer2, self.layer3 = stages
        self.upconv3 = upconv(128, 64, 3, 2)
        self.iconv3 = conv(128, 64, 3, 1)
        self.upconv2 = upconv(64, 32, 3, 2)
        self.iconv2 = conv(64, 32, 3, 1)
        self.out_conv = nn.Conv2d(32, out_dim, 1, 1)

    @staticmethod
    def _skip(skip, up):
        dy, dx = up.shape[2] - skip.shape[2], up.shape[3] - skip.shape[3]
        skip = F.pad(skip, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        return torch.cat([up, skip], 1)

    def forward(self, x):
        x = self.bn1(self.conv1(x), ACT_RELU)
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        y = self.iconv3(self._skip(x2, self.upconv3(x3)))
        y = self.iconv2(self._skip(x1, self.upconv2(y)))
        return self.out_conv(y)


class ResidualBlock(nn.Module):
    """pre-activation block: IN-ReLU-3x3-IN-ReLU-3x3 + shortcut.   ref: ops.py:43-76"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(_inorm(cin), nn.ReLU(True), _c3(cin, cout), _inorm(cout), nn.ReLU(True), _c3(cout, cout))
        self.short_cut = nn.Conv2d(cin, cout, 1, 1) if cin != cout else None

    def forward(self, x):
        c = self.conv                                 # Sequential kept for the checkpoint keys (conv.0 / .2 / .3 / .5)
        y = c[5](c[3](c[2](c[0](x, ACT_RELU)), ACT_RELU))
        return y + (x if self.short_cut is None else self.short_cut(x))


class CostVolumeInitNet(nn.Module):
    """`init_net_type: cost_volume` is, despite its name, a 2D U-Net + 3 convs.
    The ImageNet mean/std buffers exist in the checkpoint but are unused by forward
    (ref: init_net.py:16-19,33-35)."""

    def __init__(self, cfg=None):
        super().__init__()
        self.register_buffer('imagenet_mean', torch.tensor([0.485, 0.456, 0.406])[None, :, None, None])
        self.register_buffer('imagenet_std', torch.tensor([0.229, 0.224, 0.225])[None, :, None, None])
        self.res_net = ResUNetLight(out_dim=32)
        self.out_conv = nn.Sequential(_c3(32, 32), ResidualBlock(32, 32), _c1(32, 32))

    def forward(self, ref_imgs_info, src_imgs_info=None, is_train=False):
        return self.out_conv(self.res_net(ref_imgs_info['imgs']))


class DefaultVisEncoder(nn.Module):
    """ref: vis_encoder.py:6-22"""

    def __init__(self, cfg=None):
        super().__init__()
        self.out_conv = nn.Sequential(_c3(64, 32), ResidualBlock(32, 32), ResidualBlock(32, 32), _c1(32, 32))

    def forward(self, ray_feats, img_feats):
        return self.out_conv(torch.cat([img_feats, ray_feats], 1))


# ---------------------------------------------------------------------------------------------
# VGN grasp head (consumes the 1 x R^3 volume)          ref: src/gd/networks.py:39-97
# ---------------------------------------------------------------------------------------------
def _c3d(cin, cout, k, stride=1):
    return nn.Conv3d(cin, cout, k, stride=stride, padding=k // 2)


class _Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.conv2, self.conv3 = _c3d(1, 16, 5, 2), _c3d(16, 32, 3, 2), _c3d(32, 64, 3, 2)

    def forward(self, x):
        return F.relu(self.conv3(F.relu(self.conv2(F.relu(self.conv1(x))))))


class _Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.conv2, self.conv3 = _c3d(64, 64, 3), _c3d(64, 32, 3), _c3d(32, 16, 5)

    def forward(self, x):
        # nearest upsampling to the fixed sizes 10 / 20 / 40 (networks.py:88-96).  All three stride-1 convolutions run on the HIP path
        # under autograd (conv3d_same): k5 at 20^3 since round 2 (MIOpen: im2col + GEMM), the two k3 layers at 5^3 / 10^3 since round 4
        # (MIOpen spends 0.3 ms per step of 8 scenes on their im2col / col2im launches; tools/dbg/head_k3_route.py: 4.69 -> 4.38 ms)
        x = F.interpolate(F.relu(conv3d_same(x, self.conv1.weight, self.conv1.bias)), 10)
        x = F.interpolate(F.relu(conv3d_same(x, self.conv2.weight, self.conv2.bias)), 20)
        return F.interpolate(F.relu(conv3d_same(x, self.conv3.weight, self.conv3.bias)), 40)


_CONV_WS = {}


def _hip_conv3d_same(x, w, b, mode):
    """gnr_conv3d_same through the C ABI: mode 0 forward (x [B,Cin,D,H,W] -> y [B,Cout,..] + b), mode 1 backward data
    (x = dy [B,Cout,..] -> dx [B,Cin,..])."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    cout, cin, k = w.shape[0], w.shape[1], w.shape[-1]
    x, w = x.contiguous(), w.contiguous()
    B, _, D, H, W = x.shape
    need = L.gnr_conv3d_same_workspace_bytes(cin, cout, k)
    key = (x.device, need)
    if key not in _CONV_WS:
        _CONV_WS[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
    ws = _CONV_WS[key]
    y = torch.empty(B, cin if mode else cout, D, H, W, dtype=torch.float32, device=x.device)
    rc = L.gnr_conv3d_same(x.data_ptr(), w.data_ptr(), b.contiguous().data_ptr() if (b is not None and not mode) else None, y.data_ptr(),
                           B, cin, cout, D, H, W, k, mode, ws.data_ptr(), ws.numel(), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc:
        raise _lib.GnrError(f'gnr_conv3d_same failed: {_lib.ERRORS.get(rc, rc)} ({L.gnr_head_last_error().decode(errors="replace")})')
    return y


class _Conv3dSame(torch.autograd.Function):
    """F.conv3d(x, w, b, padding=k//2) (k = 3 or 5, fp32) of the grasp head under autograd on the HIP path in all three
    directions: forward and input gradient through gnr_conv3d_same (one implicit-GEMM MFMA kernel, the backward-data pass
    being the same convolution with transposed, flipped weights), weight gradient through gnr_conv3d_same_bwd_weight (LDS-staged, voxel axis as the MFMA K).  MIOpen
    runs these two k5 layers (decoder.conv3 at 20^3, the fused 16 -> 6 heads at 40^3) as im2col + GEMM: ~11 ms per step of
    8 scenes for forward + backward-data, and 75 ms for the weight gradient."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return _hip_conv3d_same(x, w, b, 0)

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from . import _lib
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        k = w.shape[-1]
        dx = _hip_conv3d_same(dy, w, None, 1) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            xc = x.contiguous()
            L = _lib.lib()
            dims = (x.shape[0], x.shape[1], w.shape[0], x.shape[2], x.shape[3], x.shape[4], k)
            need = L.gnr_conv3d_same_bwd_weight_workspace_bytes(*dims)
            key = (x.device, 'wgrad', need)
            if key not in _CONV_WS:
                _CONV_WS[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
            rc = L.gnr_conv3d_same_bwd_weight(xc.data_ptr(), dy.data_ptr(), dw.data_ptr(), *dims, _CONV_WS[key].data_ptr(), need,
                                              C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc:
                raise _lib.GnrError(f'gnr_conv3d_same_bwd_weight failed: {rc}')
        db = dy.sum((0, 2, 3, 4)) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def conv3d_same(x, w, b):
    """Stride-1 same-padding conv3d; on the GPU under autograd all three directions run in HIP (_Conv3dSame)."""
    if x.is_cuda and x.dtype == torch.float32 and w.shape[-1] in (3, 5) and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return _Conv3dSame.apply(x, w, b)
    return F.conv3d(x, w, b, padding=w.shape[-1] // 2)


class ConvNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder, self.decoder = _Encoder(), _Decoder()
        self.conv_qual, self.conv_rot, self.conv_width = _c3d(16, 1, 5), _c3d(16, 4, 5), _c3d(16, 1, 5)

    def forward(self, x):
        f = self.decoder(self.encoder(x))
        # the three heads read the same 16-channel volume: one 16 -> 6 convolution (identical per output channel) so that
        # autograd runs ONE conv3d backward instead of three (MIOpen: 9.4 ms each at 40^3)
        w = torch.cat([self.conv_qual.weight, self.conv_rot.weight, self.conv_width.weight], 0)
        b = torch.cat([self.conv_qual.bias, self.conv_rot.bias, self.conv_width.bias], 0)
        y = conv3d_same(f, w, b)
        return torch.sigmoid(y[:, :1]), F.normalize(y[:, 1:5], dim=1), y[:, 5:6]
