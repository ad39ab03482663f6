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
        for conv in (self.conv1, self.conv2, self.conv3):
            if _s2d_route(x, conv.weight):
                x = F.relu(conv3d_stride2(x, conv.weight, conv.bias))
            else:
                x = F.relu(conv(x))
        return x


class _Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.conv2, self.conv3 = _c3d(64, 64, 3), _c3d(64, 32, 3), _c3d(32, 16, 5)

    def forward(self, x, upsample_last=True):
        # nearest upsampling to the fixed sizes 10 / 20 / 40 (networks.py:88-96).  All three stride-1 convolutions run on the HIP path
        # under autograd (conv3d_same): k5 at 20^3 since round 2 (MIOpen: im2col + GEMM), the two k3 layers at 5^3 / 10^3 since round 4
        # (MIOpen spends 0.3 ms per step of 8 scenes on their im2col / col2im launches; tools/dbg/head_k3_route.py: 4.69 -> 4.38 ms)
        x = F.interpolate(F.relu(conv3d_same(x, self.conv1.weight, self.conv1.bias)), 10)
        x = F.relu(conv3d_same(x, self.conv2.weight, self.conv2.bias))
        if _fold_route(x, self.conv3.weight) and tuple(x.shape[2:]) == (10, 10, 10):
            x = F.relu(upconv5_x2(x, self.conv3.weight, self.conv3.bias))          # = conv3(F.interpolate(x, 20)), the 20^3 x 32 tensor never built
        else:
            x = F.relu(conv3d_same(F.interpolate(x, 20), self.conv3.weight, self.conv3.bias))
        return F.interpolate(x, 40) if upsample_last else x


FOLD_UPSAMPLED_K5 = True            # tests switch it off to get the plain statement of the same layers


def _fold_route(x, w):
    """The folded form is taken where conv3d_same would run its HIP kernels (GPU, fp32, under autograd)."""
    return (FOLD_UPSAMPLED_K5 and x.is_cuda and x.dtype == torch.float32 and w.shape[-1] == 5 and torch.is_grad_enabled()
            and (x.requires_grad or w.requires_grad))


STRIDE2_AS_S2D = True               # the same for the encoder's stride-2 layers
CONV3D_FIRST_GEN = False            # tests: the K = 3 convolutions of this module through the first-generation kernels (GNR_CONV3D_FIRST_GEN:
                                    # a per-call flag of the C ABI, include/gnr.h -- this switch lives on the Python side, the library keeps none)
GNR_CONV3D_FIRST_GEN = 0x100


def _s2d_route(x, w):
    return (STRIDE2_AS_S2D and x.is_cuda and x.dtype == torch.float32 and w.shape[-1] in (3, 5) and torch.is_grad_enabled()
            and (x.requires_grad or w.requires_grad) and all(d % 2 == 0 for d in x.shape[2:]))


_FOLD_MATS = {}


def _stride2_matrix(k, device):
    """0/1 matrix [k^3, 8 * 27].  A stride-2 convolution (padding k // 2, even input size) reads x[2 o + d], d = t - k // 2, for output o:
    with d = 2 s + r (r = parity, s = d // 2 in {-1, 0, 1}) that is X_r[o + s] on the parity sub-grid X_r[m] = x[2 m + r] -- a k = 3
    same-padding stride-1 convolution over the 8 Cin channels of the space-to-depth input, every k^3 tap landing in exactly one
    (parity, offset) slot (the other slots stay zero).  Row k-tap, column (parity, k3 tap)."""
    key = (k, str(device))
    if key not in _FOLD_MATS:
        ax = torch.zeros(k, 2, 3)
        for t in range(k):
            d = t - k // 2
            ax[t, d % 2, d // 2 + 1] = 1
        _FOLD_MATS[key] = torch.einsum('aip,bjq,ckr->abcijkpqr', ax, ax, ax).reshape(k ** 3, 216).to(device)
    return _FOLD_MATS[key]


def conv3d_stride2(x, w, b):
    """F.conv3d(x, w, b, stride=2, padding=k // 2) (k = 3 or 5, even sizes: the encoder of gd/networks.py:59-76) as a space-to-depth
    shuffle and ONE stride-1 k = 3 convolution on the HIP path in all three directions (MIOpen runs these three small layers as
    im2col / col2im + GEMM and a naive fallback kernel: 1.4 ms per training step of 8 volumes for 3 % of the head's MACs)."""
    co, ci, k = w.shape[0], w.shape[1], w.shape[-1]
    B, _, D, H, W = x.shape
    xs = x.view(B, ci, D // 2, 2, H // 2, 2, W // 2, 2).permute(0, 3, 5, 7, 1, 2, 4, 6).reshape(B, 8 * ci, D // 2, H // 2, W // 2)
    wf = (w.reshape(co * ci, k ** 3) @ _stride2_matrix(k, w.device)).reshape(co, ci, 8, 27).permute(0, 2, 1, 3).reshape(co, 8 * ci, 3, 3, 3)
    return conv3d_same(xs, wf, b, _stride2_tap_mask(k, ci, co, w.device) if x.is_cuda else None)


_TAP_MASKS = {}


def _stride2_tap_mask(k, ci, co, device):
    """Tap mask (include/gnr.h gnr_conv3d_tap_mask) of conv3d_stride2's k3 weights [co, 8 ci, 27]: of the 8 x 27 (parity, tap) slots per
    input channel the 27 (k = 3) or 125 (k = 5) that a weight lands in.  Built once per layer shape from the 0/1 matrix itself (the
    structure, not the weights' values); the HIP kernels skip the other taps of every 16 x 16 channel block in all three directions."""
    key = (k, ci, co, str(device))
    if key not in _TAP_MASKS:
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        pat = (_stride2_matrix(k, device).sum(0) > 0).float().reshape(1, 1, 8, 27).expand(co, ci, 8, 27).permute(0, 2, 1, 3).reshape(co, 8 * ci, 27).contiguous()
        mask = torch.zeros(L.gnr_conv3d_tap_mask_words(8 * ci, co), dtype=torch.int32, device=device)
        _lib.check(L.gnr_conv3d_tap_mask(pat.data_ptr(), mask.data_ptr(), 8 * ci, co, 3, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                   'gnr_conv3d_tap_mask')
        _TAP_MASKS[key] = mask
    return _TAP_MASKS[key]


def _upfold_matrix(device):
    """0/1 matrix [125, 8 * 27].  A k = 5 same-padding convolution on a x2 nearest-upsampled grid x[j] = u[j // 2] reads, for the output
    voxel 2 s + c (c = parity per axis), tap t at j = 2 s + c + t - 2, i.e. source voxel s + (c + t - 2) // 2: offsets -1, 0, +1 -- per
    parity class a k = 3 same-padding convolution on the SOURCE grid whose weights are sums of the k = 5 weights (the zero padding of the
    upsampled grid is the zero padding of the source grid).  Column (class, k3 tap), row k5 tap."""
    key = ('up', str(device))
    if key not in _FOLD_MATS:
        ax = torch.zeros(5, 2, 3)
        for c in range(2):
            for t in range(5):
                ax[t, c, (c + t - 2) // 2 + 1] = 1
        _FOLD_MATS[key] = torch.einsum('aip,bjq,ckr->abcijkpqr', ax, ax, ax).reshape(125, 216).to(device)
    return _FOLD_MATS[key]


def upconv5_x2(u, w, b):
    """F.conv3d(F.interpolate(u, scale 2, nearest), w, b, padding=2) for a k = 5 kernel without the upsampled tensor (networks.py:91-96
    followed by a k5 convolution: decoder.conv3 and the three heads, 92 % of the head's MACs): ONE k = 3 convolution on the source grid
    with 8 x Cout output channels (one group per output parity class, weights pre-summed by a constant 0/1 matrix, so autograd returns
    the k = 5 weight gradient through the same matrix) and a depth-to-space shuffle.  125 / 27 = 4.6x fewer MACs in all three directions,
    no Cout padding for the 6-channel heads (48 = 3 MFMA row blocks), and neither the upsampled activation nor its gradient exist."""
    co, ci = w.shape[:2]
    B, _, D, H, W = u.shape
    wf = (w.reshape(co * ci, 125) @ _upfold_matrix(w.device)).reshape(co, ci, 8, 27).permute(2, 0, 1, 3).reshape(8 * co, ci, 3, 3, 3)
    y8 = conv3d_same(u, wf, b.repeat(8))
    return y8.view(B, 2, 2, 2, co, D, H, W).permute(0, 4, 5, 1, 6, 2, 7, 3).reshape(B, co, 2 * D, 2 * H, 2 * W)


_CONV_WS = {}


def _hip_conv3d_same(x, w, b, mode, mask=None):
    """gnr_conv3d_same(_masked) through the C ABI: mode 0 forward (x [B,Cin,D,H,W] -> y [B,Cout,..] + b), mode 1 backward data
    (x = dy [B,Cout,..] -> dx [B,Cin,..]); mask = the layer's tap mask (_tap_mask) or None."""
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
    rc = L.gnr_conv3d_same_masked(x.data_ptr(), w.data_ptr(), b.contiguous().data_ptr() if (b is not None and not mode) else None, y.data_ptr(),
                                  B, cin, cout, D, H, W, k, mode | (GNR_CONV3D_FIRST_GEN if CONV3D_FIRST_GEN else 0), mask.data_ptr() if mask is not None else None, ws.data_ptr(), ws.numel(),
                                  C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
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
    def forward(ctx, x, w, b, mask=None):
        ctx.save_for_backward(x, w)
        ctx.mask = mask
        return _hip_conv3d_same(x, w, b, 0, mask)

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from . import _lib
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        k = w.shape[-1]
        mask = ctx.mask
        dx = _hip_conv3d_same(dy, w, None, 1, mask) if ctx.needs_input_grad[0] else None
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
            kflag = k | (GNR_CONV3D_FIRST_GEN if CONV3D_FIRST_GEN else 0)
            rc = L.gnr_conv3d_same_bwd_weight_masked(xc.data_ptr(), dy.data_ptr(), dw.data_ptr(), *dims[:-1], kflag, mask.data_ptr() if mask is not None else None,
                                                     _CONV_WS[key].data_ptr(), need, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc:
                raise _lib.GnrError(f'gnr_conv3d_same_bwd_weight failed: {rc}')
        db = dy.sum((0, 2, 3, 4)) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def conv3d_same(x, w, b, mask=None):
    """Stride-1 same-padding conv3d; on the GPU under autograd all three directions run in HIP (_Conv3dSame).  mask: the tap mask of a
    structurally sparse weight tensor (_tap_mask), None = dense."""
    if x.is_cuda and x.dtype == torch.float32 and w.shape[-1] in (3, 5) and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return _Conv3dSame.apply(x, w, b, mask)
    return F.conv3d(x, w, b, padding=w.shape[-1] // 2)


class ConvNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder, self.decoder = _Encoder(), _Decoder()
        self.conv_qual, self.conv_rot, self.conv_width = _c3d(16, 1, 5), _c3d(16, 4, 5), _c3d(16, 1, 5)

    def forward(self, x):
        # the three heads read the same 16-channel volume: one 16 -> 6 convolution (identical per output channel) so that
        # autograd runs ONE conv3d backward instead of three (MIOpen: 9.4 ms each at 40^3)
        w = torch.cat([self.conv_qual.weight, self.conv_rot.weight, self.conv_width.weight], 0)
        b = torch.cat([self.conv_qual.bias, self.conv_rot.bias, self.conv_width.bias], 0)
        e = self.encoder(x)
        if _fold_route(e, w):
            y = upconv5_x2(self.decoder(e, upsample_last=False), w, b)      # the heads on the 20^3 decoder output, upsampling folded in
        else:
            y = conv3d_same(self.decoder(e), w, b)
        return torch.sigmoid(y[:, :1]), F.normalize(y[:, 1:5], dim=1), y[:, 5:6]
