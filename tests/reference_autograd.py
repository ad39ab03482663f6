"""TEST INFRASTRUCTURE: the differentiable PyTorch statement of the volumetric path, the reference the
HIP backward kernels (csrc/gnr_bwd.inc) are tested against.  Not part of the product: graspnerf_amd/ trains through its HIP
twin pairs only and raises without a GPU.

* the path itself, one scene per call, view-major flat arrays [V, N, C] with N = rn*dn points, live `nn.Parameter`s under
  the reference's state-dict names, nothing detached, the in-forward VJP taken with `create_graph=True` (ibrnet.py:497-504);
  `taps` keeps intermediates (and their gradients) for the stage-by-stage checks of tests/test_bwd_twins.py
* `attn_core` / `tail_backward`: the dual-number reverse pass of the per-ray tail in plain tensor algebra, checked against
  autograd's double backward (tests/test_ray_tail.py) and what k_ray_dual_bwd / k_geo_dual_* are compared with
* `ReferenceRenderer`: NeuralRayRenderer whose TRAINING forward runs this statement (any device), so that the model
  mirror, the losses and the trainer can be pinned against the reference's own backward (golden_train_step.npz) on the CPU,
  and the HIP training path against it on the GPU.
ref: src/nr/network/renderer.py:62-220, render_ops.py, dist_decoder.py, aggregate_net.py, ibrnet.py:447-513.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _lin(x, P, name):
    y = x @ P[name + '.weight'].t()
    b = P.get(name + '.bias')
    return y if b is None else y + b


def _mlp3(x, P, pre, last):
    x = F.elu(_lin(x, P, pre + '.0'))
    x = F.elu(_lin(x, P, pre + '.2'))
    return last(_lin(x, P, pre + '.4'))


def grid_points(res, volume_size=0.3):
    """utils/field_utils.py:12-27: voxel centres, index x*res^2 + y*res + z (fp64 then cast)."""
    vs = volume_size / res
    i = np.arange(res, dtype=np.float64) * vs + vs / 2
    X, Y, Z = np.meshgrid(i, i, i, indexing='ij')
    return np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)


def project(pts, poses, Ks, h, w):
    """render_ops.py:82-130 -> uv [V,N,2], z [V,N], mask [V,N] bool, dir [V,N,3]."""
    N = pts.shape[0]
    KRt = Ks @ poses
    hp = torch.cat([pts, pts.new_ones(N, 1)], 1)
    pc = torch.einsum('vij,nj->vni', KRt, hp)
    z = pc[..., 2]
    invalid = z.abs() < 1e-4
    z = torch.where(invalid, torch.full_like(z, 1e-3), z)
    uv = pc[..., :2] / z[..., None]
    outside = (uv[..., 0] < -0.5) | (uv[..., 0] >= w - 0.5) | (uv[..., 1] < -0.5) | (uv[..., 1] >= h - 0.5)
    cam = -(poses[:, :, :3].transpose(1, 2) @ poses[:, :, 3:])[..., 0]
    d = pts[None] - cam[:, None]
    return uv, z, (~invalid) & (~outside), -d / torch.clamp_min(torch.linalg.norm(d, dim=2, keepdim=True), 1e-5)


def bilinear_border(feat, uv, h, w):
    """render_ops.py:54-70 / ops.py:14-34: feat [V,C,fh,fw], uv [V,N,2] (full-res pixels) -> [V,N,C]; border padding;
    align_corners True for full-res maps, False otherwise.  Differentiable w.r.t. feat."""
    V, C, fh, fw = feat.shape
    xn = uv[..., 0] / (w - 1) * 2 - 1
    yn = uv[..., 1] / (h - 1) * 2 - 1
    if fh == h and fw == w:
        px, py = (xn + 1) / 2 * (fw - 1), (yn + 1) / 2 * (fh - 1)
    else:
        px, py = ((xn + 1) * fw - 1) / 2, ((yn + 1) * fh - 1) / 2
    px, py = px.clamp(0, fw - 1), py.clamp(0, fh - 1)
    x0, y0 = torch.floor(px), torch.floor(py)
    wx1, wy1 = px - x0, py - y0
    x0i, y0i = x0.long(), y0.long()
    x1i, y1i = (x0i + 1).clamp(max=fw - 1), (y0i + 1).clamp(max=fh - 1)
    fl = feat.permute(0, 2, 3, 1).reshape(V, fh * fw, C)

    def tap(yi, xi, wgt):
        return torch.gather(fl, 1, (yi * fw + xi)[..., None].expand(-1, -1, C)) * wgt[..., None]
    return tap(y0i, x0i, (1 - wx1) * (1 - wy1)) + tap(y0i, x1i, wx1 * (1 - wy1)) + \
        tap(y1i, x0i, (1 - wx1) * wy1) + tap(y1i, x1i, wx1 * wy1)


def decode_hit_vis(P, dec, f_ray, z, mask, depth_range, lo, hi):
    """dist_decoder.py:99-142 + renderer.py:62-78 -> hit, vis [V,N] (masked)."""
    mean = _mlp3(f_ray, P, dec + 'mean_decoder', F.softplus)
    var = _mlp3(f_ray, P, dec + 'var_decoder', F.softplus) + 0.05
    aw = _mlp3(f_ray, P, dec + 'aw_decoder', torch.sigmoid)
    near_r, far_r = -1 / depth_range[:, 0][:, None], -1 / depth_range[:, 1][:, None]
    dhat = (-1 / torch.clamp(z, min=1e-5) - near_r) / (far_r - near_r)
    near, far = (dhat - lo)[..., None], (dhat + hi)[..., None]
    mix = torch.cat([aw, 1 - aw], -1)
    cdf0 = 0.5 + 0.5 * torch.tanh((near - mean) * var)
    cdf1 = 0.5 + 0.5 * torch.tanh((far - mean) * var)
    if dec + 'vis_decoder.0.weight' in P:            # dist_decoder_cfg.use_vis (dist_decoder.py:89-97,103-104,133-134)
        pv = _mlp3(f_ray, P, dec + 'vis_decoder', torch.sigmoid)
        cdf0, cdf1 = cdf0 * pv, cdf1 * pv
    m = mask.to(f_ray.dtype)
    return torch.sum((cdf1 - cdf0) * mix, -1) * m, torch.sum((1 - cdf0) * mix, -1) * m


def sinusoid_table(n, d=16):
    pos = np.arange(n, dtype=np.float64)[:, None]
    j = np.arange(d)
    ang = pos / np.power(10000, 2 * (j // 2) / d)
    tab = ang.copy()
    tab[:, 0::2], tab[:, 1::2] = np.sin(ang[:, 0::2]), np.cos(ang[:, 1::2])
    return torch.from_numpy(tab).float()


_PE_CACHE = {}


def sinusoid_on(n, like):
    """sinusoid_table(n) on `like`'s device / dtype, uploaded once (a host->device copy per call would serialise the
    host with the GPU queue in every training pass)."""
    key = (n, like.device, like.dtype)
    if key not in _PE_CACHE:
        _PE_CACHE[key] = sinusoid_table(n).to(like)
    return _PE_CACHE[key]


def _mean_var(x, w):
    mean = torch.sum(x * w, 0)
    return mean, torch.sum(w * (x - mean[None]) ** 2, 0)


def _tap(taps, name, t):
    """Test hook: keep an intermediate (and its gradient) so the backward twins can be checked stage by stage."""
    if taps is not None:
        if t.requires_grad:
            t.retain_grad()
        taps[name] = t
    return t


def aggregate(P, agg, f_ray, rgb, f_img, hit, vis, mask, dirv, qdir, pts, rn, dn, want_grad, want_rgb, taps=None):
    """aggregate_net.py:35-70 + ibrnet.py:447-513.  -> sdf [rn,dn], grad [rn,dn,3] | None (differentiable: taken with
    create_graph=True), rgb [rn,dn,3] | None."""
    a = agg + 'agg_impl.'
    m = mask.to(f_ray.dtype)[..., None]
    pe_in = torch.cat([f_ray, ((hit - 0.5) * 2)[..., None], ((vis - 0.5) * 2)[..., None]], -1)
    e1 = _tap(taps, 'e1', F.relu(_lin(pe_in, P, agg + 'prob_embed.0')))
    e = _tap(taps, 'e', _lin(e1, P, agg + 'prob_embed.2'))
    dd = torch.cat([dirv - qdir[None], torch.sum(dirv * qdir[None], -1, keepdim=True)], -1)
    x = _tap(taps, 'x', torch.cat([rgb, f_img], -1) + F.elu(_lin(F.elu(_lin(dd, P, a + 'ray_dir_fc.0')), P, a + 'ray_dir_fc.2')))
    w = m / (torch.sum(m, 0, keepdim=True) + 1e-8)
    gate = _tap(taps, 'gate', torch.sigmoid(_lin(F.elu(_lin(e, P, a + 'neuray_fc.0')), P, a + 'neuray_fc.2')))
    w0 = gate * w
    mean0, var0 = _mean_var(x, w0)
    mean1, var1 = _mean_var(x, w)
    W0 = P[a + 'base_fc.0.weight']
    pre = _tap(taps, 'G', _tap(taps, 'SV', torch.cat([mean0, var0, mean1, var1], -1)) @ W0[:, :140].t() + P[a + 'base_fc.0.bias'])
    h = F.elu(pre[None] + x @ W0[:, 140:175].t() + e @ W0[:, 175:].t())
    h = F.elu(_lin(h, P, a + 'base_fc.2'))
    xv = F.elu(_lin(F.elu(_lin(h * w, P, a + 'vis_fc.0')), P, a + 'vis_fc.2'))
    v1 = torch.sigmoid(xv[..., 32:]) * m
    h = _tap(taps, 'h', h + xv[..., :32])
    v2 = _tap(taps, 'v2', torch.sigmoid(_lin(F.elu(_lin(h * v1, P, a + 'vis_fc2.0')), P, a + 'vis_fc2.2')) * m)
    w2 = v2 / (torch.sum(v2, 0, keepdim=True) + 1e-8)
    mean, var = _mean_var(h, w2)
    mean, var = _tap(taps, 'mean', mean), _tap(taps, 'var', var)
    nvalid = torch.sum(m, 0)[:, 0]
    col = None
    if want_rgb:
        c = F.elu(_lin(torch.cat([h, v2, dd], -1), P, a + 'rgb_fc.0'))
        c = _lin(F.elu(_lin(c, P, a + 'rgb_fc.2')), P, a + 'rgb_fc.4').masked_fill(m == 0, -1e9)
        col = torch.sum(rgb * torch.softmax(c, 0), 0).reshape(rn, dn, 3)
    sdf, grad = sdf_tail(P, agg, mean, var, torch.mean(w2, 0), nvalid, pts, rn, dn, want_grad, taps)
    return sdf, grad, col


def sdf_tail(P, agg, mean, var, wbar, nvalid, pts, rn, dn, want_grad, taps=None):
    """The per-ray tail (ibrnet.py:485-504): geometry_fc on [mean, var, wbar, embed(p)], positional encoding, 40-token
    self-attention, LayerNorm, out_geometry_fc, clip; with want_grad the in-forward gradient of sdf w.r.t. the points
    (create_graph=True).  mean/var [N,32], wbar [N,1], nvalid [N] -> sdf [rn,dn], grad [rn,dn,3] | None."""
    a = agg + 'agg_impl.'
    p = pts.detach().clone().requires_grad_(want_grad)
    with torch.enable_grad():
        emb = torch.cat([p] + [fn(p * f) for f in (1.0, 2.0, 4.0) for fn in (torch.sin, torch.cos)], -1)
        z86 = torch.cat([mean, var, wbar, emb], -1)
        g = _tap(taps, 'g16', F.elu(_lin(F.elu(_lin(z86, P, a + 'geometry_fc.0')), P, a + 'geometry_fc.2')))
        t = g.reshape(rn, dn, 16) + sinusoid_on(dn, g)[None]
        heads = lambda name: _lin(t, P, a + 'ray_attention.' + name).reshape(rn, dn, 4, 4).transpose(1, 2)
        q, k, v = heads('w_qs'), heads('w_ks'), heads('w_vs')
        logits = ((q / 2.0) @ k.transpose(2, 3)).masked_fill(~(nvalid.reshape(rn, 1, dn, 1) > 1), -1e9)
        o = (torch.softmax(logits, -1) @ v).transpose(1, 2).reshape(rn, dn, 16)
        y = _lin(o, P, a + 'ray_attention.fc') + t
        n = F.layer_norm(y, (16,), P[a + 'ray_attention.layer_norm.weight'], P[a + 'ray_attention.layer_norm.bias'], 1e-6)
        s = _lin(_lin(n, P, a + 'out_geometry_fc.0'), P, a + 'out_geometry_fc.1')[..., 0]
        sdf = s.clip(-1.0, 1.0).masked_fill(nvalid.reshape(rn, dn) < 1, 1.0)
        grad = None
        if want_grad:
            grad = torch.autograd.grad(sdf, p, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0].reshape(rn, dn, 3)
    return sdf, grad


def _gather(ref, uv, mask):
    h, w = ref['imgs'].shape[-2:]
    m = mask.to(ref['imgs'].dtype)[..., None]
    return (bilinear_border(ref['ray_feats'], uv, h, w) * m, bilinear_border(ref['imgs'], uv, h, w) * m,
            bilinear_border(ref['img_feats'], uv, h, w) * m)


def sample_volume(P, ref, res, dec='dist_decoder.', agg='agg_net.', taps=None):
    """renderer.py:164-199 for one scene -> [1,1,res,res,res]."""
    dev = ref['imgs'].device
    h, w = ref['imgs'].shape[-2:]
    pts = (torch.from_numpy(grid_points(res)).to(dev) + ref['bbox3d'][0].to(dev)).reshape(res * res, res, 3)
    pts = torch.flip(pts, (1,)).reshape(-1, 3)
    uv, z, mask, dirv = project(pts, ref['poses'], ref['Ks'], h, w)
    f_ray, rgb, f_img = _gather(ref, uv, mask)
    hit, vis = decode_hit_vis(P, dec, f_ray, z, mask, ref['depth_range'], 0.005, 0.005)
    qdir = torch.tensor([0., 0., 1.], device=dev).expand(pts.shape[0], 3)
    if taps is not None:
        taps.update(hit=_tap(taps, 'hit', hit), vis=_tap(taps, 'vis', vis), mask=mask)
    sdf, _, _ = aggregate(P, agg, f_ray, rgb, f_img, hit, vis, mask, dirv, qdir, pts, res * res, res, False, False, taps)
    return torch.flip(sdf.reshape(1, 1, res, res, res), (-1,))


def ray_points(que, depth):
    """render_ops.py:4-39: que coords [rn,2], pose [3,4], K [3,3]; depth [rn,dn] -> pts [rn*dn,3], qdir [rn,3]."""
    rn = depth.shape[0]
    rot = que['pose'][:, :3].t()
    trans = -rot @ que['pose'][:, 3:]
    # linalg.inv_ex = torch.inverse without the singularity check (a device->host read per call)
    cam = torch.linalg.inv_ex(que['K']).inverse @ torch.cat([que['coords'], que['coords'].new_ones(rn, 1)], 1).t()
    d = (rot @ cam + trans - trans).t()
    pts = (trans.t()[:, None] + d[:, None] * depth[..., None]).reshape(-1, 3)
    return pts, -d / torch.linalg.norm(d, dim=1, keepdim=True)


class _CumprodPositive(torch.autograd.Function):
    """torch.cumprod along the last axis for strictly positive factors.  Same forward values; the backward is the
    closed form autograd itself uses when no factor is zero (reverse cumulative sum of grad*out, divided by the
    input) -- without its `(input == 0).any()` check, a device->host read in the middle of every backward."""

    @staticmethod
    def forward(ctx, x):
        y = torch.cumprod(x, -1)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        w = g * y
        return (w + torch.sum(w, -1, keepdim=True) - torch.cumsum(w, -1)) / x


def composite(P, agg, sdf, grad, col, nvalid, qdir, depth, que, ref_hw, cfg):
    """NeuS alpha (aggregate_net.py:105-121), exclusive-cumprod compositing (render_ops.py:72-80) and the output dict of
    one render pass (renderer.py:110-138).  nvalid [rn,dn] = number of views that see each sample."""
    h, w = ref_hw
    variance = P[agg + 'deviation_network.variance']
    inv_s = torch.exp(variance * 10.0).clip(1e-6, 1e6)
    dists = torch.cat([depth[:, 1:] - depth[:, :-1], torch.full_like(depth[:, :1], 1e6)], -1)
    iter_cos = -F.relu(-torch.sum(-qdir[:, None] * grad, -1))
    pc = torch.sigmoid((sdf - iter_cos * dists * 0.5) * inv_s)
    nc = torch.sigmoid((sdf + iter_cos * dists * 0.5) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)
    T = _CumprodPositive.apply(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1))   # factors >= 1e-10
    hp = alpha * T[:, :-1]
    out = {'sdf_values': sdf[None], 'alpha_values': alpha[None], 'colors_nr': col[None], 'hit_prob_nr': hp[None],
           'pixel_colors_nr': torch.sum(hp[..., None] * col, 1)[None],
           'sdf_gradient_error': torch.mean((torch.linalg.norm(grad, dim=-1) - 1.0) ** 2).reshape(1, 1),
           's': variance.reshape(1, 1), 'render_depth': torch.sum(hp * depth, -1)[None],
           'ray_mask': (torch.sum((nvalid > cfg['ray_mask_view_num']).int(), 1) > cfg['ray_mask_point_num'])[None]}
    if 'imgs' in que:                                                     # renderer.py:125-127
        xy = torch.stack([que['coords'][:, 0] / (w - 1), que['coords'][:, 1] / (h - 1)], -1)     # no host-built tensor: no sync
        gt = F.grid_sample(que['imgs'], (xy * 2 - 1)[None, None], mode='bilinear', padding_mode='zeros', align_corners=True)
        out['pixel_colors_gt'] = gt[0, :, 0].t()[None]
    return out


def render_by_depth(P, ref, que, depth, dec, agg, cfg):
    """renderer.py:90-138 for one scene; depth [rn,dn].  que: coords [rn,2], pose [3,4], K [3,3], depth_range [2]."""
    rn, dn = depth.shape
    h, w = ref['imgs'].shape[-2:]
    pts, qdir = ray_points(que, depth)
    uv, z, mask, dirv = project(pts, ref['poses'], ref['Ks'], h, w)
    f_ray, rgb, f_img = _gather(ref, uv, mask)
    near, far = -1 / que['depth_range'][0], -1 / que['depth_range'][1]
    di = (-1 / depth - near) / (far - near)
    half = torch.cat([di[:, 1:] - di[:, :-1], torch.full_like(di[:, :1], 1e6)], -1) / 2
    ext = torch.cat([half[:, :1], half], -1)
    hit, vis = decode_hit_vis(P, dec, f_ray, z, mask, ref['depth_range'], ext[:, :-1].reshape(-1), ext[:, 1:].reshape(-1))
    qd = qdir[:, None].expand(rn, dn, 3).reshape(-1, 3)
    sdf, grad, col = aggregate(P, agg, f_ray, rgb, f_img, hit, vis, mask, dirv, qd, pts, rn, dn, True, True)
    nv = torch.sum(mask.reshape(-1, rn, dn).int(), 0)
    return composite(P, agg, sdf, grad, col, nv, qdir, depth, que, (h, w), cfg)


def sample_fine_depth(depth, hit_prob, depth_range, fdn, u):
    """render_ops.py:172-229 (hit_prob arrives detached, renderer.py:141); u [rn,fdn].  depth_range: (near, far) scalars
    or per-ray [rn,1] columns (rays of several scenes)."""
    near, far = -1 / depth_range[0], -1 / depth_range[1]
    d = (-1 / depth - near) / (far - near)
    centre = torch.cat([d[:, :1], (d[:, 1:] + d[:, :-1]) / 2, d[:, -1:]], -1)
    hp = hit_prob + 1e-5
    pdf = hp / torch.sum(hp, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = (inds - 1).clamp(min=0), inds.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(centre, 1, below), torch.gather(centre, 1, above)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    fd = b0 + (u - c0) / den * (b1 - b0)
    return -1 / (fd * (far - near) + near)


def render(P, ref, que, cfg, fine_u=None):
    """renderer.py:140-162 for one chunk of rays of one scene -> dict with '' and '_fine' keys."""
    dev = ref['imgs'].device
    rn, dn, fdn = que['coords'].shape[0], cfg['depth_sample_num'], cfg['fine_depth_sample_num']
    near, far = que['depth_range'][0], que['depth_range'][1]
    diff = 1 / far - 1 / near
    ticks = torch.cat([torch.zeros(1, device=dev), diff / (dn - 1) * torch.arange(1, dn - 1, dtype=torch.float32, device=dev),
                       diff.reshape(1)])
    depth = (1 / (1 / near + ticks))[None].expand(rn, dn).contiguous()    # render_ops.py:146-170
    out = render_by_depth(P, ref, que, depth, 'dist_decoder.', 'agg_net.', cfg)
    if fine_u is None:
        fine_u = ((0.5 + torch.arange(fdn, dtype=torch.float32, device=dev)) / fdn)[None].expand(rn, fdn)
    fd = sample_fine_depth(depth, out['hit_prob_nr'][0].detach(), que['depth_range'], fdn, fine_u.to(dev))
    fd = torch.cat([depth, fd], -1) if cfg.get('fine_depth_use_all') else fd          # renderer.py:145-146
    fine = render_by_depth(P, ref, que, torch.sort(fd, -1)[0], 'fine_dist_decoder.', 'fine_agg_net.', cfg)
    out.update({k + '_fine': v for k, v in fine.items()})
    return out


def depth_mean(P, ref, coords_rc, dec):
    """predict_mean_for_depth_loss for one level (renderer.py:230-266): coords [pn,2] are fed as (x,y) although they
    hold (row,col) -> [V,pn,2]."""
    h, w = ref['imgs'].shape[-2:]
    V = ref['imgs'].shape[0]
    uv = coords_rc.to(torch.float32)[None].expand(V, -1, -1)
    f = bilinear_border(ref['ray_feats'], uv, h, w)
    return _mlp3(f, P, dec + 'mean_decoder', F.softplus)



# ----------------------------------------------------------------------------------------------------------------------
# the per-ray tail's backward incl. the second-order path, in tensor algebra (moved here from the product's ray_tail.py)
# ----------------------------------------------------------------------------------------------------------------------
from graspnerf_amd.ray_tail import TAIL_KEYS, unfold_out_geometry      # noqa: E402


def _elu_d1(x):        # ELU'(x)
    return torch.where(x > 0, torch.ones_like(x), torch.exp(x))


def _elu_d2(x):        # ELU''(x)
    return torch.where(x > 0, torch.zeros_like(x), torch.exp(x))


def embed_tangent(p, gamma):
    """d/d eps of embed(p + eps*gamma) for the 21-channel embedder (neus.py:21-66): [p, sin p, cos p, sin 2p, ...]."""
    out = [gamma]
    for f in (1.0, 2.0, 4.0):
        out += [f * torch.cos(f * p) * gamma, -f * torch.sin(f * p) * gamma]
    return torch.cat(out, -1)


def attn_core(W, g, gd, a, nvalid):
    """Dual reverse pass over [+PE, attention, fc + residual, LayerNorm, folded out_geometry_fc, clip / mask].
    W: dict with wq, wk, wv, wfc [16,16], lnw, lnb [16], weff [16], beff [].   g, gd [R,dn,16] value / tangent of the
    geometry_fc output; a [R,dn]; nvalid [R,dn].
    -> gbar, gdbar [R,dn,16] and a dict of gradients for the entries of W."""
    R, dn, _ = g.shape
    t = g + sinusoid_on(dn, g)[None]
    td = gd
    heads = lambda x, w: (x @ w.t()).reshape(R, dn, 4, 4).transpose(1, 2)           # [R,4,dn,4]
    merge = lambda x: x.transpose(1, 2).reshape(R, dn, 16)
    q, k, v = heads(t, W['wq']), heads(t, W['wk']), heads(t, W['wv'])
    qd, kd, vd = heads(td, W['wq']), heads(td, W['wk']), heads(td, W['wv'])
    rowok = (nvalid > 1).reshape(R, 1, dn, 1)
    S = (0.5 * q) @ k.transpose(2, 3)
    Sd = ((0.5 * qd) @ k.transpose(2, 3) + (0.5 * q) @ kd.transpose(2, 3)) * rowok
    Pm = torch.softmax(S.masked_fill(~rowok, -1e9), -1)
    r = torch.sum(Pm * Sd, -1, keepdim=True)
    Pd = Pm * (Sd - r)
    o, od = merge(Pm @ v), merge(Pd @ v + Pm @ vd)
    y, yd = o @ W['wfc'].t() + t, od @ W['wfc'].t() + td
    # LayerNorm (eps 1e-6) on dual numbers
    c = y - y.mean(-1, keepdim=True)
    rs = torch.rsqrt((c * c).mean(-1, keepdim=True) + 1e-6)
    xh = c * rs
    cd = yd - yd.mean(-1, keepdim=True)
    m1 = (xh * cd).mean(-1, keepdim=True)
    xhd = rs * (cd - xh * m1)
    n, nd = W['lnw'] * xh + W['lnb'], W['lnw'] * xhd
    u = n @ W['weff'] + W['beff']
    m = ((u >= -1) & (u <= 1) & (nvalid >= 1)).to(g.dtype)
    sbar, sdbar = a * m, m                                                # seeds: d Phi / d u, d Phi / d u_dot
    G = {'weff': torch.sum(sbar[..., None] * n + sdbar[..., None] * nd, (0, 1)), 'beff': sbar.sum()}
    nbar, ndbar = sbar[..., None] * W['weff'], sdbar[..., None] * W['weff']
    G['lnw'], G['lnb'] = torch.sum(nbar * xh + ndbar * xhd, (0, 1)), nbar.sum((0, 1))
    xhbar, e = W['lnw'] * nbar, W['lnw'] * ndbar
    E1 = torch.sum(e * xh, -1, keepdim=True)
    rsbar = torch.sum(e * xhd, -1, keepdim=True) / rs
    cdbar = rs * (e - xh * E1 / 16)
    xhbar = xhbar - rs * (m1 * e + (E1 / 16) * cd)
    ydbar = cdbar - cdbar.mean(-1, keepdim=True)
    cbar = rs * xhbar
    rsbar = rsbar + torch.sum(xhbar * c, -1, keepdim=True)
    cbar = cbar + (2.0 / 16) * c * (-0.5 * rs ** 3 * rsbar)
    ybar = cbar - cbar.mean(-1, keepdim=True)
    # fc + residual
    tbar, tdbar = ybar.clone(), ydbar.clone()
    G['wfc'] = torch.einsum('rio,rik->ok', ybar, o) + torch.einsum('rio,rik->ok', ydbar, od)
    split = lambda x: x.reshape(R, dn, 4, 4).transpose(1, 2)
    obar, odbar = split(ybar @ W['wfc']), split(ydbar @ W['wfc'])
    # attention
    vbar = Pm.transpose(2, 3) @ obar + Pd.transpose(2, 3) @ odbar
    vdbar = Pm.transpose(2, 3) @ odbar
    Pbar = obar @ v.transpose(2, 3) + odbar @ vd.transpose(2, 3)
    Pdbar = odbar @ v.transpose(2, 3)
    cst = torch.sum(Pdbar * Pm, -1, keepdim=True)
    Sdbar = Pm * (Pdbar - cst) * rowok
    Pbar2 = Pbar + Pdbar * (Sd - r) - cst * Sd
    Sbar = Pm * (Pbar2 - torch.sum(Pm * Pbar2, -1, keepdim=True)) * rowok
    qbar = 0.5 * (Sbar @ k + Sdbar @ kd)
    qdbar = 0.5 * (Sdbar @ k)
    kbar = 0.5 * (Sbar.transpose(2, 3) @ q + Sdbar.transpose(2, 3) @ qd)
    kdbar = 0.5 * (Sdbar.transpose(2, 3) @ q)
    for name, xb, xdb in (('wq', qbar, qdbar), ('wk', kbar, kdbar), ('wv', vbar, vdbar)):
        xb, xdb = merge(xb), merge(xdb)
        G[name] = torch.einsum('rio,rik->ok', xb, t) + torch.einsum('rio,rik->ok', xdb, td)
        tbar = tbar + xb @ W[name]
        tdbar = tdbar + xdb @ W[name]
    return tbar, tdbar, G



def tail_weights(P, agg):
    """The tail's parameters under the names attn_core uses (out_geometry_fc folded: two linears, no activation)."""
    a = agg + 'agg_impl.'
    wa, ba = P[a + 'out_geometry_fc.0.weight'], P[a + 'out_geometry_fc.0.bias']
    wb, bb = P[a + 'out_geometry_fc.1.weight'], P[a + 'out_geometry_fc.1.bias']
    return {'wq': P[a + 'ray_attention.w_qs.weight'], 'wk': P[a + 'ray_attention.w_ks.weight'],
            'wv': P[a + 'ray_attention.w_vs.weight'], 'wfc': P[a + 'ray_attention.fc.weight'],
            'lnw': P[a + 'ray_attention.layer_norm.weight'], 'lnb': P[a + 'ray_attention.layer_norm.bias'],
            'weff': (wb @ wa)[0], 'beff': (wb @ ba + bb)[0]}


def tail_backward(P, agg, stats, nvalid, pts, rn, dn, a, gamma, core=attn_core, geo=None):
    """stats [N,65] = (mean 32, var 32, wbar), nvalid [N], pts [N,3] (N = rn*dn), a [rn,dn], gamma [rn,dn,3].
    -> d stats [N,65] and {state-dict name: gradient} for geometry_fc, ray_attention, out_geometry_fc of `agg`.
    `core` = attn_core or the HIP kernel's wrapper (same signature); `geo` = None (the two geometry_fc layers in tensor
    algebra, below) or hip_geo(...): k_geo_dual_fwd / k_geo_dual_bwd."""
    assert geo is None
    pre = agg + 'agg_impl.'
    W1, b1 = P[pre + 'geometry_fc.0.weight'], P[pre + 'geometry_fc.0.bias']
    W2, b2 = P[pre + 'geometry_fc.2.weight'], P[pre + 'geometry_fc.2.bias']
    p = pts.detach()
    emb = torch.cat([p] + [fn(p * f) for f in (1.0, 2.0, 4.0) for fn in (torch.sin, torch.cos)], -1)
    embd = embed_tangent(p, gamma.reshape(-1, 3))
    x = torch.cat([stats, emb], -1)
    h1p = x @ W1.t() + b1
    e1 = _elu_d1(h1p)
    h1, h1pd = F.elu(h1p), embd @ W1[:, 65:].t()
    h1d = e1 * h1pd
    gp = h1 @ W2.t() + b2
    e2 = _elu_d1(gp)
    gpd = h1d @ W2.t()
    W = tail_weights(P, agg)
    gbar, gdbar, G = core(W, F.elu(gp).reshape(rn, dn, 16), (e2 * gpd).reshape(rn, dn, 16), a, nvalid.reshape(rn, dn))
    gbar, gdbar = gbar.reshape(-1, 16), gdbar.reshape(-1, 16)
    gpbar = e2 * gbar + _elu_d2(gp) * gpd * gdbar
    gpdbar = e2 * gdbar
    h1bar, h1dbar = gpbar @ W2, gpdbar @ W2
    h1pbar = e1 * h1bar + _elu_d2(h1p) * h1pd * h1dbar
    h1pdbar = e1 * h1dbar
    dW1 = h1pbar.t() @ x
    dW1[:, 65:] += h1pdbar.t() @ embd
    grads = {pre + 'geometry_fc.0.weight': dW1, pre + 'geometry_fc.0.bias': h1pbar.sum(0),
             pre + 'geometry_fc.2.weight': gpbar.t() @ h1 + gpdbar.t() @ h1d, pre + 'geometry_fc.2.bias': gpbar.sum(0)}
    for k, name in (('wq', 'w_qs'), ('wk', 'w_ks'), ('wv', 'w_vs'), ('wfc', 'fc')):
        grads[pre + 'ray_attention.' + name + '.weight'] = G[k]
    grads[pre + 'ray_attention.layer_norm.weight'], grads[pre + 'ray_attention.layer_norm.bias'] = G['lnw'], G['lnb']
    grads.update(unfold_out_geometry(P, agg, G['weff'], G['beff']))
    return h1pbar @ W1[:, :65], grads


# ----------------------------------------------------------------------------------------------------------------------
# NeuralRayRenderer whose training forward is this statement
# ----------------------------------------------------------------------------------------------------------------------
from graspnerf_amd.renderer import NeuralRayRenderer      # noqa: E402


class ReferenceRenderer(NeuralRayRenderer):
    """Same module (parameters, backbones, cfg, RNG draw order, output dict) as the product's NeuralRayRenderer; in
    training mode the volumetric path is differentiated through the PyTorch statement above instead of the HIP twin
    pairs.  `use_reference_statement(net)` swaps the class of an existing GraspNeRF's nr_net in place."""

    def _render_autograd(self, que, ref, _prep=None):
        P = self._params()
        rn, chunk, fdn = que['coords'].shape[1], self.cfg['ray_batch_num'], self.cfg['fine_depth_sample_num']
        parts = []
        for r0 in range(0, rn, chunk):
            u = torch.rand([1, min(chunk, rn - r0), fdn])                  # render_ops.py:204-205 (CPU generator)
            for net in (self.agg_net, self.fine_agg_net):
                net.train_step_bookkeeping()
            q = {'coords': que['coords'][0, r0:r0 + chunk], 'pose': que['poses'][0], 'K': que['Ks'][0],
                 'depth_range': que['depth_range'][0]}
            if 'imgs' in que:
                q['imgs'] = que['imgs']
            parts.append(render(P, ref, q, self._render_cfg(), u[0]))
        out = {k: torch.cat([p[k] for p in parts], 1) for k in parts[0]}
        if not self.cfg['render_depth']:
            out.pop('render_depth', None), out.pop('render_depth_fine', None)
        return out

    def sample_volume(self, ref_imgs_info, _prep=None, is_train=False):
        if self._use_autograd(is_train):
            return sample_volume(self._params(), ref_imgs_info, self.cfg['volume_resolution'])
        return super().sample_volume(ref_imgs_info, _prep, is_train)

    def predict_mean_for_depth_loss(self, ref_imgs_info, _prep=None, is_train=False):
        if not self._use_autograd(is_train):
            return super().predict_mean_for_depth_loss(ref_imgs_info, _prep, is_train)
        h, w = ref_imgs_info['imgs'].shape[-2:]
        rfn = ref_imgs_info['imgs'].shape[0]
        coords = self.gen_depth_loss_coords(h, w, ref_imgs_info['imgs'].device)
        P = self._params()
        mc = depth_mean(P, ref_imgs_info, coords, 'dist_decoder.')
        mf = depth_mean(P, ref_imgs_info, coords, 'fine_dist_decoder.')
        return {'depth_mean': mc[..., 0], 'depth_coords': coords[None].repeat(rfn, 1, 1), 'depth_mean_2': mc[..., 1],
                'depth_mean_fine': mf[..., 0], 'depth_mean_fine_2': mf[..., 1]}

    def _train_prep(self, ref_imgs_info, rn=0):
        return None                                                        # no HIP workspaces on this route

    def _need_gpu(self, t):
        pass

    def forward_scenes(self, datas, stacked=False):
        return None                                                        # scene by scene, like the reference


def use_reference_statement(net, on=True):
    """GraspNeRF `net`: route its nr_net's training forward through this module (on=True) or back through the product's HIP
    twin pairs (on=False).  Parameters, buffers and cfg are untouched (class swap)."""
    net.nr_net.__class__ = ReferenceRenderer if on else NeuralRayRenderer
    return net
