"""CPU oracle for the GraspNeRF `src/nr` volumetric hot path  --  TEST INFRASTRUCTURE ONLY.

This is a from-scratch fp32 restatement (torch-CPU tensors, view-major flat [V, N, C] arrays,
no 207-wide concat, no permutes) of the reference algorithm.  It is NOT the product: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and
only as the checker / the timed CPU baseline.  The product path (graspnerf_amd/) never
imports this module and fails loudly when the HIP library is missing.

Pinning: the reference ships no tests or golden vectors for this path, and
the arithmetic lives in PyTorch (requirements.txt:1, unpinned).  The oracle is therefore
pinned against outputs of the reference itself, imported and run in the build container
(tools/make_goldens.py -> tests/golden/*.npz; torch 2.10.0 CPU); tests/test_oracle_golden.py
replays them.  Every function cites the reference file:line it follows.

Conventions: V views, N = rn*dn points (ray/column-major, sample fastest), all fp32.
`sd` is a dict {reference state-dict key: torch tensor}; `dec`/`agg` are key prefixes such
as 'dist_decoder.' / 'agg_net.' (coarse) or 'fine_dist_decoder.' / 'fine_agg_net.' (fine).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

F32 = torch.float32          # the working dtype: torch.float32 like the reference; fp64_mode() switches it to float64


class fp64_mode:
    """`with fp64_mode():` runs the same restatement in float64 (inputs and weights are given as float64 tensors holding
    the fp32 values exactly: to_torch64).  The ARBITER of the GPU parity tests (tests/test_gpu_precision.py): the distance
    of the HIP path from these values is compared with the distance of the fp32 oracle (= the reference's arithmetic) from
    them.  Index decisions (floor of a tap, in-image tests, searchsorted) are taken in float64 too, so single points that
    sit within an fp32 rounding of such an edge can differ by a whole tap; the tests compare through quantiles."""

    def __enter__(self):
        global F32
        self.prev = F32
        F32 = torch.float64
        return self

    def __exit__(self, *exc):
        global F32
        F32 = self.prev
        return False


# ----------------------------------------------------------------------------------------
# G1  voxel-centre grid                                  ref: utils/field_utils.py:12-27
# ----------------------------------------------------------------------------------------
def grid_points(res, volume_size=0.3):
    """[res^3, 3] voxel centres, index = x*res^2 + y*res + z; computed in float64 then cast
    (field_utils.py:17-25: python-float arithmetic, np.array(...).astype(np.float32))."""
    vs = volume_size / res
    hv = vs / 2
    i = np.arange(res, dtype=np.float64) * vs + hv
    X, Y, Z = np.meshgrid(i, i, i, indexing='ij')
    return np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)


def volume_query_points(res, bbox_min):
    """Points of sample_volume in column order: [res*res columns, res samples, 3], sample 0 =
    TOP voxel (z flipped).  ref: renderer.py:167-170."""
    p = (torch.from_numpy(grid_points(res)) + torch.as_tensor(bbox_min, dtype=torch.float32)).to(F32)   # the points are fp32 INPUTS of the path
    p = p.reshape(res * res, res, 3)
    return torch.flip(p, (1,))


# ----------------------------------------------------------------------------------------
# P1-P3  projection, in-image mask, point->camera direction   ref: render_ops.py:82-130
# ----------------------------------------------------------------------------------------
def project_points(pts, poses, Ks, h, w):
    """pts [N,3], poses [V,3,4], Ks [V,3,3] -> uv [V,N,2], z [V,N], mask [V,N] bool, dir [V,N,3]."""
    V = poses.shape[0]
    N = pts.shape[0]
    KRt = Ks @ poses                                             # render_ops.py:94
    hp = torch.cat([pts, torch.ones(N, 1, dtype=F32)], 1)        # :91
    pc = torch.einsum('vij,nj->vni', KRt, hp)                    # :98-99 (rows 0..2 of H)
    z = pc[..., 2].clone()
    invalid = z.abs() < 1e-4                                     # :101
    z[invalid] = 1e-3                                            # :102
    uv = pc[..., :2] / z[..., None]                              # :103
    outside = (uv[..., 0] < -0.5) | (uv[..., 0] >= w - 0.5) | \
              (uv[..., 1] < -0.5) | (uv[..., 1] >= h - 0.5)      # :126-127
    mask = (~invalid) & (~outside)                               # :128
    cam = -(poses[:, :, :3].transpose(1, 2) @ poses[:, :, 3:])[..., 0]   # :112  [V,3]
    d = pts[None] - cam[:, None]                                 # :113
    nrm = torch.clamp_min(torch.linalg.norm(d, dim=2, keepdim=True), 1e-5)
    return uv, z, mask, -d / nrm                                 # :114


# ----------------------------------------------------------------------------------------
# I1  bilinear gather, border padding            ref: render_ops.py:54-70, ops.py:14-34
# ----------------------------------------------------------------------------------------
def bilinear_border(feat, uv, h, w):
    """feat [V,C,fh,fw], uv [V,N,2] in full-res pixel units -> [V,N,C].
    Pixel map (ops.py:29-33 + grid_sample): feature map at other resolution (align_corners
    False): px = u/(w-1)*fw - 0.5; full-res map (align_corners True): px = u/(w-1)*(fw-1).
    Border: clamp px to [0, f-1], then 4-tap lerp; taps outside contribute 0."""
    V, C, fh, fw = feat.shape
    xn = uv[..., 0] / (w - 1) * 2 - 1
    yn = uv[..., 1] / (h - 1) * 2 - 1
    if fh == h and fw == w:
        px = (xn + 1) / 2 * (fw - 1)
        py = (yn + 1) / 2 * (fh - 1)
    else:
        px = ((xn + 1) * fw - 1) / 2
        py = ((yn + 1) * fh - 1) / 2
    px = px.clamp(0, fw - 1)
    py = py.clamp(0, fh - 1)
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    wx1 = px - x0
    wy1 = py - y0
    wx0 = 1 - wx1
    wy0 = 1 - wy1
    x0i = x0.long()
    y0i = y0.long()
    x1i = (x0i + 1).clamp(max=fw - 1)      # the +1 tap only ever leaves the map with weight 0
    y1i = (y0i + 1).clamp(max=fh - 1)
    fl = feat.permute(0, 2, 3, 1).reshape(V, fh * fw, C)

    def tap(yi, xi):
        idx = (yi * fw + xi)[..., None].expand(-1, -1, C)
        return torch.gather(fl, 1, idx)
    out = tap(y0i, x0i) * (wx0 * wy0)[..., None] + tap(y0i, x1i) * (wx1 * wy0)[..., None] + \
          tap(y1i, x0i) * (wx0 * wy1)[..., None] + tap(y1i, x1i) * (wx1 * wy1)[..., None]
    return out


def gather_views(inp, uv, mask):
    """P4 + I2: masked gathers of ray_feats, rgb, img_feats.  ref: render_ops.py:137-138,
    renderer.py:80-88.  inp: dict(imgs[V,3,H,W], img_feats[V,32,fh,fw], ray_feats[...])."""
    h, w = inp['imgs'].shape[-2:]
    m = mask.to(F32)[..., None]
    f_ray = bilinear_border(inp['ray_feats'], uv, h, w) * m
    rgb = bilinear_border(inp['imgs'], uv, h, w) * m
    f_img = bilinear_border(inp['img_feats'], uv, h, w) * m
    return f_ray, rgb, f_img


# ----------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------
def _lin(x, sd, name):
    w = sd[name + '.weight']
    b = sd.get(name + '.bias')
    y = x @ w.t()
    return y if b is None else y + b


def _mlp3(x, sd, pre, last):
    """Linear-ELU-Linear-ELU-Linear-<last>    ref: dist_decoder.py:64-88"""
    x = F.elu(_lin(x, sd, pre + '.0'))
    x = F.elu(_lin(x, sd, pre + '.2'))
    return last(_lin(x, sd, pre + '.4'))


# ----------------------------------------------------------------------------------------
# D1-D4  mixture-of-logistics decoder -> hit prob / visibility per (view, point)
# ----------------------------------------------------------------------------------------
def ref_view_inv_depth(z, depth_range):
    """z [V,N] camera depth in each ref view -> normalised inverse depth.
    ref: dist_decoder.py:17-23."""
    near = -1 / depth_range[:, 0][:, None]
    far = -1 / depth_range[:, 1][:, None]
    d = -1 / torch.clamp(z, min=1e-5)
    return (d - near) / (far - near)


def decode_hit_vis(sd, dec, f_ray, z, mask, depth_range, lo, hi):
    """-> hit [V,N], vis [V,N] (already masked).  lo/hi: half-interval below/above the
    projected normalised inverse depth ([N] tensors or python floats).
    ref: dist_decoder.py:99-107, :109-142, renderer.py:62-78.  `use_vis: true` = the state dict holds `vis_decoder`:
    both cdfs are multiplied by its sigmoid output (dist_decoder.py:89-97,103-104,133-134)."""
    mean = _mlp3(f_ray, sd, dec + 'mean_decoder', F.softplus)              # [V,N,2]
    var = _mlp3(f_ray, sd, dec + 'var_decoder', F.softplus) + 0.05         # AddBias(0.05)
    aw = _mlp3(f_ray, sd, dec + 'aw_decoder', torch.sigmoid)               # [V,N,1]
    dhat = ref_view_inv_depth(z, depth_range)
    near = (dhat - lo)[..., None]
    far = (dhat + hi)[..., None]
    mix = torch.cat([aw, 1 - aw], -1)
    cdf0 = 0.5 + 0.5 * torch.tanh((near - mean) * var)
    cdf1 = 0.5 + 0.5 * torch.tanh((far - mean) * var)
    if dec + 'vis_decoder.0.weight' in sd:
        pv = _mlp3(f_ray, sd, dec + 'vis_decoder', torch.sigmoid)          # [V,N,1]
        cdf0, cdf1 = cdf0 * pv, cdf1 * pv
    vis = torch.sum((1 - cdf0) * mix, -1)
    hit = torch.sum((cdf1 - cdf0) * mix, -1)
    m = mask.to(F32)
    return hit * m, vis * m


def ray_half_intervals(que_depth, que_depth_range):
    """Render path: half-intervals from the query ray's normalised inverse-depth spacing.
    que_depth [rn,dn], que_depth_range [2] -> lo, hi [rn*dn].
    ref: render_ops.py:41-52 (depth2inv_dists, last = 1e6), dist_decoder.py:34-38."""
    near, far = -1 / que_depth_range[0], -1 / que_depth_range[1]
    di = (-1 / que_depth - near) / (far - near)
    dists = torch.cat([di[:, 1:] - di[:, :-1], torch.full_like(di[:, :1], 1e6)], -1)
    half = dists / 2
    ext = torch.cat([half[:, :1], half], -1)            # [rn, dn+1]
    return ext[:, :-1].reshape(-1), ext[:, 1:].reshape(-1)


# ----------------------------------------------------------------------------------------
# A3  positional embedding of the query point     ref: neus.py:21-66 (multires 3)
# ----------------------------------------------------------------------------------------
def embed_points(p):
    out = [p]
    for f in (1.0, 2.0, 4.0):
        out += [torch.sin(p * f), torch.cos(p * f)]
    return torch.cat(out, -1)


def sinusoid_table(n, d=16):
    """ref: ibrnet.py:437-445 (float64 numpy, then .float())."""
    pos = np.arange(n, dtype=np.float64)[:, None]
    j = np.arange(d)
    ang = pos / np.power(10000, 2 * (j // 2) / d)
    tab = ang.copy()
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(tab).float().to(F32)       # (the reference's table is fp32: rounded first)


# ----------------------------------------------------------------------------------------
# E1 + A1 + A2 + A4 + A5 + A6  the aggregation network
# ----------------------------------------------------------------------------------------
def weighted_mean_var(x, w):
    """x [V,N,C], w [V,N,1] -> mean, var [N,C].  ref: ibrnet.py:112-116."""
    mean = torch.sum(x * w, 0)
    var = torch.sum(w * (x - mean[None]) ** 2, 0)
    return mean, var


def aggregate(sd, agg, f_ray, rgb, f_img, hit, vis, mask, dirv, qdir, pts, rn, dn,
              want_grad=True, want_rgb=True, debug=None):
    """Everything after the decoder for N = rn*dn points.
    f_ray/f_img [V,N,32], rgb [V,N,3], hit/vis/mask [V,N], dirv [V,N,3], qdir [N,3], pts [N,3].
    -> dict(sdf [rn,dn], grad [rn,dn,3] or None, rgb [rn,dn,3] or None)
    ref: aggregate_net.py:35-70 (embedding), ibrnet.py:447-513 (IBRNetWithNeuRayNeus.forward)."""
    a = agg + 'agg_impl.'
    V, N = mask.shape
    m = mask.to(F32)[..., None]                                          # [V,N,1]
    # E1: prob embedding 34 -> 32 -> 32            aggregate_net.py:46-54
    pe_in = torch.cat([f_ray, ((hit - 0.5) * 2)[..., None], ((vis - 0.5) * 2)[..., None]], -1)
    e = _lin(F.relu(_lin(pe_in, sd, agg + 'prob_embed.0')), sd, agg + 'prob_embed.2')
    # dir diff                                      aggregate_net.py:11-17
    dd = torch.cat([dirv - qdir[None], torch.sum(dirv * qdir[None], -1, keepdim=True)], -1)
    # A1                                            ibrnet.py:456-482
    dfeat = F.elu(_lin(F.elu(_lin(dd, sd, a + 'ray_dir_fc.0')), sd, a + 'ray_dir_fc.2'))
    x = torch.cat([rgb, f_img], -1) + dfeat                               # [V,N,35]
    w = m / (torch.sum(m, 0, keepdim=True) + 1e-8)
    gate = _lin(F.elu(_lin(e, sd, a + 'neuray_fc.0')), sd, a + 'neuray_fc.2')
    w0 = torch.sigmoid(gate) * w
    mean0, var0 = weighted_mean_var(x, w0)
    mean1, var1 = weighted_mean_var(x, w)
    glob = torch.cat([mean0, var0, mean1, var1], -1)                      # [N,140]
    W0 = sd[a + 'base_fc.0.weight']                                       # [64,207]
    # view-invariant 140 columns applied once per point (algebraically == the 207-wide concat)
    pre = glob @ W0[:, :140].t() + sd[a + 'base_fc.0.bias']
    h = F.elu(pre[None] + x @ W0[:, 140:175].t() + e @ W0[:, 175:].t())
    h = F.elu(_lin(h, sd, a + 'base_fc.2'))                               # [V,N,32]
    xv = F.elu(_lin(F.elu(_lin(h * w, sd, a + 'vis_fc.0')), sd, a + 'vis_fc.2'))
    res, v1 = xv[..., :32], xv[..., 32:]
    v1 = torch.sigmoid(v1) * m
    h = h + res
    v2 = torch.sigmoid(_lin(F.elu(_lin(h * v1, sd, a + 'vis_fc2.0')), sd, a + 'vis_fc2.2')) * m
    w2 = v2 / (torch.sum(v2, 0, keepdim=True) + 1e-8)
    mean, var = weighted_mean_var(h, w2)                                  # [N,32]
    wbar = torch.mean(w2, 0)                                              # [N,1]
    nvalid = torch.sum(m, 0)[:, 0]                                        # [N]

    # A3-A5: SDF head + in-forward VJP             ibrnet.py:484-504
    p = pts.detach().clone().requires_grad_(want_grad)
    with torch.enable_grad():
        z86 = torch.cat([mean.detach(), var.detach(), wbar.detach(), embed_points(p)], -1)
        g = F.elu(_lin(F.elu(_lin(z86, sd, a + 'geometry_fc.0')), sd, a + 'geometry_fc.2'))
        t = g.reshape(rn, dn, 16) + sinusoid_table(dn)[None]
        q = _lin(t, sd, a + 'ray_attention.w_qs').reshape(rn, dn, 4, 4).transpose(1, 2)
        k = _lin(t, sd, a + 'ray_attention.w_ks').reshape(rn, dn, 4, 4).transpose(1, 2)
        v = _lin(t, sd, a + 'ray_attention.w_vs').reshape(rn, dn, 4, 4).transpose(1, 2)
        logits = (q / 2.0) @ k.transpose(2, 3)                            # [rn,4,dn,dn]
        row_ok = (nvalid.reshape(rn, 1, dn, 1) > 1)
        logits = logits.masked_fill(~row_ok, -1e9)                        # query-row mask (H3)
        att = torch.softmax(logits, -1)
        o = (att @ v).transpose(1, 2).reshape(rn, dn, 16)
        y = _lin(o, sd, a + 'ray_attention.fc') + t
        n = F.layer_norm(y, (16,), sd[a + 'ray_attention.layer_norm.weight'],
                         sd[a + 'ray_attention.layer_norm.bias'], 1e-6)
        s = _lin(_lin(n, sd, a + 'out_geometry_fc.0'), sd, a + 'out_geometry_fc.1')[..., 0]
        sdf = s.clip(-1.0, 1.0)
        sdf = sdf.masked_fill(nvalid.reshape(rn, dn) < 1, 1.0)
        grad = None
        if want_grad:
            grad = torch.autograd.grad(sdf, p, torch.ones_like(sdf))[0].reshape(rn, dn, 3)
    out = {'sdf': sdf.detach(), 'grad': grad, 'rgb': None, 'nvalid': nvalid.reshape(rn, dn)}
    if debug is not None:      # stage-by-stage intermediates for kernel bring-up (tests only)
        debug.update(v2=v2[..., 0].detach(), msum=nvalid, wbar=wbar[:, 0].detach(), mean_f0=mean[:, 0].detach(),
                     var_f0=var[:, 0].detach(), mean0_rgb0=mean0[:, 0].detach(), pre_f0=pre[:, 0].detach(),
                     g_f0=g[:, 0].detach(), vsum=torch.sum(v2, 0)[:, 0].detach())
    if want_rgb:                                                          # ibrnet.py:506-512
        c = torch.cat([h, v2, dd], -1)
        c = F.elu(_lin(c, sd, a + 'rgb_fc.0'))
        c = F.elu(_lin(c, sd, a + 'rgb_fc.2'))
        c = _lin(c, sd, a + 'rgb_fc.4')
        c = c.masked_fill(m == 0, -1e9)
        bw = torch.softmax(c, 0)
        out['rgb'] = torch.sum(rgb * bw, 0).reshape(rn, dn, 3)
    return out


# ----------------------------------------------------------------------------------------
# R1  sample_volume                                      ref: renderer.py:164-199
# ----------------------------------------------------------------------------------------
def sample_volume(sd, inp, res=40, dec='dist_decoder.', agg='agg_net.', debug=None):
    """inp: dict of fp32 tensors imgs[V,3,H,W], img_feats/ray_feats[V,32,fh,fw], poses[V,3,4],
    Ks[V,3,3], depth_range[V,2], bbox3d[2,3].   -> volume [1,1,res,res,res]."""
    with torch.no_grad():
        h, w = inp['imgs'].shape[-2:]
        pts = volume_query_points(res, inp['bbox3d'][0]).reshape(-1, 3)
        uv, z, mask, dirv = project_points(pts, inp['poses'], inp['Ks'], h, w)
        f_ray, rgb, f_img = gather_views(inp, uv, mask)
        hit, vis = decode_hit_vis(sd, dec, f_ray, z, mask, inp['depth_range'], 0.005, 0.005)
        qdir = torch.tensor([0., 0., 1.], dtype=F32).expand(pts.shape[0], 3)        # renderer.py:179
    o = aggregate(sd, agg, f_ray, rgb, f_img, hit, vis, mask, dirv, qdir, pts, res * res, res,
                  want_grad=False, want_rgb=False, debug=debug)
    if debug is not None:
        debug.update(mask=mask, uv=uv, z=z, hit=hit, vis=vis)
    vol = o['sdf'].reshape(1, 1, res, res, res)
    return torch.flip(vol, (-1,))                                        # renderer.py:197-198


# ----------------------------------------------------------------------------------------
# S1-S3, N1, C0, F1, R2, F2  ray rendering
# ----------------------------------------------------------------------------------------
def sample_depth(depth_range, rn, dn):
    """Uniform in disparity, endpoints exact.  ref: render_ops.py:146-170 (random_sample False)."""
    near, far = depth_range[0], depth_range[1]
    diff = 1 / far - 1 / near
    interval = diff / (dn - 1)
    val = torch.arange(1, dn - 1, dtype=F32)
    ticks = torch.cat([torch.zeros(1, dtype=F32), interval * val, diff.reshape(1)])
    return (1 / (1 / near + ticks))[None].expand(rn, dn).contiguous()


def ray_points(coords, pose, K, depth):
    """coords [rn,2], pose [3,4], K [3,3], depth [rn,dn] -> pts [rn,dn,3], qdir [rn,3].
    ref: render_ops.py:4-39 (directions unnormalised, so `depth` is camera-z)."""
    rot = pose[:, :3].t()
    trans = -rot @ pose[:, 3:]                                            # [3,1]
    hc = torch.cat([coords, torch.ones(coords.shape[0], 1, dtype=F32)], 1)
    cam = torch.inverse(K) @ hc.t()                                       # [3,rn]
    d = (rot @ cam + trans - trans).t()                                   # :22-23
    pts = trans.t()[:, None] + d[:, None] * depth[..., None]
    qdir = -d / torch.linalg.norm(d, dim=1, keepdim=True)
    return pts, qdir


def neus_alpha(sdf, grad, qdir, depth, variance):
    """ref: aggregate_net.py:105-121 (cos_anneal_ratio fixed 1.0), neus.py:15-19."""
    inv_s = torch.exp(variance * 10.0).clip(1e-6, 1e6)
    dists = torch.cat([depth[:, 1:] - depth[:, :-1], torch.full_like(depth[:, :1], 1e6)], -1)
    true_cos = torch.sum(-qdir[:, None] * grad, -1)
    iter_cos = -F.relu(-true_cos)
    nxt = sdf + iter_cos * dists * 0.5
    prv = sdf - iter_cos * dists * 0.5
    pc = torch.sigmoid(prv * inv_s)
    nc = torch.sigmoid(nxt * inv_s)
    return ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)


def alpha_to_hit_prob(alpha):
    """ref: render_ops.py:72-80 (exclusive cumprod of 1-alpha+1e-10)."""
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)
    return alpha * T[:, :-1]


def render_by_depth(sd, inp, que, depth, dec, agg, cfg, debug=None):
    """One pass (coarse or fine) over rays.  ref: renderer.py:110-138."""
    rn, dn = depth.shape
    with torch.no_grad():
        h, w = inp['imgs'].shape[-2:]
        pts, qdir = ray_points(que['coords'], que['pose'], que['K'], depth)
        pf = pts.reshape(-1, 3)
        uv, z, mask, dirv = project_points(pf, inp['poses'], inp['Ks'], h, w)
        f_ray, rgb, f_img = gather_views(inp, uv, mask)
        lo, hi = ray_half_intervals(depth, que['depth_range'])
        hit, vis = decode_hit_vis(sd, dec, f_ray, z, mask, inp['depth_range'], lo, hi)
        qd = qdir[:, None].expand(rn, dn, 3).reshape(-1, 3)
    o = aggregate(sd, agg, f_ray, rgb, f_img, hit, vis, mask, dirv, qd, pf, rn, dn)
    with torch.no_grad():
        alpha = neus_alpha(o['sdf'], o['grad'], qdir, depth, sd[agg + 'deviation_network.variance'])
        hp = alpha_to_hit_prob(alpha)
        out = {
            'sdf_values': o['sdf'][None], 'alpha_values': alpha[None], 'colors_nr': o['rgb'][None],
            'hit_prob_nr': hp[None],
            'pixel_colors_nr': torch.sum(hp[..., None] * o['rgb'], 1)[None],
            'sdf_gradient_error': torch.mean((torch.linalg.norm(o['grad'], dim=-1) - 1.0) ** 2).reshape(1, 1),
            's': sd[agg + 'deviation_network.variance'].reshape(1, 1),
            'render_depth': torch.sum(hp * depth, -1)[None],              # renderer.py:136
        }
        nv = torch.sum(mask.reshape(-1, rn, dn).int(), 0)                 # renderer.py:130-132
        out['ray_mask'] = (torch.sum((nv > cfg['ray_mask_view_num']).int(), 1)
                           > cfg['ray_mask_point_num'])[None]
        if debug is not None:
            debug.update(mask=mask, grad=o['grad'], depth=depth)
    return out


def sample_fine_depth(depth, hit_prob, depth_range, fdn, u=None, details=None):
    """Inverse-CDF resampling in normalised inverse depth.  u=None: eval mode (deterministic midpoints,
    render_ops.py:200-203); u [rn,fdn]: the is_train draws of torch.rand (render_ops.py:204-205).
    -> fine depth [rn,fdn] (unsorted), inds [rn,fdn] int64.   ref: render_ops.py:172-229.
    `details` (dict, tests): receives cdf [rn,dn+1], u, margin = min_j|u - cdf_j| [rn,fdn] (an index can only differ
    between two implementations whose cdf differ by more than this), den_raw (cdf bin width before the 1e-5 guard) and
    sens [rn,fdn] = |d depth / d cdf| of every sample (how far a cdf perturbation moves the resampled depth)."""
    near, far = -1 / depth_range[0], -1 / depth_range[1]
    d = (-1 / depth - near) / (far - near)
    centre = torch.cat([d[:, :1], (d[:, 1:] + d[:, :-1]) / 2, d[:, -1:]], -1)   # [rn,dn+1]
    hp = hit_prob + 1e-5
    pdf = hp / torch.sum(hp, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)  # [rn,dn+1]
    interval = 1 / fdn
    if u is None:
        u = (0.5 * interval + torch.arange(fdn) * interval)[None].expand(depth.shape[0], fdn)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(centre, 1, below), torch.gather(centre, 1, above)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    t = (u - c0) / den
    fd = b0 + t * (b1 - b0)
    fd = -1 / (fd * (far - near) + near)
    if details is not None:
        details.update(cdf=cdf, u=u, den_raw=c1 - c0, centre=centre,
                       margin=(u[:, :, None].double() - cdf[:, None, :].double()).abs().amin(-1).float(),
                       # fd_norm = b0 + (u-c0)/den (b1-b0): |d fd_norm / d c| <= (1 + |t|) |b1-b0| / den;  d z / d fd_norm = z^2 (far-near)
                       sens=(1 + t.abs()) * (b1 - b0).abs() / den * fd * fd * (far - near).abs())
    return fd, inds


def query_pixel_colors(que):
    """pixel_colors_gt: bilinear sample of the query image at the ray pixels, zeros padding, align_corners=True.
    que: coords [rn,2], imgs [1,3,H,W] -> [1,rn,3].   ref: renderer.py:125-127."""
    wh = torch.tensor([que['imgs'].shape[-1] - 1, que['imgs'].shape[-2] - 1], dtype=F32)
    return F.grid_sample(que['imgs'], (que['coords'] / wh * 2 - 1)[None, None], mode='bilinear', padding_mode='zeros',
                         align_corners=True)[0, :, 0].t()[None]


DEFAULT_RENDER_CFG = {'depth_sample_num': 40, 'fine_depth_sample_num': 40,
                      'ray_mask_view_num': 2, 'ray_mask_point_num': 8}


def render(sd, inp, que, cfg=None, debug=None, fine_depth_override=None, fine_u=None):
    """Coarse + fine ray rendering; eval mode unless `fine_u` [rn,fdn] supplies the is_train random draws.  `fine_depth_override` [rn,fdn] (sorted)
    teacher-forces the fine pass: inverse-CDF resampling is ill-conditioned where the coarse
    pdf is ~0 (a 3e-5 change of hit_prob moves a fine sample by ~1e-2 of the depth range), so
    tight fine-pass parity is checked on identical sample positions.  que: dict(coords[rn,2], pose[3,4], K[3,3],
    depth_range[2], optional imgs[1,3,H,W]).  Output keys/shapes follow renderer.py:90-162
    (+ '_fine' suffix).  ref: renderer.py:140-162."""
    cfg = {**DEFAULT_RENDER_CFG, **(cfg or {})}
    rn = que['coords'].shape[0]
    dbg_c = {} if debug is not None else None
    dbg_f = {} if debug is not None else None
    depth = sample_depth(que['depth_range'], rn, cfg['depth_sample_num'])
    out = render_by_depth(sd, inp, que, depth, 'dist_decoder.', 'agg_net.', cfg, dbg_c)
    f1 = {} if debug is not None else None
    fd, inds = sample_fine_depth(depth, out['hit_prob_nr'][0], que['depth_range'],
                                 cfg['fine_depth_sample_num'], fine_u, details=f1)
    if cfg.get('fine_depth_use_all', False):                               # renderer.py:145-146: coarse + resampled depths together
        fdepth = torch.sort(torch.cat([depth, fd], -1), -1)[0]
    else:
        fdepth = torch.sort(fd, -1)[0]                                     # renderer.py:148
    if fine_depth_override is not None:
        fdepth = fine_depth_override
    fine = render_by_depth(sd, inp, que, fdepth, 'fine_dist_decoder.', 'fine_agg_net.', cfg, dbg_f)
    for k, v in fine.items():
        out[k + '_fine'] = v
    if 'imgs' in que:
        gt = query_pixel_colors(que)
        out['pixel_colors_gt'] = gt
        out['pixel_colors_gt_fine'] = gt
    if debug is not None:
        debug.update(coarse=dbg_c, fine=dbg_f, fine_inds=inds, fine_depth=fdepth, coarse_depth=depth, f1=f1, fine_depth_unsorted=fd)
    return out


# ----------------------------------------------------------------------------------------
# convenience: numpy <-> torch
# ----------------------------------------------------------------------------------------
def to_torch(d):
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v)
            for k, v in d.items()}


def to_torch64(d):
    """The same fp32 values as float64 tensors (for fp64_mode)."""
    return {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in to_torch(d).items()}
