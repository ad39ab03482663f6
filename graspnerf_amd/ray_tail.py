"""Backward of the per-ray tail of the render path, including the second-order path through the in-forward SDF gradient
(ibrnet.py:485-504: geometry_fc on [mean, var, wbar, embed(p)], positional encoding, 40-token attention, LayerNorm,
out_geometry_fc, clip, and `grad = d sum(sdf) / d p` taken with create_graph=True).

With upstream gradients `a = dL/d sdf` [rn,dn] and `gamma = dL/d grad` [rn,dn,3] the parameter / input gradients are those of

    Phi = sum_i a_i sdf_i  +  <gamma, grad>  =  sum_i a_i sdf_i  +  d/d eps  sum_i sdf_i(p + eps*gamma) |_0 ,

i.e. a reverse pass over the tail evaluated on dual numbers (value, directional derivative along gamma): every op is
reversed once for its value adjoint and once for its tangent adjoint, plus one cross term where the op is non-linear
(ELU'' , the softmax Jacobian's dependence on P, LayerNorm's 1/sigma).  No double-backward graph is built.

`tail_backward` is the whole thing in plain tensor algebra; its attention / LayerNorm core (`attn_core`) is what
csrc/gnr_bwd.inc's `k_ray_dual_bwd` implements per ray.  Everything is checked against autograd's double backward of
autograd_path.sdf_tail (tests/test_ray_tail.py)."""
import torch
import torch.nn.functional as F

from .autograd_path import sinusoid_on

TAIL_KEYS = ('geometry_fc.0.weight', 'geometry_fc.0.bias', 'geometry_fc.2.weight', 'geometry_fc.2.bias',
             'ray_attention.w_qs.weight', 'ray_attention.w_ks.weight', 'ray_attention.w_vs.weight', 'ray_attention.fc.weight',
             'ray_attention.layer_norm.weight', 'ray_attention.layer_norm.bias',
             'out_geometry_fc.0.weight', 'out_geometry_fc.0.bias', 'out_geometry_fc.1.weight', 'out_geometry_fc.1.bias')


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


def hip_core(hot, level):
    """attn_core on the device: csrc/gnr_bwd.inc k_ray_dual_bwd through the C ABI (weights = the level's packed blob)."""
    def core(W, g, gd, a, nvalid):
        gbar, gdbar, dt = hot.ray_tail_dual_bwd(level, g, gd, a, nvalid)
        m = lambda k: dt[256 * k:256 * (k + 1)].reshape(16, 16)
        return gbar, gdbar, {'wq': m(0), 'wk': m(1), 'wv': m(2), 'wfc': m(3), 'lnw': dt[1024:1040], 'lnb': dt[1040:1056],
                             'weff': dt[1056:1072], 'beff': dt[1072]}
    return core


def hip_geo(hot, level, canon, split):
    """geometry_fc's two layers on dual numbers on the device: (fwd, bwd) around the attention core.  canon = the level's
    canonical blob on the device, split(dcan) -> {state-dict name without the 'nr_net.' prefix: gradient}."""
    return (lambda st66, pts, gamma: hot.geo_dual_fwd(canon, st66, pts, gamma),
            lambda st66, pts, gamma, gbar, gdbar: hot.geo_dual_bwd(canon, st66, pts, gamma, gbar, gdbar), split)


def _tail_backward_device(P, agg, stats, nvalid, pts, rn, dn, a, gamma, core, geo):
    fwd, bwd, split = geo
    pre = agg + 'agg_impl.'
    st66 = torch.cat([stats, nvalid[:, None]], 1) if stats.shape[1] == 65 else stats
    gm = gamma.reshape(-1, 3)
    g, gd = fwd(st66, pts, gm)
    gbar, gdbar, G = core(tail_weights(P, agg), g.reshape(rn, dn, 16), gd.reshape(rn, dn, 16), a, nvalid.reshape(rn, dn))
    dstats, dcan = bwd(st66, pts, gm, gbar.reshape(-1, 16), gdbar.reshape(-1, 16))
    got = split(dcan)
    grads = {pre + k: got[pre + k] for k in TAIL_KEYS[:4]}
    for k, name in (('wq', 'w_qs'), ('wk', 'w_ks'), ('wv', 'w_vs'), ('wfc', 'fc')):
        grads[pre + 'ray_attention.' + name + '.weight'] = G[k]
    grads[pre + 'ray_attention.layer_norm.weight'], grads[pre + 'ray_attention.layer_norm.bias'] = G['lnw'], G['lnb']
    grads.update(unfold_out_geometry(P, agg, G['weff'], G['beff']))
    return (dstats[:, :65] if stats.shape[1] == 65 else dstats), grads


def tail_weights(P, agg):
    """The tail's parameters under the names attn_core uses (out_geometry_fc folded: two linears, no activation)."""
    a = agg + 'agg_impl.'
    wa, ba = P[a + 'out_geometry_fc.0.weight'], P[a + 'out_geometry_fc.0.bias']
    wb, bb = P[a + 'out_geometry_fc.1.weight'], P[a + 'out_geometry_fc.1.bias']
    return {'wq': P[a + 'ray_attention.w_qs.weight'], 'wk': P[a + 'ray_attention.w_ks.weight'],
            'wv': P[a + 'ray_attention.w_vs.weight'], 'wfc': P[a + 'ray_attention.fc.weight'],
            'lnw': P[a + 'ray_attention.layer_norm.weight'], 'lnb': P[a + 'ray_attention.layer_norm.bias'],
            'weff': (wb @ wa)[0], 'beff': (wb @ ba + bb)[0]}


def unfold_out_geometry(P, agg, dweff, dbeff):
    """Gradients of out_geometry_fc.{0,1} from those of the folded row (w_eff = W1 W0, b_eff = W1 b0 + b1)."""
    a = agg + 'agg_impl.'
    wa, ba, wb = P[a + 'out_geometry_fc.0.weight'], P[a + 'out_geometry_fc.0.bias'], P[a + 'out_geometry_fc.1.weight']
    return {a + 'out_geometry_fc.0.weight': wb.t() @ dweff[None], a + 'out_geometry_fc.0.bias': wb[0] * dbeff,
            a + 'out_geometry_fc.1.weight': (wa @ dweff + ba * dbeff)[None], a + 'out_geometry_fc.1.bias': dbeff.reshape(1)}


def tail_backward(P, agg, stats, nvalid, pts, rn, dn, a, gamma, core=attn_core, geo=None):
    """stats [N,65] = (mean 32, var 32, wbar), nvalid [N], pts [N,3] (N = rn*dn), a [rn,dn], gamma [rn,dn,3].
    -> d stats [N,65] and {state-dict name: gradient} for geometry_fc, ray_attention, out_geometry_fc of `agg`.
    `core` = attn_core or the HIP kernel's wrapper (same signature); `geo` = None (the two geometry_fc layers in tensor
    algebra, below) or hip_geo(...): k_geo_dual_fwd / k_geo_dual_bwd."""
    if geo is not None:
        return _tail_backward_device(P, agg, stats, nvalid, pts, rn, dn, a, gamma, core, geo)
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
