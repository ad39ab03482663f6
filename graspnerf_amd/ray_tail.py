"""Backward of the per-ray tail of the render path, including the second-order path through the in-forward SDF gradient
(ibrnet.py:485-504: geometry_fc on [mean, var, wbar, embed(p)], positional encoding, 40-token attention, LayerNorm,
out_geometry_fc, clip, and `grad = d sum(sdf) / d p` taken with create_graph=True): host-side orchestration of the three
device kernels (csrc/gnr_bwd.inc).

With upstream gradients `a = dL/d sdf` [rn,dn] and `gamma = dL/d grad` [rn,dn,3] the parameter / input gradients are those of

    Phi = sum_i a_i sdf_i  +  <gamma, grad>  =  sum_i a_i sdf_i  +  d/d eps  sum_i sdf_i(p + eps*gamma) |_0 ,

i.e. a reverse pass over the tail evaluated on dual numbers (value, directional derivative along gamma): every op is
reversed once for its value adjoint and once for its tangent adjoint, plus one cross term where the op is non-linear
(ELU'' , the softmax Jacobian's dependence on P, LayerNorm's 1/sigma).  No double-backward graph is built.
k_geo_dual_fwd (geometry_fc's two ELU layers on dual numbers) -> k_ray_dual_bwd (attention / LayerNorm / folded
out_geometry_fc / clip core) -> k_geo_dual_bwd.  The same algebra in plain tensor form, checked against autograd's double
backward, is test infrastructure: tests/reference_autograd.py (`attn_core`, `tail_backward`), tests/test_ray_tail.py."""
import torch

TAIL_KEYS = ('geometry_fc.0.weight', 'geometry_fc.0.bias', 'geometry_fc.2.weight', 'geometry_fc.2.bias',
             'ray_attention.w_qs.weight', 'ray_attention.w_ks.weight', 'ray_attention.w_vs.weight', 'ray_attention.fc.weight',
             'ray_attention.layer_norm.weight', 'ray_attention.layer_norm.bias',
             'out_geometry_fc.0.weight', 'out_geometry_fc.0.bias', 'out_geometry_fc.1.weight', 'out_geometry_fc.1.bias')


def unfold_out_geometry(P, agg, dweff, dbeff):
    """Gradients of out_geometry_fc.{0,1} from those of the folded row the kernels use (two linears without an activation,
    ibrnet.py:410-412: w_eff = W1 W0, b_eff = W1 b0 + b1) -- the transpose of the packer's fold, 16x16 weight algebra."""
    a = agg + 'agg_impl.'
    wa, ba, wb = P[a + 'out_geometry_fc.0.weight'], P[a + 'out_geometry_fc.0.bias'], P[a + 'out_geometry_fc.1.weight']
    return {a + 'out_geometry_fc.0.weight': wb.t() @ dweff[None], a + 'out_geometry_fc.0.bias': wb[0] * dbeff,
            a + 'out_geometry_fc.1.weight': (wa @ dweff + ba * dbeff)[None], a + 'out_geometry_fc.1.bias': dbeff.reshape(1)}


def tail_backward(hot, level, P, agg, stats66, pts, rn, dn, a, gamma):
    """stats66 [N,66] = (mean 32, var 32, wbar, n_valid), pts [N,3] (N = rn*dn), a [rn,dn], gamma [rn,dn,3]; P = the level's
    tail parameters by state-dict name (for the unfold only: the kernels read the packed / canonical device blobs).
    -> d stats [N,66] and {state-dict name: gradient} for geometry_fc, ray_attention, out_geometry_fc of `agg`."""
    pre = agg + 'agg_impl.'
    canon = hot.can_dev[level]
    gm = gamma.reshape(-1, 3)
    g, gd = hot.geo_dual_fwd(canon, stats66, pts, gm)
    gbar, gdbar, dt = hot.ray_tail_dual_bwd(level, g.reshape(rn, dn, 16), gd.reshape(rn, dn, 16), a, stats66[:, 65].reshape(rn, dn))
    dstats, dcan = hot.geo_dual_bwd(canon, stats66, pts, gm, gbar.reshape(-1, 16), gdbar.reshape(-1, 16))
    from . import weights as _w
    got = _w.split_canonical(dcan, level)
    grads = {pre + k: got[pre + k] for k in TAIL_KEYS[:4]}
    m = lambda k: dt[256 * k:256 * (k + 1)].reshape(16, 16)
    for k, name in enumerate(('w_qs', 'w_ks', 'w_vs', 'fc')):
        grads[pre + 'ray_attention.' + name + '.weight'] = m(k)
    grads[pre + 'ray_attention.layer_norm.weight'], grads[pre + 'ray_attention.layer_norm.bias'] = dt[1024:1040], dt[1040:1056]
    grads.update(unfold_out_geometry(P, agg, dt[1056:1072], dt[1072]))
    return dstats, grads
