"""CPU checks of the C-ABI library that need no GPU:
  * libgnr.so loads and exports every symbol include/gnr.h declares,
  * the weight packer's MFMA A-fragment layout: every layer is emulated in numpy with the ISA
    lane mapping of v_mfma_f32_16x16x4_f32 (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
    D[i=4*(l>>4)+t][j=l&15]) and compared with the dense  W x + b  of the reference layer.
The feature layouts (phi/psi) are re-derived here independently from gnr_layout.h's prose."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from graspnerf_amd import _lib, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'gnr.h')).read()
    declared = set(re.findall(r'\b(gnr_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f'libgnr.so does not export {name}'
    assert set(_lib.EXPORTED) <= declared


def off(name):
    o = _lib.lib().gnr_layout_offset(name.encode())
    assert o >= 0, name
    return o


# ---- numpy emulation of one chained-MFMA layer on a 16-point tile ---------------------------
def frag_lane(packed, base, J, NB, j, nb, lane):
    if NB == 1:
        return packed[base + ((j // 4) * 64 + lane) * 4 + (j % 4)]
    if NB == 3:
        return packed[base + (j * 64 + lane) * 4 + nb]
    return packed[base + (j * 64 + lane) * NB + nb]


def emulate(packed, base, J, NB, B_in, acc=None):
    """B_in[j][lane] -> acc[nb][t][lane] following the MFMA semantics."""
    lanes = np.arange(64)
    out = np.zeros((NB, 4, 64), np.float64) if acc is None else acc.astype(np.float64).copy()
    for j in range(J):
        for nb in range(NB):
            A = np.array([frag_lane(packed, base, J, NB, j, nb, l) for l in lanes], np.float64)   # A[i][k] @ lane i+16k
            Bm = B_in[j].astype(np.float64)                                                    # B[k][col] @ lane col+16k
            for t in range(4):
                for l in lanes:
                    i, col = 4 * (l >> 4) + t, l & 15
                    out[nb, t, l] += sum(A[i + 16 * k] * Bm[col + 16 * k] for k in range(4))
    return out


def nat(j, g):
    return 16 * (j // 4) + 4 * g + (j % 4)


def xfeat(j, g):
    return 3 + 8 * g + j if j < 8 else (g if g < 3 else -1)


def to_B(x, J, phi):
    """x[point r][feature] -> B_in[j][lane] with lane=(r,g)."""
    B = np.zeros((J, 64), np.float32)
    for j in range(J):
        for l in range(64):
            f = phi(j, l >> 4)
            B[j, l] = x[l & 15, f] if f >= 0 else 0.0
    return B


def from_D(acc, NB, psi, nout):
    """acc[nb][t][lane] -> y[point][feature]."""
    y = np.full((16, nout), np.nan)
    for nb in range(NB):
        for t in range(4):
            for l in range(64):
                o = psi(nb, 4 * (l >> 4) + t)
                if o >= 0:
                    y[l & 15, o] = acc[nb, t, l]
                else:
                    assert acc[nb, t, l] == 0.0        # padded rows stay exactly zero
    assert not np.isnan(y).any()
    return y


def bias_acc(packed, boff, NB):
    acc = np.zeros((NB, 4, 64))
    for nb in range(NB):
        for l in range(64):
            for t in range(4):
                acc[nb, t, l] = packed[boff + nb * 16 + 4 * (l >> 4) + t]
    return acc


@pytest.fixture(scope='module')
def packed_and_sd(weights_np):
    can = weights.canonical_blob(weights_np, 'coarse')
    return weights.pack(can), weights_np


LOG2E = 1.4426950408889634
TRUE = lambda i: 1.0          # input carried at true scale
TILDE = lambda i: LOG2E       # input is a scaled-ELU output (log2e * ELU)

LAYERS = [
    # name, frag, bias, J, NB, key, phi(j,g) -> input idx, psi(nb,i) -> output idx, input scale, output scale
    ('dec1_var', ('DEC1', 1024), ('B_DEC1', 32), 8, 2, 'dist_decoder.var_decoder.0', lambda j, g: 8 * g + j, lambda nb, i: 16 * nb + i, TRUE, LOG2E),
    ('dec2_aw', ('DEC2', 2048), ('B_DEC2', 64), 8, 2, 'dist_decoder.aw_decoder.2', nat, lambda nb, i: 16 * nb + i, TILDE, LOG2E),
    ('pe1', ('PE1', 0), ('B_PE1', 0), 9, 2, 'agg_net.prob_embed.0',
     lambda j, g: 8 * g + j if j < 8 else (32 if g == 0 else (33 if g == 1 else -1)), lambda nb, i: 16 * nb + i, TRUE, 1.0),
    ('rdf1', ('RDF1', 0), ('B_RDF1', 0), 1, 1, 'agg_net.agg_impl.ray_dir_fc.0', lambda j, g: g, lambda nb, i: i, TRUE, LOG2E),
    ('rdf2', ('RDF2', 0), ('B_RDF2', 0), 4, 3, 'agg_net.agg_impl.ray_dir_fc.2', nat,
     lambda nb, i: (3 + 8 * (i >> 2) + (i & 3)) if nb == 0 else ((3 + 8 * (i >> 2) + 4 + (i & 3)) if nb == 1 else
                                                                 ((i >> 2) if (i & 3) == 0 and (i >> 2) < 3 else -1)), TILDE, LOG2E),
    ('base2', ('BASE2', 0), ('B_BASE2', 0), 16, 2, 'agg_net.agg_impl.base_fc.2', nat, lambda nb, i: 16 * nb + i, TILDE, LOG2E),
    ('vis1', ('VIS1', 0), ('B_VIS1', 0), 8, 2, 'agg_net.agg_impl.vis_fc.0', nat, lambda nb, i: 16 * nb + i, TILDE, LOG2E),
    ('visb1', ('VISB1', 0), ('B_VISB1', 0), 8, 2, 'agg_net.agg_impl.vis_fc2.0', nat, lambda nb, i: 16 * nb + i, TILDE, LOG2E),
    ('rgb1', ('RGB1', 0), ('B_RGB1', 0), 10, 1, 'agg_net.agg_impl.rgb_fc.0',
     lambda j, g: nat(j, g) if j < 8 else ((32 if g == 0 else 32 + g) if j == 8 else (36 if g == 0 else -1)), lambda nb, i: i,
     lambda i: LOG2E if i < 32 else 1.0, LOG2E),
    ('rgb2', ('RGB2', 0), ('B_RGB2', 0), 4, 1, 'agg_net.agg_impl.rgb_fc.2', nat, lambda nb, i: i if i < 8 else -1, TILDE, LOG2E),
    ('geo1', ('GEO1', 0), ('B_GEO1', 0), 23, 4, 'agg_net.agg_impl.geometry_fc.0',
     lambda j, g: nat(j, g) if j < 8 else (32 + nat(j - 8, g) if j < 16 else
                                           ((64 if j == 16 else -1) if g == 0 else 65 + 3 * (j - 16) + (g - 1))),
     lambda nb, i: 16 * nb + i, lambda i: LOG2E if i < 32 else (LOG2E ** 2 if i < 64 else 1.0), LOG2E),
    ('geo2', ('GEO2', 0), ('B_GEO2', 0), 16, 1, 'agg_net.agg_impl.geometry_fc.2', nat, lambda nb, i: i, TILDE, LOG2E),
]


@pytest.mark.parametrize('spec', LAYERS, ids=[l[0] for l in LAYERS])
def test_layer_fragments(spec, packed_and_sd):
    packed, sd = packed_and_sd
    name, (fname, fadd), (bname, badd), J, NB, key, phi, psi, iscale, oscale = spec
    W, b = sd[key + '.weight'], sd[key + '.bias']
    rng = np.random.default_rng(7)
    x = rng.standard_normal((16, W.shape[1])).astype(np.float32)
    # what the kernel holds in registers: true inputs times the scale they are carried at
    xk = (x * np.array([iscale(i) for i in range(W.shape[1])])).astype(np.float32)
    acc = emulate(packed, off(fname) + fadd, J, NB, to_B(xk, J, phi), bias_acc(packed, off(bname) + badd, NB))
    y = from_D(acc, NB, psi, W.shape[0])
    np.testing.assert_allclose(y, oscale * (x.astype(np.float64) @ W.T.astype(np.float64) + b), rtol=2e-5, atol=2e-5)


def test_neuray_gate_with_folded_prob_embed2(packed_and_sd):
    """NR1 fragments = neuray_fc.0 o prob_embed.2 applied to e1 = ReLU(prob_embed.0(...))."""
    packed, sd = packed_and_sd
    Wn, bn = sd['agg_net.agg_impl.neuray_fc.0.weight'], sd['agg_net.agg_impl.neuray_fc.0.bias']
    Wp, bp = sd['agg_net.prob_embed.2.weight'], sd['agg_net.prob_embed.2.bias']
    e1 = np.abs(np.random.default_rng(9).standard_normal((16, 32))).astype(np.float32)
    acc = emulate(packed, off('NR1'), 8, 1, to_B(e1, 8, nat), bias_acc(packed, off('B_NR1'), 1))
    y = from_D(acc, 1, lambda nb, i: i if i < 8 else -1, 8)
    e = e1.astype(np.float64) @ Wp.T.astype(np.float64) + bp
    np.testing.assert_allclose(y, LOG2E * (e @ Wn.T.astype(np.float64) + bn), rtol=2e-5, atol=2e-5)


def test_base_fc0_split(packed_and_sd):
    """HOIST (140 view-invariant columns + bias) + BASE1 (x 35, e1 32 with prob_embed.2 folded in) == base_fc.0 on the
    207-wide concat [glob, x, prob_embed.2(e1)]."""
    packed, sd = packed_and_sd
    W, b = sd['agg_net.agg_impl.base_fc.0.weight'], sd['agg_net.agg_impl.base_fc.0.bias']
    Wp, bp = sd['agg_net.prob_embed.2.weight'], sd['agg_net.prob_embed.2.bias']
    rng = np.random.default_rng(3)
    z = rng.standard_normal((16, 207)).astype(np.float32)
    B_h = to_B(z, 36, lambda j, g: (35 * (j // 9) + xfeat(j % 9, g)) if xfeat(j % 9, g) >= 0 else -1)
    G = emulate(packed, off('HOIST'), 36, 4, B_h, bias_acc(packed, off('B_HOIST'), 4))
    B_v = to_B(z, 17, lambda j, g: ((140 + xfeat(j, g)) if xfeat(j, g) >= 0 else -1) if j < 9 else 175 + nat(j - 9, g))
    acc = emulate(packed, off('BASE1'), 17, 4, B_v, G)
    y = from_D(acc, 4, lambda nb, i: 16 * nb + i, 64)
    zt = z.astype(np.float64).copy()
    zt[:, 175:] = z[:, 175:].astype(np.float64) @ Wp.T.astype(np.float64) + bp      # what the reference concatenates
    np.testing.assert_allclose(y, LOG2E * (zt @ W.T.astype(np.float64) + b), rtol=2e-5, atol=2e-5)


def test_vis_fc2_rows_and_tables(packed_and_sd):
    """vis_fc.2: rows 0..31 via MFMA fragments, row 32 via the per-group VALU table; the other
    1-row layers (decoder .4, neuray_fc.2, vis_fc2.2, rgb_fc.4) likewise."""
    packed, sd = packed_and_sd
    rng = np.random.default_rng(5)
    h = rng.standard_normal((16, 32)).astype(np.float32)

    def table_dot(toff, J, x, width):
        # lane (r,g) holds x[r][nat(j,g)] for j<J ; table T[g][j]; sum over the 4 groups
        out = np.zeros(16)
        for r in range(16):
            for g in range(4):
                for j in range(J):
                    f = nat(j, g)
                    out[r] += packed[toff + g * J + j] * (x[r, f] if f < width else 0.0)
        return out

    W, b = sd['agg_net.agg_impl.vis_fc.2.weight'], sd['agg_net.agg_impl.vis_fc.2.bias']
    hk = (h * LOG2E).astype(np.float32)       # every table below consumes a scaled-ELU vector
    acc = emulate(packed, off('VIS2'), 8, 2, to_B(hk, 8, nat), bias_acc(packed, off('B_VIS2'), 2))
    np.testing.assert_allclose(from_D(acc, 2, lambda nb, i: 16 * nb + i, 32), LOG2E * (h @ W[:32].T + b[:32]), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(table_dot(off('T_VIS2R'), 8, hk, 32) + packed[off('T_SCAL') + 1], h @ W[32] + b[32], rtol=1e-5, atol=1e-5)
    W2, b2 = sd['agg_net.agg_impl.vis_fc2.2.weight'], sd['agg_net.agg_impl.vis_fc2.2.bias']
    np.testing.assert_allclose(table_dot(off('T_VISB2'), 8, hk, 32) + packed[off('T_SCAL') + 2], h @ W2[0] + b2[0], rtol=1e-5, atol=1e-5)
    # decoder heads: order mean0 mean1 var0 var1 aw
    rows = [('mean_decoder', 0), ('mean_decoder', 1), ('var_decoder', 0), ('var_decoder', 1), ('aw_decoder', 0)]
    for o, (br, k) in enumerate(rows):
        Wd, bd = sd[f'dist_decoder.{br}.4.weight'], sd[f'dist_decoder.{br}.4.bias']
        np.testing.assert_allclose(table_dot(off('T_DEC3') + o * 32, 8, hk, 32) + packed[off('T_DEC3_B') + o],
                                   h @ Wd[k] + bd[k], rtol=1e-5, atol=1e-5)
    # 8-wide inputs living on groups 0,1 (rows 0..7 of a 16-row block): neuray_fc.2, rgb_fc.4
    h8 = np.zeros((16, 16), np.float32)
    h8[:, :8] = rng.standard_normal((16, 8))
    for tname, sidx, key in (('T_NR2', 0, 'agg_net.agg_impl.neuray_fc.2'), ('T_RGB3', 3, 'agg_net.agg_impl.rgb_fc.4')):
        Wr, br_ = sd[key + '.weight'], sd[key + '.bias']
        got = np.zeros(16)
        for r in range(16):
            for g in range(4):
                for t in range(4):
                    got[r] += packed[off(tname) + g * 4 + t] * h8[r, 4 * g + t] * LOG2E
        np.testing.assert_allclose(got + packed[off('T_SCAL') + sidx], h8[:, :8] @ Wr[0] + br_[0], rtol=1e-5, atol=1e-5)


def test_ray_section(packed_and_sd):
    packed, sd = packed_and_sd
    a = 'agg_net.agg_impl.'
    np.testing.assert_array_equal(packed[off('R_WQ'):off('R_WQ') + 256].reshape(16, 16), sd[a + 'ray_attention.w_qs.weight'])
    np.testing.assert_array_equal(packed[off('R_WFC'):off('R_WFC') + 256].reshape(16, 16), sd[a + 'ray_attention.fc.weight'])
    np.testing.assert_array_equal(packed[off('R_GEO2W'):off('R_GEO2W') + 1024].reshape(16, 64), sd[a + 'geometry_fc.2.weight'])
    e = packed[off('R_GEO1E'):off('R_GEO1E') + 64 * 24].reshape(64, 24)[:, :21]
    np.testing.assert_array_equal(e, sd[a + 'geometry_fc.0.weight'][:, 65:86])
    assert packed[off('R_VARIANCE')] == sd['agg_net.deviation_network.variance']
    from oracle.graspnerf_oracle import sinusoid_table
    np.testing.assert_allclose(packed[off('R_PE'):off('R_PE') + 40 * 16].reshape(40, 16), sinusoid_table(40).numpy(), atol=1e-7)


def test_pack_rejects_bad_input():
    L = _lib.lib()
    assert L.gnr_pack_weights(None, None) == -1
    with pytest.raises(ValueError):
        weights.pack(np.zeros(10, np.float32))
    assert L.gnr_layout_offset(b'NOPE') == -1


# ---- backward blob (gnr_pack_weights_bwd): transposed fragments, emulated like the forward ones -------------------
BWD_OFF = {}


def _bwd_offsets():
    """Offsets of the backward blob, mirrored from gnr_layout.h namespace pkb (checked against the blob size)."""
    if BWD_OFF:
        return BWD_OFF
    ff = lambda J, NB: (-(-J // 4)) * 256 if NB == 1 else (J * 256 if NB == 3 else J * 64 * NB)
    o, order = 0, [('DM_W2T', 1024), ('DM_W1T', 1024), ('GEO2T', ff(4, 4)), ('GEO1T_A', ff(16, 4)), ('GEO1T_B', ff(16, 1)),
                   ('PE2F', 1024), ('B_PE2', 32), ('VISB1T', 1024), ('VIS2T', 1024), ('VIS1T', 1024), ('BASE2T', ff(8, 4)),
                   ('BASE1XT', ff(16, 3)), ('BASE1ET', ff(16, 2)), ('HOISTT_A', ff(16, 4)), ('HOISTT_B', ff(16, 4)),
                   ('HOISTT_C', ff(16, 1)), ('DEC2T', 3072), ('DEC1T', 3072), ('V1_PE2F', 1024), ('V1_B_PE2', 32), ('PE2T', 1024),
                   ('PE0T', 1024), ('T_PE0HV', 64), ('NR0T', ff(4, 2)), ('RDF2T', ff(9, 1)), ('RGB2T', ff(4, 1)),
                   ('RGB0HT', ff(4, 2)), ('T_RGB0V', 16), ('DECV2T', 1024), ('DECV1T', 1024),
                   # fp16-pair images of the fragments the backward twins multiply on the f16 matrix cores (pkb::P2_*, P1_*, P_DECV*)
                   ('P2_PE2F', 1024), ('P2_B_PE2', 32), ('P2_VISB1T', 1024), ('P2_VIS2T', 1024), ('P2_VIS1T', 1024), ('P2_BASE2T', 2048),
                   ('P2_BASE1XT', 2 * 1536), ('P2_BASE1ET', 2048), ('P1_DEC2T', 3072), ('P1_DEC1T', 3072), ('P1_PE2F', 1024),
                   ('P1_B_PE2', 32), ('P1_PE2T', 1024), ('P1_PE0T', 1024), ('P_DECV2T', 1024), ('P_DECV1T', 1024)]
    for name, n in order:
        BWD_OFF[name] = o
        o += n
    assert o == _lib.lib().gnr_packed_bwd_floats()
    return BWD_OFF


def test_bwd_fragments_compute_transposed_products(weights_np):
    """dX = W^T dY through every transposed fragment, on one random 16-point tile, against numpy."""
    can = weights.canonical_blob(weights_np, 'coarse')
    rng = np.random.default_rng(0)
    vis = rng.standard_normal(_lib.lib().gnr_canonical_vis_floats()).astype(np.float32)       # use_vis: the fourth decoder branch
    assert vis.size == 2145
    O = _bwd_offsets()
    plain = weights.pack_bwd(can)
    vis_secs = [(O['DECV2T'], 2048), (O['P_DECV2T'], 2048)]                                   # fp32 fragments and their pair images
    for o, n in vis_secs:
        assert np.array_equal(plain[o:o + n], np.zeros(n, np.float32))                        # absent: its sections stay zero
    pb = weights.pack_bwd(can, vis)
    keep = np.ones(pb.size, bool)
    for o, n in vis_secs:
        keep[o:o + n] = False
    assert np.array_equal(pb[keep], plain[keep])
    W = lambda k: weights_np['agg_net.agg_impl.' + k]
    gather = lambda nb, i: 8 * (i // 4) + 4 * nb + (i % 4)
    natO = lambda nb, i: 16 * nb + i
    xout = lambda nb, i: (3 + 8 * (i >> 2) + (i & 3)) if nb == 0 else ((3 + 8 * (i >> 2) + 4 + (i & 3)) if nb == 1 else ((i >> 2) if (i & 3) == 0 and (i >> 2) < 3 else -1))
    first8 = lambda j, g: (4 * g + j) if 4 * g + j < 8 else -1

    def zslot(j, g):
        if j < 8:
            return nat(j, g)
        if j < 16:
            return 32 + nat(j - 8, g)
        return (64 if j == 16 else -1) if g == 0 else 65 + 3 * (j - 16) + (g - 1)
    sslot = lambda s, g: -1 if xfeat(s % 9, g) < 0 else 35 * (s // 9) + xfeat(s % 9, g)
    cases = [   # name, W^T as [out][in], J, NB, phi (input layout), psi (output layout), n_out
        ('DM_W2T', weights_np['dist_decoder.mean_decoder.2.weight'].T, 8, 2, nat, natO, 32),
        ('DM_W1T', weights_np['dist_decoder.mean_decoder.0.weight'].T, 8, 2, nat, gather, 32),
        ('GEO2T', W('geometry_fc.2.weight').T, 4, 4, nat, natO, 64),
        ('VISB1T', W('vis_fc2.0.weight').T, 8, 2, nat, natO, 32),
        ('VIS2T', W('vis_fc.2.weight')[:32].T, 8, 2, nat, natO, 32),
        ('VIS1T', W('vis_fc.0.weight').T, 8, 2, nat, natO, 32),
        ('BASE2T', W('base_fc.2.weight').T, 8, 4, nat, natO, 64),
        ('BASE1XT', W('base_fc.0.weight')[:, 140:175].T, 16, 3, nat, xout, 35),
        ('BASE1ET', W('base_fc.0.weight')[:, 175:].T, 16, 2, nat, natO, 32),
        ('PE2T', weights_np['agg_net.prob_embed.2.weight'].T, 8, 2, nat, natO, 32),
        ('PE0T', weights_np['agg_net.prob_embed.0.weight'][:, :32].T, 8, 2, nat, gather, 32),
        ('NR0T', W('neuray_fc.0.weight').T, 4, 2, first8, natO, 32),
        ('RDF2T', W('ray_dir_fc.2.weight').T, 9, 1, xfeat, natO, 16),
        ('RGB2T', W('rgb_fc.2.weight').T, 4, 1, first8, natO, 16),
        ('RGB0HT', W('rgb_fc.0.weight')[:, :32].T, 4, 2, nat, natO, 32),
        ('DECV2T', vis[1056:2080].reshape(32, 32).T, 8, 2, nat, natO, 32),
        ('DECV1T', vis[:1024].reshape(32, 32).T, 8, 2, nat, gather, 32),
        ('GEO1T_A', W('geometry_fc.0.weight').T, 16, 4, nat, lambda nb, i: zslot(4 * nb + (i & 3), i >> 2), 86),
        ('HOISTT_A', W('base_fc.0.weight')[:, :140].T, 16, 4, nat, lambda nb, i: sslot(4 * nb + (i & 3), i >> 2), 140),
        ('HOISTT_C', W('base_fc.0.weight')[:, :140].T, 16, 1, nat, lambda nb, i: sslot(32 + (i & 3), i >> 2), 140),
    ]
    for name, WT, J, NB, phi, psi, nout in cases:
        nin = WT.shape[1]
        dy = rng.standard_normal((16, nin)).astype(np.float32)
        B_in = to_B(dy, J, phi)
        acc = emulate(pb, O[name], J, NB, B_in)
        want = dy.astype(np.float64) @ WT.T.astype(np.float64)           # [16][out]
        got = np.full((16, nout), np.nan)
        for nb in range(NB):
            for t in range(4):
                for l in range(64):
                    o = psi(nb, 4 * (l >> 4) + t)
                    if o is not None and o >= 0:
                        got[l & 15, o] = acc[nb, t, l]
        seen = ~np.isnan(got[0])
        assert seen.sum() >= min(nout, 16 * NB) - 8, name                 # the fragment covers its block of rows
        assert np.abs(got[:, seen] - want[:, seen]).max() < 1e-4, name


def test_training_tail_entry_points_validate_arguments():
    """The flat-list entry points of the render pass's backward refuse null pointers and sizes outside 3..128 samples before
    touching the device (status codes of include/gnr.h: -1 bad argument, -2 bad shape); no compute call, runs without a GPU."""
    L = _lib.lib()
    p = 4096                                       # any non-null address: validation happens first
    assert L.gnr_ray_tail_grad_floats() == 1073
    big = 1 << 30
    assert L.gnr_ray_tail_dual_bwd(None, p, p, p, p, p, p, p, 4, 40, p, big, 0, None) == -1
    assert L.gnr_ray_tail_dual_bwd(p, p, p, p, p, p, p, p, 4, 40, None, 0, 0, None) == -1         # the partial-sum scratch is required
    assert L.gnr_ray_tail_dual_bwd(p, p, p, p, p, p, p, p, 4, 2, p, big, 0, None) == -2
    assert L.gnr_ray_tail_dual_bwd(p, p, p, p, p, p, p, p, 4, 129, p, big, 0, None) == -2
    assert L.gnr_ray_tail_dual_bwd(p, p, p, p, p, p, p, p, 0, 40, p, big, 0, None) == -2
    assert L.gnr_ray_tail_dual_bwd(p, p, p, p, p, p, p, p, 4, 40, p, 16, 0, None) == -4            # scratch too small
    assert L.gnr_ray_tail_dual_bwd(p, p, p, p, p, p, p, p, 4, 40, p, big, 1 << 20, None) == -1     # unknown option bits
    assert L.gnr_ray_tail_dual_bwd_workspace_bytes() >= 4 * (1024 + 34)
    assert L.gnr_composite_bwd(p, p, p, p, p, p, None, None, None, None, None, p, p, p, p, 4, 40, p, big, None) == -1      # dpix is required
    assert L.gnr_composite_bwd(p, p, p, p, p, p, p, None, None, None, None, p, p, p, p, 4, 129, p, big, None) == -2
    assert L.gnr_composite_bwd(p, p, p, p, p, p, p, None, None, None, None, p, p, p, p, 4, 40, p, 4, None) == -4
    assert L.gnr_composite_bwd_workspace_bytes(130) >= 3 * 8
    assert L.gnr_geo_dual_fwd(p, p, p, None, p, p, 10, p, 1 << 20, 0, None) == -1
    assert L.gnr_geo_dual_fwd(p, p, p, p, p, p, 0, p, 1 << 20, 0, None) == -2
    assert L.gnr_geo_dual_fwd(p, p, p, p, p, p, 10, p, 16, 0, None) == -4            # scratch too small
    assert L.gnr_geo_dual_fwd(p, p, p, p, p, p, 10, p, 1 << 20, 1 << 20, None) == -1   # unknown option bits
    assert L.gnr_geo_dual_fwd_workspace_bytes() >= 64 * 1024
    assert L.gnr_geo_dual_bwd(p, p, p, p, p, p, p, None, 10, p, 1 << 20, 0, None) == -1
    assert L.gnr_geo_dual_bwd(p, p, p, p, p, p, p, p, 10, p, 16, 0, None) == -4          # scratch too small
    assert L.gnr_geo_dual_bwd(p, p, p, p, p, p, p, p, 10, p, 1 << 30, 1 << 20, None) == -1   # unknown option bits (GNR_OPT_*) are refused, not ignored
    assert L.gnr_geo_dual_bwd_workspace_bytes(10) >= 10 * 288 * 4
    assert b'geo_dual_bwd' in L.gnr_last_error()


# ---- C16 section: the same layers as fp16 pairs for v_mfma_f32_16x16x32_f16 (gnr_layout.h) --------------------------------
# ISA lane mapping of the K = 32 instruction: A[i = l & 15][k = 8 (l >> 4) + e], B[k = 8 (l >> 4) + e][j = l & 15], e = 0..7
# (8 halfs per lane), D as for the fp32 instruction.  A layer = K32 pair blocks (+ left-over fp32 k-steps).
def pair_parts(packed, base, NB, block, nb, part):
    """[64 lanes][8] float64 values of one part (0 = h, 1 = m) of a K32 block."""
    raw = packed[base + block * NB * 512: base + (block + 1) * NB * 512].view(np.float16).reshape(NB, 2, 64, 8)
    return raw[nb, part].astype(np.float64)


def emulate_pairs(packed, base, NB, blocks_in, acc):
    """blocks_in[b][e][lane] fp32 activations of K32 block b -> acc[nb][t][lane] += W x with the kernel's arithmetic: operands as
    fp16 pairs (h = fp16(x), m = fp16((x - h) 2^11)), partial products Wh xh + (Wh xm + Wm xh) 2^-11 (Wm xm dropped: GNR_SPLIT_MM 0)."""
    out = acc.astype(np.float64).copy()
    for b, xin in enumerate(blocks_in):                       # xin [8][64]
        xh = xin.astype(np.float16)
        xm = ((xin.astype(np.float32) - xh.astype(np.float32)) * np.float32(2048)).astype(np.float16)
        xh, xm = xh.astype(np.float64), xm.astype(np.float64)
        for nb in range(NB):
            wh, wm = pair_parts(packed, base, NB, b, nb, 0), pair_parts(packed, base, NB, b, nb, 1)
            for l in range(64):
                col = l & 15
                for t in range(4):
                    i = 4 * (l >> 4) + t
                    s_h = s_l = 0.0
                    for k in range(4):
                        la, lb = i + 16 * k, col + 16 * k                  # lanes holding A[i][8k..8k+7] and B[8k..8k+7][col]
                        s_h += np.dot(wh[la], xh[:, lb])
                        s_l += np.dot(wh[la], xm[:, lb]) + np.dot(wm[la], xh[:, lb])
                    out[nb, t, l] += s_h + s_l / 2048.0
    return out


C16_LAYERS = [
    # name, frag offset (+add), bias, NB, key, K32 blocks as lists of fp32 k-steps, left-over k-steps, phi, psi, iscale, oscale
    ('dec1_mean', ('DEC1', 0), ('B_DEC1', 0), 2, 'dist_decoder.mean_decoder.0', [list(range(8))], [], lambda j, g: 8 * g + j, lambda nb, i: 16 * nb + i, TRUE, LOG2E),
    ('dec2_var', ('DEC2', 1024), ('B_DEC2', 32), 2, 'dist_decoder.var_decoder.2', [list(range(8))], [], nat, lambda nb, i: 16 * nb + i, TILDE, LOG2E),
    ('pe1', ('PE1', 0), ('B_PE1', 0), 2, 'agg_net.prob_embed.0', [list(range(8))], [8],
     lambda j, g: 8 * g + j if j < 8 else (32 if g == 0 else (33 if g == 1 else -1)), lambda nb, i: 16 * nb + i, TRUE, 1.0),
    ('base2', ('BASE2', 0), ('B_BASE2', 0), 2, 'agg_net.agg_impl.base_fc.2', [list(range(8)), list(range(8, 16))], [], nat, lambda nb, i: 16 * nb + i, TILDE, LOG2E),
    ('vis2', ('VIS2', 0), ('B_VIS2', 0), 2, None, [list(range(8))], [], nat, lambda nb, i: 16 * nb + i, TILDE, LOG2E),
    ('rgb1', ('RGB1', 0), ('B_RGB1', 0), 1, 'agg_net.agg_impl.rgb_fc.0', [list(range(8))], [8, 9],
     lambda j, g: nat(j, g) if j < 8 else ((32 if g == 0 else 32 + g) if j == 8 else (36 if g == 0 else -1)), lambda nb, i: i,
     lambda i: LOG2E if i < 32 else 1.0, LOG2E),
    ('geo1', ('GEO1', 0), ('B_GEO1', 0), 4, 'agg_net.agg_impl.geometry_fc.0', [list(range(8)), list(range(8, 16)), list(range(16, 23))], [],
     lambda j, g: nat(j, g) if j < 8 else (32 + nat(j - 8, g) if j < 16 else
                                           ((64 if j == 16 else -1) if g == 0 else 65 + 3 * (j - 16) + (g - 1))),
     lambda nb, i: 16 * nb + i, lambda i: LOG2E if i < 32 else (LOG2E ** 2 if i < 64 else 1.0), LOG2E),
    ('geo2', ('GEO2', 0), ('B_GEO2', 0), 1, 'agg_net.agg_impl.geometry_fc.2', [list(range(8)), list(range(8, 16))], [], nat, lambda nb, i: i, TILDE, LOG2E),
]


@pytest.mark.parametrize('spec', C16_LAYERS, ids=[l[0] for l in C16_LAYERS])
def test_pair_fragments(spec, packed_and_sd):
    """Every kind of layer of the C16 image (what k_chain stages into LDS): K32 pair blocks + left-over fp32 k-steps, emulated
    with the kernel's fp16-pair arithmetic, against the dense fp64 layer.  The tolerance is what separates 'pairs' from
    'one fp16 operand': 3e-7 of sum|w x| (a plain fp16 operand would be off by 5e-4)."""
    packed, sd = packed_and_sd
    name, (fname, fadd), (bname, badd), NB, key, blocks, rest, phi, psi, iscale, oscale = spec
    if key is None:
        W, b = sd['agg_net.agg_impl.vis_fc.2.weight'][:32], sd['agg_net.agg_impl.vis_fc.2.bias'][:32]
    else:
        W, b = sd[key + '.weight'], sd[key + '.bias']
    rng = np.random.default_rng(11)
    x = rng.standard_normal((16, W.shape[1])).astype(np.float32)
    xk = (x * np.array([iscale(i) for i in range(W.shape[1])])).astype(np.float32)
    J = max([k for blk in blocks for k in blk] + rest) + 1
    Bin = to_B(xk, J, phi)                                                  # [j][lane]
    base = off('C16.' + fname) + fadd
    acc = bias_acc(packed, off('C16.' + bname) + badd, NB)
    pad = lambda rows: np.concatenate([rows, np.zeros((8 - len(rows), 64), rows.dtype)]) if len(rows) < 8 else rows    # short block: zero k-steps
    acc = emulate_pairs(packed, base, NB, [pad(Bin[blk]) for blk in blocks], acc)
    if rest:                                                                # left-over k-steps: plain fp32 fragments behind the pair blocks
        acc = emulate(packed, base + len(blocks) * NB * 512, len(rest), NB, Bin[rest], acc)
    y = from_D(acc, NB, psi, W.shape[0])
    ref = oscale * (x.astype(np.float64) @ W.T.astype(np.float64) + b)
    mag = oscale * (np.abs(x.astype(np.float64)) @ np.abs(W.T.astype(np.float64)) + np.abs(b))
    assert np.max(np.abs(y - ref) / mag) < 3e-7


def test_c16_image_keeps_biases_and_tables(packed_and_sd):
    packed, _ = packed_and_sd
    fe, ce = off('FRAG_END'), off('CHAIN_END')
    assert np.array_equal(packed[off('C16.FRAG_END'): off('C16.CHAIN_END')], packed[fe:ce])
    assert np.array_equal(packed[off('C16.RDF1'): off('C16.NR1')], packed[off('RDF1'): off('NR1')])       # ray_dir_fc: stays fp32
    assert np.array_equal(packed[off('C16.RGB2'): off('C16.HOIST')], packed[off('RGB2'): off('HOIST')])   # 4 k-steps x 1 block: stays fp32
    # two slots grow by their zero-padded tail blocks; the image ends the blob and fits the LDS
    assert off('C16.GEO1') - off('C16.HOIST') == 5 * 4 * 512
    assert off('C16.GEO2') - off('C16.GEO1') == 3 * 4 * 512
    assert off('TOTAL') == off('C16') + off('C16_END') and off('C16.CHAIN_END') == off('C16') + off('C16_END')
    assert off('C16_END') * 4 <= 160 * 1024


def test_packer_marks_weights_beyond_fp16_range(weights_np):
    """A weight of 7e4 has no fp16 pair: the blob carries the mark (T_VIS + 2, both images) that makes k_chain's pair kernels hand
    every launch to their fp32-MFMA twins (include/gnr.h gnr_range_status bit 2; GPU side: tests/test_range_guard.py); blobs of
    in-range weights do not."""
    ok = weights.pack(weights.canonical_blob(weights_np, 'coarse'))
    assert ok[off('T_VIS') + 2] == 0.0 and ok[off('C16.T_VIS') + 2] == 0.0
    big = dict(weights_np)
    big['agg_net.agg_impl.base_fc.2.weight'] = big['agg_net.agg_impl.base_fc.2.weight'].copy()
    big['agg_net.agg_impl.base_fc.2.weight'][0, 0] = 7.0e4
    bad = weights.pack(weights.canonical_blob(big, 'coarse'))
    assert bad[off('T_VIS') + 2] == 1.0 and bad[off('C16.T_VIS') + 2] == 1.0
    # the fp32 sections (what the twin and the backward kernels read) hold the weight itself
    assert np.isfinite(bad[:off('CHAIN_END')]).all() and np.abs(bad[:off('CHAIN_END')]).max() > 6.9e4


def test_base_fc0_split_in_pair_form(packed_and_sd):
    """C16 image of base_fc.0: HOIST = 4 K32 pair blocks + its 4-k-step tail zero-padded into a fifth; BASE1 = [x slots 0..7 | e1 slots (k-steps 9..16)] as
    two K32 pair blocks + the rgb slot (k-step 8) as one fp32 k-step -- against the dense layer on [glob, x, prob_embed.2(e1)]."""
    packed, sd = packed_and_sd
    W, b = sd['agg_net.agg_impl.base_fc.0.weight'], sd['agg_net.agg_impl.base_fc.0.bias']
    Wp, bp = sd['agg_net.prob_embed.2.weight'], sd['agg_net.prob_embed.2.bias']
    z = np.random.default_rng(4).standard_normal((16, 207)).astype(np.float32)
    B_h = to_B(z, 36, lambda j, g: (35 * (j // 9) + xfeat(j % 9, g)) if xfeat(j % 9, g) >= 0 else -1)
    tail = np.concatenate([B_h[32:36], np.zeros((4, 64), np.float32)])
    G = emulate_pairs(packed, off('C16.HOIST'), 4, [B_h[8 * k: 8 * k + 8] for k in range(4)] + [tail], bias_acc(packed, off('C16.B_HOIST'), 4))
    B_v = to_B(z, 17, lambda j, g: ((140 + xfeat(j, g)) if xfeat(j, g) >= 0 else -1) if j < 9 else 175 + nat(j - 9, g))
    acc = emulate(packed, off('C16.BASE1') + 2 * 4 * 512, 1, 4, B_v[8:9], G)
    acc = emulate_pairs(packed, off('C16.BASE1'), 4, [B_v[0:8], B_v[9:17]], acc)
    y = from_D(acc, 4, lambda nb, i: 16 * nb + i, 64)
    zt = z.astype(np.float64).copy()
    zt[:, 175:] = z[:, 175:].astype(np.float64) @ Wp.T.astype(np.float64) + bp
    ref = LOG2E * (zt @ W.T.astype(np.float64) + b)
    mag = LOG2E * (np.abs(zt) @ np.abs(W.T.astype(np.float64)) + np.abs(b))
    assert np.max(np.abs(y - ref) / mag) < 5e-7


def test_vis_decoder_branch(weights_np):
    """gnr_pack_vis_decoder (cfg use_vis): the fourth decoder branch in both images -- fp32 fragments in the CHAIN section, fp16
    pairs in the C16 image --, its .4 row as a VALU table, and the flag HotPath reads; absent -> all zero."""
    rng = np.random.default_rng(21)
    sd = dict(weights_np)
    plain = weights.pack_state_dict(sd, 'coarse')
    assert plain[off('T_VIS') + 1] == 0 and not plain[off('DECV1'): off('FRAG_END')].any()
    for k, shape in weights.VIS_KEYS:
        sd['dist_decoder.' + k] = (0.2 * rng.standard_normal(shape)).astype(np.float32)
    packed = weights.pack_state_dict(sd, 'coarse')
    assert weights.has_vis_decoder(sd, 'coarse') and not weights.has_vis_decoder(sd, 'fine')
    assert packed[off('T_VIS') + 1] == 1 and packed[off('C16.T_VIS') + 1] == 1
    assert np.array_equal(np.delete(packed, np.r_[off('DECV1'): off('FRAG_END'), off('B_DECV1'): off('B_DECV1') + 64,
                                                  off('T_DECV3'): off('T_VIS') + 8,
                                                  off('C16.DECV1'): off('C16.FRAG_END'), off('C16.B_DECV1'): off('C16.B_DECV1') + 64,
                                                  off('C16.T_DECV3'): off('C16.T_VIS') + 8]),
                          np.delete(plain, np.r_[off('DECV1'): off('FRAG_END'), off('B_DECV1'): off('B_DECV1') + 64,
                                                 off('T_DECV3'): off('T_VIS') + 8,
                                                 off('C16.DECV1'): off('C16.FRAG_END'), off('C16.B_DECV1'): off('C16.B_DECV1') + 64,
                                                 off('C16.T_DECV3'): off('C16.T_VIS') + 8])), 'nothing else moves'
    x = rng.standard_normal((16, 32)).astype(np.float32)
    W0, b0 = sd['dist_decoder.vis_decoder.0.weight'], sd['dist_decoder.vis_decoder.0.bias']
    W2, b2 = sd['dist_decoder.vis_decoder.2.weight'], sd['dist_decoder.vis_decoder.2.bias']
    for base, boff, tag in ((off('DECV1'), off('B_DECV1'), 'fp32'), (None, None, 'pairs')):
        if tag == 'fp32':
            a1 = emulate(packed, off('DECV1'), 8, 2, to_B(x, 8, lambda j, g: 8 * g + j), bias_acc(packed, off('B_DECV1'), 2))
            a2 = emulate(packed, off('DECV2'), 8, 2, to_B((x * LOG2E).astype(np.float32), 8, nat), bias_acc(packed, off('B_DECV2'), 2))
        else:
            a1 = emulate_pairs(packed, off('C16.DECV1'), 2, [to_B(x, 8, lambda j, g: 8 * g + j)], bias_acc(packed, off('C16.B_DECV1'), 2))
            a2 = emulate_pairs(packed, off('C16.DECV2'), 2, [to_B((x * LOG2E).astype(np.float32), 8, nat)], bias_acc(packed, off('C16.B_DECV2'), 2))
        np.testing.assert_allclose(from_D(a1, 2, lambda nb, i: 16 * nb + i, 32), LOG2E * (x.astype(np.float64) @ W0.T + b0), rtol=2e-5, atol=2e-5, err_msg=tag)
        np.testing.assert_allclose(from_D(a2, 2, lambda nb, i: 16 * nb + i, 32), LOG2E * (x.astype(np.float64) @ W2.T + b2), rtol=2e-5, atol=2e-5, err_msg=tag)
    # .4 row as a per-group table over a scaled-ELU input, bias next to the flag
    h = (x * LOG2E).astype(np.float32)
    out = np.zeros(16)
    for r in range(16):
        for g in range(4):
            for j in range(8):
                out[r] += packed[off('T_DECV3') + g * 8 + j] * h[r, nat(j, g)]
    np.testing.assert_allclose(out + packed[off('T_VIS')], x.astype(np.float64) @ sd['dist_decoder.vis_decoder.4.weight'][0] + sd['dist_decoder.vis_decoder.4.bias'][0],
                               rtol=2e-5, atol=2e-5)


def test_gradient_blob_key_order_with_and_without_the_vis_decoder():
    """weights.level_keys / split_canonical: the canonical order of a level (63 tensors, 36 958 floats) and, for a use_vis level, the
    vis_decoder's six tensors BEHIND it (2 145 floats more: the layout of the *_bwd entry points' gradient blobs, include/gnr.h)."""
    L = _lib.lib()
    n, nv = L.gnr_canonical_weights_floats(), L.gnr_canonical_vis_floats()
    assert (n, nv) == (36958, 2145)
    for level, dec in (('coarse', 'dist_decoder.'), ('fine', 'fine_dist_decoder.')):
        base, ext = weights.level_keys(level), weights.level_keys(level, use_vis=True)
        assert ext[:len(base)] == base and [k for k, _ in ext[len(base):]] == [dec + k for k, _ in weights.VIS_KEYS]
        flat = np.arange(n + nv, dtype=np.float32)
        parts = weights.split_canonical(flat, level, use_vis=True)
        assert list(parts) == [k for k, _ in ext]
        assert parts[dec + 'vis_decoder.0.weight'].shape == (32, 32) and parts[dec + 'vis_decoder.0.weight'][0, 0] == n
        assert parts[dec + 'vis_decoder.4.bias'].reshape(-1)[0] == n + nv - 1
        assert sum(int(np.prod(s)) if len(s) else 1 for _, s in base) == n


def test_bwd_pair_images_encode_their_fp32_fragments(weights_np):
    """The pair images of the backward blob (what k_view2_bwd / k_view1_bwd stage into LDS for their dX chains and for prob_embed.2's
    forward): block b, output block nb of a J x NB fragment holds, per lane, 8 halfs h and 8 halfs m with h + m 2^-11 = the fp32
    fragment's k-steps 8b .. 8b+7 to an fp32 ulp or two (exactly for most values); the bias copies are copies."""
    can = weights.canonical_blob(weights_np, 'coarse')
    vis = np.random.default_rng(0).standard_normal(2145).astype(np.float32)
    pb = weights.pack_bwd(can, vis)
    O = _bwd_offsets()
    fi = lambda NB, j, nb, lane: ((j // 4) * 64 + lane) * 4 + (j % 4) if NB == 1 else ((j * 64 + lane) * 4 + nb if NB == 3 else (j * 64 + lane) * NB + nb)
    cases = [('P2_PE2F', 'PE2F', 8, 2), ('P2_VISB1T', 'VISB1T', 8, 2), ('P2_VIS2T', 'VIS2T', 8, 2), ('P2_VIS1T', 'VIS1T', 8, 2),
             ('P2_BASE2T', 'BASE2T', 8, 4), ('P2_BASE1XT', 'BASE1XT', 16, 3), ('P2_BASE1ET', 'BASE1ET', 16, 2),
             ('P1_PE2F', 'V1_PE2F', 8, 2), ('P1_PE2T', 'PE2T', 8, 2), ('P1_PE0T', 'PE0T', 8, 2), ('P_DECV2T', 'DECV2T', 8, 2), ('P_DECV1T', 'DECV1T', 8, 2)] + \
            [('P1_DEC2T', 'DEC2T', 8, 2, br) for br in range(3)] + [('P1_DEC1T', 'DEC1T', 8, 2, br) for br in range(3)]
    lane = np.arange(64)
    for c in cases:
        dst, src, J, NB = c[:4]
        br = c[4] if len(c) > 4 else 0
        d0, s0 = O[dst] + br * 1024, O[src] + br * 1024
        halfs = pb[d0:d0 + (J // 8) * NB * 512].view(np.float16)
        for b in range(J // 8):
            for nb in range(NB):
                for i in range(8):
                    h = halfs[(((b * NB + nb) * 2 + 0) * 64 + lane) * 8 + i].astype(np.float64)
                    m = halfs[(((b * NB + nb) * 2 + 1) * 64 + lane) * 8 + i].astype(np.float64)
                    w = pb[s0 + np.array([fi(NB, 8 * b + i, nb, l) for l in lane])].astype(np.float64)
                    assert np.all(np.abs(h + m / 2048.0 - w) <= 2.0 ** -22 * np.abs(w) + 2.0 ** -35), (dst, b, nb, i)      # one to two fp32 ulps; below 6e-5 the halves are fp16 subnormals: 3e-11 absolute
    assert np.array_equal(pb[O['P2_B_PE2']:O['P2_B_PE2'] + 32], pb[O['B_PE2']:O['B_PE2'] + 32])
    assert np.array_equal(pb[O['P1_B_PE2']:O['P1_B_PE2'] + 32], pb[O['V1_B_PE2']:O['V1_B_PE2'] + 32])


def test_rm_section_encodes_the_vjp_tail_of_k_ray(weights_np):
    """The RM section of the forward blob (gnr_layout.h): fp16-pair K32 blocks of the three layers k_ray<true> runs on the matrix cores.
    Block b, output block nb, lane (r = lane & 15, g = lane >> 4), element e holds h + m 2^-11 = the weight of output 16 nb + r and
      RM_DC:   input 32 b + 8 g + e of [dQ | dK | dV | dy]: [Wq; Wk; Wv][k][o] for k < 48, the identity behind (the residual),
      RM_GEOA: input 4 g + e (e < 4; the D layout of RM_DC's output): geometry_fc.2.weight[c][h],
      RM_GEOB: hidden unit 16 (2 b + e / 4) + 4 g + e % 4 (the D layout of RM_GEOA's output): geometry_fc.0.weight[h][65 + o], o < 21;
    a weight without an fp16 pair (1e5) is stored as inf in the high half (k_ray's range guard then recomputes in fp32)."""
    w = dict(weights_np)
    k2 = 'agg_net.agg_impl.geometry_fc.2.weight'
    w[k2] = w[k2].copy(); w[k2][3, 5] = 1e5
    p = weights.pack_state_dict(w, 'coarse')
    att = 'agg_net.agg_impl.ray_attention.'
    Wqkv = np.concatenate([w[att + 'w_qs.weight'], w[att + 'w_ks.weight'], w[att + 'w_vs.weight']], 0).astype(np.float64)     # [48][16]
    G2 = w[k2].astype(np.float64)                                                   # [16][64]
    G0 = w['agg_net.agg_impl.geometry_fc.0.weight'].astype(np.float64)             # [64][86]

    def decode(name, KB, NB):
        halfs = p[off(name):off(name) + KB * NB * 512].view(np.float16).astype(np.float64)
        out = np.zeros((KB, NB, 64, 8))
        for b in range(KB):
            for nb in range(NB):
                for half, sc in ((0, 1.0), (1, 1.0 / 2048.0)):
                    blk = halfs[((b * NB + nb) * 2 + half) * 512:((b * NB + nb) * 2 + half + 1) * 512].reshape(64, 8)
                    out[b, nb] += blk * sc
        return out

    dc, ga, gb = decode('RM_DC', 2, 1), decode('RM_GEOA', 1, 4), decode('RM_GEOB', 2, 2)
    tol = lambda x: 2.0 ** -22 * np.abs(x) + 2.0 ** -35
    for lane in range(64):
        r, g = lane & 15, lane >> 4
        for e in range(8):
            for b in range(2):
                k = 32 * b + 8 * g + e
                want = Wqkv[k, r] if k < 48 else (1.0 if k - 48 == r else 0.0)
                assert abs(dc[b, 0, lane, e] - want) <= tol(want), ('RM_DC', b, lane, e)
                for nb in range(2):
                    h, o = 16 * (2 * b + e // 4) + 4 * g + e % 4, 16 * nb + r
                    want = G0[h, 65 + o] if o < 21 else 0.0
                    assert abs(gb[b, nb, lane, e] - want) <= tol(want), ('RM_GEOB', b, nb, lane, e)
            for nb in range(4):
                h, c = 16 * nb + r, 4 * g + e
                if e >= 4:
                    assert ga[0, nb, lane, e] == 0.0
                elif (c, h) == (3, 5):
                    assert np.isinf(ga[0, nb, lane, e]) and ga[0, nb, lane, e] > 0          # no fp16 pair
                else:
                    assert abs(ga[0, nb, lane, e] - G2[c, h]) <= tol(G2[c, h]), ('RM_GEOA', nb, lane, e)


def test_optimiser_moved_weights_fixture_is_what_the_probe_wrote(weights_np, weights_trained_np):
    """tests/golden/weights_trained_probe.npz (tools/train_probe.py: 1 000 Adam steps of this repository's trainer on the GPU; record in
    profiles/r06_g_train_probe.json): the hot-path tensors of weights_seed0 by name and shape, finite, moved by the optimiser, every one
    inside the fp16-pair range of the packer, and no sampled training step tripped the range guard or was skipped."""
    import json
    assert list(weights_trained_np) == list(weights_np)
    for k, v in weights_np.items():
        t = weights_trained_np[k]
        assert t.shape == v.shape and t.dtype == np.float32 and np.isfinite(t).all(), k
    for lvl in ('coarse', 'fine'):
        blob = weights.pack_state_dict(weights_trained_np, lvl)
        assert np.isfinite(blob).all()                              # (a weight without an fp16 pair is stored as inf by the pair packer)
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'r06_g_train_probe.json')))
    assert rec['steps'] == 1000 and rec['status_bits_or_over_sampled_steps'] == 0 and rec['skipped_optimizer_steps'] == 0
    # how far the optimiser moved the tensors from the trainer's own initialisation (synth.synth_state_dict, not weights_seed0)
    assert 0.05 < rec['median_rel_change'] < 1.0 and rec['max_rel_change'] > 0.5
    for k, t in weights_trained_np.items():
        assert abs(float(np.abs(t).max()) - rec['tensors'][k]['absmax_after']) <= 1e-6 * max(1.0, rec['tensors'][k]['absmax_after']), k
