// The weight packer's arithmetic, written ONCE for both sides: canonical blob (reference state-dict order) -> MFMA A-fragment blobs.
//   host:   gnr_pack.cpp instantiates it with HostExec (plain loops)           -> gnr_pack_weights, gnr_pack_weights_bwd, ...
//   device: gnr_pack_dev.hip instantiates it with DevExec (grid-strided loops)  -> gnr_pack_weights_device, ... (training: the
//           parameters move every optimiser step and never leave the device; reference: train/trainer.py:146-158 keeps them there)
// Same source, IEEE double arithmetic without contraction on both sides => the device blobs equal the host blobs bit for bit
// (tests/test_pack_device.py).  See gnr_layout.h for the execution model and the layouts.
// Parameter shapes/order: ref dist_decoder.py:64-88, aggregate_net.py:29-33, ibrnet.py:382-423.
#pragma once
#include <stdint.h>
#include <string.h>

#include "gnr_layout.h"

#ifndef GNR_HD
#define GNR_HD
#endif

namespace gnr {
namespace packer {

GNR_HD inline int nat_in(int j, int g) { return 16 * (j / 4) + 4 * g + (j % 4); }
GNR_HD inline int nat_out(int nb, int i) { return 16 * nb + i; }
// layout of the 35-wide colour feature x = [r,g,b, img_feats(32)] in 9 slots
GNR_HD inline int xfeat(int j, int g) { return j < 8 ? 3 + 8 * g + j : (g < 3 ? g : -1); }
// x-layout output rows (ray_dir_fc.2, base_fc.0^T): blocks 0,1 = img channels, block 2 = rgb on register 0 of groups 0..2
GNR_HD inline int xout(int nb, int i) {
    const int g = i >> 2, r = i & 3;
    if (nb == 0) return 3 + 8 * g + r;
    if (nb == 1) return 3 + 8 * g + 4 + r;
    return (r == 0 && g < 3) ? g : -1;
}
GNR_HD inline int first8(int, int i) { return i < 8 ? i : -1; }
// output rows in the gather layout: lane group g, register 4*nb + t  <->  ray channel 8g + 4nb + t
GNR_HD inline int gather_out(int nb, int i) { return 8 * (i / 4) + 4 * nb + (i % 4); }
// Z slots of geometry_fc.0's input: [mean(32), var(32), wbar | embed(21)]; slots 16..22: group 0 carries wbar in slot 16, group g>=1
// coordinate g-1, kind k = slot-16 of [p, sin p, cos p, sin 2p, cos 2p, sin 4p, cos 4p]  (neus.py:37-45)
GNR_HD inline int zslot(int j, int g) {
    if (j < 8) return nat_in(j, g);
    if (j < 16) return 32 + nat_in(j - 8, g);
    const int k = j - 16;
    if (g == 0) return k == 0 ? 64 : -1;
    return 65 + 3 * k + (g - 1);
}
// statistic slot s (0..35) of lane group g <-> base_fc.0 column 35*(s/9) + xfeat(s%9, g)
GNR_HD inline int sslot(int s, int g) { const int x = xfeat(s % 9, g); return x < 0 ? -1 : 35 * (s / 9) + x; }

// Scaled-ELU convention (saves one VALU multiply per activation in k_chain): a layer that feeds an ELU emits
// x' = log2(e) * (W x + b), the kernel computes u~ = med3(x', log2e*(2^x' - 1), 0) = log2e * ELU(x), and every consumer of u~ has
// the factor divided out of its weight columns.  `oscale` multiplies the rows (and bias), `iscale(i)` is the factor carried by
// logical input i (weights are divided by it).
constexpr double LOG2E = 1.4426950408889634;
struct ScaleTrue { GNR_HD double operator()(int) const { return 1.0; } };
struct ScaleTilde { GNR_HD double operator()(int) const { return LOG2E; } };

struct HostExec {
    template <class F> void run(int n, F f) const { for (int t = 0; t < n; ++t) f(t); }
};
struct DevExec {
    int tid, nthreads;
    template <class F> GNR_HD void run(int n, F f) const { for (int t = tid; t < n; t += nthreads) f(t); }
};

// row-major weight matrix as the element accessor of pack_frag
struct Mat {
    const float* W; int ld;
    GNR_HD double operator()(int o, int i) const { return (double)W[o * ld + i]; }
};
// its transpose: element (o, i) of W^T where W is [rows][cols] row-major (o indexes W's columns)
struct MatT {
    const float* W; int ld;
    GNR_HD double operator()(int o, int i) const { return (double)W[i * ld + o]; }
};

GNR_HD inline int frag_index(int NB, int j, int nb, int lane) {
    if (NB == 1) return ((j / 4) * 64 + lane) * 4 + (j % 4);
    if (NB == 3) return (j * 64 + lane) * 4 + nb;
    return (j * 64 + lane) * NB + nb;
}

// frag[(j,nb,lane)] = oscale * W(psi(nb, lane&15), phi(j, lane>>4)) / iscale(phi)
template <class Ex, class WF, class Phi, class Psi, class IS>
GNR_HD void pack_frag(const Ex& ex, float* dst, WF wf, int J, int NB, Phi phi, Psi psi, double oscale, IS iscale) {
    ex.run(J * NB * 64, [&](int t) {
#pragma clang fp contract(off)
        const int lane = t & 63, nb = (t >> 6) % NB, j = (t >> 6) / NB;
        const int o = psi(nb, lane & 15), i = phi(j, lane >> 4);
        const float v = (o >= 0 && i >= 0) ? (float)(wf(o, i) * oscale / iscale(i)) : 0.f;
        dst[frag_index(NB, j, nb, lane)] = v;
    });
}
template <class Ex, class WF, class Phi, class Psi>
GNR_HD void pack_frag(const Ex& ex, float* dst, WF wf, int J, int NB, Phi phi, Psi psi) {
    pack_frag(ex, dst, wf, J, NB, phi, psi, 1.0, ScaleTrue());
}

template <class Ex, class BF, class Psi>
GNR_HD void pack_bias(const Ex& ex, float* dst, BF bf, int NB, Psi psi, double oscale) {
    ex.run(NB * 16, [&](int t) {
#pragma clang fp contract(off)
        const int nb = t >> 4, i = t & 15;
        const int o = psi(nb, i);
        dst[nb * 16 + i] = o >= 0 ? (float)(bf(o) * oscale) : 0.f;                 // i = 4g + reg
    });
}
struct Vec {
    const float* b;
    GNR_HD double operator()(int o) const { return (double)b[o]; }
};

// per-lane-group table of one output row over a natural-layout input of J slots: T[g][j]
template <class Ex>
GNR_HD void pack_row(const Ex& ex, float* dst, const float* wrow, int J, double iscale) {
    ex.run(4 * J, [&](int t) {
#pragma clang fp contract(off)
        const int g = t / J, j = t - g * J;
        dst[g * J + j] = (float)((double)wrow[nat_in(j, g)] / iscale);
    });
}

// ---- fp16 pairs (C16 section, gnr_layout.h) ---------------------------------------------------------------------
GNR_HD inline uint16_t f32_to_f16(float f) {                       // round to nearest even, subnormals kept, overflow -> inf
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, ef = (x >> 23) & 0xffu, mant = x & 0x7fffffu;
    if (ef == 0xffu) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0u));
    const int e = (int)ef - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        const uint32_t m = mant | 0x800000u;
        const int shift = 14 - e;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}
// exact; `finite` false for inf / NaN
GNR_HD inline float f16_to_f32(uint16_t h, bool& finite) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    finite = e != 31;
    uint32_t x;
    if (e == 0) {
        // m 2^-24: exact in fp32 (a normal number), built from the integer
        float f = (float)m * 5.9604644775390625e-08f;
        memcpy(&x, &f, 4);
        x |= sign;
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e - 15 + 127) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// which k-steps of a layer become K32 pair blocks (runs of <= 8 consecutive k-steps; a short run is zero-padded) and which stay
// fp32 fragments (k_chain's call sites use the same split)
#ifndef GNR_RDF2_PAIRS
#define GNR_RDF2_PAIRS 0      // see c16_pairs
#endif
struct C16Plan { int off, J, NB, nblk; int k0[5], kn[5]; int nrest; int rest[2]; };
constexpr int C16_PLANS = 18;
GNR_HD inline C16Plan c16_plan(int n) {
    using namespace gnr::pk;
    if (n < 6) {
        const int br = n >> 1;
        return C16Plan{((n & 1) ? DEC2 : DEC1) + br * frag_floats(8, 2), 8, 2, 1, {0}, {8}, 0, {0, 0}};
    }
    switch (n) {
        case 6: return C16Plan{PE1, 9, 2, 1, {0}, {8}, 1, {8, 0}};                                   // ray features | (hit, vis)
        case 7: return C16Plan{NR1, 8, 1, 1, {0}, {8}, 0, {0, 0}};
        case 8: return C16Plan{BASE1, 17, 4, 2, {0, 9}, {8, 8}, 1, {8, 0}};                          // x[0..7] | e1[0..7] | x[8] (rgb)
        case 9: return C16Plan{BASE2, 16, 2, 2, {0, 8}, {8, 8}, 0, {0, 0}};
        case 10: return C16Plan{VIS1, 8, 2, 1, {0}, {8}, 0, {0, 0}};
        case 11: return C16Plan{VIS2, 8, 2, 1, {0}, {8}, 0, {0, 0}};
        case 12: return C16Plan{VISB1, 8, 2, 1, {0}, {8}, 0, {0, 0}};
        case 13: return C16Plan{RGB1, 10, 1, 1, {0}, {8}, 2, {8, 9}};
        case 14: return C16Plan{HOIST, 36, 4, 5, {0, 8, 16, 24, 32}, {8, 8, 8, 8, 4}, 0, {0, 0}};    // 4-k-step tail zero-padded
        case 15: return C16Plan{GEO1, 23, 4, 3, {0, 8, 16}, {8, 8, 7}, 0, {0, 0}};                   // 7-k-step tail zero-padded
        case 16: return C16Plan{GEO2, 16, 1, 2, {0, 8}, {8, 8}, 0, {0, 0}};
        case 17: return C16Plan{DECV1, 8, 2, 1, {0}, {8}, 0, {0, 0}};
        default: return C16Plan{DECV2, 8, 2, 1, {0}, {8}, 0, {0, 0}};
    }
    // RDF1 (one k-step), RDF2 and RGB2 (4 k-steps) stay fp32
}
constexpr int C16_PLANS_ALL = 19;

// fp32 fragment of one layer (src) -> its C16 form at dst; EVERY float of the slot's pair blocks and left-over fragment is written
// (zero-padded tails of short blocks included).  *bad0 / *bad1 are set to 1.f when a weight is outside the fp16 range (its
// pair is left zero).
template <class Ex>
GNR_HD void to_pairs(const Ex& ex, float* dst, const float* src, const C16Plan& pl, float* bad0, float* bad1) {
    const int NB = pl.NB;
    uint16_t* o16 = reinterpret_cast<uint16_t*>(dst);
    int pos = 0;                                                                 // floats
    for (int b = 0; b < pl.nblk; ++b) {
        const int n = pl.kn[b], k0 = pl.k0[b];
        ex.run(NB * 64 * 8, [&](int t) {
#pragma clang fp contract(off)
            const int i = t & 7, lane = (t >> 3) & 63, nb = t >> 9;
            uint16_t h = 0, m = 0;
            if (i < n) {
                const float w = src[frag_index(NB, k0 + i, nb, lane)];
                const uint16_t hh = f32_to_f16(w);
                bool fin;
                const float hf = f16_to_f32(hh, fin);
                if (fin) { h = hh; m = f32_to_f16((w - hf) * 2048.f); }
                else { *bad0 = 1.f; *bad1 = 1.f; }
            }
            uint16_t* base = o16 + 2 * pos;
            base[((nb * 2 + 0) * 64 + lane) * 8 + i] = h;
            base[((nb * 2 + 1) * 64 + lane) * 8 + i] = m;
        });
        pos += pk::k32_floats(NB);
    }
    const int Jr = pl.nrest;
    if (Jr > 0) {
        ex.run(frag_floats(Jr, NB), [&](int t) {                                   // every float of the left-over fragment
            int jr, nb, lane;
            if (NB == 1) { jr = (t >> 8) * 4 + (t & 3); lane = (t >> 2) & 63; nb = 0; }
            else if (NB == 3) { jr = t >> 8; lane = (t >> 2) & 63; nb = t & 3; }
            else { nb = t % NB; lane = (t / NB) & 63; jr = (t / NB) >> 6; }
            dst[pos + t] = (jr < Jr && nb < NB) ? src[frag_index(NB, pl.rest[jr], nb, lane)] : 0.f;
        });
    }
}

// C16 image, step 1: the CHAIN section's slots copied to their places (two slots grown, gnr_layout.h c16_off)
template <class Ex>
GNR_HD void c16_copy_runs(const Ex& ex, float* p) {
    const int cuts[4] = {0, pk::GEO1, pk::GEO2, pk::CHAIN_END};                   // the slot behind each grown layer starts a new run
    for (int r = 0; r < 3; ++r) {
        float* d = p + pk::C16 + pk::c16_off(cuts[r]);
        const float* s = p + cuts[r];
        ex.run(cuts[r + 1] - cuts[r], [&](int t) { d[t] = s[t]; });
    }
}
// step 2: the wide layers' fragments as fp16 pairs.  A weight beyond the fp16 range (|w| >= 65 520) has no pair: the blob says so
// (T_VIS + 2, in both images) and k_chain's pair kernels hand every launch with this blob to their fp32-MFMA twins.
template <class Ex>
GNR_HD void c16_pairs(const Ex& ex, float* p) {
    for (int n = 0; n < C16_PLANS_ALL; ++n) {
        const C16Plan pl = c16_plan(n);
        to_pairs(ex, p + pk::C16 + pk::c16_off(pl.off), p + pl.off, pl, p + pk::T_VIS + 2, p + pk::C16 + pk::c16_off(pk::T_VIS) + 2);
    }
#if GNR_RDF2_PAIRS
    // measurement build (tools/ab_chain.py, profiles/r06_*_chain_ab.json): ray_dir_fc.2's 4 k-steps as ONE zero-padded K32 pair block.  The
    // C16 image has no room left (156.3 of 160 KB): the block overlays the vis_decoder's slots, so this build serves use_vis = 0 only.
    to_pairs(ex, p + pk::C16 + pk::c16_off(pk::DECV1), p + pk::RDF2, C16Plan{pk::RDF2, 4, 3, 1, {0}, {4}, 0, {0, 0}}, p + pk::T_VIS + 2,
             p + pk::C16 + pk::c16_off(pk::T_VIS) + 2);
#endif
}

// prob_embed.2 (32x32 + bias, no activation) folded into a consumer's rows: (W o P2)[o][i] = sum_k W[o][k] P2[k][i], products in
// double, rounded once to float (then treated like any other weight)
struct Folded {
    const float* W; int ld, col0;          // consumer rows: W[o*ld + col0 + k]
    const float* P2;                       // prob_embed.2.weight [32][32]
    GNR_HD double operator()(int o, int i) const {
#pragma clang fp contract(off)
        double a = 0;
        for (int k = 0; k < 32; ++k) a += (double)W[o * ld + col0 + k] * P2[k * 32 + i];
        return (double)(float)a;
    }
};
struct FoldedBias {
    const float* b; const float* W; int ld, col0; const float* bp2;
    GNR_HD double operator()(int o) const {
#pragma clang fp contract(off)
        double bb = b[o];
        for (int k = 0; k < 32; ++k) bb += (double)W[o * ld + col0 + k] * bp2[k];
        return (double)(float)bb;
    }
};
// base_fc.0 with its prob-embedding columns (175..206) multiplied by prob_embed.2
struct Base0Folded {
    const float* W; const float* P2;
    GNR_HD double operator()(int o, int i) const {
        if (i < 175) return (double)W[o * 207 + i];
        return Folded{W, 207, 175, P2}(o, i - 175);
    }
};

// fp16-pair K32 blocks straight from a weight accessor (the RM section: no fp32 fragment in between): block kb, output block nb,
// lane, element i  <-  w = wf(psi(nb, lane & 15), phi(8 kb + i, lane >> 4)), zero where either index is -1
template <class Ex, class WF, class Phi, class Psi>
GNR_HD void pack_pairs(const Ex& ex, float* dst, WF wf, int KB, int NB, Phi phi, Psi psi) {
    uint16_t* o16 = reinterpret_cast<uint16_t*>(dst);
    ex.run(KB * NB * 64 * 8, [&](int t) {
#pragma clang fp contract(off)
        const int i = t & 7, lane = (t >> 3) & 63, nb = (t >> 9) % NB, kb = (t >> 9) / NB;
        const int o = psi(nb, lane & 15), k = phi(8 * kb + i, lane >> 4);
        uint16_t h = 0, m = 0;
        if (o >= 0 && k >= 0) {
            const float w = (float)wf(o, k);
            const uint16_t hh = f32_to_f16(w);
            bool fin;
            const float hf = f16_to_f32(hh, fin);
            if (fin) { h = hh; m = f32_to_f16((w - hf) * 2048.f); }
            else h = hh;                                         // no pair (|w| >= 65 520 or not finite): +-inf / NaN in the high half -- the consumer's outputs
                                                                 // turn non-finite and its range guard recomputes them in fp32 (k_ray<true>)
        }
        o16[(((kb * NB + nb) * 2 + 0) * 64 + lane) * 8 + i] = h;
        o16[(((kb * NB + nb) * 2 + 1) * 64 + lane) * 8 + i] = m;
    });
}

// ---- forward blob: CHAIN + RAY sections from the canonical blob (everything except the position table R_PE and the C16 image)
template <class Ex>
GNR_HD void pack_forward(const Ex& ex, const float* c, float* p) {
    auto natI = [](int j, int g) { return nat_in(j, g); };
    auto natO = [](int nb, int i) { return nat_out(nb, i); };
    auto ray8 = [](int j, int g) { return 8 * g + j; };
    auto f8 = [](int nb, int i) { return first8(nb, i); };
    auto xo = [](int nb, int i) { return xout(nb, i); };
    const ScaleTrue kTrue;
    const ScaleTilde kTilde;
    ex.run(1, [&](int) { p[pk::T_VIS + 2] = 0.f; });         // "a weight has no fp16 pair": recomputed by c16_pairs

    // --- decoder: three branches, layers .0 (input = ray feature channels 8g+j) and .2
    const int d0w[3] = {can::MEAN0_W, can::VAR0_W, can::AW0_W}, d0b[3] = {can::MEAN0_B, can::VAR0_B, can::AW0_B};
    const int d2w[3] = {can::MEAN2_W, can::VAR2_W, can::AW2_W}, d2b[3] = {can::MEAN2_B, can::VAR2_B, can::AW2_B};
    for (int br = 0; br < 3; ++br) {
        pack_frag(ex, p + pk::DEC1 + br * frag_floats(8, 2), Mat{c + d0w[br], 32}, 8, 2, ray8, natO, LOG2E, kTrue);
        pack_bias(ex, p + pk::B_DEC1 + br * 32, Vec{c + d0b[br]}, 2, natO, LOG2E);
        pack_frag(ex, p + pk::DEC2 + br * frag_floats(8, 2), Mat{c + d2w[br], 32}, 8, 2, natI, natO, LOG2E, kTilde);
        pack_bias(ex, p + pk::B_DEC2 + br * 32, Vec{c + d2b[br]}, 2, natO, LOG2E);
    }
    // decoder .4 rows on the VALU: mean0 mean1 var0 var1 aw
    pack_row(ex, p + pk::T_DEC3 + 0 * 32, c + can::MEAN4_W, 8, LOG2E);
    pack_row(ex, p + pk::T_DEC3 + 1 * 32, c + can::MEAN4_W + 32, 8, LOG2E);
    pack_row(ex, p + pk::T_DEC3 + 2 * 32, c + can::VAR4_W, 8, LOG2E);
    pack_row(ex, p + pk::T_DEC3 + 3 * 32, c + can::VAR4_W + 32, 8, LOG2E);
    pack_row(ex, p + pk::T_DEC3 + 4 * 32, c + can::AW4_W, 8, LOG2E);
    ex.run(1, [&](int) {
        p[pk::T_DEC3_B + 0] = c[can::MEAN4_B]; p[pk::T_DEC3_B + 1] = c[can::MEAN4_B + 1];
        p[pk::T_DEC3_B + 2] = c[can::VAR4_B]; p[pk::T_DEC3_B + 3] = c[can::VAR4_B + 1];
        p[pk::T_DEC3_B + 4] = c[can::AW4_B];
    });

    // --- prob_embed: 34 -> 32 -> 32 ; slot 8 carries (hit', vis') on groups 0,1
    pack_frag(ex, p + pk::PE1, Mat{c + can::PE0_W, 34}, 9, 2,
              [](int j, int g) { return j < 8 ? 8 * g + j : (g == 0 ? 32 : (g == 1 ? 33 : -1)); }, natO);
    pack_bias(ex, p + pk::B_PE1, Vec{c + can::PE0_B}, 2, natO, 1.0);
    // prob_embed.2 (32x32 + bias, no activation) is folded into its two linear consumers (see gnr_layout.h)
    const float* Wp2 = c + can::PE2_W;      // [32][32]
    const float* bp2 = c + can::PE2_B;

    // --- ray_dir_fc: 4 -> 16 -> 35, output laid out like x (see xfeat)
    pack_frag(ex, p + pk::RDF1, Mat{c + can::RDF0_W, 4}, 1, 1, [](int, int g) { return g; }, natO, LOG2E, kTrue);
    pack_bias(ex, p + pk::B_RDF1, Vec{c + can::RDF0_B}, 1, natO, LOG2E);
    pack_frag(ex, p + pk::RDF2, Mat{c + can::RDF2_W, 16}, 4, 3, natI, xo, LOG2E, kTilde);
    pack_bias(ex, p + pk::B_RDF2, Vec{c + can::RDF2_B}, 3, xo, LOG2E);

    // --- neuray_fc: 32 -> 8 (MFMA) -> 1 (VALU);  neuray_fc.0 o prob_embed.2 : [8][32], input = ReLU output of prob_embed.0
    pack_frag(ex, p + pk::NR1, Folded{c + can::NR0_W, 32, 0, Wp2}, 8, 1, natI, f8, LOG2E, kTrue);
    pack_bias(ex, p + pk::B_NR1, FoldedBias{c + can::NR0_B, c + can::NR0_W, 32, 0, bp2}, 1, f8, LOG2E);
    ex.run(16, [&](int t) {
#pragma clang fp contract(off)
        p[pk::T_NR2 + t] = (t < 8) ? (float)(c[can::NR2_W + t] / LOG2E) : 0.f;
    });
    ex.run(1, [&](int) { p[pk::T_SCAL + 0] = c[can::NR2_B]; });

    // --- base_fc.0 split: view-invariant 140 columns (HOIST) + per-view 67 columns (BASE1)
    pack_frag(ex, p + pk::HOIST, Mat{c + can::BASE0_W, 207}, 36, 4, [](int j, int g) { return sslot(j, g); }, natO, LOG2E, kTrue);
    pack_bias(ex, p + pk::B_HOIST, FoldedBias{c + can::BASE0_B, c + can::BASE0_W, 207, 175, bp2}, 4, natO, LOG2E);
    pack_frag(ex, p + pk::BASE1, Base0Folded{c + can::BASE0_W, Wp2}, 17, 4,
              [](int j, int g) {
                  if (j < 9) { const int x = xfeat(j, g); return x < 0 ? -1 : 140 + x; }
                  return 175 + nat_in(j - 9, g);
              }, natO, LOG2E, kTrue);
    pack_frag(ex, p + pk::BASE2, Mat{c + can::BASE2_W, 64}, 16, 2, natI, natO, LOG2E, kTilde);
    pack_bias(ex, p + pk::B_BASE2, Vec{c + can::BASE2_B}, 2, natO, LOG2E);

    // --- vis_fc (32 -> 32 -> 32+1) and vis_fc2 (32 -> 32 -> 1)
    pack_frag(ex, p + pk::VIS1, Mat{c + can::VIS0_W, 32}, 8, 2, natI, natO, LOG2E, kTilde);
    pack_bias(ex, p + pk::B_VIS1, Vec{c + can::VIS0_B}, 2, natO, LOG2E);
    pack_frag(ex, p + pk::VIS2, Mat{c + can::VIS2_W, 32}, 8, 2, natI, natO, LOG2E, kTilde);
    pack_bias(ex, p + pk::B_VIS2, Vec{c + can::VIS2_B}, 2, natO, LOG2E);
    pack_row(ex, p + pk::T_VIS2R, c + can::VIS2_W + 32 * 32, 8, LOG2E);
    pack_frag(ex, p + pk::VISB1, Mat{c + can::VISB0_W, 32}, 8, 2, natI, natO, LOG2E, kTilde);
    pack_bias(ex, p + pk::B_VISB1, Vec{c + can::VISB0_B}, 2, natO, LOG2E);
    pack_row(ex, p + pk::T_VISB2, c + can::VISB2_W, 8, LOG2E);
    ex.run(1, [&](int) { p[pk::T_SCAL + 1] = c[can::VIS2_B + 32]; p[pk::T_SCAL + 2] = c[can::VISB2_B]; p[pk::T_SCAL + 3] = c[can::RGB4_B]; });

    // --- rgb_fc: [h(32), vis(1), dir_diff(4)] -> 16 -> 8 -> 1
    pack_frag(ex, p + pk::RGB1, Mat{c + can::RGB0_W, 37}, 10, 1,
              [](int j, int g) {
                  if (j < 8) return nat_in(j, g);
                  if (j == 8) return g == 0 ? 32 : 33 + (g - 1);
                  return g == 0 ? 36 : -1;
              }, natO, LOG2E, [](int i) { return i < 32 ? LOG2E : 1.0; });
    pack_bias(ex, p + pk::B_RGB1, Vec{c + can::RGB0_B}, 1, natO, LOG2E);
    pack_frag(ex, p + pk::RGB2, Mat{c + can::RGB2_W, 16}, 4, 1, natI, f8, LOG2E, kTilde);
    pack_bias(ex, p + pk::B_RGB2, Vec{c + can::RGB2_B}, 1, f8, LOG2E);
    ex.run(16, [&](int t) {
#pragma clang fp contract(off)
        p[pk::T_RGB3 + t] = (t < 8) ? (float)(c[can::RGB4_W + t] / LOG2E) : 0.f;
    });

    // --- geometry_fc: [mean(32), var(32), wbar, embed(21)] -> 64 -> 16
    pack_frag(ex, p + pk::GEO1, Mat{c + can::GEO0_W, 86}, 23, 4, [](int j, int g) { return zslot(j, g); }, natO, LOG2E,
              [](int i) { return i < 32 ? LOG2E : (i < 64 ? LOG2E * LOG2E : 1.0); });
    pack_bias(ex, p + pk::B_GEO1, Vec{c + can::GEO0_B}, 4, natO, LOG2E);
    pack_frag(ex, p + pk::GEO2, Mat{c + can::GEO2_W, 64}, 16, 1, natI, natO, LOG2E, kTilde);
    pack_bias(ex, p + pk::B_GEO2, Vec{c + can::GEO2_B}, 1, natO, LOG2E);

    // --- RAY section (the position table R_PE is a constant of the blob: gnr_pack.cpp fills it)
    auto copy = [&](int dst, int src, int n) { ex.run(n, [&](int t) { p[dst + t] = c[src + t]; }); };
    copy(pk::R_WQ, can::WQ, 256); copy(pk::R_WK, can::WK, 256); copy(pk::R_WV, can::WV, 256); copy(pk::R_WFC, can::WFC, 256);
    copy(pk::R_LNW, can::LN_W, 16); copy(pk::R_LNB, can::LN_B, 16);
    copy(pk::R_OUT0W, can::OUT0_W, 256); copy(pk::R_OUT0B, can::OUT0_B, 16); copy(pk::R_OUT1W, can::OUT1_W, 16);
    copy(pk::R_OUT1B, can::OUT1_B, 1);
    copy(pk::R_GEO2W, can::GEO2_W, 1024);
    ex.run(64 * 21, [&](int t) { const int h = t / 21, e = t - h * 21; p[pk::R_GEO1E + h * 24 + e] = c[can::GEO0_W + h * 86 + 65 + e]; });
    copy(pk::R_VARIANCE, can::VARIANCE, 1);
    ex.run(256, [&](int t) {
        const int i = t >> 4, o = t & 15;
        p[pk::R_WQT + i * 16 + o] = c[can::WQ + o * 16 + i];
        p[pk::R_WKT + i * 16 + o] = c[can::WK + o * 16 + i];
        p[pk::R_WVT + i * 16 + o] = c[can::WV + o * 16 + i];
        p[pk::R_WFCT + i * 16 + o] = c[can::WFC + o * 16 + i];
    });
    ex.run(1024, [&](int t) { const int h = t >> 4, o = t & 15; p[pk::R_GEO2WT + h * 16 + o] = c[can::GEO2_W + o * 64 + h]; });
    ex.run(16, [&](int i) {
#pragma clang fp contract(off)
        double acc = 0;
        for (int f = 0; f < 16; ++f) acc += (double)c[can::OUT0_W + f * 16 + i] * (double)c[can::OUT1_W + f];
        p[pk::R_OUTVJP + i] = (float)acc;
    });
    ex.run(1, [&](int) {
#pragma clang fp contract(off)
        double acc = c[can::OUT1_B];
        for (int f = 0; f < 16; ++f) acc += (double)c[can::OUT1_W + f] * (double)c[can::OUT0_B + f];
        p[pk::R_OUTB] = (float)acc;
    });
    // --- RM section: the tail of k_ray<true>'s in-forward VJP as pair blocks (gnr_layout.h)
    auto k64 = [](int j, int g) { return 32 * (j >> 3) + 8 * g + (j & 7); };                     // input 32 b + 8 g + e of [dQ | dK | dV | dy]
    auto k16d = [](int j, int g) { return j < 4 ? 4 * g + j : -1; };                             // the D layout of RM_DC's output as RM_GEOA's B operand
    auto khid = [](int j, int g) { const int b = j >> 3, e = j & 7; return 16 * (2 * b + (e >> 2)) + 4 * g + (e & 3); };   // D layout of RM_GEOA's output
    const float* wqkv = c + can::WQ;                                                             // [Wq; Wk; Wv] rows 0..47 (contiguous in the blob)
    pack_pairs(ex, p + pk::RM_DC, [wqkv](int o, int k) { return k < 48 ? wqkv[k * 16 + o] : (k - 48 == o ? 1.f : 0.f); }, 2, 1, k64, natO);
    pack_pairs(ex, p + pk::RM_GEOA, MatT{c + can::GEO2_W, 64}, 1, 4, k16d, natO);               // (o = h, k = c): geometry_fc.2.weight[c][h]
    pack_pairs(ex, p + pk::RM_GEOB, MatT{c + can::GEO0_W + 65, 86}, 2, 2, khid, [](int nb, int i) { const int e = 16 * nb + i; return e < 21 ? e : -1; });
}

// The optional fourth decoder branch of a level (dist_decoder_cfg.use_vis: true, dist_decoder.py:89-97,103-104,133-134) into an
// already packed blob.  v = vis_decoder.{0.weight [32][32], 0.bias [32], 2.weight [32][32], 2.bias [32], 4.weight [1][32],
// 4.bias [1]} in state-dict order (2 145 floats).  Sets the flag k_chain tests.
template <class Ex>
GNR_HD void pack_vis(const Ex& ex, const float* v, float* p) {
    auto natI = [](int j, int g) { return nat_in(j, g); };
    auto natO = [](int nb, int i) { return nat_out(nb, i); };
    auto ray8 = [](int j, int g) { return 8 * g + j; };
    pack_frag(ex, p + pk::DECV1, Mat{v, 32}, 8, 2, ray8, natO, LOG2E, ScaleTrue());
    pack_bias(ex, p + pk::B_DECV1, Vec{v + 1024}, 2, natO, LOG2E);
    pack_frag(ex, p + pk::DECV2, Mat{v + 1056, 32}, 8, 2, natI, natO, LOG2E, ScaleTilde());
    pack_bias(ex, p + pk::B_DECV2, Vec{v + 2080}, 2, natO, LOG2E);
    pack_row(ex, p + pk::T_DECV3, v + 2112, 8, LOG2E);
    ex.run(1, [&](int) { p[pk::T_VIS] = v[2144]; p[pk::T_VIS + 1] = 1.f; });
}

// ---- backward blob: transposed fragments for the backward twins (gnr_layout.h, namespace pkb); true scale
template <class Ex>
GNR_HD void pack_backward(const Ex& ex, const float* c, float* p) {
    auto natI = [](int j, int g) { return nat_in(j, g); };
    auto natO = [](int nb, int i) { return nat_out(nb, i); };
    auto gatherO = [](int nb, int i) { return gather_out(nb, i); };
    pack_frag(ex, p + pkb::DM_W2T, MatT{c + can::MEAN2_W, 32}, 8, 2, natI, natO);
    pack_frag(ex, p + pkb::DM_W1T, MatT{c + can::MEAN0_W, 32}, 8, 2, natI, gatherO);
    // geometry_fc.2^T [64][16], geometry_fc.0^T [86][64]; output rows in the Z-slot layout of the forward
    pack_frag(ex, p + pkb::GEO2T, MatT{c + can::GEO2_W, 64}, 4, 4, natI, natO);
    pack_frag(ex, p + pkb::GEO1T_A, MatT{c + can::GEO0_W, 86}, 16, 4, natI, [](int nb, int i) { return zslot(4 * nb + (i & 3), i >> 2); });
    pack_frag(ex, p + pkb::GEO1T_B, MatT{c + can::GEO0_W, 86}, 16, 1, natI, [](int, int i) { return zslot(16 + (i & 3), i >> 2); });
    // second view loop
    pack_frag(ex, p + pkb::PE2F, Mat{c + can::PE2_W, 32}, 8, 2, natI, natO);
    pack_bias(ex, p + pkb::B_PE2, Vec{c + can::PE2_B}, 2, natO, 1.0);
    pack_frag(ex, p + pkb::VISB1T, MatT{c + can::VISB0_W, 32}, 8, 2, natI, natO);
    pack_frag(ex, p + pkb::VIS2T, MatT{c + can::VIS2_W, 32}, 8, 2, natI, natO);          // rows 0..31 of the 33 (first 32x32 block)
    pack_frag(ex, p + pkb::VIS1T, MatT{c + can::VIS0_W, 32}, 8, 2, natI, natO);
    pack_frag(ex, p + pkb::BASE2T, MatT{c + can::BASE2_W, 64}, 8, 4, natI, natO);
    pack_frag(ex, p + pkb::BASE1XT, MatT{c + can::BASE0_W + 140, 207}, 16, 3, natI, [](int nb, int i) { return xout(nb, i); });
    pack_frag(ex, p + pkb::BASE1ET, MatT{c + can::BASE0_W + 175, 207}, 16, 2, natI, natO);
    // hoisted columns: statistic slot s (0..35) of lane group g <-> column 35*(s/9) + xfeat(s%9, g)
    pack_frag(ex, p + pkb::HOISTT_A, MatT{c + can::BASE0_W, 207}, 16, 4, natI, [](int nb, int i) { return sslot(4 * nb + (i & 3), i >> 2); });
    pack_frag(ex, p + pkb::HOISTT_B, MatT{c + can::BASE0_W, 207}, 16, 4, natI, [](int nb, int i) { return sslot(16 + 4 * nb + (i & 3), i >> 2); });
    pack_frag(ex, p + pkb::HOISTT_C, MatT{c + can::BASE0_W, 207}, 16, 1, natI, [](int, int i) { return sslot(32 + (i & 3), i >> 2); });
    // first view loop
    const int d0w[3] = {can::MEAN0_W, can::VAR0_W, can::AW0_W}, d2w[3] = {can::MEAN2_W, can::VAR2_W, can::AW2_W};
    for (int br = 0; br < 3; ++br) {
        pack_frag(ex, p + pkb::DEC2T + br * 1024, MatT{c + d2w[br], 32}, 8, 2, natI, natO);
        pack_frag(ex, p + pkb::DEC1T + br * 1024, MatT{c + d0w[br], 32}, 8, 2, natI, gatherO);
    }
    pack_frag(ex, p + pkb::V1_PE2F, Mat{c + can::PE2_W, 32}, 8, 2, natI, natO);
    pack_bias(ex, p + pkb::V1_B_PE2, Vec{c + can::PE2_B}, 2, natO, 1.0);
    pack_frag(ex, p + pkb::PE2T, MatT{c + can::PE2_W, 32}, 8, 2, natI, natO);
    pack_frag(ex, p + pkb::PE0T, MatT{c + can::PE0_W, 34}, 8, 2, natI, gatherO);          // columns 0..31 of the 34
    ex.run(64, [&](int t) {                                                             // prob_embed.0[:, 32], [:, 33] as [4][8] row tables
        const int which = t >> 5, q = t & 31, g = q >> 3, j = q & 7;
        p[pkb::T_PE0HV + which * 32 + g * 8 + j] = c[can::PE0_W + nat_in(j, g) * 34 + 32 + which];
    });
    pack_frag(ex, p + pkb::NR0T, MatT{c + can::NR0_W, 32}, 4, 2, [](int j, int g) { const int f = 4 * g + j; return f < 8 ? f : -1; }, natO);
    pack_frag(ex, p + pkb::RDF2T, MatT{c + can::RDF2_W, 16}, 9, 1, [](int j, int g) { return xfeat(j, g); }, natO);
    // colour head
    pack_frag(ex, p + pkb::RGB2T, MatT{c + can::RGB2_W, 16}, 4, 1, [](int j, int g) { const int f = 4 * g + j; return f < 8 ? f : -1; }, natO);
    pack_frag(ex, p + pkb::RGB0HT, MatT{c + can::RGB0_W, 37}, 4, 2, natI, natO);          // columns 0..31 of the 37
    ex.run(16, [&](int t) { p[pkb::T_RGB0V + t] = c[can::RGB0_W + t * 37 + 32]; });
}

// fp16-pair images of the backward blob (pkb::P2_BEGIN ..): second phase of the backward pack, reads the fp32 fragments
// pack_backward wrote (on the device: its own launch)
GNR_HD inline C16Plan pair_plan(int J, int NB) {
    C16Plan pl{0, J, NB, J / 8, {0, 8, 16, 24, 32}, {8, 8, 8, 8, 8}, 0, {0, 0}};
    return pl;
}
template <class Ex>
GNR_HD void pack_backward_pairs(const Ex& ex, float* p) {
    float* bad = p + pkb::P2_B_PE2 + 31;           // (a transposed weight beyond the fp16 range is a forward weight beyond it: the forward
    (void)bad;                                     //  blob's flag already sends every launch to the fp32 kernels; nothing to record here)
    float sink0 = 0.f, sink1 = 0.f;
    auto pairs = [&](int dst, int src, int J, int NB) { to_pairs(ex, p + dst, p + src, pair_plan(J, NB), &sink0, &sink1); };
    pairs(pkb::P2_PE2F, pkb::PE2F, 8, 2);
    ex.run(32, [&](int t) { p[pkb::P2_B_PE2 + t] = p[pkb::B_PE2 + t]; p[pkb::P1_B_PE2 + t] = p[pkb::V1_B_PE2 + t]; });
    pairs(pkb::P2_VISB1T, pkb::VISB1T, 8, 2);
    pairs(pkb::P2_VIS2T, pkb::VIS2T, 8, 2);
    pairs(pkb::P2_VIS1T, pkb::VIS1T, 8, 2);
    pairs(pkb::P2_BASE2T, pkb::BASE2T, 8, 4);
    pairs(pkb::P2_BASE1XT, pkb::BASE1XT, 16, 3);
    pairs(pkb::P2_BASE1ET, pkb::BASE1ET, 16, 2);
    for (int br = 0; br < 3; ++br) {
        pairs(pkb::P1_DEC2T + br * pk::k32_floats(2), pkb::DEC2T + br * 1024, 8, 2);
        pairs(pkb::P1_DEC1T + br * pk::k32_floats(2), pkb::DEC1T + br * 1024, 8, 2);
    }
    pairs(pkb::P1_PE2F, pkb::V1_PE2F, 8, 2);
    pairs(pkb::P1_PE2T, pkb::PE2T, 8, 2);
    pairs(pkb::P1_PE0T, pkb::PE0T, 8, 2);
}
template <class Ex>
GNR_HD void pack_vis_backward_pairs(const Ex& ex, float* p) {
    float sink0 = 0.f, sink1 = 0.f;
    to_pairs(ex, p + pkb::P_DECV2T, p + pkb::DECV2T, pair_plan(8, 2), &sink0, &sink1);
    to_pairs(ex, p + pkb::P_DECV1T, p + pkb::DECV1T, pair_plan(8, 2), &sink0, &sink1);
}

// The fourth decoder branch's transposed fragments (use_vis training) into a blob of pack_backward; output rows of
// vis_decoder.0^T land in the gather layout like DEC1T.
template <class Ex>
GNR_HD void pack_vis_backward(const Ex& ex, const float* v, float* p) {
    auto natI = [](int j, int g) { return nat_in(j, g); };
    auto natO = [](int nb, int i) { return nat_out(nb, i); };
    pack_frag(ex, p + pkb::DECV2T, MatT{v + 1056, 32}, 8, 2, natI, natO);
    pack_frag(ex, p + pkb::DECV1T, MatT{v, 32}, 8, 2, natI, [](int nb, int i) { return gather_out(nb, i); });
}

}  // namespace packer
}  // namespace gnr
