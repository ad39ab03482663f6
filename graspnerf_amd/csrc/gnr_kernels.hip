// gfx950 (MI355X / CDNA4) kernels of the GraspNeRF volumetric hot path.
//
//   k_repack_feats   NCHW img_feats + ray_feats -> [B*V][fh][fw][64] channel-last (one 256-B
//                    line per bilinear tap)
//   k_view_setup     per view: H = K[R|t], camera centre, inverse-depth range
//   k_points_volume  per voxel centre descriptor, column order, top->down  (renderer.py:167-170)
//   k_coarse_depth / k_points_rays   ray samples (render_ops.py:4-52,146-170)
//   k_chain<V,RENDER,SAVE,USEVIS,SP> THE hot kernel (SAVE: training forward keeping the per-view states; USEVIS: 4th decoder
//                    branch; SP: fp16-pair operands, false = the fp32-MFMA twin): per-(point,view) projection + gather + mixture decoder +
//                    prob-embed + IBRNet aggregation + geometry MLP, all layers as chained MFMAs with
//                    activations resident in registers: fp32 values, multiplied on the f16 matrix cores
//                    as fp16 pairs (v_mfma_f32_16x16x32_f16; short remainders on v_mfma_f32_16x16x4_f32)
//                    (dist_decoder.py:99-142, aggregate_net.py:35-70, ibrnet.py:456-489,506-512)
//   k_ray<RENDER>    per ray / voxel column: 40-token self-attention + SDF head, and for rays the
//                    in-forward VJP, NeuS alpha, compositing, ray mask and inverse-CDF resampling
//                    (ibrnet.py:490-504, aggregate_net.py:105-121, render_ops.py:72-80,172-229)
//
// Layout conventions are documented in gnr_layout.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gnr_layout.h"

// GNR_ABLATE: timing-only diagnostic builds of k_chain (tools/ab_chain.py; results are WRONG, never shipped): bit 0 no rotation of
// the per-view state, bit 1 projection / tap arithmetic of view 0 reused for every view, bit 2 no residual half (m = 0),
// bit 3 ELU -> identity, bit 5 no cross-term fold.  What a launch loses with a piece removed is that piece's cost.
#ifndef GNR_ABLATE
#define GNR_ABLATE 0
#endif

namespace gnr {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define DEV __device__ __forceinline__

DEV f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// fp32-input MFMA and fp32 VALU share the SIMD's FMA datapath on gfx950 (tools/ubench/mfma_valu.hip:
// MFMA-only 1.23 ms + VALU-only 0.93 ms = 2.03 ms when issued by partner waves of one SIMD), so every
// VALU instruction of the chain costs throughput: activations are written for minimum instruction count.
// ELU(x) = max(x, min(e^x, 1) - 1): e^x - 1 >= x everywhere, and for x > 0 the clamped exponential gives exactly 0.  The clamp
// is the output modifier of v_exp_f32, so an activation is v_exp (clamp), v_fma, v_max (v_max issues faster than the
// v_med3 of the earlier  med3(x, e^x - 1, 0)  form; same values bit for bit).
DEV float elu1(float x) { return fmaxf(x, __builtin_amdgcn_fmed3f(__expf(x), 0.f, 1.f) - 1.f); }
// scaled form used inside the MFMA chain: input x' = log2e*x (the packer folds log2e into the producing
// layer), output log2e*ELU(x) (divided out of the consumer's weights): 3 VALU ops per activation.
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
#if GNR_ABLATE & 8
DEV float elu_s(float xs) { return xs; }
#else
DEV float elu_s(float xs) { return fmaxf(xs, fmaf(__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(xs), 0.f, 1.f), kLog2e, -kLog2e)); }
#endif
// the same activation as med3(x', log2e (2^x' - 1), 0): for inputs that come straight out of an MFMA (fmaxf would first
// canonicalise them with an extra v_max x, x)
#if GNR_ABLATE & 8
DEV float elu_m(float xs) { return xs; }
#else
DEV float elu_m(float xs) { return __builtin_amdgcn_fmed3f(xs, fmaf(__builtin_amdgcn_exp2f(xs), kLog2e, -kLog2e), 0.f); }
#endif
DEV float rcp1(float x) { return __builtin_amdgcn_rcpf(x); }
DEV float sigmoid1(float x) { return rcp1(1.f + __expf(-x)); }
// softplus = max(x,0) + log(1 + e^-|x|)   (== torch's threshold-20 form to fp32 rounding)
DEV float softplus1(float x) { return fmaxf(x, 0.f) + __logf(1.f + __expf(-fabsf(x))); }
DEV float tanh1(float x) {            // 1 - 2/(e^{2x}+1); saturates cleanly to +-1
    return 1.f - 2.f * rcp1(__expf(2.f * x) + 1.f);
}
// sin / cos of x, 2x, 4x (the embedder's three octaves, ibrnet.py:118-131): one sincosf, the octaves by the double-angle
// identities (each doubling adds <= 2 ulp of 1.0, 4e-7 absolute, on values the MLPs consume at 1e-3 relative); three precise
// sincosf cost 377 VALU per tile of k_chain.  Every forward and backward kernel uses this one function (same bits everywhere).
DEV void sincos_octaves(float x, float& s1, float& c1, float& s2, float& c2, float& s4, float& c4) {
    sincosf(x, &s1, &c1);
    s2 = 2.f * s1 * c1; c2 = fmaf(-2.f * s1, s1, 1.f);
    s4 = 2.f * s2 * c2; c4 = fmaf(-2.f * s2, s2, 1.f);
}
// sum over the 4 lane groups (lanes l, l^16, l^32, l^48); identical bits on all four
DEV float gsum(float x) {
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

// ---------------------------------------------------------------------------------------
// chained-MFMA layer:  acc[nb] += W_frag(j, nb) * in[j]   for k-steps j in [J0, J0+J)
// ---------------------------------------------------------------------------------------
// The weight image in LDS is loop-invariant; without a compiler barrier the fragment loads are hoisted out of the
// view/tile loops (LICM) and hundreds of registers spill.  Every loop iteration starts with GNR_ITER_FENCE();
// LF (template flag of mm/load_bias) additionally fences every layer: needed for V >= 7 (S[V][20] leaves too few
// registers for loads pulled across layers), 0.5 % slower for V <= 6.  -DGNR_LICM_FENCE=1 forces it everywhere.
#ifndef GNR_LICM_FENCE
#define GNR_LICM_FENCE 0
#endif
#define GNR_ITER_FENCE() asm volatile("" ::: "memory")

template <int J, int NB, int J0 = 0, bool LF = true>
DEV void mm(const float* __restrict__ w, int lane, const float (&in)[J], f4 (&acc)[NB]) {
    if constexpr (LF) asm volatile("" ::: "memory");
    if constexpr (NB == 4 || NB == 3) {
        const f4* w4 = reinterpret_cast<const f4*>(w) + J0 * 64 + lane;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const f4 a = w4[j * 64];
            acc[0] = mfma16(a.x, in[j], acc[0]);
            acc[1] = mfma16(a.y, in[j], acc[1]);
            acc[2] = mfma16(a.z, in[j], acc[2]);
            if constexpr (NB == 4) acc[3] = mfma16(a.w, in[j], acc[3]);
        }
    } else if constexpr (NB == 2) {
        const f2* w2 = reinterpret_cast<const f2*>(w) + J0 * 64 + lane;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const f2 a = w2[j * 64];
            acc[0] = mfma16(a.x, in[j], acc[0]);
            acc[1] = mfma16(a.y, in[j], acc[1]);
        }
    } else {
        static_assert(NB == 1 && J0 % 4 == 0, "NB==1 parts must start on a 4-slot boundary");
        const f4* w4 = reinterpret_cast<const f4*>(w) + (J0 / 4) * 64 + lane;
        // two interleaved accumulators hide the 40-cycle dependent-MFMA latency
        f4 alt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j4 = 0; j4 < cdiv(J, 4); ++j4) {
            const f4 a = w4[j4 * 64];
            if (4 * j4 + 0 < J) acc[0] = mfma16(a.x, in[4 * j4 + 0], acc[0]);
            if (4 * j4 + 1 < J) alt = mfma16(a.y, in[4 * j4 + 1], alt);
            if (4 * j4 + 2 < J) acc[0] = mfma16(a.z, in[4 * j4 + 2], acc[0]);
            if (4 * j4 + 3 < J) alt = mfma16(a.w, in[4 * j4 + 3], alt);
        }
        acc[0] += alt;
    }
}

template <int NB, bool LF = true>
DEV void load_bias(const float* __restrict__ b, int g, f4 (&acc)[NB]) {
    if constexpr (LF) asm volatile("" ::: "memory");
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = reinterpret_cast<const f4*>(b)[nb * 4 + g];
}

template <int NB, bool FROM_MFMA = true>
DEV void elu_to(const f4 (&acc)[NB], float (&out)[NB * 4]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if constexpr (FROM_MFMA) {
            out[nb * 4 + 0] = elu_m(acc[nb].x); out[nb * 4 + 1] = elu_m(acc[nb].y);
            out[nb * 4 + 2] = elu_m(acc[nb].z); out[nb * 4 + 3] = elu_m(acc[nb].w);
        } else {
            out[nb * 4 + 0] = elu_s(acc[nb].x); out[nb * 4 + 1] = elu_s(acc[nb].y);
            out[nb * 4 + 2] = elu_s(acc[nb].z); out[nb * 4 + 3] = elu_s(acc[nb].w);
        }
    }
}

// partial dot of a natural-layout 8-slot vector with a per-group table row T[g][8]
DEV float dot8(const float* __restrict__ T, int g, const float (&h)[8]) {
    const f4 a = reinterpret_cast<const f4*>(T)[g * 2], b = reinterpret_cast<const f4*>(T)[g * 2 + 1];
    return a.x * h[0] + a.y * h[1] + a.z * h[2] + a.w * h[3] + b.x * h[4] + b.y * h[5] + b.z * h[6] + b.w * h[7];
}
DEV float dot4(const float* __restrict__ T, int g, const float (&h)[4]) {
    const f4 a = reinterpret_cast<const f4*>(T)[g];
    return a.x * h[0] + a.y * h[1] + a.z * h[2] + a.w * h[3];
}

// ---------------------------------------------------------------------------------------
// fp16-pair layers (C16 section of the packed blob, gnr_layout.h): the same chained layers on the f16 matrix cores.
// The fp32-input MFMA runs at the fp32 VECTOR rate on gfx950 and shares its datapath (tools/ubench/mfma_valu.hip); the
// f16 MFMA is 16x faster.  Every fp32 operand is carried as a pair  x = h + m 2^-11  (h = fp16(x), m = fp16((x - h) 2^11),
// round to nearest: 23 significant bits + sign, i.e. x to 1 fp32 ulp and exactly for 3 of 4 values; x - h is exact in
// fp32), and  W x  is the sum of the four partial products  Wh xh + (Wh xm + Wm xh) 2^-11 + Wm xm 2^-22 , each exact in the
// fp32 accumulator of the matrix core.  Measured against fp64 (tools/ubench/split_mfma.hip): max error 5.3e-8 of
// sum|w x| against 1.5e-7 for the chain of eight v_mfma_f32_16x16x4_f32 (fewer roundings of the running sum), fp16
// subnormals are honoured by the MFMA, a 32->32 ELU layer takes 448 instead of 696 cycles per SIMD.
// The lane mapping is the fp32 form's: element i of lane group g of a K32 block is the block's k-step i, the output
// lands in B-operand layout again.
// ---------------------------------------------------------------------------------------
#ifndef GNR_SPLIT16
#define GNR_SPLIT16 1
#endif
#ifndef GNR_SPLIT_MM
#define GNR_SPLIT_MM 0          // 1: also the fourth partial product Wm xm 2^-22 (one more MFMA per block and a multiply per output:
                                // +10 % kernel time).  OFF: it is <= 2^-22 |w x|, 2^-25 on average, and without it the pair form is still
                                // closer to fp64 than the fp32 MFMA chain (tools/ubench/split_mfma.hip, 51 200 dot products of K = 32:
                                // max / rms error relative to sum|w x|  fp32 MFMA 2.0e-7 / 2.4e-8, 4 products 9.9e-8 / 1.8e-8, 3 products 1.2e-7 / 2.1e-8)
#endif
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
struct P8 { h8 h, m; };
constexpr float kPairS = 2048.f, kPairSi = 1.f / 2048.f;

DEV void split2(float x0, float x1, h2& h, h2& m) {
    const f2 x = {x0, x1};
    h = __builtin_convertvector(x, h2);                                              // v_cvt_pk_f16_f32, round to nearest
    // (x - h) 2^11, exact: written as an fma on the scaled x so that it selects v_fma_mix_f32 (reads the fp16 half directly)
    const f2 r = {__builtin_fmaf((float)h.x, -kPairS, x0 * kPairS), __builtin_fmaf((float)h.y, -kPairS, x1 * kPairS)};
#if GNR_ABLATE & 4
    m = (h2){(_Float16)0.f, (_Float16)0.f};
#else
    m = __builtin_convertvector(r, h2);
#endif
}
template <int O, int N>
DEV P8 split8(const float (&v)[N]) {
    static_assert(O + 8 <= N, "split8 reads 8 slots");
    h2 h[4], m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split2(v[O + 2 * q], v[O + 2 * q + 1], h[q], m[q]);
    P8 p;
    p.h = (h8){h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
    p.m = (h8){m[0].x, m[0].y, m[1].x, m[1].y, m[2].x, m[2].y, m[3].x, m[3].y};
    return p;
}
// CNT < 8 slots starting at O, the rest of the block zero (pairs of zeros cost nothing)
template <int O, int CNT, int N>
DEV P8 split8z(const float (&v)[N]) {
    static_assert(O + CNT <= N && CNT >= 1 && CNT < 8, "split8z reads CNT slots");
    h2 h[4], m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (2 * q + 1 < CNT) split2(v[O + 2 * q], v[O + 2 * q + 1], h[q], m[q]);
        else if (2 * q < CNT) split2(v[O + 2 * q], 0.f, h[q], m[q]);
        else { h[q] = (h2){(_Float16)0.f, (_Float16)0.f}; m[q] = h[q]; }
    }
    P8 p;
    p.h = (h8){h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
    p.m = (h8){m[0].x, m[0].y, m[1].x, m[1].y, m[2].x, m[2].y, m[3].x, m[3].y};
    return p;
}
// a pair block parked in eight 32-bit registers (the per-view state keeps e1 in this form: it is consumed twice as a B operand,
// by neuray_fc.0 in the first view loop and by base_fc.0 in the second, and nowhere as fp32 values)
DEV void p8_store(const P8& p, float* r) {
    const f4 a = __builtin_bit_cast(f4, p.h), b = __builtin_bit_cast(f4, p.m);
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
}
DEV P8 p8_load(const float* r) {
    P8 p;
    p.h = __builtin_bit_cast(h8, (f4){r[0], r[1], r[2], r[3]});
    p.m = __builtin_bit_cast(h8, (f4){r[4], r[5], r[6], r[7]});
    return p;
}
DEV f4 mfma32h(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// ---- GNR_UNSCALED_ACT: the cheaper pair of an ACTIVATION (inner layers only: operands that are ELU outputs, O(1) vectors).
//   x = h + m   with  h = fp16(x),  m = fp16(x - h)  -- the residual UNSCALED, so that both products with the weight's high half
//   accumulate in the layer's own accumulator:  W x = wh h + wh m + wm (h 2^-11)  (wm = the weight's residual, scaled by 2^11 as
//   before; its B operand is the high half times 2^-11, exact unless subnormal).  No second accumulator, no fold: a split costs 5
//   VALU per two operands instead of 6 and an output nothing instead of one fma.  Price: the residual of |x| < 2^-3 falls into the
//   fp16 subnormal range (quantum 2^-24), so such an operand carries an ABSOLUTE error of up to 2^-25 instead of a relative one of
//   2^-24: a pre-activation is off by at most 2^-25 sum|w| (1e-7 for these layers) on top of the fp32-like rounding.  Against the
//   float64 arbiter the path is as close as with scaled residuals (tests/test_range_guard.py::test_fp64_arbiter).  Layers fed by
//   feature maps or cross-view statistics (any magnitude) keep the scaled form.
#ifndef GNR_UNSCALED_ACT
#define GNR_UNSCALED_ACT 1
#endif
struct P8U { h8 h, m, s; };
DEV void split2u(float x0, float x1, h2& h, h2& m, h2& s) {
    const f2 x = {x0, x1};
    h = __builtin_convertvector(x, h2);
    // x - h, exact, as ONE instruction reading the fp16 half in place (the compiler turns fma(h, -1, x) back into a subtraction
    // and lowers that to v_cvt_f32_f16 + v_sub_f32)
    f2 r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(h), "v"(x1));
    m = __builtin_convertvector(r, h2);
    s = h * (h2){(_Float16)kPairSi, (_Float16)kPairSi};                 // v_pk_mul_f16
}
template <int O, int N>
DEV P8U split8u(const float (&v)[N]) {
    static_assert(O + 8 <= N, "split8u reads 8 slots");
    h2 h[4], m[4], s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split2u(v[O + 2 * q], v[O + 2 * q + 1], h[q], m[q], s[q]);
    P8U p;
    p.h = (h8){h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
    p.m = (h8){m[0].x, m[0].y, m[1].x, m[1].y, m[2].x, m[2].y, m[3].x, m[3].y};
    p.s = (h8){s[0].x, s[0].y, s[1].x, s[1].y, s[2].x, s[2].y, s[3].x, s[3].y};
    return p;
}

// ---- range guard of the pair form.  The high half of a pair is an fp16: an operand of 65 520 or more becomes +-inf there and
// the layer's outputs garbage (ELU then maps -inf and NaN to finite values, so nothing downstream would show it).  The fp32
// reference has no such limit.  Every pair block whose operands are not bounded by construction is therefore watched:
//  * a non-finite high half (+-inf, NaN) in column r of a B operand makes EVERY output of point r non-finite in that layer's
//    accumulator (w inf = +-inf, 0 inf = NaN, inf - inf = NaN), so one accumulator register per lane and layer tells: chk =
//    fma(acc, 0, chk) turns NaN at the first non-finite value and stays NaN -- one VALU instruction per watched LAYER
//    (measured: 0.6 % of a launch; v_dot2c_f32_f16 over the halves, four operands per instruction, cost 2 %: it issues in ~10
//    cycles; v_pk_max_i16 on the halves as signed integers, two per instruction, the same);
//  * the gathered features are convex combinations of feature-map values, which k_repack_feats tests once per prepare.
// A flagged launch sets a bit in GnrWorkspace's range word; the fp32-MFMA instantiation of the same kernel, launched right
// behind every pair launch, returns at once unless a bit is set and otherwise recomputes the launch (gnr_capi.inc).
#ifndef GNR_WATCH_LAST
#define GNR_WATCH_LAST 0
#endif
#ifndef GNR_RANGE_GUARD
#define GNR_RANGE_GUARD 1                     // 0: measurement builds without the watch and without the fp32 twin launches
#endif
constexpr float kFeatLimit = 60000.f;          // |feature| below this (k_repack_feats); leaves room for x = feature + ray_dir_fc
// The running word is the tile's own valid-view count `msum` (live from the first view loop to the record, so the watch costs no
// register): it turns NaN with the first non-finite accumulator and is tested once per tile.
DEV void range_watch(float& word, const f4& a) {   // a: an output block of a pair-form layer (after the cross terms were folded in)
    if constexpr (GNR_RANGE_GUARD) word = __builtin_fmaf(a.x, 0.f, word);
}

// acc[nb] += W x over KB K32 blocks (inputs x8[0..KB)), NB output blocks.  w: the layer's slot in the C16 image.  The
// left-over fp32 k-steps of a layer are added by the caller with mm<>.
// Only the K = 32 instruction: with the legacy v_mfma_f32_16x16x16_f16 (tried for the 4-k-step tails) next to it the chain's
// outputs differed from launch to launch on the same inputs (tools/dbg/chain_det.py: 2 000 - 36 000 of 64 000 voxels by
// 2e-7 ... 2e-5, at one and at two wavefronts per SIMD) -- a dependency the compiler's hazard tables (ROCm 7.2) do not
// cover; tails are zero-padded into a K32 block instead.
// WATCH: the B operands are not bounded by construction (range guard, RangeWatch)
#ifndef GNR_MFMA_PRIO
#define GNR_MFMA_PRIO 0      // measurement builds: 1 = the younger half of a workgroup (wavefronts 4-7) at s_setprio 1 for the whole kernel; 2 = s_setprio 1 around every pair-form MFMA cluster
#endif
template <int KB, int NB, bool LF, bool WATCH = false>
DEV void mm16(const float* __restrict__ w, int lane, const P8* __restrict__ x8, f4 (&acc)[NB], float* rw = nullptr) {
    if constexpr (LF) asm volatile("" ::: "memory");
    const h8* w8 = reinterpret_cast<const h8*>(w) + lane;
#if GNR_MFMA_PRIO == 2
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        f4 lo = {0.f, 0.f, 0.f, 0.f};
#if GNR_SPLIT_MM
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) lo = mfma32h(w8[((kb * NB + nb) * 2 + 1) * 64], x8[kb].m, lo);
        lo *= kPairSi;
#endif
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const h8 wh = w8[((kb * NB + nb) * 2) * 64], wm = w8[((kb * NB + nb) * 2 + 1) * 64];
            lo = mfma32h(wh, x8[kb].m, lo);
            lo = mfma32h(wm, x8[kb].h, lo);
            acc[nb] = mfma32h(wh, x8[kb].h, acc[nb]);
        }
#if GNR_ABLATE & 32
        acc[nb] += lo;
#else
        acc[nb].x = fmaf(lo.x, kPairSi, acc[nb].x); acc[nb].y = fmaf(lo.y, kPairSi, acc[nb].y);
        acc[nb].z = fmaf(lo.z, kPairSi, acc[nb].z); acc[nb].w = fmaf(lo.w, kPairSi, acc[nb].w);
#endif
    }
#if GNR_MFMA_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (WATCH) range_watch(*rw, acc[GNR_WATCH_LAST ? NB - 1 : 0]);
}
// the same layer on unscaled activation pairs: three products into ONE accumulator (the k-blocks outermost, so that
// consecutive MFMAs go to different output blocks)
template <int KB, int NB, bool LF, bool WATCH = false>
DEV void mm16u(const float* __restrict__ w, int lane, const P8U* __restrict__ x8, f4 (&acc)[NB], float* rw = nullptr) {
    if constexpr (LF) asm volatile("" ::: "memory");
    const h8* w8 = reinterpret_cast<const h8*>(w) + lane;
#if GNR_MFMA_PRIO == 2
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        h8 wh[NB], wm[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { wh[nb] = w8[((kb * NB + nb) * 2) * 64]; wm[nb] = w8[((kb * NB + nb) * 2 + 1) * 64]; }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma32h(wh[nb], x8[kb].h, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma32h(wh[nb], x8[kb].m, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma32h(wm[nb], x8[kb].s, acc[nb]);
    }
#if GNR_MFMA_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (WATCH) range_watch(*rw, acc[0]);
}

// ---------------------------------------------------------------------------------------
// setup kernels
// ---------------------------------------------------------------------------------------
// [BV][32][npix] x2 (NCHW) -> [BV][npix][64]; block = 256 threads handles 64 pixels
__global__ __launch_bounds__(256) void k_repack_feats(const float* __restrict__ ray_feats,
                                                      const float* __restrict__ img_feats,
                                                      float* __restrict__ out, int npix, unsigned* __restrict__ range_flag) {
    __shared__ float tile[64][65];
    const int bv = blockIdx.y, p0 = blockIdx.x * 64, t = threadIdx.x;
    const int tx = t & 63, ty = t >> 6;
    const size_t in_base = (size_t)bv * 32 * npix;
#pragma unroll
    for (int c0 = 0; c0 < 64; c0 += 4) {
        const int c = c0 + ty, p = p0 + tx;
        float v = 0.f;
        if (p < npix) v = (c < 32) ? ray_feats[in_base + (size_t)c * npix + p] : img_feats[in_base + (size_t)(c - 32) * npix + p];
        tile[c][tx] = v;
    }
    __syncthreads();
    float* o = out + ((size_t)bv * npix + p0) * 64;
    // range guard of k_chain's pair form (4.1d): bit 0 = a feature beyond the fp16-pair range or not finite.  Tested on the way
    // out, four values per test: the largest magnitude of a float4 against the limit, and its sum for NaN (max drops NaN, + keeps it)
    float amax = 0.f, nansum = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + t;           // float4 index within the 64x64 tile
        const int p = idx >> 4, c4 = (idx & 15) * 4;
        if (p0 + p < npix) {
            f4 v = {tile[c4][p], tile[c4 + 1][p], tile[c4 + 2][p], tile[c4 + 3][p]};
            amax = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), amax);
            nansum += (v.x + v.y) + (v.z + v.w);
            reinterpret_cast<f4*>(o)[idx] = v;
        }
    }
    const bool bad = !(amax < kFeatLimit) || nansum != nansum;
    if (range_flag && __ballot(bad) != 0 && (t & 63) == 0) atomicOr(range_flag, 1u);
}

// ref: render_ops.py:94 (K @ Rt), :112 (camera centre), dist_decoder.py:17-20
__global__ void k_view_setup(const float* __restrict__ poses, const float* __restrict__ Ks,
                             const float* __restrict__ dr, float* __restrict__ viewp, int nviews, unsigned* __restrict__ range_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 64 && range_flag) { range_flag[i] = 0u; range_flag[64 + i] = 0u; }   // gnr_prepare launches this kernel first: new watch words per prepare (word 0 + one per launch slot), and the
                                                                                 // tile counters of the chain launches behind them (words 64.., ChainArgs::tile_ctr)
    if (i >= nviews) return;
    const float* P = poses + i * 12;
    const float* K = Ks + i * 9;
    float* o = viewp + i * VIEWP_FLOATS;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
            o[r * 4 + c] = __fmaf_rn(K[r * 3 + 2], P[8 + c], __fmaf_rn(K[r * 3 + 1], P[4 + c], __fmul_rn(K[r * 3], P[c])));
    for (int c = 0; c < 3; ++c)
        o[12 + c] = -__fmaf_rn(P[8 + c], P[11], __fmaf_rn(P[4 + c], P[7], __fmul_rn(P[c], P[3])));
    const float nr = -1.f / dr[i * 2], fr = -1.f / dr[i * 2 + 1];
    o[15] = nr;
    o[16] = fr - nr;
    o[17] = 1.f / (fr - nr);
    for (int c = 18; c < VIEWP_FLOATS; ++c) o[c] = 0.f;
}

// ref: field_utils.py:17-27 (float64 arithmetic then cast), renderer.py:167-170,179
__global__ void k_points_volume(const float* __restrict__ bbox_min, float* __restrict__ desc, int R, int B) {
    const int P = R * R * R;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * P) return;
    const int b = i / P, n = i % P;
    const int col = n / R, s = n % R;
    const int ix = col / R, iy = col % R, iz = R - 1 - s;
    const double vs = 0.3 / (double)R, hv = vs / 2;
    float* d = desc + (size_t)i * DESC_FLOATS;
    d[0] = (float)(ix * vs + hv) + bbox_min[b * 3 + 0];
    d[1] = (float)(iy * vs + hv) + bbox_min[b * 3 + 1];
    d[2] = (float)(iz * vs + hv) + bbox_min[b * 3 + 2];
    d[3] = 0.f; d[4] = 0.f; d[5] = 1.f;                 // que_dir = (0,0,1)
    d[6] = 0.005f; d[7] = 0.005f;                       // fixed interval 0.01 (dist_decoder.py:47-49)
}

// ref: render_ops.py:146-170 (random_sample False)
__global__ void k_coarse_depth(const float* __restrict__ que_dr, float* __restrict__ depth, int rn, int dn, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * rn * dn) return;
    const int b = i / (rn * dn), k = i % dn;
    const float near = que_dr[b * 2], far = que_dr[b * 2 + 1];
    const float diff = __fsub_rn(__fdiv_rn(1.f, far), __fdiv_rn(1.f, near));
    const float interval = __fdiv_rn(diff, (float)(dn - 1));
    float tick = __fmul_rn(interval, (float)k);
    if (k == 0) tick = 0.f;
    if (k == dn - 1) tick = diff;
    depth[i] = __fdiv_rn(1.f, __fadd_rn(__fdiv_rn(1.f, near), tick));
}

// fine_depth_use_all (renderer.py:145-146): the fine pass renders sort(cat(coarse depths, resampled depths)).  Both inputs are
// ascending per ray (coarse by construction, fine out of the resampler's rank sort): a rank merge, coarse first on ties.
__global__ void k_merge_depths(const float* __restrict__ dc, const float* __restrict__ df, float* __restrict__ out, int nrays, int dn, int fdn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int T = dn + fdn;
    if (i >= nrays * T) return;
    const int ray = i / T, e = i - ray * T;
    const float* c = dc + (size_t)ray * dn;
    const float* f = df + (size_t)ray * fdn;
    int rank;
    float v;
    if (e < dn) { v = c[e]; rank = e; for (int j = 0; j < fdn; ++j) rank += f[j] < v ? 1 : 0; }
    else { v = f[e - dn]; rank = e - dn; for (int j = 0; j < dn; ++j) rank += c[j] <= v ? 1 : 0; }
    out[(size_t)ray * T + rank] = v;
}

DEV void inv3x3(const float* K, float* o) {
    const float a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const float A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
    const float det = a * A + b * Bc + c * C;
    const float id = 1.f / det;
    o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
    o[3] = Bc * id; o[4] = (a * i - c * g) * id; o[5] = -(a * f - c * d) * id;
    o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}

// Traversal order of a scene's rays for the inference render passes: perm[b][slot] = index of the ray that sits at position `slot`
// of the scene's rays sorted by the Morton code of their pixel (render_ops.py:4-39: a ray is a pixel of the query view).
// The kernels' internal per-point arrays (descriptors, records) are laid out in that order, so the 16-point tiles of neighbouring
// workgroups hold neighbouring pixels' rays, whose samples project onto nearly the same epipolar lines of every reference view:
// the feature-map lines they gather are shared in L1 / L2 instead of being fetched once per ray (random training rays: the order
// of `coords` carries no locality).  Every user-visible array keeps the caller's ray order (outputs are written through perm).
// One workgroup per scene, bitonic sort of (morton << 12 | index) keys in LDS; rn <= 4096.
constexpr int MAX_SORT_RAYS = 4096;
DEV unsigned spread10(unsigned v) {           // 10 bits -> every other bit
    v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u;
    return v;
}
__global__ __launch_bounds__(256) void k_ray_order(const float* __restrict__ coords, int* __restrict__ perm, int rn, int shift) {
    __shared__ unsigned key[MAX_SORT_RAYS];
    const int b = blockIdx.x;
    int M = 1;
    while (M < rn) M <<= 1;
    for (int i = threadIdx.x; i < M; i += 256) {
        unsigned k = 0xFFFFFFFFu;
        if (i < rn) {
            const float x = coords[((size_t)b * rn + i) * 2], y = coords[((size_t)b * rn + i) * 2 + 1];
            const unsigned xi = (unsigned)min(max((int)x, 0) >> shift, 1023), yi = (unsigned)min(max((int)y, 0) >> shift, 1023);
            k = ((spread10(xi) | (spread10(yi) << 1)) << 12) | (unsigned)i;
        }
        key[i] = k;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= M; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < M; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned a = key[i], c = key[l];
                    const bool up = (i & k2) == 0;
                    if ((a > c) == up) { key[i] = c; key[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < rn; i += 256) perm[(size_t)b * rn + i] = (int)(key[i] & 0xFFFu);
}

// ref: render_ops.py:4-39 (rays, unnormalised directions), :41-52 + dist_decoder.py:34-38 (intervals)
// perm (nullable): descriptor slot (b, s, k) holds sample k of ray perm[b][s] (k_ray_order); coords / depth are in the caller's order
__global__ void k_points_rays(const float* __restrict__ coords, const float* __restrict__ que_pose,
                              const float* __restrict__ que_K, const float* __restrict__ que_dr,
                              const float* __restrict__ depth, float* __restrict__ desc, int rn, int dn, int B,
                              const int* __restrict__ perm = nullptr, float* __restrict__ depth_gen = nullptr) {
    // depth_gen: the coarse pass -- the depths are k_coarse_depth's (same arithmetic), computed here and written to depth_gen; `depth` is not read
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * rn * dn) return;
    const int b = i / (rn * dn), slot = (i / dn) % rn, k = i % dn;
    const int ray = perm ? perm[(size_t)b * rn + slot] : slot;
    const float* P = que_pose + b * 12;
    float Ki[9];
    inv3x3(que_K + b * 9, Ki);
    const float u = coords[((size_t)b * rn + ray) * 2], v = coords[((size_t)b * rn + ray) * 2 + 1];
    float cam[3], tr[3], dir[3];
    for (int r = 0; r < 3; ++r) cam[r] = Ki[r * 3] * u + Ki[r * 3 + 1] * v + Ki[r * 3 + 2];
    for (int c = 0; c < 3; ++c) tr[c] = -(P[c] * P[3] + P[4 + c] * P[7] + P[8 + c] * P[11]);
    for (int c = 0; c < 3; ++c) {
        const float rc = P[c] * cam[0] + P[4 + c] * cam[1] + P[8 + c] * cam[2];
        dir[c] = __fsub_rn(__fadd_rn(rc, tr[c]), tr[c]);          // render_ops.py:22-23
    }
    const float* zp = depth_gen ? nullptr : depth + ((size_t)b * rn + ray) * dn;
    const float dnear = que_dr[b * 2], dfar = que_dr[b * 2 + 1];
    auto z = [&](int j) -> float {
        if (!depth_gen) return zp[j];
        const float diff = __fsub_rn(__fdiv_rn(1.f, dfar), __fdiv_rn(1.f, dnear));              // k_coarse_depth (render_ops.py:146-170, random_sample False)
        float tick = __fmul_rn(__fdiv_rn(diff, (float)(dn - 1)), (float)j);
        if (j == 0) tick = 0.f;
        if (j == dn - 1) tick = diff;
        return __fdiv_rn(1.f, __fadd_rn(__fdiv_rn(1.f, dnear), tick));
    };
    const float zk = z(k);
    if (depth_gen) depth_gen[((size_t)b * rn + ray) * dn + k] = zk;
    const float nrm = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    const float near = -1.f / dnear, far = -1.f / dfar;
    auto inv = [&](int j) { return __fdiv_rn(__fsub_rn(__fdiv_rn(-1.f, z(j)), near), __fsub_rn(far, near)); };
    auto half = [&](int j) { return (j == dn - 1) ? 0.5e6f : __fmul_rn(__fsub_rn(inv(j + 1), inv(j)), 0.5f); };
    float* d = desc + (size_t)i * DESC_FLOATS;
    for (int c = 0; c < 3; ++c) {
        d[c] = tr[c] + dir[c] * zk;
        d[3 + c] = -dir[c] / nrm;
    }
    d[6] = half(k == 0 ? 0 : k - 1);
    d[7] = half(k);
}

// sample points and query directions of a render pass as plain tensors (the inputs of the training pass's backward kernels):
// pts [B*rn*dn][3] and qdir [B*rn][3] out of the descriptors k_points_rays wrote (same bits the chain consumed)
__global__ void k_desc_unpack(const float* __restrict__ desc, float* __restrict__ pts, float* __restrict__ qdir, int dn, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* d = desc + (size_t)i * DESC_FLOATS;
    if (pts) { pts[(size_t)i * 3] = d[0]; pts[(size_t)i * 3 + 1] = d[1]; pts[(size_t)i * 3 + 2] = d[2]; }
    if (qdir && i % dn == 0) { float* q = qdir + (size_t)(i / dn) * 3; q[0] = d[3]; q[1] = d[4]; q[2] = d[5]; }
}

// ---------------------------------------------------------------------------------------
// k_chain
// ---------------------------------------------------------------------------------------
struct ChainArgs {
    const float* wpk;      // packed level blob (CHAIN section first)
    const float* feat64;   // [B*V][fh][fw][64]
    const float* imgs;     // [B*V][3][H][W]
    const float* viewp;    // [B*V][VIEWP_FLOATS]
    const float* desc;     // [B*P][DESC_FLOATS]
    float* rec;            // [B*P][REC_*]
    float* colors;         // [B*P][3]      (RENDER)
    unsigned char* vmask;  // [B*P] or null
    float* dbg;            // [B*P][32] or null
    int B, P, H, W, fh, fw;
    int vol_res;           // volume launches: grid resolution R (tile order is brick-permuted); 0 for ray points
    // training forward (backward twins): per-view state after each view loop and the hoisted pre-activation, in
    // register order [tile][..][64 lanes]; all null in inference
    float* save1;          // [tiles][V][19][64]  X[9], e1[8], gate, m
    float* save2;          // [tiles][V][9 | 11][64]   h~[8], v2 (, colour logit, raw rgb: RENDER)
    float* saveG;          // [tiles][17][64]     G[16] (log2e-scaled pre-activation incl. bias), 1/(sum m + 1e-8)
    float* saveZ;          // [tiles][18][64]     mean~[8], var~[8] (log2e / log2e^2 scaled), wbar, number of valid views
    // range guard (RangeWatch): the pair kernels OR bit 1 into *range_flag when an operand leaves the fp16-pair range; a
    // launch with only_if_flagged != 0 (the fp32-MFMA twin behind every pair launch) returns at once while the word is zero
    unsigned* range_flag;
    int only_if_flagged;
    unsigned* range_launch;       // this launch's own watch word (bit 1: an activation / statistic of THIS launch left the fp16-pair range); word 0
                                  // (range_flag) keeps the bits that concern every launch of the prepared scene: 0 = a feature, 2 = a weight
    // inference render passes: the points are laid out in the Morton order of their rays (k_ray_order); the user-visible
    // per-point outputs (colours, view masks) go to the caller's ray order: point (slot s, sample k) -> ray_perm[b][s] * dn + k
    const int* ray_perm; int perm_rn, perm_dn;
    // dynamic tile hand-out (null: static round-robin).  [x] = next tile of XCD x's chunk, [8] = wavefronts of the launch that are through:
    // zero before the first launch (gnr_prepare), set back to zero by the last wavefront of every launch
    unsigned* tile_ctr;
    const float* bbox_min; // volume launches of the inference path: [B][3]; the points are computed in the kernel (k_points_volume's arithmetic), desc is not read
    int pts_R;             // grid resolution of those points (0: read desc)
    int b0;                // first scene of the launch (a launch over a part of the batch: scenes b0 .. b0 + B - 1 of every array)
};

struct ViewGeom {          // per (point, view) quantities that are cheap to recompute
    float u, v, z, m;
    float dd[4];
};

template <bool FULL>
DEV void project_view(const float* __restrict__ vp, const float (&p)[3], const float (&qd)[3], int H, int W, ViewGeom& o) {
    // ref: render_ops.py:98-104 (projection, depth guard), :126-128 (in-image test), :112-114 (direction)
    const float pcx = __fmaf_rn(vp[2], p[2], __fmaf_rn(vp[1], p[1], __fmul_rn(vp[0], p[0]))) + vp[3];
    const float pcy = __fmaf_rn(vp[6], p[2], __fmaf_rn(vp[5], p[1], __fmul_rn(vp[4], p[0]))) + vp[7];
    float z = __fmaf_rn(vp[10], p[2], __fmaf_rn(vp[9], p[1], __fmul_rn(vp[8], p[0]))) + vp[11];
    const bool inval = fabsf(z) < 1e-4f;
    if (inval) z = 1e-3f;
    o.z = z;
    o.u = __fdiv_rn(pcx, z);
    o.v = __fdiv_rn(pcy, z);
    const bool outside = (o.u < -0.5f) | (o.u >= (float)W - 0.5f) | (o.v < -0.5f) | (o.v >= (float)H - 0.5f);
    o.m = (!inval && !outside) ? 1.f : 0.f;
    if constexpr (FULL) {
        float d[3] = {p[0] - vp[12], p[1] - vp[13], p[2] - vp[14]};
        // -1 / max(|d|, 1e-5)
        const float inr = -__builtin_amdgcn_rsqf(fmaxf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], 1e-10f));
        d[0] *= inr; d[1] *= inr; d[2] *= inr;
        o.dd[0] = d[0] - qd[0]; o.dd[1] = d[1] - qd[1]; o.dd[2] = d[2] - qd[2];     // aggregate_net.py:13
        o.dd[3] = d[0] * qd[0] + d[1] * qd[1] + d[2] * qd[2];                       // :14
    }
}

struct Taps { int o00, o01, o10, o11; float w00, w01, w10, w11; };
// bilinear, border padding.  ref: ops.py:29-33 + grid_sample (see oracle bilinear_border).
//   feature map (align_corners False): px = u*fw/(W-1) - 0.5 ; full-res image (align_corners True): px = u
// (the reference's normalise/unnormalise round trip, folded; differs from it by fp32 rounding only)
DEV Taps make_taps(float u, float v, float sx, float sy, float off, int fh, int fw) {
    float px = fmaf(u, sx, off), py = fmaf(v, sy, off);
    px = fminf(fmaxf(px, 0.f), (float)(fw - 1));
    py = fminf(fmaxf(py, 0.f), (float)(fh - 1));
    const float x0 = floorf(px), y0 = floorf(py);
    const float wx1 = px - x0, wy1 = py - y0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const int x0i = (int)x0, y0i = (int)y0;
    const int x1i = min(x0i + 1, fw - 1), y1i = min(y0i + 1, fh - 1);
    Taps t;
    t.o00 = y0i * fw + x0i; t.o01 = y0i * fw + x1i; t.o10 = y1i * fw + x0i; t.o11 = y1i * fw + x1i;
    t.w00 = wx0 * wy0; t.w01 = wx1 * wy0; t.w10 = wx0 * wy1; t.w11 = wx1 * wy1;
    return t;
}

// ---- cooperative projection (k_chain's first view loop).  The four lanes of a point (one per lane group) would each compute the
// same projection, two IEEE divisions and both sets of bilinear taps.  Instead lane group g does ONE axis of ONE map -- axis
// g & 1 (u / v), map g >> 1 (feature map / full-resolution image): its row of K[R|t], its division, its 1-D tap (two indices
// and a weight) and its in-image test -- and the groups exchange the results through ds_bpermute_b32 (12 per view; the LDS
// crossbar, no VALU slot).  Every lane then assembles the 2 x 4 offsets and weights with the operations make_taps() uses,
// so offsets, weights and the in-image mask are bit for bit the ones of project_view() + make_taps() (the parity suites pass
// unchanged, all outputs bitwise equal).  The per-group constants (tap scale / offset, last index, image bound, row stride) sit
// in a 32-float table behind the weight image in LDS.
// MEASURED NEGATIVE, kept as a switch (default off): 41 vector instructions per view fewer (907 -> 866 in the first view loop,
// -246 per tile) and the same time -- 3.36 - 3.40 ms per volume launch and 2.45 - 2.46 ms per pair of render launches either
// way: the exchange puts an LDS round trip at the head of each view's dependency chain.
#ifndef GNR_COOP_PROJ
#define GNR_COOP_PROJ 0
#endif
constexpr int COOP_TAB_FLOATS = 32;
struct CoopOut { Taps f, i; float m, z; };
DEV void project_coop(const float* __restrict__ vp, const float* __restrict__ tab, int row_bytes, int bp, const float (&p)[3], CoopOut& o) {
    const f4 row = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(vp) + row_bytes);
    const float pc = __fmaf_rn(row.z, p[2], __fmaf_rn(row.y, p[1], __fmul_rn(row.x, p[0]))) + row.w;
    float z = __fmaf_rn(vp[10], p[2], __fmaf_rn(vp[9], p[1], __fmul_rn(vp[8], p[0]))) + vp[11];
    const bool inval = fabsf(z) < 1e-4f;
    if (inval) z = 1e-3f;
    o.z = z;
    const float q = __fdiv_rn(pc, z);
    const f4 c = reinterpret_cast<const f4*>(tab)[0];                  // tap scale, tap offset, last index, image bound
    const f4 ci = reinterpret_cast<const f4*>(tab)[1];                 // (as ints) row stride, last index
    const int stride = __float_as_int(ci.x), n1 = __float_as_int(ci.y);
    const bool outside = (q < -0.5f) | (q >= c.w);
    float px = fmaf(q, c.x, c.y);
    px = fminf(fmaxf(px, 0.f), c.z);
    const float x0 = floorf(px);
    const float w1 = px - x0;                                          // in [0, 1): the sign bit carries "not visible along this axis"
    const int i0 = (int)x0, i1 = min(i0 + 1, n1);
    const int o0 = __mul24(i0, stride), o1 = __mul24(i1, stride);
    const int w1b = __float_as_int(w1) | ((outside | inval) ? (int)0x80000000 : 0);
    int O0[4], O1[4], W1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        O0[k] = __builtin_amdgcn_ds_bpermute(bp + 64 * k, o0);
        O1[k] = __builtin_amdgcn_ds_bpermute(bp + 64 * k, o1);
        W1[k] = __builtin_amdgcn_ds_bpermute(bp + 64 * k, w1b);
    }
    o.m = ((W1[0] | W1[1]) >= 0) ? 1.f : 0.f;
    auto assemble = [&](int kx, int ky, Taps& t) {
        t.o00 = O0[ky] + O0[kx]; t.o01 = O0[ky] + O1[kx]; t.o10 = O1[ky] + O0[kx]; t.o11 = O1[ky] + O1[kx];
        const float wx1 = fabsf(__int_as_float(W1[kx])), wy1 = fabsf(__int_as_float(W1[ky])), wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        t.w00 = wx0 * wy0; t.w01 = wx1 * wy0; t.w10 = wx0 * wy1; t.w11 = wx1 * wy1;
    };
    assemble(0, 1, o.f);
    assemble(2, 3, o.i);
}
// point-to-camera direction against the query direction (the FULL part of project_view)
DEV void view_dir(const float* __restrict__ vp, const float (&p)[3], const float (&qd)[3], float (&dd)[4]) {
    float d[3] = {p[0] - vp[12], p[1] - vp[13], p[2] - vp[14]};
    const float inr = -__builtin_amdgcn_rsqf(fmaxf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], 1e-10f));
    d[0] *= inr; d[1] *= inr; d[2] *= inr;
    dd[0] = d[0] - qd[0]; dd[1] = d[1] - qd[1]; dd[2] = d[2] - qd[2];
    dd[3] = d[0] * qd[0] + d[1] * qd[1] + d[2] * qd[2];
}

constexpr int SW = 20;     // per-view state width: X[9] E[8] gate m rgb  /  H2[8] v2 c . . . rgb
// The per-view state lives in registers, so its rows need static indices while the view loops stay rolled (code size): the
// state is a QUEUE that advances one row per view: (V-1) x 20 v_mov_b32 per view and loop, 1 320 per tile at V = 6 (14 % of the
// kernel's vector instructions).  They cost next to nothing: GNR_TWO_QUEUES 1 splits the views into two halves with a queue
// each and writes every view loop out twice (one rolled loop per queue: (V/2 - 1) x 20 moves per view, 600 per tile, 7 KB more
// code) and changes a volume launch from 3.35 to 3.32 ms and a render launch from 2.43 to 2.51 (more scratch: 88 B) -- the
// moves issue while the SIMD's other wavefront owns the matrix pipe.  (An earlier "no moves" ablation that showed 8 - 11 % had
// let the compiler hoist the second loop's now view-invariant operand splits out of the loop.)  Kept as a switch, default off.
// Other measured negatives: addressing the row through wave-uniform branches on the view index (V arms of 20
// moves; every variant spills, 3.63 - 4.31 ms against 3.56); full unrolling (code 83 KB, 900 B of scratch: 6.6 ms); two / three
// bodies per trip with the queue advancing two / three rows (200 - 400 B of scratch: 4.85 ms); v_swap_b32 (6.4 cycles per dword),
// v_mov_b64 / v_pk_mov_b32 (2.8 / 3.1 cycles per dword against 3.2 for v_mov_b32, tools/ubench/mov_rates.hip).
#ifndef GNR_TWO_QUEUES
#define GNR_TWO_QUEUES 0
#endif
#ifndef GNR_RDF2_PAIRS
#define GNR_RDF2_PAIRS 0      // measurement build: ray_dir_fc.2 on the f16 matrix cores (gnr_pack_body.h c16_pairs; round-6 A/B record)
#endif
// rows of the two queues
constexpr int rows_a(int V) { return GNR_TWO_QUEUES ? (V + 1) / 2 : V; }
constexpr int rows_b(int V) { return GNR_TWO_QUEUES ? V / 2 : 0; }

#ifndef GNR_DYN_TILES
#define GNR_DYN_TILES 1         // k_chain takes its tiles from a per-XCD counter (ChainArgs::tile_ctr); 0: static round-robin shares; 2: a wavefront whose XCD's chunk is
                                // empty goes on with the next XCD's -- MEASURED NEGATIVE (profiles/r06_c_dyn_tiles_ab.json: step 6.40 -> 6.83 ms, render launches
                                // 2.28 -> 2.58 ms: the time between two tiles of a wavefront goes from 2.4 % to 6.7 % of its life)
#endif
#ifndef GNR_VOLUME_POINTS_INLINE
#define GNR_VOLUME_POINTS_INLINE 1
#endif
#ifndef GNR_TAIL_OLD_ONLY
#define GNR_TAIL_OLD_ONLY 0      // in quarters of the XCD's wavefront count: 4 = the last 256 tiles of a 256-wavefront XCD
#endif
#ifndef GNR_DESC_PREFETCH
#define GNR_DESC_PREFETCH 0      // 1: the next tile's point descriptors are loaded behind the second view loop (eight registers carried through the tile's tail).
                                 // MEASURED NEGATIVE: step 6.45 -> 6.55 ms, render launches 2.28 -> 2.35 (scratch 72 -> 108 B, volume 12 -> 48 B)
#endif
#ifndef GNR_DYN_CTR_INC
#define GNR_DYN_CTR_INC 0
#endif
#ifndef GNR_DYN_TRAIN
#define GNR_DYN_TRAIN 1       // 1: the training-forward volume instantiation too; 2: and the render one (measured negative); 0: inference only
#endif
#ifndef GNR_SKIP_MASKED
#define GNR_SKIP_MASKED 1      // skip the (tile, view) pairs without a single valid point (inference kernels)
#endif
#ifndef GNR_CHAIN_THREADS
#define GNR_CHAIN_THREADS 512     // 8 wavefronts = 2 per SIMD at 250 registers; 256 (1 per SIMD, 512 registers) was measured too;
                                  // 768 / 1024 (3 / 4 per SIMD) would spill 356 / 524 B per lane: S[V][20] alone is 120 registers
#endif
#define GNR_CHAIN_MIN_BLOCKS (GNR_CHAIN_THREADS >= 1024 ? 1 : GNR_CHAIN_THREADS / 256)
// SAVE: training forward (writes the states the backward twins need); compiled out of the inference kernels
// USEVIS: the optional fourth decoder branch (dist_decoder_cfg.use_vis; GnrScene.use_vis), its own instantiations: as a
// run-time branch it cost the default kernels 5 - 18 % (3.65 -> 3.82 ms volume, 1.29 -> 1.52 ms render launch)
// SP: the wide layers as fp16 pairs on the f16 matrix cores (the product's launches) / on the fp32-input MFMA (the range
// guard's fallback, and every launch of the -DGNR_SPLIT16=0 companion build)
#ifndef GNR_WAVE_CLOCK
#define GNR_WAVE_CLOCK 0      // measurement build (tools/wave_clock.py): per-wavefront start / staged / end stamps and per-tile ticks of the inference launches
#endif
#if GNR_WAVE_CLOCK
__device__ unsigned long long g_wave_clock[2][4096][8];      // [RENDER][workgroup * 8 + wavefront][start, staged, end, tiles, ticks in: head + view loop 1, reduction 1 + hoist, view loop 2, tail]   (100 MHz ticks)
__device__ unsigned g_tile_clock[2][1 << 17];                // [RENDER][tile] ticks of one tile
#endif
template <int V, bool RENDER, bool SAVE = false, bool USEVIS = false, bool SP = (GNR_SPLIT16 != 0)>
__global__ __launch_bounds__(GNR_CHAIN_THREADS, GNR_CHAIN_MIN_BLOCKS) void k_chain(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool LF = (GNR_LICM_FENCE != 0) || V > 6;      // per-layer LICM fences (see mm())
#if GNR_WAVE_CLOCK
    const unsigned long long wc_t0 = wall_clock64();
#endif
    if (a.only_if_flagged) {                                // the fp32 twin: runs only when the scene's or this launch's watch word is set
        const unsigned w0 = a.range_flag ? __builtin_nontemporal_load(a.range_flag) : 0u;
        const unsigned w1 = a.range_launch ? __builtin_nontemporal_load(a.range_launch) : 0u;
        if ((w0 | w1) == 0u) return;                        // wave-uniform
    }
    if constexpr (SP && GNR_RANGE_GUARD != 0) {
        // a weight without an fp16 pair (gnr_pack.cpp build_c16): bit 2, and the whole launch is the fp32 twin's
        if (a.range_flag && a.wpk[pk::T_VIS + 2] != 0.f) {
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.range_flag, 4u);
            return;
        }
        // a feature beyond the pair range (bit 0, k_repack_feats) or a weight flagged by an earlier launch: the twin recomputes the
        // whole launch anyway, the pair kernel's work would be wasted (and would run on inf / NaN operands)
        if (a.range_flag && (__builtin_nontemporal_load(a.range_flag) & 5u) != 0u) return;
    }
    bool range_tripped = false;                           // wave-uniform
    constexpr bool UA = SP && GNR_UNSCALED_ACT != 0;       // inner layers on unscaled activation pairs (split8u / mm16u)
    constexpr bool EP = SP && !SAVE;                       // e1 travels between the view loops as its fp16 pair (the training saves keep fp32)
#define LO(o) (SP ? pk::c16_off(o) : (o))                  /* offset of a CHAIN-section name inside the staged image */                    // fp16-pair layers on the f16 matrix cores (C16 image) / fp32 MFMA (CHAIN image)
    // ---- stage the C16 (or CHAIN) section of the packed weights into LDS (once per workgroup)
    {
        const f4* src = reinterpret_cast<const f4*>(a.wpk + (SP ? pk::C16 : 0));
        f4* dst = reinterpret_cast<f4*>(lds);
        constexpr int N4 = (SP ? pk::C16_END : pk::CHAIN_END) / 4, SU = 6;      // six loads in flight per lane (one per trip: 10 us per launch, tools/wave_clock.py)
        for (int i0 = threadIdx.x; i0 < N4; i0 += blockDim.x * SU) {
            f4 tmp[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) { const int i = i0 + u * (int)blockDim.x; tmp[u] = src[i < N4 ? i : N4 - 1]; }
#pragma unroll
            for (int u = 0; u < SU; ++u) { const int i = i0 + u * (int)blockDim.x; if (i < N4) dst[i] = tmp[u]; }
        }
    }
    constexpr int COOP_TAB = SP ? pk::C16_END : pk::CHAIN_END;            // per-lane-group constants of project_coop
    if (threadIdx.x < 4) {
        const int gg = threadIdx.x, axis = gg & 1, img = gg >> 1;
        const int n = img ? (axis ? a.H : a.W) : (axis ? a.fh : a.fw);
        float* tb = lds + COOP_TAB + gg * 8;
        tb[0] = img ? 1.f : (axis ? (float)a.fh / (float)(a.H - 1) : (float)a.fw / (float)(a.W - 1));
        tb[1] = img ? 0.f : -0.5f;
        tb[2] = (float)(n - 1);
        tb[3] = (float)(axis ? a.H : a.W) - 0.5f;
        tb[4] = __int_as_float(axis ? (img ? a.W : a.fw) : 1);
        tb[5] = __int_as_float(n - 1);
        tb[6] = 0.f; tb[7] = 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if GNR_MFMA_PRIO == 1
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#if GNR_WAVE_CLOCK
    const unsigned long long wc_t1 = wall_clock64();
    unsigned long long wc_n = 0, wc_ph[4] = {0, 0, 0, 0};
#endif
    const int waves_per_block = blockDim.x >> 6;
    const int tps = (a.P + 15) >> 4;                       // tiles per scene
    const int ntiles = a.B * tps;
    constexpr int REC = RENDER ? REC_RAY : REC_VOL;
    const float fsx = (float)a.fw / (float)(a.W - 1), fsy = (float)a.fh / (float)(a.H - 1);

    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on
    // it).  Each XCD sweeps its own contiguous eighth of the tile list, so consecutive tiles (= neighbouring
    // voxels / ray samples, i.e. neighbouring feature-map lines) share ONE L2 instead of being fetched into
    // all eight (measured before this: 7.7 GB fetched per launch against 0.8 GB of inputs).
    const int NXCD = min(8, (int)gridDim.x);                               // tiny launches: fewer chunks than XCDs
    const int xcd = blockIdx.x % NXCD, lblk = blockIdx.x / NXCD;
    const int nlblk = ((int)gridDim.x - xcd + NXCD - 1) / NXCD;            // workgroups that share this XCD
    const int chunk = (ntiles + NXCD - 1) / NXCD;
    const int t_end = min(ntiles, (xcd + 1) * chunk);
    // Tiles are handed out by an atomic counter per XCD, not round-robin: the wavefronts of a launch do not run at one speed (the SIMD
    // issues its older wavefront first; tools/wave_clock.py: tile times 37 - 66 us at equal work, the fastest wavefront through after
    // 75 % of a volume launch), and a slot that stands empty while its SIMD partner finishes alone is the launch's tail: mean
    // wavefront life 0.88 (volume) / 0.85 (render) of the launch with static shares.  The next index is fetched one stage ahead
    // (after the second view loop), so its latency is hidden; the order in which an XCD sweeps its chunk is unchanged.
    // (the training-forward render instantiation keeps its static shares: five tiles per wavefront at 8 scenes, and the fetch's wait drains
    // the state stores in front of it -- 378 us per launch static, 412 dynamic; the volume one gains 2 %: profiles/r06_c_dyn_tiles_ab.json)
    const bool dyn = GNR_DYN_TILES != 0 && (GNR_DYN_TRAIN > 1 || !(SAVE && RENDER)) && (GNR_DYN_TRAIN != 0 || !SAVE) && a.tile_ctr != nullptr;
    int cx = xcd;                                          // the chunk this wavefront draws from
    // (GNR_DYN_CTR_INC 1: atomicInc instead of atomicAdd.  The compiler's atomic optimiser rewrites a uniform atomicAdd under a lane mask into
    // its wave-aggregated form, whose broadcast -- s_waitcnt vmcnt(0) + v_readfirstlane -- sits right behind the instruction, so the fetch is
    // waited for in the middle of the tile; a wrapping increment with the bound 2^32 - 1 is the same operation, is left alone, and with the
    // index pinned in its vector register until the end of the tile the wait moves there.  MEASURED NEGATIVE: render launches 2.35 -> 2.46 ms,
    // volume launch unchanged -- the other wavefront of the SIMD covers the wait, the longer live range costs the render instantiation more.)
#if GNR_DYN_CTR_INC
    auto grab = [&]() -> unsigned { unsigned t = 0u; if (lane == 0) t = atomicInc(a.tile_ctr + cx, 0xFFFFFFFFu); return t; };
#else
    auto grab = [&]() -> unsigned { unsigned t = 0u; if (lane == 0) t = atomicAdd(a.tile_ctr + cx, 1u); return t; };
#endif
    auto resolve = [&](unsigned t_raw) -> int {            // fetched index -> tile; -1: nothing left
        int t = __builtin_amdgcn_readfirstlane((int)t_raw);
#if GNR_DYN_TILES > 1
        for (int hops = 0; t >= min(ntiles, (cx + 1) * chunk) - cx * chunk;) {      // measured negative: walk on to the next XCD's chunk
            if (++hops >= NXCD) return -1;
            cx = cx + 1 == NXCD ? 0 : cx + 1;
            t = __builtin_amdgcn_readfirstlane((int)grab());
        }
        return cx * chunk + t;
#else
        return t < t_end - xcd * chunk ? xcd * chunk + t : -1;
#endif
    };
    // scene and (brick-permuted) tile-in-scene of a tile of the list
    auto locate = [&](int tile_, int& b_, int& ts_) {
        b_ = tile_ / tps;
        ts_ = tile_ - b_ * tps;
        b_ += a.b0;
        if (a.vol_res > 0) {
            // Volume points are stored column-major (x, y, then z top->down).  Visit them in bricks of
            // (R/4) x (R/4) columns instead of whole x-planes: at R = 40 a brick (100 columns, 250 tiles) is what the
            // 256 wavefronts of an XCD have in flight, and a compact block of columns projects to the smallest union
            // of feature-map lines (an x-plane covers ~8 MB of them, an XCD's L2 holds 4 MiB).  Measured L2 fill per
            // launch: x-planes 2.45 GB, 4x4-column bricks 1.83 GB, 8x8 1.50 GB, 10x10 1.44 GB, 20x20 1.54 GB.
            // Bijection on 2-column groups (2R points = R/8 tiles); requires R % 8 == 0 (host checks).
            const int R = a.vol_res, tpg = R >> 3, gpr = R >> 1;
            const int BX = R >> 2, GY = R >> 3, GB = BX * GY;             // a plane = 4 x 4 bricks of (R/4) x (R/4) columns
            const int sidx = ts_ / tpg, tin = ts_ - sidx * tpg;
            const int brick = sidx / GB, wi = sidx - brick * GB;
            const int bx = brick >> 2, by = brick & 3;
            ts_ = ((BX * bx + wi / GY) * gpr + GY * by + wi % GY) * tpg + tin;
        }
    };
    auto point_of = [&](int b_, int ts_) -> size_t { const int nr = ts_ * 16 + r; return (size_t)b_ * a.P + (nr < a.P ? nr : a.P - 1); };
    int tile = dyn ? resolve(grab()) : xcd * chunk + lblk * waves_per_block + wave;
    constexpr bool DPF = GNR_DESC_PREFETCH != 0 && !SAVE;      // the next tile's point descriptors are fetched behind the second view loop
    f4 d0n = {0.f, 0.f, 0.f, 0.f}, d1n = d0n;
    bool have_nxt = false;                                  // wave-uniform
    for (; dyn ? tile >= 0 : tile < t_end;) {
        GNR_ITER_FENCE();
        unsigned tile_nxt = 0u;
        int tile_after = -1;
#if GNR_WAVE_CLOCK
        const unsigned long long wc_tt = wall_clock64();
#endif
        int b, ts;
        locate(tile, b, ts);
        const int n_raw = ts * 16 + r;
        const bool row_ok = n_raw < a.P;
        const int n = row_ok ? n_raw : a.P - 1;
        const size_t pt = (size_t)b * a.P + n;
        size_t opt = pt;                                   // where the point's user-visible outputs go (caller's ray order)
        if (RENDER && !SAVE && a.ray_perm) {
            const int s_ = n / a.perm_dn;
            opt = (size_t)b * a.P + (size_t)a.ray_perm[(size_t)b * a.perm_rn + s_] * a.perm_dn + (n - s_ * a.perm_dn);
        }
        float p[3], qd[3], lo, hi;
        {
            f4 d0, d1;
            if (DPF && have_nxt) { d0 = d0n; d1 = d1n; }
            else if (!RENDER && !SAVE && GNR_VOLUME_POINTS_INLINE != 0 && a.pts_R > 0) {
                // the grid point of voxel n of scene b, as k_points_volume writes it (field_utils.py:17-27: float64 arithmetic, then the cast;
                // renderer.py:167-170: + bbox_min, z from the top down): one launch and one exposed load per tile less
                const int R = a.pts_R;
                const float invR = 1.f / (float)R;
                const int col = (int)(((float)n + 0.5f) * invR), sz = n - col * R;          // exact for n < 2^22
                const int ix = (int)(((float)col + 0.5f) * invR), iy = col - ix * R, iz = R - 1 - sz;
                const double vs = 0.3 / (double)R, hv = vs / 2;
                const float* bm = a.bbox_min + b * 3;
                d0 = (f4){(float)(ix * vs + hv) + bm[0], (float)(iy * vs + hv) + bm[1], (float)(iz * vs + hv) + bm[2], 0.f};
                d1 = (f4){0.f, 1.f, 0.005f, 0.005f};
            }
            else { d0 = reinterpret_cast<const f4*>(a.desc)[pt * 2]; d1 = reinterpret_cast<const f4*>(a.desc)[pt * 2 + 1]; }
            p[0] = d0.x; p[1] = d0.y; p[2] = d0.z; qd[0] = d0.w; qd[1] = d1.x; qd[2] = d1.y; lo = d1.z; hi = d1.w;
        }
        constexpr int VA = rows_a(V), VB = rows_b(V);
        float SA[VA][SW], SB[VB > 0 ? VB : 1][SW];
        // row of view v (v a compile-time constant after unrolling: the branch folds)
        auto S = [&](int v, int q) -> float& { return (VB == 0 || v < VA) ? SA[v < VA ? v : 0][q] : SB[v >= VA ? v - VA : 0][q]; };
#pragma unroll
        for (int k = 0; k < V; ++k)
#pragma unroll
#if GNR_ABLATE & 1
            for (int q = 0; q < SW; ++q) S(k, q) = p[0] * (float)(k * SW + q + 1);      // opaque values: nothing downstream folds away
#else
            for (int q = 0; q < SW; ++q) S(k, q) = 0.f;
#endif
        float msum = 0.f;
        unsigned vbits = 0;
#if GNR_ABLATE & 2
        ViewGeom vg0;
        project_view<true>(a.viewp + b * V * VIEWP_FLOATS, p, qd, a.H, a.W, vg0);
        const Taps t0 = make_taps(vg0.u, vg0.v, fsx, fsy, -0.5f, a.fh, a.fw);
        const Taps ti0 = make_taps(vg0.u, vg0.v, 1.f, 1.f, 0.f, a.H, a.W);
#endif

        // ================= phase 1: per view, everything up to the first cross-view reduction
        // Q: one of the two queues; the views v0 .. v0 + rows - 1 enter it in order (the loop is instantiated once per queue)
        auto phase1 = [&](auto& Q, const int v0) __attribute__((always_inline)) {
          constexpr int NQ = (int)(sizeof(Q) / sizeof(Q[0]));
#pragma unroll 1
          for (int v = v0; v < v0 + NQ; ++v) {
#if !(GNR_ABLATE & 1)
#pragma unroll
            for (int k = 0; k < NQ - 1; ++k)
#pragma unroll
                for (int q = 0; q < SW; ++q) Q[k][q] = Q[k + 1][q];
#endif
            GNR_ITER_FENCE();
            float (&Sv)[SW] = Q[NQ - 1];
            const int bv = b * V + v;
            const float* vp = a.viewp + bv * VIEWP_FLOATS;
            ViewGeom vg;
            Taps t, ti;
#if GNR_ABLATE & 2
            vg = vg0; t = t0; ti = ti0;
#elif GNR_COOP_PROJ
            {
                CoopOut co;
                project_coop(vp, lds + COOP_TAB + g * 8, (g & 1) * 16, (lane & 15) << 2, p, co);
                view_dir(vp, p, qd, vg.dd);
                vg.z = co.z; vg.m = co.m; t = co.f; ti = co.i;
            }
#else
            project_view<true>(vp, p, qd, a.H, a.W, vg);
            t = make_taps(vg.u, vg.v, fsx, fsy, -0.5f, a.fh, a.fw);
            ti = make_taps(vg.u, vg.v, 1.f, 1.f, 0.f, a.H, a.W);
#endif
            const float m = vg.m;
            msum += m;
            vbits |= (m != 0.f ? 1u : 0u) << v;
#if GNR_SKIP_MASKED
            // No point of the tile lies inside this view (render_ops.py:24-31 mask == 0 for all 16): everything the view would hand to
            // the cross-view reductions carries the weight m = 0 (ibrnet.py:466-472: x, e1 and the gate enter as value * weight), so its
            // row is zeros and the gathers, the decoder and the embedding are not run.  A quarter of a ray's (sample, view) pairs are
            // masked on the benchmark's cameras and 7 % of its (tile, view) pairs entirely so; on the volume points next to none (1.3 % of
            // the pairs), and there the branch costs more than it saves (volume launch 3.28 -> 3.35 ms): ray points only.
            if constexpr (!SAVE && RENDER) {
                if (__ballot(m != 0.f) == 0ull) {
#pragma unroll
                    for (int q = 0; q < SW; ++q) Sv[q] = 0.f;
                    if (a.dbg && g == 0 && row_ok) { a.dbg[pt * 32 + v] = 0.f; a.dbg[pt * 32 + 8 + v] = 0.f; }
                    continue;
                }
            }
#endif

            // ---- gather: ray channels 8g..8g+7, image-feature channels 8g..8g+7, rgb channel g
            float FR[8], XI[9];
            {
                const float* fb = a.feat64 + (size_t)bv * a.fh * a.fw * 64 + 8 * g;
                const f4* q00 = reinterpret_cast<const f4*>(fb + (size_t)t.o00 * 64);
                const f4* q01 = reinterpret_cast<const f4*>(fb + (size_t)t.o01 * 64);
                const f4* q10 = reinterpret_cast<const f4*>(fb + (size_t)t.o10 * 64);
                const f4* q11 = reinterpret_cast<const f4*>(fb + (size_t)t.o11 * 64);
                const float w00 = t.w00 * m, w01 = t.w01 * m, w10 = t.w10 * m, w11 = t.w11 * m;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f4 a0 = q00[h], a1 = q01[h], a2 = q10[h], a3 = q11[h];
                    const f4 c0 = q00[8 + h], c1 = q01[8 + h], c2 = q10[8 + h], c3 = q11[8 + h];
                    const f4 fr = a0 * w00 + a1 * w01 + a2 * w10 + a3 * w11;
                    const f4 fi = c0 * w00 + c1 * w01 + c2 * w10 + c3 * w11;
                    FR[4 * h] = fr.x; FR[4 * h + 1] = fr.y; FR[4 * h + 2] = fr.z; FR[4 * h + 3] = fr.w;
                    XI[4 * h] = fi.x; XI[4 * h + 1] = fi.y; XI[4 * h + 2] = fi.z; XI[4 * h + 3] = fi.w;
                }
                const float* ib = a.imgs + ((size_t)bv * 3 + min(g, 2)) * a.H * a.W;
                const float rgb = (ib[ti.o00] * ti.w00 + ib[ti.o01] * ti.w01 + ib[ti.o10] * ti.w10 + ib[ti.o11] * ti.w11) * m;
                XI[8] = g < 3 ? rgb : 0.f;
            }
            Sv[19] = XI[8];                              // raw rgb for the colour blend

            // ---- mixture-of-logistics decoder (dist_decoder.py:99-142)
            float o5[5];
            P8 frp;
            if constexpr (SP) frp = split8<0>(FR);
#pragma unroll
            for (int br = 0; br < 3; ++br) {
                f4 acc[2];
                float h1[8], h2[8];
                load_bias<2, LF>(lds + LO(pk::B_DEC1) + br * 32, g, acc);
                if constexpr (SP) mm16<1, 2, LF>(lds + LO(pk::DEC1) + br * frag_floats(8, 2), lane, &frp, acc);
                else mm<8, 2, 0, LF>(lds + LO(pk::DEC1) + br * frag_floats(8, 2), lane, FR, acc);
                elu_to<2, !SP>(acc, h1);
                load_bias<2, LF>(lds + LO(pk::B_DEC2) + br * 32, g, acc);
                if constexpr (SP) {
                    if constexpr (UA) { const P8U hp = split8u<0>(h1); mm16u<1, 2, LF, true>(lds + LO(pk::DEC2) + br * frag_floats(8, 2), lane, &hp, acc, &msum); }
                    else { const P8 hp = split8<0>(h1); mm16<1, 2, LF, true>(lds + LO(pk::DEC2) + br * frag_floats(8, 2), lane, &hp, acc, &msum); }
                }
                else mm<8, 2, 0, LF>(lds + LO(pk::DEC2) + br * frag_floats(8, 2), lane, h1, acc);
                elu_to<2, !SP || UA>(acc, h2);
                if (br < 2) {
                    o5[2 * br] = gsum(dot8(lds + LO(pk::T_DEC3) + (2 * br) * 32, g, h2)) + lds[LO(pk::T_DEC3_B) + 2 * br];
                    o5[2 * br + 1] = gsum(dot8(lds + LO(pk::T_DEC3) + (2 * br + 1) * 32, g, h2)) + lds[LO(pk::T_DEC3_B) + 2 * br + 1];
                } else {
                    o5[4] = gsum(dot8(lds + LO(pk::T_DEC3) + 4 * 32, g, h2)) + lds[LO(pk::T_DEC3_B) + 4];
                }
            }
            float hit, vis;
            {
                // The four lane groups of a point hold the same o5: each evaluates ONE of the four logistic CDFs (mixture
                // component g & 1 at the near (g < 2) or far edge) -- 2 softplus + 1 tanh per lane instead of 4 + 4 -- and the
                // mixture sums over components / edges are the cross-group sums the dot products use anyway.
                const bool m1 = (g & 1) != 0, far_edge = g >= 2;
                const float meang = softplus1(m1 ? o5[1] : o5[0]);
                const float varg = softplus1(m1 ? o5[3] : o5[2]) + 0.05f;
                const float mix = sigmoid1(m1 ? -o5[4] : o5[4]);             // aw or 1 - aw = sigmoid(-logit)   (:128)
                const float dinv = -rcp1(fmaxf(vg.z, 1e-5f));                 // dist_decoder.py:21-23
                const float dhat = (dinv - vp[15]) * vp[17];
                const float edge = far_edge ? dhat + hi : dhat - lo;
                float cg = 0.5f + 0.5f * tanh1((edge - meang) * varg);
                if constexpr (USEVIS) {   // dist_decoder_cfg.use_vis (dist_decoder.py:89-97,103-104,133-134); off in nrvgn_sdf.yaml
                    f4 acc[2];
                    float h1[8], h2[8];
                    load_bias<2, LF>(lds + LO(pk::B_DECV1), g, acc);
                    if constexpr (SP) mm16<1, 2, LF>(lds + LO(pk::DECV1), lane, &frp, acc);
                    else mm<8, 2, 0, LF>(lds + LO(pk::DECV1), lane, FR, acc);
                    elu_to<2, !SP>(acc, h1);
                    load_bias<2, LF>(lds + LO(pk::B_DECV2), g, acc);
                    if constexpr (SP) {
                        if constexpr (UA) { const P8U hp = split8u<0>(h1); mm16u<1, 2, LF, true>(lds + LO(pk::DECV2), lane, &hp, acc, &msum); }
                        else { const P8 hp = split8<0>(h1); mm16<1, 2, LF, true>(lds + LO(pk::DECV2), lane, &hp, acc, &msum); }
                    }
                    else mm<8, 2, 0, LF>(lds + LO(pk::DECV2), lane, h1, acc);
                    elu_to<2, !SP || UA>(acc, h2);
                    const float pv = sigmoid1(gsum(dot8(lds + LO(pk::T_DECV3), g, h2)) + lds[LO(pk::T_VIS)]);
                    cg *= pv;
                }
                // visibility = sum_mix (1 - cdf0) mix,  hit_prob = sum_mix (cdf1 - cdf0) mix     (dist_decoder.py:135-138)
                const float mc = mix * cg;
                vis = gsum(far_edge ? 0.f : mix - mc) * m;
                hit = gsum(far_edge ? mc : -mc) * m;
            }
            // ---- prob embedding 34 -> 32 -> 32 (aggregate_net.py:46-54)
            {
                f4 acc[2];
                float e1[8];
                load_bias<2, LF>(lds + LO(pk::B_PE1), g, acc);
                if constexpr (SP) mm16<1, 2, LF>(lds + LO(pk::PE1), lane, &frp, acc);
                else mm<8, 2, 0, LF>(lds + LO(pk::PE1), lane, FR, acc);
                const float extra[1] = {g == 0 ? (hit - 0.5f) * 2.f : (g == 1 ? (vis - 0.5f) * 2.f : 0.f)};
                mm<1, 2, 8, LF>(lds + LO(pk::PE1), lane, extra, acc);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    e1[nb * 4 + 0] = fmaxf(acc[nb].x, 0.f); e1[nb * 4 + 1] = fmaxf(acc[nb].y, 0.f);
                    e1[nb * 4 + 2] = fmaxf(acc[nb].z, 0.f); e1[nb * 4 + 3] = fmaxf(acc[nb].w, 0.f);
                }
                // prob_embed.2 is folded into neuray_fc.0 / base_fc.0 on the host: the state kept per view is e1 -- as its fp16
                // pair where nothing needs the fp32 values (inference, pair build), so that it is split once, not twice
                if constexpr (EP) p8_store(split8<0>(e1), &Sv[9]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) Sv[9 + j] = e1[j];
                }
            }
            // ---- x = [rgb, img_feats] + ray_dir_fc(dir_diff)  (ibrnet.py:457-459)
            {
                f4 acc1[1], acc3[3];
                float d1[4], df[12];
                load_bias<1, LF>(lds + LO(pk::B_RDF1), g, acc1);
                const float ddg[1] = {g == 0 ? vg.dd[0] : (g == 1 ? vg.dd[1] : (g == 2 ? vg.dd[2] : vg.dd[3]))};
                mm<1, 1, 0, LF>(lds + LO(pk::RDF1), lane, ddg, acc1);
                elu_to<1, true>(acc1, d1);
                load_bias<3, LF>(lds + LO(pk::B_RDF2), g, acc3);
                // 4 k-steps: stays on the fp32 MFMA (zero-padded into a K32 pair block: 3.68 vs 3.71 ms alone, but 3.80 together
                // with the padded tails of HOIST / GEO1, which gain more: 3.63)
#if GNR_RDF2_PAIRS
                if constexpr (SP && !USEVIS) { const P8 dp = split8z<0, 4>(d1); mm16<1, 3, LF>(lds + pk::c16_off(pk::DECV1), lane, &dp, acc3); }
                else
#endif
                mm<4, 3, 0, LF>(lds + LO(pk::RDF2), lane, d1, acc3);
                elu_to<3, true>(acc3, df);
#pragma unroll
                for (int j = 0; j < 9; ++j) Sv[j] = fmaf(df[j], kLn2, XI[j]);
            }
            // ---- gate of the first weighted mean/var: sigmoid(neuray_fc(e))  (ibrnet.py:469)
            {
                f4 acc1[1];
                float e[8], n1[4];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = Sv[9 + j];
                load_bias<1, LF>(lds + LO(pk::B_NR1), g, acc1);
                if constexpr (SP) { const P8 ep = EP ? p8_load(e) : split8<0>(e); mm16<1, 1, LF, true>(lds + LO(pk::NR1), lane, &ep, acc1, &msum); }
                else mm<8, 1, 0, LF>(lds + LO(pk::NR1), lane, e, acc1);
                elu_to<1, !SP>(acc1, n1);
                Sv[17] = sigmoid1(gsum(dot4(lds + LO(pk::T_NR2), g, n1)) + lds[LO(pk::T_SCAL) + 0]);
            }
            Sv[18] = m;
            if (SAVE && a.save1) {
                float* sp = a.save1 + (((size_t)b * tps + ts) * V + v) * 19 * 64 + lane;
#pragma unroll
                for (int q = 0; q < 19; ++q) sp[q * 64] = Sv[q];
            }
            if (a.dbg && g == 0 && row_ok) { a.dbg[pt * 32 + v] = hit; a.dbg[pt * 32 + 8 + v] = vis; }
          }
        };
        phase1(SA, 0);
        if constexpr (VB > 0) phase1(SB, VA);
#if GNR_WAVE_CLOCK
        const unsigned long long wc_p1 = wall_clock64();
#endif

        // ================= cross-view reduction 1 (ibrnet.py:466-472), in-lane
        float SV[36];
        const float inv_msum = rcp1(msum + 1e-8f);
        {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                float mu0 = 0.f, mu1 = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float w = S(v, 18) * inv_msum;
                    mu0 += S(v, j) * (S(v, 17) * w);
                    mu1 += S(v, j) * w;
                }
                float va0 = 0.f, va1 = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float w = S(v, 18) * inv_msum;
                    const float d0 = S(v, j) - mu0, d1 = S(v, j) - mu1;
                    va0 += (S(v, 17) * w) * d0 * d0;
                    va1 += w * d1 * d1;
                }
                SV[j] = mu0; SV[9 + j] = va0; SV[18 + j] = mu1; SV[27 + j] = va1;
            }
        }
        // view-invariant 140 columns of base_fc.0, once per point (+ bias)
        f4 G[4];
        load_bias<4, LF>(lds + LO(pk::B_HOIST), g, G);
        if constexpr (SP) {
            const P8 sp[5] = {split8<0>(SV), split8<8>(SV), split8<16>(SV), split8<24>(SV), split8z<32, 4>(SV)};
            mm16<5, 4, LF, true>(lds + LO(pk::HOIST), lane, sp, G, &msum);
        } else mm<36, 4, 0, LF>(lds + LO(pk::HOIST), lane, SV, G);
        if (SAVE && a.saveG) {
            float* sp = a.saveG + ((size_t)b * tps + ts) * 17 * 64 + lane;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                sp[(4 * nb) * 64] = G[nb].x; sp[(4 * nb + 1) * 64] = G[nb].y; sp[(4 * nb + 2) * 64] = G[nb].z; sp[(4 * nb + 3) * 64] = G[nb].w;
            }
            sp[16 * 64] = inv_msum;
        }

#if GNR_WAVE_CLOCK
        const unsigned long long wc_p2 = wall_clock64();
#endif
        // ================= phase 2: per view, base_fc / vis_fc / vis_fc2 (/ rgb_fc)
        float vsum = 0.f;
        // the oldest row of the queue is copied out, the queue advances, the view's results become its last row
        auto phase2 = [&](auto& Q, const int v0) __attribute__((always_inline)) {
          constexpr int NQ = (int)(sizeof(Q) / sizeof(Q[0]));
#pragma unroll 1
          for (int v = v0; v < v0 + NQ; ++v) {
            constexpr int RN = 20;
            float Rv[RN];
#pragma unroll
            for (int j = 0; j < RN; ++j) Rv[j] = Q[0][j];
#if !(GNR_ABLATE & 1)
#pragma unroll
            for (int k = 0; k < NQ - 1; ++k)
#pragma unroll
                for (int q = 0; q < SW; ++q) Q[k][q] = Q[k + 1][q];
#endif
            GNR_ITER_FENCE();
            float X[9], E[8];
#pragma unroll
            for (int j = 0; j < 9; ++j) X[j] = Rv[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) E[j] = Rv[9 + j];
            const float m = Rv[18];
            const float w = m * inv_msum;
#if GNR_SKIP_MASKED
            if constexpr (!SAVE && RENDER) {                                    // the same (tile, view) pairs as in phase 1: visibility 0,
                if (__ballot(m != 0.f) == 0ull) {                               // colour logit -1e9, features with weight 0 (ibrnet.py:479-484,509)
#pragma unroll
                    for (int j = 0; j < 9; ++j) Q[NQ - 1][j] = 0.f;
                    Q[NQ - 1][9] = RENDER ? -1e9f : 0.f;
                    Q[NQ - 1][19] = Rv[19];
                    if (a.dbg && g == 0 && row_ok) a.dbg[pt * 32 + 16 + v] = 0.f;
                    continue;
                }
            }
#endif
            float Hh[8];
            {
                f4 acc4[4] = {G[0], G[1], G[2], G[3]};
                float b1[16];
                if constexpr (SP) {
                    const float x8[1] = {X[8]};
                    mm<1, 4, 0, LF>(lds + LO(pk::BASE1) + 2 * pk::k32_floats(4), lane, x8, acc4);
                    const P8 xe[2] = {split8<0>(X), EP ? p8_load(E) : split8<0>(E)};
                    mm16<2, 4, LF, true>(lds + LO(pk::BASE1), lane, xe, acc4, &msum);
                } else {
                    mm<9, 4, 0, LF>(lds + LO(pk::BASE1), lane, X, acc4);
                    mm<8, 4, 9, LF>(lds + LO(pk::BASE1), lane, E, acc4);
                }
                elu_to<4, !SP>(acc4, b1);
                f4 acc[2];
                load_bias<2, LF>(lds + LO(pk::B_BASE2), g, acc);
                if constexpr (SP) {
                    if constexpr (UA) { const P8U bp[2] = {split8u<0>(b1), split8u<8>(b1)}; mm16u<2, 2, LF, true>(lds + LO(pk::BASE2), lane, bp, acc, &msum); }
                    else { const P8 bp[2] = {split8<0>(b1), split8<8>(b1)}; mm16<2, 2, LF, true>(lds + LO(pk::BASE2), lane, bp, acc, &msum); }
                }
                else mm<16, 2, 0, LF>(lds + LO(pk::BASE2), lane, b1, acc);
                elu_to<2, !SP || UA>(acc, Hh);
            }
            float vis1;
            {
                f4 acc[2];
                float xin[8], v1[8], res[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xin[j] = Hh[j] * w;
                load_bias<2, LF>(lds + LO(pk::B_VIS1), g, acc);
                if constexpr (SP) {
                    if constexpr (UA) { const P8U xp = split8u<0>(xin); mm16u<1, 2, LF, true>(lds + LO(pk::VIS1), lane, &xp, acc, &msum); }
                    else { const P8 xp = split8<0>(xin); mm16<1, 2, LF, true>(lds + LO(pk::VIS1), lane, &xp, acc, &msum); }
                }
                else mm<8, 2, 0, LF>(lds + LO(pk::VIS1), lane, xin, acc);
                elu_to<2, !SP || UA>(acc, v1);
                load_bias<2, LF>(lds + LO(pk::B_VIS2), g, acc);
                if constexpr (SP) {
                    if constexpr (UA) { const P8U vp8 = split8u<0>(v1); mm16u<1, 2, LF, true>(lds + LO(pk::VIS2), lane, &vp8, acc, &msum); }
                    else { const P8 vp8 = split8<0>(v1); mm16<1, 2, LF, true>(lds + LO(pk::VIS2), lane, &vp8, acc, &msum); }
                }
                else mm<8, 2, 0, LF>(lds + LO(pk::VIS2), lane, v1, acc);
                elu_to<2, !SP || UA>(acc, res);
                const float logit = elu1(gsum(dot8(lds + LO(pk::T_VIS2R), g, v1)) + lds[LO(pk::T_SCAL) + 1]);
                vis1 = sigmoid1(logit) * m;                                    // ibrnet.py:479
#pragma unroll
                for (int j = 0; j < 8; ++j) Hh[j] += res[j];                    // :480
            }
            float v2;
            {
                f4 acc[2];
                float xin[8], t1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xin[j] = Hh[j] * vis1;
                load_bias<2, LF>(lds + LO(pk::B_VISB1), g, acc);
                if constexpr (SP) {
                    if constexpr (UA) { const P8U xp = split8u<0>(xin); mm16u<1, 2, LF, true>(lds + LO(pk::VISB1), lane, &xp, acc, &msum); }
                    else { const P8 xp = split8<0>(xin); mm16<1, 2, LF, true>(lds + LO(pk::VISB1), lane, &xp, acc, &msum); }
                }
                else mm<8, 2, 0, LF>(lds + LO(pk::VISB1), lane, xin, acc);
                elu_to<2, !SP || UA>(acc, t1);
                v2 = sigmoid1(gsum(dot8(lds + LO(pk::T_VISB2), g, t1)) + lds[LO(pk::T_SCAL) + 2]) * m;   // :481
            }
            vsum += v2;
            float clog = 0.f;
            if constexpr (RENDER) {                                             // ibrnet.py:507-509
                const float* vp = a.viewp + (b * V + v) * VIEWP_FLOATS;
                ViewGeom vg;
                project_view<true>(vp, p, qd, a.H, a.W, vg);
                f4 acc1[1];
                float c1[4], c2[4];
                load_bias<1, LF>(lds + LO(pk::B_RGB1), g, acc1);
                const float ex[2] = {g == 0 ? v2 : (g == 1 ? vg.dd[0] : (g == 2 ? vg.dd[1] : vg.dd[2])), g == 0 ? vg.dd[3] : 0.f};
                mm<2, 1, 8, LF>(lds + LO(pk::RGB1), lane, ex, acc1);
                if constexpr (SP) { const P8 hp = split8<0>(Hh); mm16<1, 1, LF, true>(lds + LO(pk::RGB1), lane, &hp, acc1, &msum); }
                else mm<8, 1, 0, LF>(lds + LO(pk::RGB1), lane, Hh, acc1);
                elu_to<1, !SP>(acc1, c1);
                load_bias<1, LF>(lds + LO(pk::B_RGB2), g, acc1);
                mm<4, 1, 0, LF>(lds + LO(pk::RGB2), lane, c1, acc1);               // 4 k-steps, one output block: stays on the fp32 MFMA
                elu_to<1, true>(acc1, c2);
                clog = gsum(dot4(lds + LO(pk::T_RGB3), g, c2)) + lds[LO(pk::T_SCAL) + 3];
                if (m == 0.f) clog = -1e9f;
            }
            float Ov[10];
#pragma unroll
            for (int j = 0; j < 8; ++j) Ov[j] = Hh[j];
            Ov[8] = v2;
            Ov[9] = clog;
            if (SAVE && a.save2) {
                constexpr int S2W = RENDER ? 11 : 9;
                float* sp = a.save2 + (((size_t)b * tps + ts) * V + v) * S2W * 64 + lane;
#pragma unroll
                for (int q = 0; q < (S2W < 10 ? S2W : 10); ++q) sp[q * 64] = Ov[q];
                if constexpr (RENDER) sp[10 * 64] = Rv[RN - 1];
            }
#pragma unroll
            for (int j = 0; j < 10; ++j) Q[NQ - 1][j] = Ov[j];
            Q[NQ - 1][19] = Rv[19];
            if (a.dbg && g == 0 && row_ok) a.dbg[pt * 32 + 16 + v] = v2;
          }
        };
        phase2(SA, 0);
        if constexpr (VB > 0) phase2(SB, VA);
#if GNR_WAVE_CLOCK
        const unsigned long long wc_p3 = wall_clock64();
#endif
        if (dyn) {
            // GNR_TAIL_OLD_ONLY: the younger half of a workgroup (wavefronts 4-7: the SIMD issues them second, a tile takes them ~1.5x as long)
            // leaves the chunk's last tiles to the older half -- a wavefront whose current index is within TAIL_K of the chunk's end fetches no further one
            bool more = true;
#if GNR_TAIL_OLD_ONLY
            if (wave >= 4) more = (t_end - tile) > (GNR_TAIL_OLD_ONLY * nlblk * waves_per_block) / 4;
#endif
            tile_nxt = more ? grab() : 0x7fffffffu;
        }
        if constexpr (DPF) {
            tile_after = dyn ? resolve(tile_nxt) : tile + nlblk * waves_per_block;
            have_nxt = dyn ? tile_after >= 0 : tile_after < t_end;
            if (have_nxt) {
                int bn, tsn;
                locate(tile_after, bn, tsn);
                const size_t ptn = point_of(bn, tsn);
                d0n = reinterpret_cast<const f4*>(a.desc)[ptn * 2]; d1n = reinterpret_cast<const f4*>(a.desc)[ptn * 2 + 1];
            }
        }

        // ================= cross-view reduction 2 (ibrnet.py:482-484,488) + colour blend (:510-511)
        float Z[23];
        float wbar = 0.f;
        {
            const float inv_vsum = rcp1(vsum + 1e-8f);
#pragma unroll
            for (int v = 0; v < V; ++v) wbar += S(v, 8) * inv_vsum;
            wbar *= (1.f / (float)V);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float mu = 0.f, va = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) mu += S(v, j) * (S(v, 8) * inv_vsum);
#pragma unroll
                for (int v = 0; v < V; ++v) { const float d = S(v, j) - mu; va += (S(v, 8) * inv_vsum) * d * d; }
                Z[j] = mu; Z[8 + j] = va;
            }
        }
        if constexpr (RENDER) {
            float cmax = -3.0e38f;
#pragma unroll
            for (int v = 0; v < V; ++v) cmax = fmaxf(cmax, S(v, 9));
            float den = 0.f, num = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) { const float e = __expf(S(v, 9) - cmax); den += e; num += S(v, 19) * e; }
            if (g < 3 && row_ok) a.colors[opt * 3 + g] = num * rcp1(den);
        }
        if (SAVE && a.saveZ) {
            float* sp = a.saveZ + ((size_t)b * tps + ts) * 18 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) sp[q * 64] = Z[q];
            sp[16 * 64] = wbar; sp[17 * 64] = msum;
        }
        // ================= geometry_fc on [mean, var, wbar, embed(p)]  (ibrnet.py:487-489)
        {
            const float pc = g == 1 ? p[0] : (g == 2 ? p[1] : p[2]);
            float s1, c1, s2, c2, s4, c4;
            sincos_octaves(pc, s1, c1, s2, c2, s4, c4);
            const bool g0 = g == 0;
            Z[16] = g0 ? wbar : pc; Z[17] = g0 ? 0.f : s1; Z[18] = g0 ? 0.f : c1; Z[19] = g0 ? 0.f : s2;
            Z[20] = g0 ? 0.f : c2; Z[21] = g0 ? 0.f : s4; Z[22] = g0 ? 0.f : c4;
        }
        f4 U[4];
        float u64[16];
        load_bias<4, LF>(lds + LO(pk::B_GEO1), g, U);
        if constexpr (SP) {
            const P8 zp[3] = {split8<0>(Z), split8<8>(Z), split8z<16, 7>(Z)};
            mm16<3, 4, LF, true>(lds + LO(pk::GEO1), lane, zp, U, &msum);
        } else mm<23, 4, 0, LF>(lds + LO(pk::GEO1), lane, Z, U);
        elu_to<4, !SP>(U, u64);
        f4 g16[1];
        load_bias<1, LF>(lds + LO(pk::B_GEO2), g, g16);
        if constexpr (SP) { const P8 up[2] = {split8<0>(u64), split8<8>(u64)}; mm16<2, 1, LF, true>(lds + LO(pk::GEO2), lane, up, g16, &msum); }
        else mm<16, 1, 0, LF>(lds + LO(pk::GEO2), lane, u64, g16);
        float gg[4];
        elu_to<1, !SP>(g16, gg);

        // ================= record
        if constexpr (SP && GNR_RANGE_GUARD != 0) range_tripped |= __ballot(msum != msum) != 0;
        if (row_ok) {
            float* rec = a.rec + pt * REC;
            const f4 gv = {gg[0] * kLn2, gg[1] * kLn2, gg[2] * kLn2, gg[3] * kLn2};    // back to true scale
            reinterpret_cast<f4*>(rec)[g] = gv;
            if constexpr (RENDER) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const f4 uv = {u64[nb * 4] * kLn2, u64[nb * 4 + 1] * kLn2, u64[nb * 4 + 2] * kLn2, u64[nb * 4 + 3] * kLn2};
                    reinterpret_cast<f4*>(rec + 16 + 16 * nb)[g] = uv;
                }
                if (g == 0) rec[80] = msum;
            } else {
                if (g == 0) rec[16] = msum;
            }
            if (a.vmask && g == 0) a.vmask[opt] = (unsigned char)vbits;
            if (a.dbg && g == 0) {
                float* d = a.dbg + pt * 32;
                d[24] = msum; d[25] = wbar; d[26] = Z[0] * kLn2; d[27] = Z[8] * kLn2 * kLn2; d[28] = SV[8]; d[29] = G[0].x * kLn2; d[30] = gg[0] * kLn2; d[31] = vsum;
            }
        }
#if GNR_WAVE_CLOCK
        if constexpr (SP && !SAVE) { const unsigned long long wc_e = wall_clock64(); ++wc_n; wc_ph[0] += wc_p1 - wc_tt; wc_ph[1] += wc_p2 - wc_p1; wc_ph[2] += wc_p3 - wc_p2; wc_ph[3] += wc_e - wc_p3;
            if (lane == 0 && tile < (1 << 17)) g_tile_clock[RENDER ? 1 : 0][tile] = (unsigned)(wc_e - wc_tt); }
#endif
#if GNR_DYN_CTR_INC
        if (dyn) asm volatile("" : "+v"(tile_nxt));       // the fetched index stays in its vector register until here
#endif
        if constexpr (DPF) tile = have_nxt ? tile_after : (dyn ? -1 : t_end);
        else tile = dyn ? resolve(tile_nxt) : tile + nlblk * waves_per_block;
    }
    if (dyn && lane == 0) {        // the launch's last wavefront (every other one has made its last, failing, fetches) re-arms the counters for the next launch
        if (atomicAdd(a.tile_ctr + 8, 1u) == gridDim.x * (unsigned)waves_per_block - 1u) {
#pragma unroll
            for (int x = 0; x < 9; ++x) atomicExch(a.tile_ctr + x, 0u);
        }
    }
#if GNR_WAVE_CLOCK
    if constexpr (SP && !SAVE) {
        if (lane == 0 && !a.only_if_flagged && blockIdx.x * 8 + wave < 4096) {
            unsigned long long* w = g_wave_clock[RENDER ? 1 : 0][blockIdx.x * 8 + wave];
            w[0] = wc_t0; w[1] = wc_t1; w[2] = wall_clock64(); w[3] = wc_n;
            w[4] = wc_ph[0]; w[5] = wc_ph[1]; w[6] = wc_ph[2]; w[7] = wc_ph[3];
        }
    }
#endif
    if constexpr (SP) {      // range guard
        if (range_tripped && lane == 0) {
            if (a.range_launch) atomicOr(a.range_launch, 2u);
            else if (a.range_flag) atomicOr(a.range_flag, 2u);
        }
    }
}

#undef LO

// ---------------------------------------------------------------------------------------
// k_ray: one lane per sample of a ray / voxel column.  A 256-thread workgroup packs floor(256/slots) rays
// back to back (slots = lanes per ray = max(dn, fdn)); rays straddle wavefront boundaries, so everything
// "per ray" goes through LDS + workgroup barriers, never through wave shuffles: 94 % of the lanes work at
// dn = 40 (one ray per wavefront would use 40 of 64).
// ---------------------------------------------------------------------------------------
struct RayArgs {
    const float* wpk;        // packed level blob (RAY section used)
    const float* rec;        // [nrays*dn][REC]
    const float* desc;       // [nrays*dn][8]
    const float* depth;      // [nrays*dn]                     (RENDER)
    const float* colors;     // [nrays*dn][3]                  (RENDER)
    const float* que_dr;     // [B][2]                         (RENDER)
    const unsigned char* vmask_pts;  // [nrays*dn] point order, or null
    int nrays, dn, rays_per_scene;
    int slots, rays_per_block, ray_stride;       // lanes per ray, rays per workgroup, LDS floats per ray (host: ray_geometry)
    // volume outputs
    float* volume;           // [B][R*R][R] with z flipped back
    unsigned char* vmask_out;
    // render outputs (nullable)
    float *sdf, *alpha, *hit, *pix, *rdepth, *gerr_part, *grad;
    unsigned char* rmask;
    int view_num, point_num;
    // inverse-CDF resampling (coarse pass only; nullable)
    float* fine_depth; int* fine_inds; int fdn;
    const float* fine_u;     // [nrays][fdn] caller-drawn samples (is_train) or null (eval midpoints)
    unsigned* range_word;    // RENDER, MM: bit 1 is set when the matrix-core tail produced a non-finite value (an operand or a weight beyond the fp16 range)
    const unsigned* only_if; // RENDER, !MM (the fp32 twin launch): run only if bit 1 of this word is set; null: always
    const int* ray_perm;     // RENDER, nullable: rec / desc are in Morton order of the rays (k_ray_order); slot s of scene b holds ray
                             // ray_perm[b][s]; every other array (inputs depth / colours / fine_u, all outputs) is in the caller's order
};

// LDS floats per ray after the attention scratch is dead (RENDER tail): per-ray reductions + resampling arrays
namespace rt {
constexpr int W = MAX_DN_FWD;    // row width of the per-ray arrays
constexpr int TF = 0;            // [W]  transmittance factors 1-alpha+1e-10
constexpr int RED = W;           // [6][W] hit*r, hit*g, hit*b, hit*z, (|grad|-1)^2, view-count flag
constexpr int HP = 7 * W;        // [W]  hit_prob + 1e-5
constexpr int DN = 8 * W;        // [W]  normalised inverse depth
constexpr int PD = 9 * W;        // [W]  pdf
constexpr int CD = 10 * W;       // [W+8]  cdf (dn+1)
constexpr int CE = 11 * W + 8;   // [W+8]  bin centres (dn+1)
constexpr int FD = 12 * W + 16;  // [W]  unsorted fine depth
constexpr int END = 13 * W + 16;
}

// logits of one query against key j for the 4 heads (natural domain; the backward twins' recomputation, gnr_bwd.inc); `ok`
// false reproduces the reference's query-row mask (whole row = -1e9 -> uniform softmax; ibrnet.py:19-23,492-493)
DEV void head_logits(const float (&q)[16], const float* __restrict__ Kj, bool ok, float (&s)[4]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f4 k = reinterpret_cast<const f4*>(Kj)[h];
        const float d = 0.5f * (q[4 * h] * k.x + q[4 * h + 1] * k.y + q[4 * h + 2] * k.z + q[4 * h + 3] * k.w);
        s[h] = ok ? d : -1e9f;
    }
}

// log2-domain logits of one query against key j for the 4 heads.  `q` arrives pre-scaled: qs = (log2e / 2) q, so that
// s_h = log2e (q_h . k_h) / 2 feeds v_exp_f32 directly, and qs = 0 on rows the reference masks (query-row mask: the whole
// row = -1e9 -> uniform softmax, ibrnet.py:19-23,492-493, SURVEY H3 -- equal logits of any value give the same softmax,
// and exp(0) = 1 is what exp(-1e9 - (-1e9)) gave)
DEV void head_logits_s(const float (&qs)[16], const float* __restrict__ Kj, float (&s)[4]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f4 k = reinterpret_cast<const f4*>(Kj)[h];
        s[h] = qs[4 * h] * k.x + qs[4 * h + 1] * k.y + qs[4 * h + 2] * k.z + qs[4 * h + 3] * k.w;
    }
}

// the same logits with the softmax shift as the seed of each dot product's FMA chain (SD): s_h - shift_h in four instructions per
// head instead of five, and one rounding less.  The seeded chain and the plain one differ by the rounding of the partial sums, so a
// row whose shift is its EXACT maximum (found with the plain chain) must subtract it from the plain chain too: only then is the
// largest exponent exactly 0 whatever the magnitude of the logits -- k_ray's `exact` rows run with SD = false.  (GNR_RAY_SEED=0: never seeded.)
#ifndef GNR_RAY_SEED
#define GNR_RAY_SEED 1
#endif
template <bool SD>
DEV void head_logits_shifted(const float (&qs)[16], const float* __restrict__ Kj, const float (&shift)[4], float (&s)[4]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f4 k = reinterpret_cast<const f4*>(Kj)[h];
        if constexpr (SD && GNR_RAY_SEED)
            s[h] = fmaf(qs[4 * h + 3], k.w, fmaf(qs[4 * h + 2], k.z, fmaf(qs[4 * h + 1], k.y, fmaf(qs[4 * h], k.x, -shift[h]))));
        else
            s[h] = (qs[4 * h] * k.x + qs[4 * h + 1] * k.y + qs[4 * h + 2] * k.z + qs[4 * h + 3] * k.w) - shift[h];
    }
}
// a . b - c for float4 a, b the same way
template <bool SD = true>
DEV float dot4_minus(float a0, float a1, float a2, float a3, const f4& b, float c) {
    if constexpr (SD && GNR_RAY_SEED) return fmaf(a3, b.w, fmaf(a2, b.z, fmaf(a1, b.y, fmaf(a0, b.x, -c))));
    else return (a0 * b.x + a1 * b.y + a2 * b.z + a3 * b.w) - c;
}

#ifndef GNR_RAY_UNROLL
#define GNR_RAY_UNROLL 1
#endif
// LDS per ray: K [dn][16], V [dn][16], squared key norms [dn][4]; the render kernel's column pass re-uses the same floats for
// Q [dn][16], dO [dn][16] and the row statistics [dn][12] once every lane is past the row pass (RAY_PER floats per sample: 7.2 KB
// per 40-sample ray, 43 KB per workgroup of six rays), which lets THREE workgroups share a CU (launch bounds 3: <= 168
// registers) where the separate Q / dO / statistics arrays (76 floats per sample) allowed two.
#ifndef GNR_RAY_BLOCKS
#define GNR_RAY_BLOCKS 3
#endif
constexpr int ray_per(bool render) { return render ? (GNR_RAY_BLOCKS >= 3 ? 44 : 80) : 36; }
// GNR_RAY_GEO_MFMA: geometry_fc's backward inside k_ray<true> (2 632 of the ~12 500 vector instructions of a wavefront as fp32 FMAs) on the
// f16 matrix cores, 16 samples as MFMA columns; 0: the fp32 FMA form
#ifndef GNR_RAY_GEO_MFMA
#define GNR_RAY_GEO_MFMA (GNR_RAY_BLOCKS >= 3)
#endif
// wave-level ordering of LDS traffic between lanes of the same wavefront
DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// MM (RENDER only): the VJP's dense tail on the f16 matrix cores (the product); false: as fp32 FMAs -- the instantiation launched right
// behind every MM launch, which returns at once unless that launch's range watch tripped (RayArgs::only_if) and otherwise recomputes it
template <bool RENDER, bool MM = (GNR_RAY_GEO_MFMA != 0)>
__global__ __launch_bounds__(256, (RENDER ? GNR_RAY_BLOCKS : 2)) void k_ray(RayArgs a) {
    if constexpr (RENDER && !MM) {
        if (a.only_if && (__builtin_nontemporal_load(a.only_if) & 2u) == 0u) return;
    }
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dn = a.dn, S = a.slots, rpb = a.rays_per_block;
    // thread -> (ray in block, slot).  Threads beyond the last ray of the block / launch shadow the last valid
    // ray (clamped indices, no stores) so that every thread reaches every barrier.
    int rl = (int)threadIdx.x / S;
    const int slot = (int)threadIdx.x - rl * S;
    const bool in_blk = rl < rpb;
    rl = min(rl, rpb - 1);
    const int ray_u = blockIdx.x * rpb + rl;
    const bool rvalid = in_blk && ray_u < a.nrays;
    const int ray = min(ray_u, a.nrays - 1);
    const bool act = rvalid && slot < dn;
    const int i = min(slot, dn - 1);
    constexpr bool OVL = RENDER && GNR_RAY_BLOCKS >= 3;   // Q / dO / statistics overlay K / V / key norms
    float* sc = sm + (size_t)rl * a.ray_stride;
    float* Kb = sc;                                   // [dn][16]
    float* Vb = sc + dn * 16;                         // [dn][16]
    float* KN = sc + dn * 32;                         // [dn][4]    squared key norms per head
    constexpr int KNS = 4;
    float* Qb = OVL ? sc : sc + dn * 36;              // [dn][16]   (RENDER, column pass)
    float* Ob = OVL ? sc + dn * 16 : sc + dn * 52;    // [dn][16]   dO
    float* St = OVL ? sc + dn * 32 : sc + dn * 68;    // [dn][12]   shift[4] (log2 domain), rs[4] / sum[4] (dO is stored divided by the row sum as well); a workgroup with an `exact` row: shift, rs, 1/sum
    (void)Qb; (void)St; (void)Ob;
    const size_t pt = (size_t)ray * dn + i;
    // the ray in the caller's order (RENDER with sorted descriptors): index of every user-visible array
    int oray = ray;
    if (RENDER && a.ray_perm) oray = (ray / a.rays_per_scene) * a.rays_per_scene + a.ray_perm[ray];
    const size_t opt = (size_t)oray * dn + i;
    (void)opt;
    constexpr int REC = RENDER ? REC_RAY : REC_VOL;
    const float* rec = a.rec + pt * REC;
    const float* W = a.wpk;
    constexpr int UB = GNR_RAY_UNROLL;                    // (no measurable effect: the kernel is VALU-throughput bound)
    constexpr int UA = RENDER ? UB : 4;                   // key-loop unroll (register pressure vs LDS latency)

    float g16[16], t[16];
    {
        const f4* r4 = reinterpret_cast<const f4*>(rec);
        const f4* pe = reinterpret_cast<const f4*>(W + pk::R_PE + i * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4 x = r4[q], e = pe[q];
            g16[4 * q] = x.x; g16[4 * q + 1] = x.y; g16[4 * q + 2] = x.z; g16[4 * q + 3] = x.w;
            t[4 * q] = x.x + e.x; t[4 * q + 1] = x.y + e.y; t[4 * q + 2] = x.z + e.z; t[4 * q + 3] = x.w + e.w;
        }
    }
    const float nvalid = rec[RENDER ? 80 : 16];
    const bool rowok = nvalid > 1.f;                  // ibrnet.py:493 query-row mask

    // ---- q, k, v projections (no bias)      ibrnet.py:81-86
    float q[16], kk[16], vv[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) {
        float sq = 0.f, sk = 0.f, sv = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            sq = fmaf(W[pk::R_WQ + f * 16 + c], t[c], sq);
            sk = fmaf(W[pk::R_WK + f * 16 + c], t[c], sk);
            sv = fmaf(W[pk::R_WV + f * 16 + c], t[c], sv);
        }
        q[f] = sq; kk[f] = sk; vv[f] = sv;
    }
    // the query as the three sweeps use it (see head_logits): (log2e / 2) q, zero on masked rows
    float qs[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) qs[f] = rowok ? q[f] * (0.5f * kLog2e) : 0.f;
    if (act) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f4 k4 = {kk[4 * c], kk[4 * c + 1], kk[4 * c + 2], kk[4 * c + 3]};
            const f4 v4 = {vv[4 * c], vv[4 * c + 1], vv[4 * c + 2], vv[4 * c + 3]};
            reinterpret_cast<f4*>(Kb + i * 16)[c] = k4;
            reinterpret_cast<f4*>(Vb + i * 16)[c] = v4;
            KN[i * KNS + c] = k4.x * k4.x + k4.y * k4.y + k4.z * k4.z + k4.w * k4.w;
            if constexpr (RENDER && !OVL) {
                const f4 q4 = {qs[4 * c], qs[4 * c + 1], qs[4 * c + 2], qs[4 * c + 3]};
                reinterpret_cast<f4*>(Qb + i * 16)[c] = q4;
            }
        }
    }
    __syncthreads();

    // ---- attention, lane = query row (ibrnet.py:15-27): softmax over dn keys, 4 heads per key visit.
    // Softmax shift without a pass over the logits: s_ij = q_i.k_j/2 <= |q_i| max_j|k_j| / 2 (Cauchy-Schwarz), so that
    // bound replaces the row maximum (softmax is shift-invariant; exp(s - shift) <= 1 cannot overflow).  If the bound is
    // so loose that a whole row underflows (shift - max s > ~87) the lane falls back to the true maximum.
    float amax[4];
    {
        f4 kmax = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int j = 0; j < dn; ++j) {
            const f4 n = *reinterpret_cast<const f4*>(KN + j * KNS);
            kmax.x = fmaxf(kmax.x, n.x); kmax.y = fmaxf(kmax.y, n.y); kmax.z = fmaxf(kmax.z, n.z); kmax.w = fmaxf(kmax.w, n.w);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float qn = q[4 * h] * q[4 * h] + q[4 * h + 1] * q[4 * h + 1] + q[4 * h + 2] * q[4 * h + 2] + q[4 * h + 3] * q[4 * h + 3];
            amax[h] = rowok ? (0.5f * kLog2e) * sqrtf(qn * kmax[h]) : 0.f;          // log2 domain, like the logits
        }
    }
    // accumulators of the sweeps are float4 per head: the compiler issues them as v_pk_fma_f32 (two lanes of fp32 per
    // instruction; no MFMA in this kernel, so the packed form is a gain here)
    float o[16], ainv[4];
    bool exact = false;
    {
        float l[4];
        f4 o4[4];
        auto sweep = [&](auto seeded) {
            constexpr bool SD = decltype(seeded)::value;
#pragma unroll
            for (int h = 0; h < 4; ++h) { l[h] = 0.f; o4[h] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll UA
            for (int j = 0; j < dn; ++j) {
                float sj[4];
                head_logits_shifted<SD>(qs, Kb + j * 16, amax, sj);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const f4 vj = reinterpret_cast<const f4*>(Vb + j * 16)[h];
                    const float pj = __builtin_amdgcn_exp2f(sj[h]);
                    l[h] += pj;
                    o4[h] += pj * vj;
                }
            }
        };
        sweep(std::true_type{});
        // Rare: a second sweep on the exact row maxima (`exact`: plain chain, see head_logits_shifted) -- when the bound was so loose
        // that a head's row sum fell below 2^-40 (this also bounds 1 / l <= 1.1e12, which the VJP below multiplies into dO and rs before
        // the column pass: no overflow for |dO| < 3e26), or when the logits are large (bound > 2^12: one-hot rows): the seeded and the
        // plain chain differ by the rounding of the partial sums, ~2^-22 of the logit, and a workgroup that holds an `exact` row sends
        // its other rows through the plain column pass (below) -- under the threshold their P there is within 2^-10 of the forward's,
        // the resolution fp32 gives such a logit anyway; far above it the rounding could lift an exponent over the bound by whole units.
        if (!(fminf(fminf(l[0], l[1]), fminf(l[2], l[3])) > 9.0e-13f) || fmaxf(fmaxf(amax[0], amax[1]), fmaxf(amax[2], amax[3])) > 4096.f) {
            exact = true;
#pragma unroll
            for (int h = 0; h < 4; ++h) amax[h] = -3.0e38f;
            for (int j = 0; j < dn; ++j) {
                float sj[4];
                head_logits_s(qs, Kb + j * 16, sj);
#pragma unroll
                for (int h = 0; h < 4; ++h) amax[h] = fmaxf(amax[h], sj[h]);
            }
            sweep(std::false_type{});
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            ainv[h] = 1.f / l[h];
            o[4 * h] = o4[h].x * ainv[h]; o[4 * h + 1] = o4[h].y * ainv[h]; o[4 * h + 2] = o4[h].z * ainv[h]; o[4 * h + 3] = o4[h].w * ainv[h];
        }
    }
    // ---- fc + residual, LayerNorm(eps 1e-6), out_geometry_fc (two linears), clip   ibrnet.py:97-100,494-495
    float y[16], xh[16], nrm[16];
    float mean = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float s = t[c];
#pragma unroll
        for (int f = 0; f < 16; ++f) s = fmaf(W[pk::R_WFC + c * 16 + f], o[f], s);
        y[c] = s; mean += s;
    }
    mean *= (1.f / 16.f);
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { const float d = y[c] - mean; var += d * d; }
    var *= (1.f / 16.f);
    const float rstd = 1.f / sqrtf(var + 1e-6f);
#pragma unroll
    for (int c = 0; c < 16; ++c) { xh[c] = (y[c] - mean) * rstd; nrm[c] = xh[c] * W[pk::R_LNW + c] + W[pk::R_LNB + c]; }
    float sraw = W[pk::R_OUTB];                       // out_geometry_fc.1(out_geometry_fc.0(n)), folded on the host
#pragma unroll
    for (int c = 0; c < 16; ++c) sraw = fmaf(W[pk::R_OUTVJP + c], nrm[c], sraw);
    float sdf = fminf(fmaxf(sraw, -1.f), 1.f);
    if (nvalid < 1.f) sdf = 1.f;

    if constexpr (!RENDER) {
        // renderer.py:197-198: column (x,y), sample i is voxel z = R-1-i
        if (act) {
            const size_t o_idx = (size_t)ray * dn + (dn - 1 - i);
            a.volume[o_idx] = sdf;
            if (a.vmask_out && a.vmask_pts) a.vmask_out[o_idx] = a.vmask_pts[pt];
        }
        return;
    } else {
        // ================= in-forward VJP of sum(sdf) w.r.t. the sample points (ibrnet.py:497-504)
        const float ds = (sraw >= -1.f && sraw <= 1.f && nvalid >= 1.f && act) ? 1.f : 0.f;
        float dy[16];
        {
            float m1 = 0.f, m2 = 0.f, dxh[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                dxh[c] = ds * W[pk::R_OUTVJP + c] * W[pk::R_LNW + c];
                m1 += dxh[c]; m2 += dxh[c] * xh[c];
            }
            m1 *= (1.f / 16.f); m2 *= (1.f / 16.f);
#pragma unroll
            for (int c = 0; c < 16; ++c) dy[c] = rstd * (dxh[c] - m1 - xh[c] * m2);
        }
        float dO[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) s = fmaf(dy[c], W[pk::R_WFCT + f * 16 + c], s);
            dO[f] = s;
        }
        // row pass: dQ = sum_j P_ij (dA_ij - rs_i) k_j / 2 with dA_ij = dO_i . v_j and rs_i = sum_j P_ij dA_ij = dO_i . o_i (the attention's
        // own output: no sweep needed for it, and none for sum_j P k_j)
        float dQ[16], rsv[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) rsv[h] = dO[4 * h] * o[4 * h] + dO[4 * h + 1] * o[4 * h + 1] + dO[4 * h + 2] * o[4 * h + 2] + dO[4 * h + 3] * o[4 * h + 3];
        {
            f4 A1[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) A1[h] = (f4){0.f, 0.f, 0.f, 0.f};
            // FAST: shift / rs as the seeds of the FMA chains, P un-normalised (1 / l_i multiplies the finished sum).  !FAST (the lane's row
            // is `exact`): the plain chains and the normalised P of the forward's second sweep -- for a row that is (nearly) one-hot,
            // dA_ij - rs_i then cancels EXACTLY on the key that holds the weight, whatever the magnitudes (rs_i = dO_i . o_i is the same
            // chain over the same numbers); a seeded chain would leave the rounding of a huge dA times a huge k there.
            auto row_pass = [&](auto fast) {
                constexpr bool FAST = decltype(fast)::value;
#pragma unroll UB
                for (int j = 0; j < dn; ++j) {
                    float sj[4];
                    head_logits_shifted<FAST>(qs, Kb + j * 16, amax, sj);
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const f4 kj = reinterpret_cast<const f4*>(Kb + j * 16)[h];
                        const f4 vj = reinterpret_cast<const f4*>(Vb + j * 16)[h];
                        float pj = __builtin_amdgcn_exp2f(sj[h]);
                        if constexpr (!FAST) pj *= ainv[h];
                        const float dAr = dot4_minus<FAST>(dO[4 * h], dO[4 * h + 1], dO[4 * h + 2], dO[4 * h + 3], vj, rsv[h]);      // dA_ij - rs_i
                        A1[h] += (pj * dAr) * kj;
                    }
                }
            };
            if (exact) row_pass(std::false_type{}); else row_pass(std::true_type{});
            const float sc2 = rowok ? 0.5f : 0.f;       // masked query rows: d logits = 0
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f4 d4 = A1[h] * (exact ? sc2 : sc2 * ainv[h]);
                dQ[4 * h] = d4.x; dQ[4 * h + 1] = d4.y; dQ[4 * h + 2] = d4.z; dQ[4 * h + 3] = d4.w;
            }
        }
        if constexpr (OVL) {                              // the lane's own key / value row, before the arrays are re-used
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f4 k4 = reinterpret_cast<const f4*>(Kb + i * 16)[c], v4 = reinterpret_cast<const f4*>(Vb + i * 16)[c];
                kk[4 * c] = k4.x; kk[4 * c + 1] = k4.y; kk[4 * c + 2] = k4.z; kk[4 * c + 3] = k4.w;
                vv[4 * c] = v4.x; vv[4 * c + 1] = v4.y; vv[4 * c + 2] = v4.z; vv[4 * c + 3] = v4.w;
            }
        }
        // The barrier carries a vote: ONE `exact` query row among the workgroup's rays sends every lane of the workgroup through the
        // plain column pass (lanes that shadow a ray beyond the launch ran on stale LDS: they do not vote).
        __shared__ int exact_wave[4];
        {
#ifdef GNR_DBG_SHADOW_VOTE      // (the bug as found: tests/test_range_guard.py::test_stale_lds_does_not_reach_the_outputs must FAIL on this build)
            const bool w_exact = __ballot(exact) != 0ull;
#else
            const bool w_exact = __ballot(exact && act) != 0ull;
#endif
            if ((threadIdx.x & 63) == 0) exact_wave[threadIdx.x >> 6] = w_exact ? 1 : 0;
        }
        __syncthreads();                                  // every lane is past the row pass (and the key-norm sweep): K / V / KN are free
        const bool any_exact = (exact_wave[0] | exact_wave[1] | exact_wave[2] | exact_wave[3]) != 0;
        if (act) {
            // fast column pass: dO_i / l_i and rs_i / l_i are stored, so that it needs no 1 / l_i per (query, head); plain: dO, 1 / l, rs
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sc = any_exact ? 1.f : ainv[c];
                const f4 d4 = {dO[4 * c] * sc, dO[4 * c + 1] * sc, dO[4 * c + 2] * sc, dO[4 * c + 3] * sc};
                reinterpret_cast<f4*>(Ob + i * 16)[c] = d4;
                if constexpr (OVL) {
                    const f4 q4 = {qs[4 * c], qs[4 * c + 1], qs[4 * c + 2], qs[4 * c + 3]};
                    reinterpret_cast<f4*>(Qb + i * 16)[c] = q4;
                }
            }
            const f4 m4 = {amax[0], amax[1], amax[2], amax[3]}, i4 = {ainv[0], ainv[1], ainv[2], ainv[3]};
            const f4 r4 = any_exact ? (f4){rsv[0], rsv[1], rsv[2], rsv[3]} : (f4){rsv[0] * ainv[0], rsv[1] * ainv[1], rsv[2] * ainv[2], rsv[3] * ainv[3]};
            reinterpret_cast<f4*>(St + i * 12)[0] = m4;
            reinterpret_cast<f4*>(St + i * 12)[1] = r4;
            if (any_exact) reinterpret_cast<f4*>(St + i * 12)[2] = i4;
        }
        __syncthreads();
        // column pass (lane = key j = i): dK_j = sum_i dL_ij q_i / 2 ; dV_j = sum_i P_ij dO_i
        // Qb holds the pre-scaled queries (log2e / 2) q_i (zero rows where masked): the logit needs no further factor, and
        // sum_i dL_ij q_i / 2 = ln2 sum_i [P_ij (dA_ij - rs_i)] qs_i ; masked rows contribute nothing to dK (their qs is zero)
        float dK[16], dV[16];
        {
            f4 dK4[4], dV4[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) { dK4[h] = (f4){0.f, 0.f, 0.f, 0.f}; dV4[h] = (f4){0.f, 0.f, 0.f, 0.f}; }
            auto col_pass = [&](auto fast) {
                constexpr bool FAST = decltype(fast)::value;
#pragma unroll UB
                for (int qi = 0; qi < dn; ++qi) {
                    const f4 mx4 = reinterpret_cast<const f4*>(St + qi * 12)[0];
                    const f4 rs4 = reinterpret_cast<const f4*>(St + qi * 12)[1];
                    f4 il4 = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (!FAST) il4 = reinterpret_cast<const f4*>(St + qi * 12)[2];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const f4 qv = reinterpret_cast<const f4*>(Qb + qi * 16)[h];
                        const f4 dov = reinterpret_cast<const f4*>(Ob + qi * 16)[h];
                        const float s = dot4_minus<FAST>(qv.x, qv.y, qv.z, qv.w, (f4){kk[4 * h], kk[4 * h + 1], kk[4 * h + 2], kk[4 * h + 3]}, mx4[h]);   // logit - shift of query qi
                        float pj = __builtin_amdgcn_exp2f(s);                           // FAST: P_ij l_i (dO and rs arrive divided by l_i)
                        if constexpr (!FAST) pj *= il4[h];
                        const float dAr = dot4_minus<FAST>(dov.x, dov.y, dov.z, dov.w, (f4){vv[4 * h], vv[4 * h + 1], vv[4 * h + 2], vv[4 * h + 3]}, rs4[h]);   // dA - rs of query qi
                        dK4[h] += (pj * dAr) * qv;
                        dV4[h] += pj * dov;
                    }
                }
            };
            if (any_exact) col_pass(std::false_type{}); else col_pass(std::true_type{});
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                dK[4 * h] = dK4[h].x * kLn2; dK[4 * h + 1] = dK4[h].y * kLn2; dK[4 * h + 2] = dK4[h].z * kLn2; dK[4 * h + 3] = dK4[h].w * kLn2;
                dV[4 * h] = dV4[h].x; dV[4 * h + 1] = dV4[h].y; dV[4 * h + 2] = dV4[h].z; dV[4 * h + 3] = dV4[h].w;
            }
        }
        // dT = dy (residual) + Wq^T dQ + Wk^T dK + Wv^T dV ; dc = dT * ELU'(c) with ELU' = g>0 ? 1 : g+1 ;
        // geometry_fc backward (two layers, 16 -> 64 -> the 21 embed columns; only the embed columns matter)
        float de[21];
        // the same tail as fp32 FMAs, one lane per sample (MM = false: the range fallback of the matrix-core form, launched as its twin)
        auto tail_fp32 = [&]() {
            float g16b[16];                                   // geometry_fc's output again (re-read: 16 registers less across the sweeps)
            {
                const f4* r4 = reinterpret_cast<const f4*>(rec);
    #pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f4 x = __builtin_nontemporal_load(r4 + q4);
                    g16b[4 * q4] = x.x; g16b[4 * q4 + 1] = x.y; g16b[4 * q4 + 2] = x.z; g16b[4 * q4 + 3] = x.w;
                }
            }
            float dc[16];
    #pragma unroll
            for (int c = 0; c < 16; ++c) {
                float s = dy[c];
    #pragma unroll
                for (int f = 0; f < 16; ++f) {
                    s = fmaf(W[pk::R_WQT + c * 16 + f], dQ[f], s);
                    s = fmaf(W[pk::R_WKT + c * 16 + f], dK[f], s);
                    s = fmaf(W[pk::R_WVT + c * 16 + f], dV[f], s);
                }
                dc[c] = s * (g16b[c] > 0.f ? 1.f : g16b[c] + 1.f);
            }
    #pragma unroll
            for (int e = 0; e < 21; ++e) de[e] = 0.f;
    #pragma unroll 2
            for (int h4 = 0; h4 < 16; ++h4) {
                const f4 u4 = reinterpret_cast<const f4*>(rec + 16)[h4];
    #pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    const int h = 4 * h4 + hh;
                    float du = 0.f;
    #pragma unroll
                    for (int c = 0; c < 16; ++c) du = fmaf(W[pk::R_GEO2WT + h * 16 + c], dc[c], du);
                    const float uh = u4[hh];
                    const float da = du * (uh > 0.f ? 1.f : uh + 1.f);
    #pragma unroll
                    for (int e = 0; e < 21; ++e) de[e] = fmaf(W[pk::R_GEO1E + h * 24 + e], da, de[e]);
                }
            }
        };
        if constexpr (MM) {
        // The three layers on the f16 matrix cores, 16 samples as the columns of an MFMA (fp16-pair fragments: the RM section of the blob,
        // read from global memory / L2).  [dQ | dK | dV | dy] crosses from "lane = sample" to the B-operand layout through a 32-float
        // row per sample in the attention scratch (dead once every lane has left the column pass), in two halves; from there on the
        // chain stays in registers: the D layout of each layer is the B layout the next one was packed for (dT -> ELU' -> dc -> du ->
        // ELU' -> da -> de), and de comes back through the same row.
        {
            static_assert(OVL, "the transfer rows re-use the overlaid attention scratch");
            const int lane = threadIdx.x & 63, mr = lane & 15, mg = lane >> 4;
            const unsigned magic = (65536u + (unsigned)S - 1u) / (unsigned)S;      // T / S for T < 256 as (T * magic) >> 16
            constexpr int GS = 36;                            // row stride (floats): 16-byte aligned, rows of a group on different banks
            float* Gr = sc + i * GS;
            // column mr of group q4 is the sample of thread T = 64 wave + 16 q4 + mr
            auto col = [&](int q4, int& rayT, int& iT, float*& row) -> bool {
                const int T = ((int)threadIdx.x & ~63) + 16 * q4 + mr;
                int rlT = (int)(((unsigned)T * magic) >> 16);
                const int sl = T - rlT * S;
                const bool okT = rlT < rpb && blockIdx.x * rpb + rlT < a.nrays && sl < dn;
                rlT = min(rlT, rpb - 1);
                iT = min(sl, dn - 1);
                rayT = min((int)(blockIdx.x * rpb) + rlT, a.nrays - 1);
                row = sm + (size_t)rlT * a.ray_stride + iT * GS;
                return okT;
            };
            auto row_block = [&](const float* row) -> P8 {    // inputs 8 mg .. 8 mg + 7 of the row's 32
                const f4 a0 = reinterpret_cast<const f4*>(row + 8 * mg)[0], a1 = reinterpret_cast<const f4*>(row + 8 * mg)[1];
                const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                return split8<0>(x);
            };
            auto put_row = [&](const float (&lo)[16], const float (&hi)[16]) {
                if (act) {                                    // (lanes out of range shadow another lane's row: no stores)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        reinterpret_cast<f4*>(Gr)[c] = (f4){lo[4 * c], lo[4 * c + 1], lo[4 * c + 2], lo[4 * c + 3]};
                        reinterpret_cast<f4*>(Gr + 16)[c] = (f4){hi[4 * c], hi[4 * c + 1], hi[4 * c + 2], hi[4 * c + 3]};
                    }
                }
            };
            __syncthreads();                                  // Q / dO / statistics are free
            f4 dT[4];
            put_row(dQ, dK);
            wave_sync();
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                int rayT, iT; float* row;
                (void)col(q4, rayT, iT, row);
                const P8 xp = row_block(row);
                f4 acc[1] = {(f4){0.f, 0.f, 0.f, 0.f}};
                mm16<1, 1, false>(W + pk::RM_DC, lane, &xp, acc);
                dT[q4] = acc[0];
            }
            wave_sync();
            put_row(dV, dy);
            wave_sync();
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                int rayT, iT; float* row;
                const bool okT = col(q4, rayT, iT, row);
                const float* recT = a.rec + ((size_t)rayT * dn + iT) * REC + 4 * mg;
                {
                    const P8 xp = row_block(row);
                    f4 acc[1] = {dT[q4]};
                    mm16<1, 1, false>(W + pk::RM_DC + pk::k32_floats(1), lane, &xp, acc);     // dT: channel 4 mg + t of column mr
                    dT[q4] = acc[0];
                }
                const f4 gq = *reinterpret_cast<const f4*>(recT);                             // geometry_fc's output of that sample
                const float dc4[4] = {dT[q4].x * (gq.x > 0.f ? 1.f : gq.x + 1.f), dT[q4].y * (gq.y > 0.f ? 1.f : gq.y + 1.f),
                                      dT[q4].z * (gq.z > 0.f ? 1.f : gq.z + 1.f), dT[q4].w * (gq.w > 0.f ? 1.f : gq.w + 1.f)};
                const P8 xp = split8z<0, 4>(dc4);
                f4 du[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) du[nb] = (f4){0.f, 0.f, 0.f, 0.f};
                mm16<1, 4, false>(W + pk::RM_GEOA, lane, &xp, du);              // du[nb][t]: hidden unit 16 nb + 4 mg + t of column mr
                float da[16];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const f4 u = *reinterpret_cast<const f4*>(recT + 16 + 16 * nb);           // geometry_fc's hidden layer of that sample
                    da[4 * nb] = du[nb].x * (u.x > 0.f ? 1.f : u.x + 1.f); da[4 * nb + 1] = du[nb].y * (u.y > 0.f ? 1.f : u.y + 1.f);
                    da[4 * nb + 2] = du[nb].z * (u.z > 0.f ? 1.f : u.z + 1.f); da[4 * nb + 3] = du[nb].w * (u.w > 0.f ? 1.f : u.w + 1.f);
                }
                const P8 yp[2] = {split8<0>(da), split8<8>(da)};
                f4 d2[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
                mm16<2, 2, false>(W + pk::RM_GEOB, lane, yp, d2);               // d2[nb][t]: embed column 16 nb + 4 mg + t
                if (okT) {
                    *reinterpret_cast<f4*>(row + 4 * mg) = d2[0];
                    if (mg < 2) *reinterpret_cast<f4*>(row + 16 + 4 * mg) = d2[1];
                }
            }
            wave_sync();
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const f4 v = reinterpret_cast<const f4*>(Gr)[c];
                de[4 * c] = v.x; de[4 * c + 1] = v.y; de[4 * c + 2] = v.z; de[4 * c + 3] = v.w;
            }
            de[20] = Gr[20];
            // Range guard.  The high half of a pair is an fp16: an operand of 65 520 or more, or a weight without a pair (the packer stores
            // +-inf for it), makes the outputs of its sample non-finite (w inf = inf, 0 inf = NaN).  The launch's watch word then makes the
            // fp32 instantiation behind this launch recompute it: that form has no such limit (the reference computes in fp32,
            // ibrnet.py:497-504).
            float chk = 0.f;
#pragma unroll
            for (int e = 0; e < 21; ++e) chk += de[e];
            if (__ballot(act && !(fabsf(chk) < 3.0e38f)) != 0ull && lane == 0 && a.range_word) atomicOr(a.range_word, 2u);
        }
        } else {
            tail_fp32();
        }
        const float* dsc = a.desc + pt * DESC_FLOATS;
        const f4 ds0 = reinterpret_cast<const f4*>(dsc)[0], ds1 = reinterpret_cast<const f4*>(dsc)[1];
        const float pp[3] = {ds0.x, ds0.y, ds0.z}, qd[3] = {ds0.w, ds1.x, ds1.y};
        float grad[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float pc = pp[c];
            float s1, c1, s2, c2, s4, c4;
            sincos_octaves(pc, s1, c1, s2, c2, s4, c4);
            grad[c] = de[c] + c1 * de[3 + c] - s1 * de[6 + c] + 2.f * c2 * de[9 + c] - 2.f * s2 * de[12 + c]
                      + 4.f * c4 * de[15 + c] - 4.f * s4 * de[18 + c];
        }
        // ================= NeuS alpha (aggregate_net.py:105-121), compositing (render_ops.py:72-80)
        const float z = a.depth[opt];
        const float znext = a.depth[opt + ((i + 1 < dn) ? 1 : 0)];
        const float dist = (i + 1 < dn) ? znext - z : 1e6f;
        const float inv_s = fminf(fmaxf(__expf(W[pk::R_VARIANCE] * 10.f), 1e-6f), 1e6f);
        const float tcos = -(qd[0] * grad[0] + qd[1] * grad[1] + qd[2] * grad[2]);
        const float icos = fminf(tcos, 0.f);                        // -relu(-cos)
        const float nxt = sdf + icos * dist * 0.5f, prv = sdf - icos * dist * 0.5f;
        const float pcdf = sigmoid1(prv * inv_s), ncdf = sigmoid1(nxt * inv_s);
        const float alpha = fminf(fmaxf((pcdf - ncdf + 1e-5f) / (pcdf + 1e-5f), 0.f), 1.f);
        // the attention scratch is dead from here on: the ray's LDS region is re-used for the per-ray reductions
        float* Tf = sc + rt::TF;
        float* Red = sc + rt::RED;
        float* Hp = sc + rt::HP;
        __syncthreads();
        if (act) Tf[i] = 1.f - alpha + 1e-10f;
        __syncthreads();
        // exclusive transmittance product in sample order (the order of the reference's cumprod)
        float T = 1.f;
#pragma unroll 8
        for (int j = 0; j < dn; ++j) { const float f = Tf[j]; T = (j < i) ? T * f : T; }
        const float hitp = act ? alpha * T : 0.f;
        const float gn = sqrtf(grad[0] * grad[0] + grad[1] * grad[1] + grad[2] * grad[2]) - 1.f;
        const float hp = __fadd_rn(hitp, 1e-5f);
        if (act) {
            const float cr = a.colors[opt * 3], cg = a.colors[opt * 3 + 1], cb = a.colors[opt * 3 + 2];
            Red[i] = hitp * cr; Red[rt::W + i] = hitp * cg; Red[2 * rt::W + i] = hitp * cb; Red[3 * rt::W + i] = hitp * z;
            Red[4 * rt::W + i] = gn * gn; Red[5 * rt::W + i] = (nvalid > (float)a.view_num) ? 1.f : 0.f;
            Hp[i] = hp;
            if (a.sdf) a.sdf[opt] = sdf;
            if (a.alpha) a.alpha[opt] = alpha;
            if (a.hit) a.hit[opt] = hitp;
            if (a.grad) { a.grad[opt * 3] = grad[0]; a.grad[opt * 3 + 1] = grad[1]; a.grad[opt * 3 + 2] = grad[2]; }
        }
        __syncthreads();
        if (rvalid) {
            for (int qn = slot; qn < 6; qn += S) {      // slot q sums quantity q (q, q+S, .. when a ray has < 6 slots)
                float s = 0.f;
                for (int j = 0; j < dn; ++j) s += Red[qn * rt::W + j];
                if (qn < 3) { if (a.pix) a.pix[(size_t)oray * 3 + qn] = s; }
                else if (qn == 3) { if (a.rdepth) a.rdepth[oray] = s; }
                else if (qn == 4) { if (a.gerr_part) a.gerr_part[oray] = s; }
                else { if (a.rmask) a.rmask[oray] = (s > (float)a.point_num) ? 1 : 0; }     // renderer.py:130-132
            }
        }
        // ================= inverse-CDF resampling for the fine pass (render_ops.py:172-229)
        if (a.fine_depth) {                             // uniform over the launch
            const int fdn = a.fdn;
            const int b = ray / a.rays_per_scene;
            const float near = __fdiv_rn(-1.f, a.que_dr[b * 2]), far = __fdiv_rn(-1.f, a.que_dr[b * 2 + 1]);
            const float span = __fsub_rn(far, near);
            float* Dn = sc + rt::DN; float* Pd = sc + rt::PD; float* Cd = sc + rt::CD; float* Ce = sc + rt::CE; float* Fd = sc + rt::FD;
            // torch.sum (render_ops.py:194) has no defined add order (vectorised cascade on the CPU): the sum is formed in
            // fp64 and rounded once, i.e. the correctly rounded value every fp32 order is within 1-2 ulp of.
            // torch.cumsum (render_ops.py:195) on the CPU accumulates fp32 inputs in double and rounds every prefix to
            // fp32 (at::acc_type<float, false>): reproduced exactly.
            double hsum_d = 0.0;
#pragma unroll 8
            for (int j = 0; j < dn; ++j) hsum_d += (double)Hp[j];
            const float hsum = (float)hsum_d;
            if (act) {
                Dn[i] = __fdiv_rn(__fsub_rn(__fdiv_rn(-1.f, z), near), span);
                Pd[i] = __fdiv_rn(hp, hsum);
            }
            __syncthreads();
            {
                double c = 0.0;
#pragma unroll 8
                for (int j = 0; j < dn; ++j) {
                    const double pj = (double)Pd[j];
                    if (j <= i) c += pj;
                }
                if (act) {
                    Cd[i + 1] = (float)c;
                    Ce[i + 1] = (i + 1 < dn) ? __fmul_rn(__fadd_rn(Dn[i + 1], Dn[i]), 0.5f) : Dn[dn - 1];
                    if (i == 0) { Cd[0] = 0.f; Ce[0] = Dn[0]; }
                }
            }
            __syncthreads();
            const bool fact = rvalid && slot < fdn;
            const int fs = min(slot, fdn - 1);
            const float interval = 1.f / (float)fdn;
            const float u = a.fine_u ? a.fine_u[(size_t)oray * fdn + fs]
                                     : __fadd_rn(__fmul_rn(0.5f, interval), __fmul_rn((float)fs, interval));
            int inds = 0;
#pragma unroll 8
            for (int j = 0; j <= dn; ++j) inds += (Cd[j] <= u) ? 1 : 0;          // searchsorted(right=True)
            const int below = max(inds - 1, 0), above = min(inds, dn);
            const float c0 = Cd[below], c1 = Cd[above], b0 = Ce[below], b1 = Ce[above];
            float den = __fsub_rn(c1, c0);
            if (den < 1e-5f) den = 1.f;
            const float tt = __fdiv_rn(__fsub_rn(u, c0), den);
            float fd = __fadd_rn(b0, __fmul_rn(tt, __fsub_rn(b1, b0)));
            fd = __fadd_rn(__fmul_rn(fd, span), near);
            fd = __fdiv_rn(-1.f, fd);
            if (fact) Fd[fs] = fd;
            __syncthreads();
            if (fact) {
                int rank = 0;                                                    // stable rank sort (renderer.py:148)
#pragma unroll 8
                for (int j = 0; j < fdn; ++j) { const float o2 = Fd[j]; rank += (o2 < fd || (o2 == fd && j < fs)) ? 1 : 0; }
                a.fine_depth[(size_t)oray * fdn + rank] = fd;
                if (a.fine_inds) a.fine_inds[(size_t)oray * fdn + fs] = inds;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_depth_mean: predict_mean_for_depth_loss (renderer.py:222-266): bilinear gather of ray_feats at
// pixel coordinates shared by all views of a scene + the decoder's mean branch (32->32->32->2,
// Softplus).  Same chained-MFMA tile as k_chain; one wavefront per 16 points of one view.
// ---------------------------------------------------------------------------------------
struct DepthMeanArgs {
    const float* wpk; const float* feat64; const float* coords;   // coords [B][pn][2] (x, y) in full-res pixels
    float* out;                                                    // [B][V][pn][2]
    int B, V, pn, H, W, fh, fw;
};

__global__ __launch_bounds__(256) void k_depth_mean(DepthMeanArgs a) {
    __shared__ __attribute__((aligned(16))) float w[2 * 1024 + 64 + 64 + 8];
    float* W1 = w; float* W2 = w + 1024; float* B1 = w + 2048; float* B2 = w + 2080; float* T3 = w + 2112; float* TB = w + 2176;
    for (int i = threadIdx.x; i < 1024; i += 256) { W1[i] = a.wpk[pk::DEC1 + i]; W2[i] = a.wpk[pk::DEC2 + i]; }
    if (threadIdx.x < 32) { B1[threadIdx.x] = a.wpk[pk::B_DEC1 + threadIdx.x]; B2[threadIdx.x] = a.wpk[pk::B_DEC2 + threadIdx.x]; }
    if (threadIdx.x < 64) T3[threadIdx.x] = a.wpk[pk::T_DEC3 + threadIdx.x];
    if (threadIdx.x < 2) TB[threadIdx.x] = a.wpk[pk::T_DEC3_B + threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int tpv = (a.pn + 15) >> 4;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.B * a.V * tpv) return;
    const int bv = tile / tpv, b = bv / a.V;
    const int n_raw = (tile - bv * tpv) * 16 + r;
    const bool ok = n_raw < a.pn;
    const int n = ok ? n_raw : a.pn - 1;
    const float x = a.coords[((size_t)b * a.pn + n) * 2], y = a.coords[((size_t)b * a.pn + n) * 2 + 1];
    const float fsx = (float)a.fw / (float)(a.W - 1), fsy = (float)a.fh / (float)(a.H - 1);
    const Taps t = make_taps(x, y, fsx, fsy, -0.5f, a.fh, a.fw);
    const float* fb = a.feat64 + (size_t)bv * a.fh * a.fw * 64 + 8 * g;
    float FR[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f4 a0 = reinterpret_cast<const f4*>(fb + (size_t)t.o00 * 64)[h], a1 = reinterpret_cast<const f4*>(fb + (size_t)t.o01 * 64)[h];
        const f4 a2 = reinterpret_cast<const f4*>(fb + (size_t)t.o10 * 64)[h], a3 = reinterpret_cast<const f4*>(fb + (size_t)t.o11 * 64)[h];
        const f4 fr = a0 * t.w00 + a1 * t.w01 + a2 * t.w10 + a3 * t.w11;
        FR[4 * h] = fr.x; FR[4 * h + 1] = fr.y; FR[4 * h + 2] = fr.z; FR[4 * h + 3] = fr.w;
    }
    f4 acc[2];
    float h1[8], h2[8];
    load_bias<2>(B1, g, acc);
    mm<8, 2>(W1, lane, FR, acc);
    elu_to<2, true>(acc, h1);
    load_bias<2>(B2, g, acc);
    mm<8, 2>(W2, lane, h1, acc);
    elu_to<2, true>(acc, h2);
    const float m0 = softplus1(gsum(dot8(T3, g, h2)) + TB[0]);
    const float m1 = softplus1(gsum(dot8(T3 + 32, g, h2)) + TB[1]);
    if (ok && g == 0) {
        float* o = a.out + ((size_t)bv * a.pn + n) * 2;
        o[0] = m0; o[1] = m1;
    }
}

// mean over one chunk of a scene's rays of the per-ray partial sums / (rays*dn)   (aggregate_net.py:139; the
// reference renders chunks of ray_batch_num rays, renderer.py:203-215).  grid (B, n_chunks); out [B][n_chunks]
__global__ void k_gerr_reduce(const float* __restrict__ part, float* __restrict__ out, int rn, int dn, int chunk) {
    __shared__ float red[256];
    const int b = blockIdx.x, c = blockIdx.y;
    const int r0 = c * chunk, r1 = min(rn, r0 + chunk);
    float s = 0.f;
    for (int i = r0 + threadIdx.x; i < r1; i += 256) s += part[(size_t)b * rn + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[(size_t)b * gridDim.y + c] = red[0] / ((float)(r1 - r0) * (float)dn);
}

// pixel_colors_gt: bilinear, zeros padding, align_corners=True  (renderer.py:125-127, ops.py:29-33)
// out2: nullable second copy of the result (the coarse and the fine pass of one call render the same rays: one launch serves both)
__global__ void k_pixel_gt(const float* __restrict__ imgs, const float* __restrict__ coords, float* __restrict__ out,
                           int rn, int H, int W, int B, float* __restrict__ out2 = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * rn) return;
    const int b = i / rn;
    const float x = coords[(size_t)i * 2], y = coords[(size_t)i * 2 + 1];
    const float xn = x / (float)(W - 1) * 2.f - 1.f, yn = y / (float)(H - 1) * 2.f - 1.f;
    const float px = (xn + 1.f) * 0.5f * (float)(W - 1), py = (yn + 1.f) * 0.5f * (float)(H - 1);
    const float x0 = floorf(px), y0 = floorf(py);
    const int x0i = (int)x0, y0i = (int)y0;
    const float wx1 = px - x0, wy1 = py - y0;
    for (int c = 0; c < 3; ++c) {
        const float* im = imgs + ((size_t)b * 3 + c) * H * W;
        auto tap = [&](int yy, int xx) { return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? im[yy * W + xx] : 0.f; };
        const float v = tap(y0i, x0i) * (1.f - wx1) * (1.f - wy1) + tap(y0i, x0i + 1) * wx1 * (1.f - wy1) +
                        tap(y0i + 1, x0i) * (1.f - wx1) * wy1 + tap(y0i + 1, x0i + 1) * wx1 * wy1;
        out[(size_t)i * 3 + c] = v;
        if (out2) out2[(size_t)i * 3 + c] = v;
    }
}

}  // namespace gnr

// (The round-5 phase-1 prototype of the 8-point x (view pair) tile -- measured 6 % slower at three wavefronts per SIMD -- lives with its
// A/B record under docs/experiments/r05_chain_p1/; its README says how to splice it back in for a measurement build.)
#define GNR_HD __host__ __device__
#include "gnr_pack_body.h"      // (k_pack_geo_dual: the packer's pair-block builder on the device)
#include "gnr_bwd.inc"
#ifndef GNR_DEV_NO_CAPI         // tools/isa_one.sh: the kernels alone + one explicit instantiation (register / spill checks in seconds)
#include "gnr_capi.inc"
#endif
