// Packed-weight layout shared by the host packer (gnr_pack.cpp) and the gfx950 kernels.
//
// Execution model of the per-(point, view) chain (k_chain in gnr_kernels.hip):
//   one wavefront owns a TILE of 16 points and all V views of those points.
//   lane l = (r = l & 15 : point within the tile, g = l >> 4 : feature group 0..3).
//   A feature vector is held "B-operand style": float v[J]; on lane (r,g) v[j] is logical
//   feature phi(j,g) of point r.  A linear layer  y = W x  is a chain of
//   v_mfma_f32_16x16x4_f32 with  A = weight fragment (row i = output feature, k = g),
//   B = v[j]  ->  D[i][r]; lane (r,g) register t of block nb then holds output feature
//   psi(nb, 4g+t), i.e. the output is again in B-operand layout.  No cross-lane movement
//   between layers; only the k-order of each dot product is permuted (fp32 reassociation).
//   The wide layers execute the same scheme on v_mfma_f32_16x16x32_f16 with operands as fp16 pairs: see the C16 section
//   below -- 8 consecutive k-steps of the fp32 form are one K32 block, lane mapping and output layout unchanged.
//   "natural" layout: phi(j,g) = 16*(j/4) + 4*g + (j%4)   <->   psi(nb,i) = 16*nb + i.
//
// A-fragment storage of a layer with J k-steps and NB output blocks (floats):
//   NB = 2 : [J][64 lanes][2]      ds_read_b64  per k-step
//   NB = 4 : [J][64][4]            ds_read_b128 per k-step
//   NB = 3 : [J][64][4] (4th = 0)  ds_read_b128 per k-step
//   NB = 1 : [ceil(J/4)][64][4]    ds_read_b128 per 4 k-steps (component = j % 4)
#pragma once

namespace gnr {

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int frag_floats(int J, int NB) {
    return NB == 1 ? cdiv(J, 4) * 256 : (NB == 3 ? J * 256 : J * 64 * NB);
}

// ---- canonical (reference state-dict order) blob of one level: decoder + agg net --------
// ref: dist_decoder.py:64-88, aggregate_net.py:29-33, ibrnet.py:382-423, neus.py:9
namespace can {
constexpr int DEC_BRANCH = 32 * 32 + 32 + 32 * 32 + 32;       // .0 and .2 of one branch
constexpr int MEAN0_W = 0, MEAN0_B = MEAN0_W + 1024, MEAN2_W = MEAN0_B + 32, MEAN2_B = MEAN2_W + 1024;
constexpr int MEAN4_W = MEAN2_B + 32, MEAN4_B = MEAN4_W + 64;
constexpr int VAR0_W = MEAN4_B + 2, VAR0_B = VAR0_W + 1024, VAR2_W = VAR0_B + 32, VAR2_B = VAR2_W + 1024;
constexpr int VAR4_W = VAR2_B + 32, VAR4_B = VAR4_W + 64;
constexpr int AW0_W = VAR4_B + 2, AW0_B = AW0_W + 1024, AW2_W = AW0_B + 32, AW2_B = AW2_W + 1024;
constexpr int AW4_W = AW2_B + 32, AW4_B = AW4_W + 32;
constexpr int PE0_W = AW4_B + 1, PE0_B = PE0_W + 32 * 34, PE2_W = PE0_B + 32, PE2_B = PE2_W + 1024;
constexpr int RDF0_W = PE2_B + 32, RDF0_B = RDF0_W + 64, RDF2_W = RDF0_B + 16, RDF2_B = RDF2_W + 35 * 16;
constexpr int BASE0_W = RDF2_B + 35, BASE0_B = BASE0_W + 64 * 207, BASE2_W = BASE0_B + 64, BASE2_B = BASE2_W + 32 * 64;
constexpr int VIS0_W = BASE2_B + 32, VIS0_B = VIS0_W + 1024, VIS2_W = VIS0_B + 32, VIS2_B = VIS2_W + 33 * 32;
constexpr int VISB0_W = VIS2_B + 33, VISB0_B = VISB0_W + 1024, VISB2_W = VISB0_B + 32, VISB2_B = VISB2_W + 32;
constexpr int GEO0_W = VISB2_B + 1, GEO0_B = GEO0_W + 64 * 86, GEO2_W = GEO0_B + 64, GEO2_B = GEO2_W + 16 * 64;
constexpr int WQ = GEO2_B + 16, WK = WQ + 256, WV = WK + 256, WFC = WV + 256, LN_W = WFC + 256, LN_B = LN_W + 16;
constexpr int OUT0_W = LN_B + 16, OUT0_B = OUT0_W + 256, OUT1_W = OUT0_B + 16, OUT1_B = OUT1_W + 16;
constexpr int RGB0_W = OUT1_B + 1, RGB0_B = RGB0_W + 16 * 37, RGB2_W = RGB0_B + 16, RGB2_B = RGB2_W + 128;
constexpr int RGB4_W = RGB2_B + 8, RGB4_B = RGB4_W + 8;
constexpr int NR0_W = RGB4_B + 1, NR0_B = NR0_W + 256, NR2_W = NR0_B + 8, NR2_B = NR2_W + 8;
constexpr int VARIANCE = NR2_B + 1;
constexpr int TOTAL = VARIANCE + 1;
static_assert(TOTAL == 36958, "canonical size must equal the reference parameter count");
// optional fourth decoder branch (dist_decoder_cfg.use_vis, dist_decoder.py:89-97), BEHIND the level's blob: a gradient blob of a
// use_vis level has TOTAL_VIS floats (gnr_canonical_vis_floats() more), its weights travel separately (gnr_pack_vis_decoder*)
constexpr int VISD0_W = TOTAL, VISD0_B = VISD0_W + 1024, VISD2_W = VISD0_B + 32, VISD2_B = VISD2_W + 1024;
constexpr int VISD4_W = VISD2_B + 32, VISD4_B = VISD4_W + 32;
constexpr int TOTAL_VIS = VISD4_B + 1;
static_assert(TOTAL_VIS - TOTAL == 2145, "vis_decoder: 32x32 + 32 + 32x32 + 32 + 32 + 1");
}  // namespace can

// ---- packed blob of one level ------------------------------------------------------------
// CHAIN section: copied verbatim into LDS by k_chain.  All offsets in floats.
namespace pk {
// MFMA A-fragments                 J   NB
constexpr int DEC1 = 0;          //  8   2   x3 branches (mean, var, aw), 1024 floats each
constexpr int DEC2 = DEC1 + 3 * frag_floats(8, 2);        //  8   2   x3
constexpr int PE1 = DEC2 + 3 * frag_floats(8, 2);         //  9   2
// prob_embed.2 has no activation behind it and its output is only consumed linearly (neuray_fc.0, base_fc.0
// columns 175..206), so it is multiplied into those consumers on the host: 16 MFMAs per (tile, view) less.
constexpr int RDF1 = PE1 + frag_floats(9, 2);             //  1   1
constexpr int RDF2 = RDF1 + frag_floats(1, 1);            //  4   3
constexpr int NR1 = RDF2 + frag_floats(4, 3);             //  8   1
constexpr int BASE1 = NR1 + frag_floats(8, 1);            // 17   4
constexpr int BASE2 = BASE1 + frag_floats(17, 4);         // 16   2
constexpr int VIS1 = BASE2 + frag_floats(16, 2);          //  8   2
constexpr int VIS2 = VIS1 + frag_floats(8, 2);            //  8   2   (rows 0..31 of the 33)
constexpr int VISB1 = VIS2 + frag_floats(8, 2);           //  8   2
constexpr int RGB1 = VISB1 + frag_floats(8, 2);           // 10   1
constexpr int RGB2 = RGB1 + frag_floats(10, 1);           //  4   1
constexpr int HOIST = RGB2 + frag_floats(4, 1);           // 36   4
constexpr int GEO1 = HOIST + frag_floats(36, 4);          // 23   4
constexpr int GEO2 = GEO1 + frag_floats(23, 4);           // 16   1
// optional fourth decoder branch (dist_decoder_cfg.use_vis, dist_decoder.py:89-97): zero unless gnr_pack_vis_decoder filled it
constexpr int DECV1 = GEO2 + frag_floats(16, 1);          //  8   2   vis_decoder.0
constexpr int DECV2 = DECV1 + frag_floats(8, 2);          //  8   2   vis_decoder.2
constexpr int FRAG_END = DECV2 + frag_floats(8, 2);
// bias tables: [NB][4 groups][4 regs] floats per layer (lane reads float4 at nb*4+g)
constexpr int B_DEC1 = FRAG_END;            // 3 x 32
constexpr int B_DEC2 = B_DEC1 + 96;         // 3 x 32
constexpr int B_PE1 = B_DEC2 + 96;
constexpr int B_RDF1 = B_PE1 + 32;
constexpr int B_RDF2 = B_RDF1 + 16;         // 48 (3 blocks)
constexpr int B_NR1 = B_RDF2 + 48;
constexpr int B_HOIST = B_NR1 + 16;         // base_fc.0 bias, 64
constexpr int B_BASE2 = B_HOIST + 64;
constexpr int B_VIS1 = B_BASE2 + 32;
constexpr int B_VIS2 = B_VIS1 + 32;
constexpr int B_VISB1 = B_VIS2 + 32;
constexpr int B_RGB1 = B_VISB1 + 32;
constexpr int B_RGB2 = B_RGB1 + 16;
constexpr int B_GEO1 = B_RGB2 + 16;         // 64
constexpr int B_GEO2 = B_GEO1 + 64;         // 16
constexpr int B_DECV1 = B_GEO2 + 16;        // 32
constexpr int B_DECV2 = B_DECV1 + 32;       // 32
constexpr int BIAS_END = B_DECV2 + 32;
// VALU tables for the 1..2-row output layers: per lane-group weights [g][8] (or [g][4])
constexpr int T_DEC3 = BIAS_END;            // 5 outputs x [4 g][8 j] = 160 ; order mean0 mean1 var0 var1 aw
constexpr int T_DEC3_B = T_DEC3 + 160;      // 5 biases (+3 pad)
constexpr int T_NR2 = T_DEC3_B + 8;         // [4][4]
constexpr int T_VIS2R = T_NR2 + 16;         // row 32 of vis_fc.2: [4][8]
constexpr int T_VISB2 = T_VIS2R + 32;       // vis_fc2.2: [4][8]
constexpr int T_RGB3 = T_VISB2 + 32;        // rgb_fc.4: [4][4]
constexpr int T_SCAL = T_RGB3 + 16;         // scalars: [0]=nr2 bias [1]=vis2 row32 bias [2]=visb2 bias [3]=rgb3 bias
constexpr int T_DECV3 = T_SCAL + 8;         // vis_decoder.4: [4][8]
constexpr int T_VIS = T_DECV3 + 32;         // [0] = vis_decoder.4 bias, [1] = 1.0 when the branch is present (use_vis), else 0
constexpr int CHAIN_END = T_VIS + 8;
static_assert(CHAIN_END % 4 == 0, "CHAIN section must be float4 copyable");
static_assert(CHAIN_END * 4 <= 160 * 1024, "CHAIN section must fit the 160 KiB LDS");

constexpr int k32_floats(int NB) { return NB * 512; }     // one K32 pair block (C16 section below)
// RAY section (k_ray): plain row-major matrices, read with wave-uniform (scalar) loads
constexpr int R_WQ = CHAIN_END, R_WK = R_WQ + 256, R_WV = R_WK + 256, R_WFC = R_WV + 256;
constexpr int R_LNW = R_WFC + 256, R_LNB = R_LNW + 16;
constexpr int R_OUT0W = R_LNB + 16, R_OUT0B = R_OUT0W + 256, R_OUT1W = R_OUT0B + 16, R_OUT1B = R_OUT1W + 16;
constexpr int R_GEO2W = R_OUT1B + 4;        // [16][64]
constexpr int R_GEO1E = R_GEO2W + 1024;     // geometry_fc.0 weight columns 65..85: [64][21 -> 24]
constexpr int R_VARIANCE = R_GEO1E + 64 * 24;
constexpr int R_PE = R_VARIANCE + 4;        // sinusoid table [128 positions][16]
// transposed copies / products for the VJP so that every inner loop reads contiguous scalars
constexpr int R_WQT = R_PE + 128 * 16, R_WKT = R_WQT + 256, R_WVT = R_WKT + 256, R_WFCT = R_WVT + 256;   // [in][out]
constexpr int R_GEO2WT = R_WFCT + 256;      // geometry_fc.2 weight transposed: [64][16]
constexpr int R_OUTVJP = R_GEO2WT + 1024;   // w = out_geometry_fc.0^T @ out_geometry_fc.1 [16]: both d sdf / d LayerNorm output
                                            // and the folded forward (two linears without activation, ibrnet.py:410-412)
constexpr int R_OUTB = R_OUTVJP + 16;       // folded bias: out_fc.1 . out_fc.0.bias + out_fc.1.bias
// C16 section: the image k_chain stages into LDS.  The CHAIN section's slots in the same order with the same biases / tables,
// but the A fragments of every layer with >= 4 k-steps are stored as FP16 PAIRS for the f16 matrix cores
// (v_mfma_f32_16x16x32_f16 runs at 16x the rate of the fp32-input MFMA, which on gfx950 is the vector rate): a weight w is
// carried as  h = fp16(w), m = fp16((w - h) * 2^11)  (round to nearest: h + m 2^-11 equals w to 1 fp32 ulp, exactly for 3 of
// 4 values), an activation the same way (split in registers), and  w x  is the sum of the partial products, each exact in
// the fp32 accumulator.  Inside a layer's slot:
//     [K32 blocks: up to 8 k-steps each, missing ones zero][left-over k-steps as fp32 fragments, layout of frag_floats(J', NB)]
//     K32 block b, output block nb, part p (0 = h, 1 = m): 8 halfs per lane at 16-byte index ((b NB + nb) 2 + p) 64 + lane,
//     element i <-> the block's i-th k-step (lane group g holds input slot phi(k_i, g), as in the fp32 form).
// A full block takes exactly the room of its 8 fp32 k-steps (4 bytes per weight), so most slots keep their size; the
// two per-point layers whose 4- / 7-k-step tails are zero-padded into a block of their own grow (c16_off shifts what
// follows).  ray_dir_fc.2 (4 k-steps, per view) stays fp32: padded as well it costs more than it saves (measured).
// Which k-steps form the blocks of which layer: gnr_pack.cpp `c16_plan` and the call sites in k_chain.  The fp32 CHAIN
// section stays in the blob: k_depth_mean and the backward twins read their forward fragments from it.
// (k32_floats: defined above, in front of the RAY section)
constexpr int C16_GROW_HOIST = 5 * k32_floats(4) - frag_floats(36, 4);      // 36 k-steps -> 4 blocks + a padded one
constexpr int C16_GROW_GEO1 = 3 * k32_floats(4) - frag_floats(23, 4);       // 23 k-steps -> 2 blocks + a padded one
// offset inside the C16 image of what sits at offset `o` of the CHAIN section
constexpr int c16_off(int o) { return o + (o > HOIST ? C16_GROW_HOIST : 0) + (o > GEO1 ? C16_GROW_GEO1 : 0); }
// RM section (round 5): the tail of the in-forward VJP inside k_ray<true> (ibrnet.py:497-504: the transposed q / k / v projections +
// residual, then the two layers of geometry_fc's backward) as fp16-pair
// K32 blocks for the f16 matrix cores -- 16 samples are the 16 columns of an MFMA; read by k_ray straight from global memory (the
// kernel's LDS belongs to the attention).  Element e of lane group g of block b <-> input k = 32 b + 8 g + e (RM_GEOB: the D layout
// of RM_GEOA's output, hidden unit 16 (2b + e/4) + 4g + e%4); output block nb, row i <-> output 16 nb + i.
constexpr int RM_DC = R_OUTB + 4;                   // dT = [Wq^T Wk^T Wv^T I] [dQ; dK; dV; dy]    K = 64 (two blocks), NB = 1
constexpr int RM_GEOA = RM_DC + 2 * k32_floats(1);  // du = geometry_fc.2^T dc              K = 16: input 4 g + e in elements e < 4 of lane group g (RM_DC's D layout), NB = 4
constexpr int RM_GEOB = RM_GEOA + k32_floats(4);    // de = geometry_fc.0[:, 65:86]^T da    K = 64 (two blocks), NB = 2 (rows 21.. zero)
constexpr int RM_END = RM_GEOB + 2 * k32_floats(2);
constexpr int C16 = RM_END;
constexpr int C16_END = c16_off(CHAIN_END);
static_assert(C16_END % 4 == 0 && C16_END * 4 <= 160 * 1024, "the C16 image must fit the 160 KiB LDS");
constexpr int TOTAL = C16 + C16_END;
}  // namespace pk

// per-point descriptor (k_points_* -> k_chain): 8 floats
//   [0..2] world point, [3..5] query direction, [6] lo, [7] hi  (half-intervals in the
//   reference views' normalised inverse depth; ref: dist_decoder.py:34-49)
constexpr int DESC_FLOATS = 8;
// per-view parameter block (k_view_setup -> k_chain): 24 floats
//   [0..11] H = K*[R|t] row-major 3x4, [12..14] camera centre, [15] -1/near, [16] (-1/far)-(-1/near), [17] 1/[16]
constexpr int VIEWP_FLOATS = 24;
// per-point record (k_chain -> k_ray)
constexpr int REC_VOL = 20;     // g16[16], nvalid, pad
constexpr int REC_RAY = 84;     // g16[16], u[64], nvalid, pad
constexpr int MAX_DN = 64;      // samples per ray / column in the backward twins and in the resampler (coarse pass)
constexpr int MAX_DN_FWD = 128; // samples per ray of a forward render pass (fine_depth_use_all: dn + fdn, renderer.py:145-146)

// Backward blob (gnr_pack_weights_bwd): transposed weights as chained-MFMA A fragments, true scale (gradients are
// kept in the true domain; the forward's log2e-scaled activations are rescaled where they enter a product).
namespace pkb {
constexpr int DM_W2T = 0;                  // mean_decoder.2^T : 32 (natural) -> 32 (natural)
constexpr int DM_W1T = DM_W2T + 1024;      // mean_decoder.0^T : 32 (natural) -> 32 ray channels in gather layout (8g + j)
constexpr int GEO2T = DM_W1T + 1024;                         // geometry_fc.2^T : 16 (natural, J=4) -> 64 (natural)   4 x 4
constexpr int GEO1T_A = GEO2T + frag_floats(4, 4);           // geometry_fc.0^T : 64 (natural) -> Z slots 0..15       16 x 4
constexpr int GEO1T_B = GEO1T_A + frag_floats(16, 4);        //                                   -> Z slots 16..19    16 x 1
// second view loop (k_view2_bwd), one contiguous section copied into LDS
constexpr int V2_BEGIN = GEO1T_B + frag_floats(16, 1);
constexpr int PE2F = V2_BEGIN;                               // prob_embed.2 forward (unfolded), natural -> natural      8 x 2
constexpr int B_PE2 = PE2F + 1024;                           // its bias (32, bias-table layout)
constexpr int VISB1T = B_PE2 + 32;                           // vis_fc2.0^T                                              8 x 2
constexpr int VIS2T = VISB1T + 1024;                         // vis_fc.2[:32]^T                                          8 x 2
constexpr int VIS1T = VIS2T + 1024;                          // vis_fc.0^T                                               8 x 2
constexpr int BASE2T = VIS1T + 1024;                         // base_fc.2^T : 32 -> 64                                   8 x 4
constexpr int BASE1XT = BASE2T + frag_floats(8, 4);          // base_fc.0[:,140:175]^T : 64 -> x slots                  16 x 3
constexpr int BASE1ET = BASE1XT + frag_floats(16, 3);        // base_fc.0[:,175:207]^T : 64 -> 32                       16 x 2
constexpr int V2_END = BASE1ET + frag_floats(16, 2);
// hoisted base_fc.0 columns: base_fc.0[:, :140]^T, 64 (natural) -> the 36 statistic slots per lane group
constexpr int HOISTT_A = V2_END;                             // slots  0..15   16 x 4
constexpr int HOISTT_B = HOISTT_A + frag_floats(16, 4);      // slots 16..31   16 x 4
constexpr int HOISTT_C = HOISTT_B + frag_floats(16, 4);      // slots 32..35   16 x 1
// first view loop (k_view1_bwd), one contiguous section copied into LDS
constexpr int V1_BEGIN = HOISTT_C + frag_floats(16, 1);
constexpr int DEC2T = V1_BEGIN;                              // {mean,var,aw}_decoder.2^T                               3 x (8 x 2)
constexpr int DEC1T = DEC2T + 3072;                          // {mean,var,aw}_decoder.0^T, rows in the gather layout    3 x (8 x 2)
constexpr int V1_PE2F = DEC1T + 3072;                        // prob_embed.2 forward + bias (as PE2F / B_PE2)
constexpr int V1_B_PE2 = V1_PE2F + 1024;
constexpr int PE2T = V1_B_PE2 + 32;                          // prob_embed.2^T                                           8 x 2
constexpr int PE0T = PE2T + 1024;                            // prob_embed.0[:, :32]^T, rows in the gather layout        8 x 2
constexpr int T_PE0HV = PE0T + 1024;                         // prob_embed.0[:, 32], [:, 33] as [4][8] row tables
constexpr int NR0T = T_PE0HV + 64;                           // neuray_fc.0^T : 8 (padded to 16) -> 32                  4 x 2
constexpr int RDF2T = NR0T + frag_floats(4, 2);              // ray_dir_fc.2^T : 35 (x slots) -> 16                      9 x 1
constexpr int V1_END = RDF2T + frag_floats(9, 1);
// colour head rgb_fc (render path), one contiguous section
constexpr int RGB2T = V1_END;                                // rgb_fc.2^T : 8 (padded to 16) -> 16                      4 x 1
constexpr int RGB0HT = RGB2T + frag_floats(4, 1);            // rgb_fc.0[:, :32]^T : 16 -> 32                            4 x 2
constexpr int T_RGB0V = RGB0HT + frag_floats(4, 2);          // rgb_fc.0[:, 32] as a [4][4] row table
// fourth decoder branch (use_vis), zero unless gnr_pack_vis_decoder_bwd filled it; k_view1_bwd<true> copies it behind its image
constexpr int DECV2T = T_RGB0V + 16;                         // vis_decoder.2^T                                          8 x 2
constexpr int DECV1T = DECV2T + 1024;                        // vis_decoder.0^T, rows in the gather layout               8 x 2
constexpr int F32_END = DECV1T + 1024;
// fp16-pair images (layout of the C16 section: K32 blocks of (h, m) x NB x 64 lanes x 8 halfs) of the transposed fragments and of
// prob_embed.2's forward fragment: the backward twins k_view2_bwd / k_view1_bwd run their forward recompute (from the forward
// blob's C16 image) and their dX chains on the f16 matrix cores with fp32 operands as fp16 pairs, like k_chain (round 5).
// Each of the two sections is one contiguous run the kernel copies into LDS; the bias / table copies keep them contiguous.
constexpr int P2_BEGIN = F32_END;                            // second view loop (k_view2_bwd)
constexpr int P2_PE2F = P2_BEGIN;                            // prob_embed.2 forward                                  1 x 2
constexpr int P2_B_PE2 = P2_PE2F + pk::k32_floats(2);            // its bias (32)
constexpr int P2_VISB1T = P2_B_PE2 + 32;                     //                                                       1 x 2
constexpr int P2_VIS2T = P2_VISB1T + pk::k32_floats(2);
constexpr int P2_VIS1T = P2_VIS2T + pk::k32_floats(2);
constexpr int P2_BASE2T = P2_VIS1T + pk::k32_floats(2);          //                                                       1 x 4
constexpr int P2_BASE1XT = P2_BASE2T + pk::k32_floats(4);        //                                                       2 x 3
constexpr int P2_BASE1ET = P2_BASE1XT + 2 * pk::k32_floats(3);   //                                                       2 x 2
constexpr int P2_END = P2_BASE1ET + 2 * pk::k32_floats(2);
constexpr int P1_BEGIN = P2_END;                             // first view loop (k_view1_bwd)
constexpr int P1_DEC2T = P1_BEGIN;                           // 3 x (1 x 2)
constexpr int P1_DEC1T = P1_DEC2T + 3 * pk::k32_floats(2);
constexpr int P1_PE2F = P1_DEC1T + 3 * pk::k32_floats(2);
constexpr int P1_B_PE2 = P1_PE2F + pk::k32_floats(2);
constexpr int P1_PE2T = P1_B_PE2 + 32;
constexpr int P1_PE0T = P1_PE2T + pk::k32_floats(2);
constexpr int P1_END = P1_PE0T + pk::k32_floats(2);
constexpr int P_DECV2T = P1_END;                             // use_vis branch (zero unless gnr_pack_vis_decoder_bwd filled it)
constexpr int P_DECV1T = P_DECV2T + pk::k32_floats(2);
constexpr int TOTAL = P_DECV1T + pk::k32_floats(2);
static_assert(P2_BEGIN % 4 == 0 && P1_BEGIN % 4 == 0 && P_DECV2T % 4 == 0, "pair blocks are read as 16-byte vectors");
}  // namespace pkb

}  // namespace gnr
