/* gnr.h -- C ABI of the MI355X-native GraspNeRF volumetric hot path (libgnr.so).
 *
 * Drop-in boundary for the reference's `NeuralRayRenderer.sample_volume` / `.render`
 * (ref: src/nr/network/renderer.py:164-199, :201-220, :110-138) for a BATCH of scenes.
 * Plain pointers and sizes only: every `const float*` below is a DEVICE pointer to
 * contiguous fp32 unless the name says `_host`.  `stream` is a hipStream_t passed as void*.
 * All entry points are stateless and return 0 (GNR_OK) or a negative error code; nothing
 * throws across the ABI.  The caller owns every buffer, including the workspace.  STATELESS is
 * meant literally: the library keeps no mutable switch -- what a call computes depends on its arguments alone (the per-call
 * switches are GnrScene.options / the `options` argument of the entry points without a scene), so two callers, or two streams,
 * of one process can hold different settings.  (The only process-wide state is the opt-in measurement tooling at the end of
 * this header, which adds event brackets around launches and changes nothing they compute.)
 *
 * Typical call sequence (what graspnerf_amd/renderer.py does through ctypes):
 *   gnr_pack_weights(canonical_host, packed_host)         once per load_state_dict, per level
 *   gnr_workspace_bytes(&scene, R, rn, dn)                once per shape
 *   gnr_prepare(&scene, ws, ws_bytes, stream)             per forward (feature-map repack)
 *   gnr_sample_volume_fwd(...) / gnr_render_rays_fwd(...) per forward
 */
#ifndef GNR_H
#define GNR_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNR_OK 0
#define GNR_ERR_ARG (-1)       /* null pointer / bad enum                                   */
#define GNR_ERR_SHAPE (-2)     /* unsupported V, dn, feature width ...                      */
#define GNR_ERR_HIP (-3)       /* a HIP call or kernel launch failed                        */
#define GNR_ERR_WORKSPACE (-4) /* workspace too small                                       */

/* Reference views of B scenes.  Replaces the `ref_imgs_info` dict
 * (ref: src/nr/utils/imgs_info.py:120, src/nr/main.py:228-242) after the 2D backbones
 * (renderer.py:275-279) have produced img_feats / ray_feats. */
typedef struct GnrScene {
    int B, V;                 /* scenes, views per scene (V in 2..8)                        */
    int H, W;                 /* image size                                                 */
    int fh, fw;               /* feature-map size (reference: H/4, W/4)                     */
    const float* imgs;        /* [B,V,3,H,W]  in [0,1]                                      */
    const float* img_feats;   /* [B,V,32,fh,fw]  image_encoder output   (renderer.py:275)   */
    const float* ray_feats;   /* [B,V,32,fh,fw]  vis_encoder output     (renderer.py:279)   */
    const float* poses;       /* [B,V,3,4] world->camera, OpenCV                            */
    const float* Ks;          /* [B,V,3,3]                                                  */
    const float* depth_range; /* [B,V,2] near, far                                          */
    int use_vis;              /* cfg dist_decoder_cfg.use_vis (both levels, dist_decoder.py:89-97): 1 -> the packed level
                                 blobs were completed with gnr_pack_vis_decoder (training: the backward blobs with
                                 gnr_pack_vis_decoder_bwd) and the chain runs the fourth decoder branch; the gradient
                                 blobs of the *_bwd entry points then have gnr_canonical_weights_floats() +
                                 gnr_canonical_vis_floats() floats; 0 -> configs/nrvgn_sdf.yaml                          */
    unsigned options;         /* per-call switches, an OR of GNR_OPT_* below; 0 = the product path.  Unknown bits: GNR_ERR_ARG   */
} GnrScene;

/* ---- per-call options (GnrScene.options; `options` of gnr_ray_tail_dual_bwd / gnr_geo_dual_bwd) -------------
 * Product switches:
 *   GNR_OPT_FEATURE_GRAD_FIXED  the feature-map gradients of this call (gnr_depth_mean_bwd, gnr_sample_volume_bwd,
 *                               gnr_render_chain_bwd) are BIT-REPRODUCIBLE: 64-bit fixed-point adds instead of float sums in
 *                               arrival order (see "backward twins" below).  Since round 6 the view kernels' binned scatter sums
 *                               the same parked rows as integers: the mode costs what the float mode costs.
 * Measurement / test switches (same results to rounding unless stated):
 *   GNR_OPT_FP32_CHAIN          every chain launch of the call -- forward, training forward and the backward's view kernels -- runs
 *                               its fp32-input-MFMA instantiation (what the range guard falls back to); the pair kernels are skipped
 *   GNR_OPT_VIEW1_ONE_WAVEFRONT / GNR_OPT_VIEW2_ONE_WAVEFRONT   the backward of the first / second view loop as one wavefront per tile
 *                               (k_view1_bwd / k_view2_bwd) instead of a compute wavefront and its partner (k_view*_bwd_pw)
 *   GNR_OPT_DIRECT_SCATTER      the feature-map gradient of k_view1_bwd_pw as direct float atomics instead of the binned scatter
 *                               (csrc/gnr_bwd_scatter.inc: rows parked in HBM, summed per feature-map pixel first)
 *   GNR_OPT_RAY_ORDER_MORTON    the inference render passes traverse a scene's rays in the Morton order of their pixels (internal
 *                               layout only; every array of the ABI keeps the caller's ray order, results are bit-identical)
 *   GNR_OPT_GEO_DUAL_FP32       gnr_geo_dual_fwd and the per-point half of gnr_geo_dual_bwd as fp32 FMAs, one lane per point, instead of the fp16-pair chain
 *   GNR_OPT_POISON_PARTIALS     the partial-gradient buffers are filled with NaN patterns before the kernels run (an entry no wavefront
 *                               stores would show in the reduced gradient)
 *   GNR_OPT_STATIC_TILES        the chain launches of the call give every wavefront its static round-robin share of the tiles instead of
 *                               handing the tiles out through the workspace's per-XCD counters (round 6; bit-identical outputs, the
 *                               launches' tails are longer: tests, measurements)
 *   GNR_OPT_SPLIT_LAUNCH        (experiment) gnr_sample_volume_fwd runs the two halves of the batch as two launch sequences on two streams (the
 *                               caller's and a library-owned one, joined before the call returns its stream), so that each half's
 *                               launches fill the other's tails; bit-identical outputs
 *   GNR_OPT_TEST_LOSE_PARTNER   (tests) the partner wavefronts of k_view1_bwd_pw / k_view2_bwd_pw return at once: the compute wavefronts'
 *                               bounded waits give up, the call's gradients are garbage and bit 4 of gnr_range_status says so */
#define GNR_OPT_FP32_CHAIN 0x001u
#define GNR_OPT_FEATURE_GRAD_FIXED 0x002u
#define GNR_OPT_VIEW1_ONE_WAVEFRONT 0x004u
#define GNR_OPT_VIEW2_ONE_WAVEFRONT 0x008u
#define GNR_OPT_RAY_ORDER_MORTON 0x010u
#define GNR_OPT_POISON_PARTIALS 0x020u
#define GNR_OPT_DIRECT_SCATTER 0x040u
#define GNR_OPT_GEO_DUAL_FP32 0x080u
#define GNR_OPT_TEST_LOSE_PARTNER 0x100u
#define GNR_OPT_STATIC_TILES 0x200u
#define GNR_OPT_SPLIT_LAUNCH 0x400u
#define GNR_OPT_ALL 0x7ffu

/* Query rays of B scenes.  Replaces the `que_imgs_info` dict (imgs_info.py:126-135). */
typedef struct GnrRays {
    int rn;                   /* rays per scene                                             */
    int dn;                   /* coarse samples per ray   (cfg depth_sample_num)            */
    int fdn;                  /* fine samples per ray     (cfg fine_depth_sample_num)       */
    int ray_mask_view_num;    /* cfg ray_mask_view_num  (renderer.py:37)                    */
    int ray_mask_point_num;   /* cfg ray_mask_point_num (renderer.py:38)                    */
    const float* coords;      /* [B,rn,2] pixel (x,y)                                       */
    const float* que_pose;    /* [B,3,4]                                                    */
    const float* que_K;       /* [B,3,3]                                                    */
    const float* que_depth_range; /* [B,2]                                                  */
    const float* que_imgs;    /* [B,3,H,W] or NULL (then pixel_colors_gt is not written)    */
    const float* fine_u;      /* [B,rn,fdn] in [0,1) or NULL.  is_train=True draws the inverse-CDF
                                 samples with torch.rand (render_ops.py:204-205): the caller draws them
                                 (same generator, same shape as the reference) and passes them here;
                                 NULL = eval mode, u_i = (i+.5)/fdn (render_ops.py:200-203)            */
    int ray_batch_num;        /* cfg ray_batch_num (renderer.py:203-215): the reference renders rays in
                                 chunks and returns one sdf_gradient_error per chunk; >0 -> the output
                                 is [B, ceil(rn/ray_batch_num)] chunk means, 0 -> one mean per scene   */
    int fine_depth_use_all;   /* cfg fine_depth_use_all (renderer.py:145-146): 1 -> the fine pass renders the dn coarse and
                                 the fdn resampled depths of a ray, merged and sorted (dn + fdn <= 128 samples; the fine level's
                                 positional table must have been built for that length, i.e. fine_agg_net_cfg.sample_num =
                                 dn + fdn in the reference); 0 -> the fdn resampled depths only                              */
} GnrRays;

/* Outputs of one render pass (coarse or fine).  Keys follow renderer.py:90-138; any
 * pointer may be NULL to skip that output.  `dn` below = GnrRays.dn (coarse) / .fdn (fine). */
typedef struct GnrRenderOut {
    float* depth;              /* [B,rn,dn] sample depths used by this pass                 */
    float* sdf_values;         /* [B,rn,dn]                                                 */
    float* alpha_values;       /* [B,rn,dn]                                                 */
    float* colors_nr;          /* [B,rn,dn,3]  (required: also an internal hand-off)        */
    float* hit_prob_nr;        /* [B,rn,dn]                                                 */
    float* pixel_colors_nr;    /* [B,rn,3]                                                  */
    float* pixel_colors_gt;    /* [B,rn,3]                                                  */
    float* render_depth;       /* [B,rn]                                                    */
    unsigned char* ray_mask;   /* [B,rn] 0/1                                                */
    float* sdf_gradient_error; /* [B, n_chunks] mean((|grad|-1)^2) per chunk of ray_batch_num rays
                                  (n_chunks = 1 when GnrRays.ray_batch_num == 0)            */
    float* sdf_gradient;       /* [B,rn,dn,3] optional (debug / eikonal loss)               */
    unsigned char* view_mask;  /* [B,rn,dn] optional: bit v = point inside view v's image   */
} GnrRenderOut;

/* ---- weights ---------------------------------------------------------------------------
 * canonical blob = the parameters of ONE level (decoder + aggregation net) flattened and
 * concatenated in reference state-dict order:
 *   <dec>.mean_decoder.{0,2,4}.{weight,bias}, <dec>.var_decoder..., <dec>.aw_decoder...,
 *   <agg>.prob_embed.{0,2}, <agg>.agg_impl.{ray_dir_fc,base_fc,vis_fc,vis_fc2,geometry_fc}.{0,2},
 *   <agg>.agg_impl.ray_attention.{w_qs,w_ks,w_vs,fc}.weight, .layer_norm.{weight,bias},
 *   <agg>.agg_impl.out_geometry_fc.{0,1}, .rgb_fc.{0,2,4}, .neuray_fc.{0,2},
 *   <agg>.deviation_network.variance
 * with <dec>,<agg> = dist_decoder,agg_net (coarse) or fine_dist_decoder,fine_agg_net (fine).
 * (ref: dist_decoder.py:64-88, aggregate_net.py:29-33, ibrnet.py:382-423, neus.py:9)      */
int gnr_canonical_weights_floats(void);   /* 36958 */
int gnr_packed_weights_floats(void);
/* Host-side packer: MFMA fragments of every layer (fp32), the tables of the per-ray kernel, and the image the chain kernel
 * stages into LDS, in which the wide layers' weights are stored as fp16 pairs w = h + m 2^-11 (1 fp32 ulp; csrc/gnr_layout.h
 * section C16).  GNR_ERR_ARG if a pointer is null.  An effective weight outside the fp16 range (|w| >= 65520) has no pair: the
 * blob is marked and every chain launch with it runs on the fp32-input MFMA (gnr_range_status bit 2). */
int gnr_pack_weights(const float* canonical_host, float* packed_host);
/* Optional: the fourth decoder branch of a level (cfg dist_decoder_cfg.use_vis: true; dist_decoder.py:89-97,103-104,133-134:
 * its sigmoid output multiplies both cdfs) into a blob gnr_pack_weights has filled.  vis_decoder_host = vis_decoder.{0.weight
 * [32][32], 0.bias [32], 2.weight [32][32], 2.bias [32], 4.weight [1][32], 4.bias [1]} in state-dict order (2145 floats).
 * Training (GnrScene.use_vis = 1 in gnr_sample_volume_fwd_train / _bwd, gnr_render_chain_fwd_train / _bwd): the same six tensors
 * go into the backward blob with gnr_pack_vis_decoder_bwd (after gnr_pack_weights_bwd), and the gradient blob `d_canonical`
 * carries their gradients BEHIND the level's 36958 floats, in the same state-dict order: gnr_canonical_vis_floats() = 2145
 * more.  (gnr_depth_mean_* reads mean_decoder only and takes no notice.) */
int gnr_pack_vis_decoder(const float* vis_decoder_host, float* packed_host);
int gnr_pack_vis_decoder_bwd(const float* vis_decoder_host, float* packed_bwd_host);
int gnr_canonical_vis_floats(void);       /* 2145 */
/* Device-side packers (csrc/gnr_pack_dev.hip): the same blobs from canonical blobs that are ALREADY ON THE DEVICE -- the training
 * loop's parameters move every optimiser step and, as in the reference (train/trainer.py:146-158: nn.Module parameters, optimiser
 * step on the device), never visit the host: no device-to-host copy, no host pack, no upload, no wait.  Stream-ordered kernels,
 * nothing synchronises.  The arithmetic is the host packer's own source (csrc/gnr_pack_body.h) and the results are bit-identical
 * to it.  The blobs are re-packed IN PLACE: `packed_dev` / `packed_bwd_dev` must once have been filled from a blob of
 * gnr_pack_weights / gnr_pack_weights_bwd (any weights) -- the constant parts (structural zeros, the position table) are kept.
 * gnr_pack_vis_decoder_device / _bwd_device: the use_vis branch, after the level's own re-pack, as on the host. */
int gnr_pack_weights_device(const float* canonical_dev, float* packed_dev, void* stream);
int gnr_pack_weights_bwd_device(const float* canonical_dev, float* packed_bwd_dev, void* stream);
int gnr_pack_vis_decoder_device(const float* vis_decoder_dev, float* packed_dev, void* stream);
int gnr_pack_vis_decoder_bwd_device(const float* vis_decoder_dev, float* packed_bwd_dev, void* stream);
/* float offset of a named section of the packed blob (see csrc/gnr_layout.h), -1 if unknown */
int gnr_layout_offset(const char* name);

/* ---- workspace -------------------------------------------------------------------------*/
size_t gnr_workspace_bytes(const GnrScene* scene, int volume_res, int rn, int dn_max);

/* Repack img_feats/ray_feats to channel-last and build the per-view projection blocks.
 * Must precede the forward calls that use the same workspace (stream ordered). */
int gnr_prepare(const GnrScene* scene, void* workspace, size_t workspace_bytes, void* stream);

/* Range guard.  The reference computes in fp32 everywhere (ibrnet.py:474-482; SURVEY.md 5 "mixed precision: none").  The chain
 * kernel multiplies on the f16 matrix cores with every fp32 operand carried as an fp16 pair, whose high half cannot hold a
 * magnitude of 65 520 or more.  Each forward (and training-forward) chain launch therefore watches its operands (feature maps in
 * gnr_prepare, activations and cross-view statistics in the kernel) and is followed by its fp32-input-MFMA twin, which returns
 * immediately unless the watch tripped and otherwise recomputes the launch: out-of-range scenes get the fp32 kernel's values,
 * in-range scenes pay ~5 us per chain launch, and no call synchronises with the host.
 * gnr_range_status reads the status words of the last gnr_prepare on this workspace (synchronises `stream`):
 *   bit 0: a feature-map value >= 6e4 in magnitude or not finite;  bit 1: an activation / statistic beyond the fp16 range (k_chain), or a
 *   non-finite value in the matrix-core tail of the per-ray kernel (an operand or a per-ray weight beyond the fp16 range: that launch and the
 *   later per-ray launches on the prepared scene are recomputed by their fp32 instantiation);
 *   bit 2: a weight beyond the fp16 range (gnr_pack_weights marks such a blob instead of refusing it);
 *   bit 3: (backward, GNR_OPT_FEATURE_GRAD_FIXED only) a feature-map gradient contribution was clamped to the fixed-point range;
 *   bit 4: (backward) GNR_STATUS_LOST_PARTNER -- a wavefront of k_view1_bwd_pw / k_view2_bwd_pw waited ~2^22 polls for its partner and
 *          gave up: THE GRADIENTS OF THAT BACKWARD CALL ARE INVALID.  The entry points still return GNR_OK (nothing synchronises with the
 *          host); a training loop must not apply such a step (graspnerf_amd/trainer.py skips the optimiser step on the device).
 * Bits 0 and 2 hold for every launch on the prepared scene (the pair kernel then returns at once and the twin computes the launch);
 * bit 1 is watched per launch slot (volume, coarse pass, fine pass, the training forwards), so a scene whose coarse pass tripped it
 * does not pay the recomputation on its volume or fine pass; the status word is the OR over the slots.
 * The backward twins follow the same guard (round 6): gnr_sample_volume_bwd / gnr_render_chain_bwd launch their fp16-pair view kernels
 * and, behind each, the fp32-input-MFMA instantiation; the former return at once when bits 0 / 2 of the scene or bit 1 of the pass's
 * training forward are set, the latter unless -- a range-tripped pass gets bitwise the gradients of GNR_OPT_FP32_CHAIN.
 * gnr_status_words_offset: byte offset of the 64 status words (unsigned) inside the workspace, for callers that test them on the device
 * (OR of the words & GNR_STATUS_LOST_PARTNER) instead of synchronising. */
#define GNR_STATUS_LOST_PARTNER 16u
int gnr_range_status(const GnrScene* scene, const void* workspace, size_t workspace_bytes, unsigned* flags_out, void* stream);
size_t gnr_status_words_offset(const GnrScene* scene);

/* sample_volume (renderer.py:164-199), volume_type [sdf]:
 *   sdf_out[b,x,y,z] for voxel centre bbox_min[b] + ((x,y,z)+.5)*(0.3/res).
 *   view_mask_out: optional [B, res^3] bytes in (x,y,z) order, bit v = in-image mask of view v. */
int gnr_sample_volume_fwd(const GnrScene* scene, const float* bbox_min /*[B,3]*/, int volume_res,
                          const float* packed_coarse /*device*/, float* sdf_out /*[B,res,res,res]*/,
                          unsigned char* view_mask_out, void* workspace, size_t workspace_bytes,
                          void* stream);

/* render_by_depth (renderer.py:110-138): one pass over given sample depths [B,rn,dn]
 * (ascending along each ray).  `level_weights` = packed coarse or fine blob (device). */
int gnr_render_by_depth_fwd(const GnrScene* scene, const GnrRays* rays, const float* depth, int dn,
                            const float* level_weights, GnrRenderOut* out, void* workspace,
                            size_t workspace_bytes, void* stream);

/* render (renderer.py:140-162, :201-220), eval mode: coarse pass on disparity-uniform
 * depths, inverse-CDF fine resampling (render_ops.py:172-229) + sort, fine pass.
 *   fine_depth_in : optional [B,rn,fdn]; when given it replaces the resampled depths
 *   fine_inds_out : optional [B,rn,fdn] int32 searchsorted indices of the resampling
 *   fine == NULL  : cfg use_hierarchical_sampling false (renderer.py:153-162): the coarse pass only -- no resampling, no fine
 *                   pass, packed_fine is not read (fine_depth_in / fine_inds_out must be NULL)                                  */
int gnr_render_rays_fwd(const GnrScene* scene, const GnrRays* rays, const float* packed_coarse,
                        const float* packed_fine, GnrRenderOut* coarse, GnrRenderOut* fine,
                        const float* fine_depth_in, int* fine_inds_out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* fine_depth_use_all under training (renderer.py:145-146: the fine pass renders torch.sort(torch.cat([coarse depths, resampled depths])))
 * -- depth_a [nrays,na], depth_b [nrays,nb], each ascending per ray -> out [nrays,na+nb] ascending (na + nb <= 128).  The inference
 * entry point gnr_render_rays_fwd does this merge itself (GnrRays.fine_depth_use_all). */
int gnr_merge_depths(const float* depth_a, int na, const float* depth_b, int nb, float* out, int nrays, void* stream);

/* predict_mean_for_depth_loss (renderer.py:222-266) for one level: bilinear gather of ray_feats at
 * `coords` [B,pn,2] (x,y in full-res pixels, shared by the V views of a scene; the caller keeps the
 * reference's (row,col)-as-(x,y) quirk) + the decoder mean branch.  mean_out [B,V,pn,2]. */
int gnr_depth_mean_fwd(const GnrScene* scene, const float* coords, int pn, const float* level_weights,
                       float* mean_out, void* workspace, size_t workspace_bytes, void* stream);

/* Bring-up aid: per-point intermediates of the chain kernel on the volume points (column order,
 * top->down); dbg is [B*res^3][32] floats, layout documented at the definition (gnr_capi.inc). */
int gnr_debug_volume_chain(const GnrScene* scene, const float* bbox_min, int volume_res,
                           const float* packed_coarse, float* dbg, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- grasp head: gd.networks.ConvNet.forward (src/gd/networks.py:48-54) --------
 * canonical blob = ConvNet state dict flattened in order: encoder.conv{1,2,3}, decoder.conv{1,2,3}, conv_qual,
 * conv_rot, conv_width (each .weight [Cout,Cin,k,k,k] then .bias).  volume [B,1,R,R,R] -> qual [B,1,40,40,40],
 * rot [B,4,40,40,40] (unit quaternions), width [B,1,40,40,40]. */
int gnr_head_canonical_floats(void);
int gnr_head_packed_floats(void);
int gnr_pack_grasp_head(const float* canonical_host, float* packed_host);
size_t gnr_grasp_head_workspace_bytes(int B, int volume_res);
int gnr_grasp_head_fwd(int B, int volume_res, const float* volume, const float* packed_head, float* qual, float* rot,
                       float* width, void* workspace, size_t workspace_bytes, void* stream);
const char* gnr_head_last_error(void);
/* Weight gradient of a stride-1, padding K/2 3D convolution (the grasp head under autograd; MIOpen spends 75 ms on the
 * fused 16 -> 6 k5 head at 40^3, batch 8): dw [Cout,Cin,K,K,K] is ACCUMULATED; x [B,Cin,D,H,W], dy [B,Cout,D,H,W]. */
int gnr_conv3d_bwd_weight(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int D, int H, int W, int K,
                          void* stream);

/* Stride-1, same-padding (K/2) 3D convolution on fp32 MFMA for the grasp head under autograd (gd/networks.py:30-37,91-96: the
 * two k5 layers at 20^3 / 40^3), K = 3 or 5.  w = the layer's canonical weights [Cout][Cin][K][K][K] on the device; the MFMA
 * fragments are packed on the device into `workspace` (gnr_conv3d_same_workspace_bytes) at every call, the weights move
 * every optimiser step.
 *   mode 0 (forward):        x [B][Cin][D][H][W]       -> y  [B][Cout][D][H][W]  (+ bias [Cout], or NULL)
 *   mode 1 (backward data):  x = dy [B][Cout][D][H][W] -> y = dx [B][Cin][D][H][W]  (transposed, flipped weights; bias ignored) */
/*   gnr_conv3d_same_bwd_weight: dw [Cout][Cin][K][K][K] is ACCUMULATED from x [B][Cin][D][H][W] and dy [B][Cout][D][H][W]
 *   (LDS-staged successor of gnr_conv3d_bwd_weight for K = 3 / 5; the workgroups' partial blocks go through `workspace` and
 *   are summed by a second kernel instead of hundreds of workgroups adding atomically into the same few thousand weights). */
size_t gnr_conv3d_same_workspace_bytes(int Cin, int Cout, int K);
size_t gnr_conv3d_same_bwd_weight_workspace_bytes(int B, int Cin, int Cout, int D, int H, int W, int K);
int gnr_conv3d_same_bwd_weight(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int D, int H, int W, int K,
                               void* workspace, size_t workspace_bytes, void* stream);
int gnr_conv3d_same(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int D, int H, int W, int K,
                    int mode, void* workspace, size_t workspace_bytes, void* stream);
/* Structurally sparse weights (round 4).  The encoder's stride-2 layers (gd/networks.py:33-37,59-76) run as ONE stride-1 k3 convolution
 * over the space-to-depth input (8 Cin channels, backbone.conv3d_stride2): of the 8 x 27 (parity, tap) slots per input channel 27 hold a
 * weight.  gnr_conv3d_tap_mask reads a pattern tensor [Cout][Cin][K^3] (non-zero = the weight exists; the STRUCTURE, not the values: a
 * weight that happens to be 0.0 still has a gradient) and writes gnr_conv3d_tap_mask_words(Cin, Cout) uint32 words: per (16 input
 * channels, 16 output channels) block the set of taps that exist.  The _masked entry points are gnr_conv3d_same /
 * gnr_conv3d_same_bwd_weight that skip the absent taps (mask == NULL: dense, the same as the plain entry points). */
#define GNR_CONV3D_FIRST_GEN 0x100   /* tests: OR into `mode` (gnr_conv3d_same*) / `K` (gnr_conv3d_same_bwd_weight*): the K = 3 call runs the
                                        first-generation kernels (the > 32 M voxel path) */
size_t gnr_conv3d_tap_mask_words(int Cin, int Cout);
int gnr_conv3d_tap_mask(const float* pattern, unsigned* mask, int Cin, int Cout, int K, void* stream);
int gnr_conv3d_same_masked(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int D, int H, int W,
                           int K, int mode, const unsigned* mask, void* workspace, size_t workspace_bytes, void* stream);
int gnr_conv3d_same_bwd_weight_masked(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int D, int H, int W, int K,
                                      const unsigned* mask, void* workspace, size_t workspace_bytes, void* stream);

/* ---- image-side streaming ops of the 2D feature extractors (csrc/gnr_img.hip) ----------------------------------
 * The residual U-Nets in front of the hot path (src/nr/network/ops.py:96-230, init_net.py:8-35, vis_encoder.py:6-22) are
 * convolutions (left to MIOpen) glued by instance norms, activations, reflect paddings and two bilinear upsamplings; these
 * entry points run the glue as single passes over the activation.  Layout: contiguous planes [planes = N*C][H][W] fp32.
 *
 * gnr_instnorm_act: y = act(InstanceNorm2d(x; weight, bias, eps) + res)   (ops.py:135-138: ELU; :101-121,215: ReLU, the
 *   block's identity as `res`; biased variance, eps inside the square root).  res may be NULL.  mean / rstd [planes] are
 *   written for the backward.  act: GNR_ACT_NONE / GNR_ACT_RELU / GNR_ACT_ELU (alpha = 1).
 * gnr_instnorm_act_bwd: with g = dy * act'(out):  dx = weight * rstd * (g - mean_hw(g) - xhat * mean_hw(g * xhat)),
 *   dres = g (NULL: not wanted), dbias[c] = sum_n sum_hw g, dweight[c] = sum_n sum_hw g * xhat (summed in a fixed order, no
 *   atomics; s1 / s2 [planes]: scratch for the per-plane sums).  `out` = the forward's y (may be NULL for GNR_ACT_NONE).
 * gnr_reflect_pad2d: F.pad(x, (pad,)*4, mode='reflect') as nn.Conv2d(padding_mode='reflect') applies it (ops.py:8,134,163):
 *   y [planes][H+2 pad][W+2 pad];  pad < min(H, W).  _bwd: dx[h][w] = sum of dy over the padded positions reading x[h][w].
 * gnr_upsample2x_bilinear: F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True) (ops.py:147): y [planes][2H][2W].
 * All return GNR_OK / GNR_ERR_ARG (null pointer) / GNR_ERR_SHAPE / GNR_ERR_HIP; text in gnr_img_last_error(). */
#define GNR_ACT_NONE 0
#define GNR_ACT_RELU 1
#define GNR_ACT_ELU 2
const char* gnr_img_last_error(void);
int gnr_instnorm_act(const float* x, const float* res, const float* weight, const float* bias, float* y, float* mean, float* rstd,
                     long long planes, int C, int HW, float eps, int act, void* stream);
int gnr_instnorm_act_bwd(const float* dy, const float* out, const float* x, const float* mean, const float* rstd, const float* weight,
                         float* dx, float* dres, float* s1, float* s2, float* dweight, float* dbias, long long planes, int C, int HW, int act,
                         void* stream);
int gnr_reflect_pad2d(const float* x, float* y, long long planes, int H, int W, int pad, void* stream);
int gnr_reflect_pad2d_bwd(const float* dy, float* dx, long long planes, int H, int W, int pad, void* stream);
int gnr_upsample2x_bilinear(const float* x, float* y, long long planes, int H, int W, void* stream);

/* ---- backward twins ------------------------------------------------------------------------
 * Parameter gradients are DETERMINISTIC: the kernels use no float atomics on them.  Points / rays are assigned to
 * wavefronts statically, every wavefront stores its partial sums into its own slot of a partial buffer inside the
 * caller-owned scratch / training workspace, and one reduction kernel per entry point adds the slots in slot order (in
 * double) into d_canonical / dtail.  Two calls on the same inputs return the same bits (the reference's CPU backward is
 * deterministic as well: ibrnet.py:497-504 + autograd).  The feature-map gradients (d_ray_feats, d_img_feats: a bilinear
 * scatter, like ATen's grid_sampler backward on a GPU) are accumulated with float atomics by default: their low bits depend on
 * the arrival order.  GNR_OPT_FEATURE_GRAD_FIXED (GnrScene.options of the backward call) makes them BIT-REPRODUCIBLE too: every
 * contribution is rounded once to a multiple of a launch-wide power-of-two quantum (2^-28 of the launch's largest upstream gradient,
 * found by an order-independent maximum) and added as a 64-bit integer (integer adds commute); a contribution more than 2^14 times the
 * upstream maximum is clamped and raises bit 3 of gnr_range_status.
 * Every entry point that produces feature-map gradients honours it (gnr_depth_mean_bwd, gnr_sample_volume_bwd,
 * gnr_render_chain_bwd); their workspaces are sized for either mode.  In the float mode the view kernels' scatter is BINNED (round 6,
 * csrc/gnr_bwd_scatter.inc): the per-(view, point) rows are parked in the training workspace and summed per feature-map pixel before
 * anything is added atomically (786 M -> ~40 M atomic dwords per 8-scene volume launch).  In the fixed-point mode the same rows are
 * summed as 64-bit integers (same rounding per contribution, same bits as the direct fixed-point scatter): no extra cost.
 * gnr_depth_mean_bwd: the backward of gnr_depth_mean_fwd (predict_mean_for_depth_loss, renderer.py:230-266,
 * consumed by DepthLoss, loss.py:87-144).  sample_volume and the per-view chain of the render passes follow below;
 * the render path's twin pairs (per-view chain, per-ray tail, compositing) follow further down.
 *   gnr_pack_weights_bwd: canonical blob -> transposed MFMA fragments [gnr_packed_bwd_floats()]
 *   gnr_depth_mean_bwd:   dmean [B,V,pn,2] -> d_canonical [gnr_canonical_weights_floats()] (ACCUMULATED:
 *                         mean_decoder.{0,2,4}.{weight,bias} entries, state-dict order) and
 *                         d_ray_feats [B,V,32,fh,fw] (overwritten; NULL to skip).  Needs gnr_prepare's
 *                         workspace (feature maps in channel-last form) and a separate scratch buffer of
 *                         gnr_depth_mean_bwd_workspace_bytes(scene) for the partial parameter gradients and the
 *                         channel-last feature gradient (required, also with d_ray_feats == NULL).            */
int gnr_packed_bwd_floats(void);
int gnr_pack_weights_bwd(const float* canonical_host, float* packed_bwd_host);
size_t gnr_depth_mean_bwd_workspace_bytes(const GnrScene* scene);
int gnr_depth_mean_bwd(const GnrScene* scene, const float* coords, int pn, const float* level_weights,
                       const float* level_weights_bwd, const float* dmean, float* d_canonical, float* d_ray_feats,
                       void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, void* stream);

/* sample_volume for training: the forward that also saves the per-view states (same values and speed as
 * gnr_sample_volume_fwd), and its backward in five stages (csrc/gnr_bwd.inc): bit 4 = attention / LayerNorm /
 * out_geometry_fc tail, bit 3 = geometry_fc + second cross-view reduction, bit 2 = second view loop (base_fc, vis_fc,
 * vis_fc2), bit 1 = hoisted base_fc.0 columns + first reduction, bit 0 = first view loop (decoder, prob_embed,
 * ray_dir_fc, neuray gate) + feature-map gradients.  `stages` = 0x1f runs all of it; partial masks exist for the staged
 * tests, which read the inter-stage gradients through gnr_train_workspace_layout().  d_canonical is ACCUMULATED
 * (state-dict order, coarse level); d_ray_feats / d_img_feats [B,V,32,fh,fw] are overwritten (NULL to skip).  The
 * training workspace and the regular workspace must survive untouched from the forward to the backward.              */
size_t gnr_sample_volume_train_workspace_bytes(const GnrScene* scene, int volume_res);
int gnr_train_workspace_layout(const GnrScene* scene, int volume_res, size_t* offsets_out9);
int gnr_sample_volume_fwd_train(const GnrScene* scene, const float* bbox_min, int volume_res, const float* level_weights,
                                float* sdf_out, void* workspace, size_t workspace_bytes, void* train_workspace,
                                size_t train_workspace_bytes, void* stream);
int gnr_sample_volume_bwd(const GnrScene* scene, int volume_res, const float* level_weights, const float* level_weights_bwd,
                          const float* canonical_weights_dev, const float* dvol, float* d_canonical, float* d_ray_feats,
                          float* d_img_feats, void* workspace, size_t workspace_bytes, void* train_workspace,
                          size_t train_workspace_bytes, int stages, void* stream);

/* Render path for training: the per-view chain of one render pass in both directions; the per-ray tail (renderer.py:90-108,
 * ibrnet.py:485-504) and NeuS alpha / compositing have their own twin pairs below; the ray geometry (S1-S3), the inverse-CDF
 * resampling and the sort (F1/F2) run in the forward kernels.  PyTorch autograd only connects the pairs with the losses.
 *   stats_out [B, rn*dn, 66] = mean(32) var(32) wbar n_valid_views (true scale); colors_out [B, rn*dn, 3]           */
size_t gnr_render_chain_train_workspace_bytes(const GnrScene* scene, int rn, int dn);
/*   depth == NULL: the coarse pass, sample_depth (render_ops.py:146-170) on the device (dn must be rays->dn, 3..64).
 *   depth != NULL: dn caller-given depths per ray, 3..128 (fine_depth_use_all, renderer.py:145-146: the fine pass renders the
 *   coarse and the resampled depths together); gnr_render_tail_fwd_train, gnr_ray_tail_dual_bwd and gnr_composite_bwd take the
 *   same range (the inverse-CDF resampler, fine_depth_out, at most 64).
 *   depth_out [B,rn,dn], pts_out [B,rn*dn,3], qdir_out [B,rn,3] (each nullable): the depths the pass used and its ray
 *   geometry (render_ops.py:4-39: sample points, normalised query directions) for the tail / compositing backward.  */
int gnr_render_chain_fwd_train(const GnrScene* scene, const GnrRays* rays, const float* depth, int dn, const float* level_weights,
                               float* stats_out, float* colors_out, float* depth_out, float* pts_out, float* qdir_out,
                               void* workspace, size_t workspace_bytes, void* train_workspace, size_t train_workspace_bytes,
                               void* stream);
int gnr_render_chain_bwd(const GnrScene* scene, int rn, int dn, const float* level_weights, const float* level_weights_bwd,
                         const float* dstats, const float* dcolors, float* d_canonical, float* d_ray_feats, float* d_img_feats,
                         void* workspace, size_t workspace_bytes, void* train_workspace, size_t train_workspace_bytes,
                         void* stream);

/* Per-ray tail of a training render pass (ibrnet.py:485-504: geometry_fc, positional encoding, 40-token attention,
 * LayerNorm, out_geometry_fc, clip, and the in-forward gradient of sdf w.r.t. the ray points, create_graph=True).
 * Forward: gnr_render_tail_fwd_train runs the inference tail kernel (tail + NeuS alpha + compositing) on the records
 * gnr_render_chain_fwd_train just left in `workspace` (call it right after that chain): out->colors_nr is the INPUT (the
 * chain's colours), every other non-null member of `out` is written as by gnr_render_by_depth_fwd.
 * Backward: with a = dL/d sdf and gamma = dL/d grad the gradients are those of  sum a*sdf + <gamma, grad>  =  a reverse
 * pass over the tail evaluated on dual numbers (value, derivative along gamma).  gnr_ray_tail_dual_bwd is its attention /
 * LayerNorm / out_geometry_fc core for a flat list of rays: g, gd [nrays*dn,16] value and tangent of geometry_fc's output,
 * a, nvalid [nrays*dn] -> gbar, gdbar [nrays*dn,16] and dtail [gnr_ray_tail_grad_floats()] = dWq, dWk, dWv, dWfc [16][16],
 * dLNw, dLNb [16], d w_eff [16], d b_eff (out_geometry_fc folded into one row; unfolded by the caller).  The two ELU
 * layers of geometry_fc around it: gnr_geo_dual_fwd / gnr_geo_dual_bwd below.                                        */
/* fine_depth_out [B,rn,rays->fdn] (nullable): this (coarse) pass's inverse-CDF resampling, sorted (render_ops.py:172-229,
 * renderer.py:146-148); rays->fine_u = the is_train draws, NULL = eval midpoints. */
int gnr_render_tail_fwd_train(const GnrScene* scene, const GnrRays* rays, const float* depth, int dn, const float* level_weights,
                              GnrRenderOut* out, float* fine_depth_out, void* workspace, size_t workspace_bytes,
                              void* train_workspace, size_t train_workspace_bytes, void* stream);
int gnr_ray_tail_grad_floats(void);
/* geometry_fc (86 -> 64 -> 16, two ELUs; ibrnet.py:488-489) on dual numbers, around gnr_ray_tail_dual_bwd.  canonical_dev =
 * the level's parameters in state-dict order on the device; stats [P,66] (mean 32, var 32, wbar, n_valid), pts, gamma [P,3].
 * fwd: g, gd [P,16] = value / derivative along gamma of geometry_fc's output.  bwd: gbar, gdbar [P,16] -> dstats [P,66]
 * (overwritten) and the gradients of geometry_fc.{0,2}.{weight,bias} ACCUMULATED into d_canonical (state-dict order).   */
/* fwd runs on the f16 matrix cores like the backward's per-point half (16 points as the columns of a chained fp16-pair MFMA, fragments
 * packed per call into caller-owned scratch of gnr_geo_dual_fwd_workspace_bytes() bytes; a call whose outputs come out non-finite -- a
 * weight or an operand beyond the fp16 range -- is recomputed by the fp32 kernel launched behind it); GNR_OPT_GEO_DUAL_FP32: the fp32 FMA
 * kernel alone (scratch may then be NULL). */
size_t gnr_geo_dual_fwd_workspace_bytes(void);
int gnr_geo_dual_fwd(const float* canonical_dev, const float* stats, const float* pts, const float* gamma, float* g, float* gd,
                     int P, void* scratch, size_t scratch_bytes, unsigned options, void* stream);
/* bwd runs as two kernels (the per-point reverse pass -- 16 points as the columns of a chained fp16-pair MFMA, its fragments packed per
 * call from canonical_dev -- then the weight-gradient outer products over the points on the matrix cores) with the per-point adjoints
 * between them (288 floats per point), the fragment image (64 KB) and the partial weight gradients in caller-owned scratch of
 * gnr_geo_dual_bwd_workspace_bytes(P) bytes. */
size_t gnr_geo_dual_bwd_workspace_bytes(int P);
int gnr_geo_dual_bwd(const float* canonical_dev, const float* stats, const float* pts, const float* gamma, const float* gbar,
                     const float* gdbar, float* dstats, float* d_canonical, int P, void* scratch, size_t scratch_bytes, unsigned options,
                     void* stream);
/* Backward of NeuS alpha + compositing (aggregate_net.py:105-121, render_ops.py:72-80, renderer.py:110-123) for a flat list
 * of rays, forward values taken from the tensors the forward wrote: sdf [nrays*dn], grad, col [nrays*dn,3], depth
 * [nrays*dn], qdir [nrays,3].  Upstream: dpix [nrays,3]; ddepth [nrays], wgerr [nrays] (d L / d sum_k (|grad_k|-1)^2 of the
 * ray), dalpha, dhit [nrays*dn] may be null.  Out: a_out = dL/d sdf, gamma_out = dL/d grad (the upstreams of
 * gnr_ray_tail_dual_bwd), dcol_out, dvar_out[0] = dL/d deviation_network.variance (all overwritten).
 * scratch: gnr_composite_bwd_workspace_bytes(nrays) / gnr_ray_tail_dual_bwd_workspace_bytes() bytes, caller-owned (the
 * per-wavefront partial sums of the parameter gradients).  `options` of gnr_ray_tail_dual_bwd / gnr_geo_dual_fwd / gnr_geo_dual_bwd: GNR_OPT_* (these
 * entry points take no GnrScene): GNR_OPT_POISON_PARTIALS, and GNR_OPT_GEO_DUAL_FP32 for gnr_geo_dual_fwd / gnr_geo_dual_bwd.                 */
size_t gnr_composite_bwd_workspace_bytes(int nrays);
int gnr_composite_bwd(const float* level_weights, const float* sdf, const float* grad, const float* col, const float* depth,
                      const float* qdir, const float* dpix, const float* ddepth, const float* wgerr, const float* dalpha,
                      const float* dhit, float* a_out, float* gamma_out, float* dcol_out, float* dvar_out, int nrays, int dn,
                      void* scratch, size_t scratch_bytes, void* stream);
size_t gnr_ray_tail_dual_bwd_workspace_bytes(void);
int gnr_ray_tail_dual_bwd(const float* level_weights, const float* g, const float* gd, const float* a, const float* nvalid,
                          float* gbar, float* gdbar, float* dtail, int nrays, int dn, void* scratch, size_t scratch_bytes,
                          unsigned options, void* stream);

/* ---- host helper ---------------------------------------------------------------------------
 * The first k entries of torch.randperm(n) on the CPU generator, bit-exact, in O(k + n/624) instead of n random-access swaps:
 * the reference draws the depth-loss pixels as torch.randperm(h*w)[:8192] (src/nr/network/renderer.py:222-228), 5-15 ms per
 * scene on the host.  torch_cpu_rng_state = the bytes of torch.get_rng_state() (5056), advanced in place exactly as
 * torch.randperm(n) would advance the generator; out[k] int64.  Returns GNR_ERR_SHAPE for state layouts / sizes it does not
 * know (the caller then falls back to torch.randperm).  No device work.                                          */
int gnr_host_randperm_prefix(unsigned char* torch_cpu_rng_state, long long state_bytes, long long n, int k, long long* out);

/* ---- grasp post-processing on the device -------------------------------------------------
 * Replaces the reference planner's `process` + `select` (src/nr/main.py:23-57, 60-84), which run
 * scipy.ndimage (gaussian_filter sigma=1 mode='nearest'; binary_dilation iterations=2 with mask;
 * maximum_filter size=4) on the host for every plan.  Bit-exact with scipy: fp64 accumulation in
 * scipy's order, fp32 rounding after every Gaussian axis.
 *   tsdf, qual, width [B,R,R,R], rot [B,4,R,R,R]  (outputs of sample_volume / the grasp head)
 *   qual_out [B,R,R,R]      processed quality volume                        (main.py:41-55)
 *   count [B]               selected grasps per scene (may exceed max_n: only the first max_n are stored)
 *   index [B,max_n,3] (i,j,k), score [B,max_n], quat [B,max_n,4] (x,y,z,w as stored in rot),
 *   width_out [B,max_n]     in np.argwhere order                            (main.py:70-84)              */
#define GNR_GAUSS_MAX_RADIUS 16
typedef struct GnrSelectParams {
    int gauss_radius;                          /* int(4*sigma + 0.5)                                    */
    double gauss_w[GNR_GAUSS_MAX_RADIUS + 1];  /* w[k] = exp(-k^2/(2 sigma^2)) / sum, k = 0..radius     */
    float tsdf_thres_high, tsdf_thres_low;     /* planner: 0, -0.85 (main.py:93-94)                     */
    float min_width, max_width;                /* 1.33, 9.33 (main.py:29-30)                            */
    float threshold;                           /* 0.90 (main.py:60)                                     */
    int dilate_iterations;                     /* 2 (main.py:48)                                        */
    int max_filter_size;                       /* 4 (main.py:60)                                        */
} GnrSelectParams;
size_t gnr_grasp_select_workspace_bytes(int B, int R);
int gnr_grasp_select_fwd(const float* tsdf, const float* qual, const float* rot, const float* width, int B, int R,
                         const GnrSelectParams* params, float* qual_out, int* count, int* index, float* score,
                         float* quat, float* width_out, int max_n, void* workspace, size_t workspace_bytes, void* stream);
const char* gnr_post_last_error(void);

/* ---- introspection / measurement -------------------------------------------------------*/
/* name of the dominant kernel as it appears in rocprofv3 traces, and the last HIP error text */
const char* gnr_dominant_kernel_name(void);
const char* gnr_last_error(void);
/* In-situ timing (process-wide switch, measurement only): between gnr_timing_begin() and gnr_timing_end() every kernel
 * launch of the library is bracketed by a pair of HIP events recorded on ITS launch stream (nothing else is added to the
 * path).  gnr_timing_end() waits for the events and writes one line per label -- "label count total_ms\n", label =
 * "kernel@entry point" (k_chain / k_ray: "k_chain.volume", "k_chain.render", "...train") -- into `report` (NUL-terminated,
 * truncated to report_bytes); returns the number of launches recorded or a negative error code.
 * Event pairs around EVERY launch cost a few microseconds of pipeline bubble each (11 launches per forward step: ~2 % of the
 * step), so a headline measurement brackets its dominant kernel only (gnr_timing_begin_only) and takes the full table in a
 * separate pass.  gnr_chain_timing_begin/end: the same switch for the k_chain launches on volume points, reduced to their
 * average duration. */
int gnr_timing_begin(void);
int gnr_timing_begin_only(const char* label_substring);   /* bracket only the launches whose label contains the substring */
int gnr_timing_end(char* report, size_t report_bytes);
int gnr_chain_timing_begin(void);
int gnr_chain_timing_end(float* avg_ms_out, int* count_out);
/* Time `iters` launches of the dominant kernel alone (volume points of `scene`) with HIP
 * events on `stream`; returns average milliseconds per launch in *ms_out. */
int gnr_time_chain_kernel(const GnrScene* scene, int volume_res, const float* packed_coarse,
                          void* workspace, size_t workspace_bytes, int iters, float* ms_out, void* stream);

/* Test tooling: fill the LDS of every CU with a 32-bit pattern (LDS is not cleared between kernels; a kernel that reads LDS it did
 * not write would compute on it).  tests/test_range_guard.py runs the path after two patterns and asks for identical bits. */
int gnr_debug_fill_lds(unsigned pattern, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GNR_H */
