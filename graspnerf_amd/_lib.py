"""ctypes binding of libgnr.so (include/gnr.h).  The product path has NO CPU fallback: if the
library is missing or fails to load, importing the hot path raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', os.environ.get('GNR_LIB', 'libgnr.so'))   # GNR_LIB: A/B builds

GNR_OK, GNR_ERR_ARG, GNR_ERR_SHAPE, GNR_ERR_HIP, GNR_ERR_WORKSPACE = 0, -1, -2, -3, -4
ERRORS = {-1: 'GNR_ERR_ARG', -2: 'GNR_ERR_SHAPE', -3: 'GNR_ERR_HIP', -4: 'GNR_ERR_WORKSPACE'}

c_float_p = C.POINTER(C.c_float)


class GnrScene(C.Structure):
    _fields_ = [('B', C.c_int), ('V', C.c_int), ('H', C.c_int), ('W', C.c_int), ('fh', C.c_int), ('fw', C.c_int),
                ('imgs', C.c_void_p), ('img_feats', C.c_void_p), ('ray_feats', C.c_void_p),
                ('poses', C.c_void_p), ('Ks', C.c_void_p), ('depth_range', C.c_void_p), ('use_vis', C.c_int), ('options', C.c_uint)]


# per-call options (GnrScene.options; the `options` argument of gnr_ray_tail_dual_bwd / gnr_geo_dual_bwd): include/gnr.h GNR_OPT_*
OPTIONS = {'fp32_chain': 0x001, 'feature_grad_fixed': 0x002, 'view1_one_wavefront': 0x004, 'view2_one_wavefront': 0x008,
           'ray_order_morton': 0x010, 'poison_partials': 0x020, 'direct_scatter': 0x040, 'geo_dual_fp32': 0x080, 'test_lose_partner': 0x100, 'static_tiles': 0x200, 'split_launch': 0x400}
GNR_STATUS_LOST_PARTNER = 16


class GnrRays(C.Structure):
    _fields_ = [('rn', C.c_int), ('dn', C.c_int), ('fdn', C.c_int),
                ('ray_mask_view_num', C.c_int), ('ray_mask_point_num', C.c_int),
                ('coords', C.c_void_p), ('que_pose', C.c_void_p), ('que_K', C.c_void_p),
                ('que_depth_range', C.c_void_p), ('que_imgs', C.c_void_p), ('fine_u', C.c_void_p),
                ('ray_batch_num', C.c_int), ('fine_depth_use_all', C.c_int)]


RENDER_OUT_FIELDS = ['depth', 'sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr',
                     'pixel_colors_gt', 'render_depth', 'ray_mask', 'sdf_gradient_error', 'sdf_gradient', 'view_mask']


class GnrRenderOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in RENDER_OUT_FIELDS]


GNR_GAUSS_MAX_RADIUS = 16


class GnrSelectParams(C.Structure):
    _fields_ = [('gauss_radius', C.c_int), ('gauss_w', C.c_double * (GNR_GAUSS_MAX_RADIUS + 1)),
                ('tsdf_thres_high', C.c_float), ('tsdf_thres_low', C.c_float), ('min_width', C.c_float),
                ('max_width', C.c_float), ('threshold', C.c_float), ('dilate_iterations', C.c_int),
                ('max_filter_size', C.c_int)]


class GnrError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libgnr.so once.  Raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnrError(f'{LIB_PATH} not found: build it with graspnerf_amd/csrc/build.sh '
                       f'(or __graft_entry__.build()); there is no CPU fallback for the hot path')
    # torch first: the PyTorch-ROCm wheel ships its own ROCm runtime libraries, and device pointers / streams cross between the two
    # (plumbing, include/gnr.h).  A process that dlopen()s libgnr.so before importing torch binds /opt/rocm's runtime, torch then
    # loads its own next to it, and the second runtime finds no device (`python __graft_entry__.py smoke`: build() then smoke()).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.gnr_canonical_weights_floats.restype = C.c_int
    L.gnr_packed_weights_floats.restype = C.c_int
    L.gnr_pack_weights.argtypes = [c_float_p, c_float_p]
    L.gnr_pack_weights.restype = C.c_int
    L.gnr_pack_vis_decoder.argtypes = [c_float_p, c_float_p]
    L.gnr_pack_vis_decoder.restype = C.c_int
    L.gnr_pack_vis_decoder_bwd.argtypes = [c_float_p, c_float_p]
    L.gnr_pack_vis_decoder_bwd.restype = C.c_int
    L.gnr_canonical_vis_floats.restype = C.c_int
    for name in ('gnr_pack_weights_device', 'gnr_pack_weights_bwd_device', 'gnr_pack_vis_decoder_device', 'gnr_pack_vis_decoder_bwd_device'):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        getattr(L, name).restype = C.c_int
    L.gnr_layout_offset.argtypes = [C.c_char_p]
    L.gnr_layout_offset.restype = C.c_int
    L.gnr_workspace_bytes.argtypes = [C.POINTER(GnrScene), C.c_int, C.c_int, C.c_int]
    L.gnr_workspace_bytes.restype = C.c_size_t
    L.gnr_prepare.argtypes = [C.POINTER(GnrScene), C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_prepare.restype = C.c_int
    L.gnr_range_status.argtypes = [C.POINTER(GnrScene), C.c_void_p, C.c_size_t, C.POINTER(C.c_uint), C.c_void_p]
    L.gnr_range_status.restype = C.c_int
    L.gnr_status_words_offset.argtypes = [C.POINTER(GnrScene)]
    L.gnr_status_words_offset.restype = C.c_size_t
    L.gnr_sample_volume_fwd.argtypes = [C.POINTER(GnrScene), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_sample_volume_fwd.restype = C.c_int
    L.gnr_debug_volume_chain.argtypes = [C.POINTER(GnrScene), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_debug_volume_chain.restype = C.c_int
    L.gnr_depth_mean_fwd.argtypes = [C.POINTER(GnrScene), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_depth_mean_fwd.restype = C.c_int
    L.gnr_merge_depths.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.gnr_merge_depths.restype = C.c_int
    L.gnr_render_by_depth_fwd.argtypes = [C.POINTER(GnrScene), C.POINTER(GnrRays), C.c_void_p, C.c_int, C.c_void_p,
                                          C.POINTER(GnrRenderOut), C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_render_by_depth_fwd.restype = C.c_int
    L.gnr_render_rays_fwd.argtypes = [C.POINTER(GnrScene), C.POINTER(GnrRays), C.c_void_p, C.c_void_p,
                                      C.POINTER(GnrRenderOut), C.POINTER(GnrRenderOut), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_render_rays_fwd.restype = C.c_int
    L.gnr_head_canonical_floats.restype = C.c_int
    L.gnr_head_packed_floats.restype = C.c_int
    L.gnr_pack_grasp_head.argtypes = [c_float_p, c_float_p]
    L.gnr_pack_grasp_head.restype = C.c_int
    L.gnr_grasp_head_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.gnr_grasp_head_workspace_bytes.restype = C.c_size_t
    L.gnr_grasp_head_fwd.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_grasp_head_fwd.restype = C.c_int
    L.gnr_head_last_error.restype = C.c_char_p
    L.gnr_conv3d_bwd_weight.argtypes = [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p]
    L.gnr_conv3d_same_bwd_weight.argtypes = [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_conv3d_same_bwd_weight_workspace_bytes.argtypes = [C.c_int] * 7
    L.gnr_conv3d_same_bwd_weight_workspace_bytes.restype = C.c_size_t
    L.gnr_conv3d_same_bwd_weight.restype = C.c_int
    L.gnr_conv3d_same_workspace_bytes.argtypes = [C.c_int] * 3
    L.gnr_conv3d_same_workspace_bytes.restype = C.c_size_t
    L.gnr_conv3d_same.argtypes = [C.c_void_p] * 4 + [C.c_int] * 8 + [C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_conv3d_same.restype = C.c_int
    L.gnr_conv3d_tap_mask_words.argtypes = [C.c_int] * 2
    L.gnr_conv3d_tap_mask_words.restype = C.c_size_t
    L.gnr_conv3d_tap_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.gnr_conv3d_tap_mask.restype = C.c_int
    L.gnr_conv3d_same_masked.argtypes = [C.c_void_p] * 4 + [C.c_int] * 8 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_conv3d_same_masked.restype = C.c_int
    L.gnr_conv3d_same_bwd_weight_masked.argtypes = [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_conv3d_same_bwd_weight_masked.restype = C.c_int
    L.gnr_conv3d_bwd_weight.restype = C.c_int
    L.gnr_packed_bwd_floats.restype = C.c_int
    L.gnr_pack_weights_bwd.argtypes = [c_float_p, c_float_p]
    L.gnr_pack_weights_bwd.restype = C.c_int
    L.gnr_depth_mean_bwd_workspace_bytes.argtypes = [C.POINTER(GnrScene)]
    L.gnr_depth_mean_bwd_workspace_bytes.restype = C.c_size_t
    L.gnr_depth_mean_bwd.argtypes = [C.POINTER(GnrScene), C.c_void_p, C.c_int] + [C.c_void_p] * 5 + \
                                    [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_depth_mean_bwd.restype = C.c_int
    L.gnr_sample_volume_train_workspace_bytes.argtypes = [C.POINTER(GnrScene), C.c_int]
    L.gnr_sample_volume_train_workspace_bytes.restype = C.c_size_t
    L.gnr_train_workspace_layout.argtypes = [C.POINTER(GnrScene), C.c_int, C.POINTER(C.c_size_t)]
    L.gnr_train_workspace_layout.restype = C.c_int
    L.gnr_sample_volume_fwd_train.argtypes = [C.POINTER(GnrScene), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_sample_volume_fwd_train.restype = C.c_int
    L.gnr_sample_volume_bwd.argtypes = [C.POINTER(GnrScene), C.c_int] + [C.c_void_p] * 7 + [C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_size_t, C.c_int, C.c_void_p]
    L.gnr_sample_volume_bwd.restype = C.c_int
    L.gnr_render_chain_train_workspace_bytes.argtypes = [C.POINTER(GnrScene), C.c_int, C.c_int]
    L.gnr_render_chain_train_workspace_bytes.restype = C.c_size_t
    L.gnr_render_chain_fwd_train.argtypes = [C.POINTER(GnrScene), C.POINTER(GnrRays), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                             C.c_size_t, C.c_void_p]
    L.gnr_render_chain_fwd_train.restype = C.c_int
    L.gnr_render_chain_bwd.argtypes = [C.POINTER(GnrScene), C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_render_chain_bwd.restype = C.c_int
    L.gnr_render_tail_fwd_train.argtypes = [C.POINTER(GnrScene), C.POINTER(GnrRays), C.c_void_p, C.c_int, C.c_void_p,
                                            C.POINTER(GnrRenderOut), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_void_p]
    L.gnr_composite_bwd.argtypes = [C.c_void_p] * 15 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_composite_bwd_workspace_bytes.argtypes = [C.c_int]
    L.gnr_composite_bwd_workspace_bytes.restype = C.c_size_t
    L.gnr_host_randperm_prefix.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_void_p]
    L.gnr_host_randperm_prefix.restype = C.c_int
    L.gnr_geo_dual_fwd.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p]
    L.gnr_geo_dual_fwd_workspace_bytes.argtypes = []
    L.gnr_geo_dual_fwd_workspace_bytes.restype = C.c_size_t
    L.gnr_geo_dual_fwd.restype = C.c_int
    L.gnr_geo_dual_bwd.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p]
    L.gnr_geo_dual_bwd_workspace_bytes.argtypes = [C.c_int]
    L.gnr_geo_dual_bwd_workspace_bytes.restype = C.c_size_t
    L.gnr_geo_dual_bwd.restype = C.c_int
    L.gnr_composite_bwd.restype = C.c_int
    L.gnr_render_tail_fwd_train.restype = C.c_int
    L.gnr_ray_tail_grad_floats.restype = C.c_int
    L.gnr_ray_tail_dual_bwd.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p]
    L.gnr_ray_tail_dual_bwd_workspace_bytes.restype = C.c_size_t
    L.gnr_ray_tail_dual_bwd.restype = C.c_int
    L.gnr_grasp_select_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.gnr_grasp_select_workspace_bytes.restype = C.c_size_t
    L.gnr_grasp_select_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.POINTER(GnrSelectParams)] + \
                                      [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.gnr_grasp_select_fwd.restype = C.c_int
    L.gnr_post_last_error.restype = C.c_char_p
    L.gnr_time_chain_kernel.argtypes = [C.POINTER(GnrScene), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                        c_float_p, C.c_void_p]
    L.gnr_time_chain_kernel.restype = C.c_int
    L.gnr_chain_timing_begin.restype = C.c_int
    L.gnr_timing_begin.restype = C.c_int
    L.gnr_timing_begin_only.argtypes = [C.c_char_p]
    L.gnr_timing_begin_only.restype = C.c_int
    L.gnr_timing_end.argtypes = [C.c_char_p, C.c_size_t]
    L.gnr_timing_end.restype = C.c_int
    L.gnr_chain_timing_end.argtypes = [c_float_p, C.POINTER(C.c_int)]
    L.gnr_chain_timing_end.restype = C.c_int
    L.gnr_img_last_error.restype = C.c_char_p
    L.gnr_instnorm_act.argtypes = [C.c_void_p] * 7 + [C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
    L.gnr_instnorm_act.restype = C.c_int
    L.gnr_instnorm_act_bwd.argtypes = [C.c_void_p] * 12 + [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.gnr_instnorm_act_bwd.restype = C.c_int
    for fn in (L.gnr_reflect_pad2d, L.gnr_reflect_pad2d_bwd):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
        fn.restype = C.c_int
    L.gnr_upsample2x_bilinear.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
    L.gnr_upsample2x_bilinear.restype = C.c_int
    L.gnr_last_error.restype = C.c_char_p
    L.gnr_dominant_kernel_name.restype = C.c_char_p
    L.gnr_debug_fill_lds.argtypes = [C.c_uint, C.c_void_p]
    L.gnr_debug_fill_lds.restype = C.c_int
    _lib = L
    return L


EXPORTED = ['gnr_canonical_weights_floats', 'gnr_packed_weights_floats', 'gnr_pack_weights', 'gnr_pack_vis_decoder', 'gnr_pack_vis_decoder_bwd', 'gnr_canonical_vis_floats', 'gnr_pack_weights_device', 'gnr_pack_weights_bwd_device', 'gnr_pack_vis_decoder_device', 'gnr_pack_vis_decoder_bwd_device', 'gnr_layout_offset', 'gnr_workspace_bytes',
            'gnr_prepare', 'gnr_range_status', 'gnr_status_words_offset', 'gnr_sample_volume_fwd', 'gnr_debug_volume_chain', 'gnr_depth_mean_fwd', 'gnr_merge_depths', 'gnr_render_by_depth_fwd', 'gnr_render_rays_fwd',
            'gnr_dominant_kernel_name', 'gnr_last_error', 'gnr_time_chain_kernel', 'gnr_head_canonical_floats',
            'gnr_head_packed_floats', 'gnr_pack_grasp_head', 'gnr_grasp_head_workspace_bytes', 'gnr_grasp_head_fwd',
            'gnr_head_last_error', 'gnr_chain_timing_begin', 'gnr_chain_timing_end', 'gnr_timing_begin', 'gnr_timing_begin_only', 'gnr_timing_end', 'gnr_grasp_select_workspace_bytes',
            'gnr_grasp_select_fwd', 'gnr_post_last_error', 'gnr_packed_bwd_floats', 'gnr_pack_weights_bwd',
            'gnr_depth_mean_bwd_workspace_bytes', 'gnr_depth_mean_bwd', 'gnr_sample_volume_train_workspace_bytes',
            'gnr_train_workspace_layout', 'gnr_sample_volume_fwd_train', 'gnr_sample_volume_bwd',
            'gnr_render_chain_train_workspace_bytes', 'gnr_render_chain_fwd_train', 'gnr_render_chain_bwd', 'gnr_conv3d_bwd_weight', 'gnr_conv3d_same_workspace_bytes', 'gnr_conv3d_same', 'gnr_conv3d_tap_mask_words', 'gnr_conv3d_tap_mask', 'gnr_conv3d_same_masked', 'gnr_conv3d_same_bwd_weight_masked', 'gnr_conv3d_same_bwd_weight', 'gnr_conv3d_same_bwd_weight_workspace_bytes',
            'gnr_render_tail_fwd_train', 'gnr_ray_tail_grad_floats', 'gnr_ray_tail_dual_bwd', 'gnr_ray_tail_dual_bwd_workspace_bytes', 'gnr_composite_bwd', 'gnr_composite_bwd_workspace_bytes', 'gnr_geo_dual_fwd', 'gnr_geo_dual_fwd_workspace_bytes', 'gnr_geo_dual_bwd', 'gnr_geo_dual_bwd_workspace_bytes', 'gnr_host_randperm_prefix',
            'gnr_debug_fill_lds', 'gnr_img_last_error', 'gnr_instnorm_act', 'gnr_instnorm_act_bwd', 'gnr_reflect_pad2d', 'gnr_reflect_pad2d_bwd', 'gnr_upsample2x_bilinear']


def check(rc, what):
    if rc != GNR_OK:
        msg = lib().gnr_last_error().decode(errors='replace')
        raise GnrError(f'{what} failed: {ERRORS.get(rc, rc)} ({msg})')


def timing_begin(only=None):
    """Start bracketing the kernel launches of libgnr.so (all of them, or those whose label contains `only`) with HIP events
    on their launch stream (include/gnr.h)."""
    check(lib().gnr_timing_begin_only(only.encode() if only else None), 'gnr_timing_begin')


def timing_end():
    """-> {label: (launches, total_ms)} of the launches since timing_begin(); waits for them."""
    buf = C.create_string_buffer(1 << 16)
    n = lib().gnr_timing_end(buf, len(buf))
    if n < 0:
        check(n, 'gnr_timing_end')
    out = {}
    for line in buf.value.decode().splitlines():
        label, cnt, ms = line.rsplit(' ', 2)
        out[label] = (int(cnt), float(ms))
    return out
