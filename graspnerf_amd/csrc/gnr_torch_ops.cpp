// torch.ops.graspnerf.*: the hot path registered with the PyTorch dispatcher ("Torch side: torch.ops.graspnerf.
// sample_volume / render_rays registered via TORCH_LIBRARY").  A thin shim over the C ABI of include/gnr.h: every op allocates its
// outputs / workspaces as tensors on the inputs' device, takes the current HIP stream and calls the same entry points the ctypes
// route (graspnerf_amd/_lib.py, hotpath.py) calls -- same kernels, same bits.  No arithmetic lives here.
//   sample_volume        NeuralRayRenderer.sample_volume (renderer.py:164-199)                         -> volume [B,1,R,R,R]
//   render_rays          NeuralRayRenderer.render (renderer.py:201-220 + 140-162), both levels         -> 2 x 10 tensors
//   sample_volume_train  the same forward with the saved states of its backward                         -> (volume, ws, tws)
//   sample_volume_bwd    its backward twins (csrc/gnr_bwd.inc)                                          -> (d canonical, d ray_feats, d img_feats)
// Scene tensors as in hotpath.py: imgs [B,V,3,H,W], img_feats / ray_feats [B,V,32,fh,fw], poses [B,V,3,4], Ks [B,V,3,3],
// depth_range [B,V,2]; float32, contiguous, one device.  Built by csrc/build.sh into libgnr_torch.so (links libgnr.so).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>      // the ROCm build of PyTorch presents HIP devices / streams under the "cuda" device type
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>
#include <torch/csrc/autograd/autograd_not_implemented_fallback.h>

#include <tuple>
#include <vector>

#include "../../include/gnr.h"

namespace {

using at::Tensor;

void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

void ok(int rc, const char* what) { TORCH_CHECK(rc == GNR_OK, what, " failed (", rc, "): ", gnr_last_error()); }

const float* fp(const Tensor& t) { return t.data_ptr<float>(); }

Tensor f32(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "graspnerf ops: ", name, " must be on the GPU (there is no CPU fallback)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "graspnerf ops: ", name, " must be float32");
    return t.contiguous();
}

// every further argument lives on the scene's device and has the size the C ABI will read (the entry points take raw pointers:
// a short or misplaced tensor would be an out-of-bounds device read, not an error)
Tensor arg(const Tensor& t, const char* name, const Tensor& ref, int64_t numel) {
    Tensor c = f32(t, name);
    TORCH_CHECK(c.device() == ref.device(), "graspnerf ops: ", name, " is on ", c.device(), ", the scene on ", ref.device());
    TORCH_CHECK(numel < 0 || c.numel() == numel, "graspnerf ops: ", name, " has ", c.numel(), " elements, expected ", numel);
    return c;
}

struct Scene {
    Tensor imgs, img_feats, ray_feats, poses, Ks, depth_range;
    GnrScene s;
};

Scene make_scene(const Tensor& imgs, const Tensor& img_feats, const Tensor& ray_feats, const Tensor& poses, const Tensor& Ks,
                 const Tensor& depth_range, bool use_vis) {
    Scene sc{f32(imgs, "imgs"), f32(img_feats, "img_feats"), f32(ray_feats, "ray_feats"), f32(poses, "poses"), f32(Ks, "Ks"),
             f32(depth_range, "depth_range"), {}};
    for (const Tensor* t : {&sc.img_feats, &sc.ray_feats, &sc.poses, &sc.Ks, &sc.depth_range})
        TORCH_CHECK(t->device() == sc.imgs.device(), "graspnerf ops: every scene tensor must be on ", sc.imgs.device(), " (found ", t->device(), ")");
    TORCH_CHECK(sc.imgs.dim() == 5 && sc.imgs.size(2) == 3 && sc.img_feats.dim() == 5 && sc.ray_feats.sizes() == sc.img_feats.sizes(), "graspnerf ops: imgs [B,V,3,H,W], feats [B,V,32,fh,fw]");
    const int B = (int)sc.imgs.size(0), V = (int)sc.imgs.size(1);
    TORCH_CHECK(sc.img_feats.size(0) == B && sc.img_feats.size(1) == V && sc.img_feats.size(2) == 32, "graspnerf ops: feature maps must be [B,V,32,fh,fw]");
    TORCH_CHECK(sc.poses.numel() == (int64_t)B * V * 12 && sc.Ks.numel() == (int64_t)B * V * 9 && sc.depth_range.numel() == (int64_t)B * V * 2, "graspnerf ops: poses [B,V,3,4], Ks [B,V,3,3], depth_range [B,V,2]");
    sc.s = GnrScene{B, V, (int)sc.imgs.size(3), (int)sc.imgs.size(4), (int)sc.img_feats.size(3), (int)sc.img_feats.size(4),
                    fp(sc.imgs), fp(sc.img_feats), fp(sc.ray_feats), fp(sc.poses), fp(sc.Ks), fp(sc.depth_range), use_vis ? 1 : 0};
    return sc;
}

Tensor bytes(size_t n, const Tensor& like) { return at::empty({(int64_t)n}, like.options().dtype(at::kByte)); }

Tensor sample_volume(const Tensor& imgs, const Tensor& img_feats, const Tensor& ray_feats, const Tensor& poses, const Tensor& Ks,
                     const Tensor& depth_range, const Tensor& bbox_min, const Tensor& weights, int64_t res, bool use_vis) {
    c10::DeviceGuard guard(imgs.device());
    Scene sc = make_scene(imgs, img_feats, ray_feats, poses, Ks, depth_range, use_vis);
    const Tensor bb = arg(bbox_min, "bbox_min [B,3]", sc.imgs, (int64_t)sc.s.B * 3), w = arg(weights, "weights (packed level blob)", sc.imgs, gnr_packed_weights_floats());
    const size_t wsb = gnr_workspace_bytes(&sc.s, (int)res, 0, 0);
    Tensor ws = bytes(wsb, sc.imgs);
    void* st = cur_stream(sc.imgs);
    ok(gnr_prepare(&sc.s, ws.data_ptr(), wsb, st), "gnr_prepare");
    Tensor vol = at::empty({sc.s.B, 1, res, res, res}, sc.imgs.options());
    ok(gnr_sample_volume_fwd(&sc.s, fp(bb), (int)res, fp(w), vol.data_ptr<float>(), nullptr, ws.data_ptr(), wsb, st), "gnr_sample_volume_fwd");
    return vol;
}

// outputs of one level, in this order (an undefined pixel_colors_gt when the query images are not given comes back as an empty tensor)
const char* const kRenderKeys[10] = {"depth", "sdf_values", "alpha_values", "colors_nr", "hit_prob_nr", "pixel_colors_nr", "pixel_colors_gt",
                                     "render_depth", "ray_mask", "sdf_gradient_error"};

std::vector<Tensor> alloc_level(GnrRenderOut& o, int B, int rn, int dn, int nch, bool with_gt, const Tensor& like) {
    auto f = like.options();
    std::vector<Tensor> t = {at::empty({B, rn, dn}, f), at::empty({B, rn, dn}, f), at::empty({B, rn, dn}, f), at::empty({B, rn, dn, 3}, f),
                             at::empty({B, rn, dn}, f), at::empty({B, rn, 3}, f), with_gt ? at::empty({B, rn, 3}, f) : at::empty({0}, f),
                             at::empty({B, rn}, f), at::empty({B, rn}, f.dtype(at::kByte)), at::empty({B, nch}, f)};
    o = GnrRenderOut{t[0].data_ptr<float>(), t[1].data_ptr<float>(), t[2].data_ptr<float>(), t[3].data_ptr<float>(), t[4].data_ptr<float>(),
                     t[5].data_ptr<float>(), with_gt ? t[6].data_ptr<float>() : nullptr, t[7].data_ptr<float>(), t[8].data_ptr<uint8_t>(),
                     t[9].data_ptr<float>(), nullptr, nullptr};
    return t;
}

std::vector<Tensor> render_rays(const Tensor& imgs, const Tensor& img_feats, const Tensor& ray_feats, const Tensor& poses, const Tensor& Ks,
                                const Tensor& depth_range, const Tensor& coords, const Tensor& que_pose, const Tensor& que_K,
                                const Tensor& que_depth_range, const c10::optional<Tensor>& que_imgs, const Tensor& weights_coarse,
                                const Tensor& weights_fine, int64_t dn, int64_t fdn, int64_t ray_mask_view_num, int64_t ray_mask_point_num,
                                int64_t ray_batch_num, bool fine_depth_use_all, bool use_vis) {
    c10::DeviceGuard guard(imgs.device());
    Scene sc = make_scene(imgs, img_feats, ray_feats, poses, Ks, depth_range, use_vis);
    const Tensor co = arg(coords, "coords", sc.imgs, -1);
    TORCH_CHECK(co.dim() == 3 && co.size(0) == sc.s.B && co.size(2) == 2, "graspnerf::render_rays: coords [B,rn,2]");
    const int B = sc.s.B, rn = (int)co.size(1);
    const Tensor qp = arg(que_pose, "que_pose [B,3,4]", sc.imgs, (int64_t)B * 12), qk = arg(que_K, "que_K [B,3,3]", sc.imgs, (int64_t)B * 9),
                 qd = arg(que_depth_range, "que_depth_range [B,2]", sc.imgs, (int64_t)B * 2);
    const Tensor wc = arg(weights_coarse, "weights_coarse (packed level blob)", sc.imgs, gnr_packed_weights_floats()),
                 wf = arg(weights_fine, "weights_fine (packed level blob)", sc.imgs, gnr_packed_weights_floats());
    Tensor qi;
    if (que_imgs.has_value() && que_imgs->defined()) qi = arg(*que_imgs, "que_imgs [B,3,H,W]", sc.imgs, (int64_t)B * 3 * sc.s.H * sc.s.W);
    GnrRays rays{rn, (int)dn, (int)fdn, (int)ray_mask_view_num, (int)ray_mask_point_num, fp(co), fp(qp), fp(qk), fp(qd),
                 qi.defined() ? fp(qi) : nullptr, nullptr, (int)ray_batch_num, fine_depth_use_all ? 1 : 0};
    const int fine_dn = fine_depth_use_all ? (int)(dn + fdn) : (int)fdn;
    const int nch = ray_batch_num > 0 ? (rn + (int)ray_batch_num - 1) / (int)ray_batch_num : 1;
    const size_t wsb = gnr_workspace_bytes(&sc.s, 1, rn, dn > fine_dn ? (int)dn : fine_dn);
    Tensor ws = bytes(wsb, sc.imgs);
    void* st = cur_stream(sc.imgs);
    ok(gnr_prepare(&sc.s, ws.data_ptr(), wsb, st), "gnr_prepare");
    GnrRenderOut oc, of;
    std::vector<Tensor> out = alloc_level(oc, B, rn, (int)dn, nch, qi.defined(), sc.imgs);
    std::vector<Tensor> fine = alloc_level(of, B, rn, fine_dn, nch, qi.defined(), sc.imgs);
    ok(gnr_render_rays_fwd(&sc.s, &rays, fp(wc), fp(wf), &oc, &of, nullptr, nullptr, ws.data_ptr(), wsb, st), "gnr_render_rays_fwd");
    out.insert(out.end(), fine.begin(), fine.end());
    return out;
}

std::tuple<Tensor, Tensor, Tensor> sample_volume_train(const Tensor& imgs, const Tensor& img_feats, const Tensor& ray_feats, const Tensor& poses,
                                                       const Tensor& Ks, const Tensor& depth_range, const Tensor& bbox_min, const Tensor& weights,
                                                       int64_t res, bool use_vis) {
    c10::DeviceGuard guard(imgs.device());
    Scene sc = make_scene(imgs, img_feats, ray_feats, poses, Ks, depth_range, use_vis);
    const Tensor bb = arg(bbox_min, "bbox_min [B,3]", sc.imgs, (int64_t)sc.s.B * 3), w = arg(weights, "weights (packed level blob)", sc.imgs, gnr_packed_weights_floats());
    const size_t wsb = gnr_workspace_bytes(&sc.s, (int)res, 0, 0), twb = gnr_sample_volume_train_workspace_bytes(&sc.s, (int)res);
    TORCH_CHECK(twb > 0, "graspnerf::sample_volume_train: bad volume resolution");
    Tensor ws = bytes(wsb, sc.imgs), tws = bytes(twb, sc.imgs);
    void* st = cur_stream(sc.imgs);
    ok(gnr_prepare(&sc.s, ws.data_ptr(), wsb, st), "gnr_prepare");
    Tensor vol = at::empty({sc.s.B, 1, res, res, res}, sc.imgs.options());
    ok(gnr_sample_volume_fwd_train(&sc.s, fp(bb), (int)res, fp(w), vol.data_ptr<float>(), ws.data_ptr(), wsb, tws.data_ptr(), twb, st),
       "gnr_sample_volume_fwd_train");
    return {vol, ws, tws};
}

std::tuple<Tensor, Tensor, Tensor> sample_volume_bwd(const Tensor& imgs, const Tensor& img_feats, const Tensor& ray_feats, const Tensor& poses,
                                                     const Tensor& Ks, const Tensor& depth_range, const Tensor& ws, const Tensor& tws,
                                                     const Tensor& dvol, const Tensor& weights, const Tensor& weights_bwd,
                                                     const Tensor& canonical, int64_t res, bool use_vis) {
    c10::DeviceGuard guard(imgs.device());
    Scene sc = make_scene(imgs, img_feats, ray_feats, poses, Ks, depth_range, use_vis);
    const int n = gnr_canonical_weights_floats() + (use_vis ? gnr_canonical_vis_floats() : 0);
    const Tensor dv = arg(dvol, "dvol [B,1,R,R,R]", sc.imgs, (int64_t)sc.s.B * res * res * res), w = arg(weights, "weights (packed level blob)", sc.imgs, gnr_packed_weights_floats()),
                 wb = arg(weights_bwd, "weights_bwd (gnr_pack_weights_bwd blob)", sc.imgs, gnr_packed_bwd_floats()),
                 can = arg(canonical, "canonical (the level's canonical blob)", sc.imgs, gnr_canonical_weights_floats());
    TORCH_CHECK(ws.is_cuda() && tws.is_cuda() && ws.scalar_type() == at::kByte && tws.scalar_type() == at::kByte && ws.is_contiguous() && tws.is_contiguous() &&
                ws.device() == sc.imgs.device() && tws.device() == sc.imgs.device(), "graspnerf::sample_volume_bwd: ws / tws are the byte tensors sample_volume_train returned");
    TORCH_CHECK((size_t)ws.numel() >= gnr_workspace_bytes(&sc.s, (int)res, 0, 0) && (size_t)tws.numel() >= gnr_sample_volume_train_workspace_bytes(&sc.s, (int)res),
                "graspnerf::sample_volume_bwd: ws / tws are smaller than this scene and resolution need");
    Tensor dcan = at::zeros({n}, sc.imgs.options());
    Tensor dray = at::empty_like(sc.ray_feats), dimg = at::empty_like(sc.img_feats);
    ok(gnr_sample_volume_bwd(&sc.s, (int)res, fp(w), fp(wb), fp(can), fp(dv), dcan.data_ptr<float>(), dray.data_ptr<float>(), dimg.data_ptr<float>(),
                             ws.data_ptr(), (size_t)ws.numel(), tws.data_ptr(), (size_t)tws.numel(), 0x1f, cur_stream(sc.imgs)),
       "gnr_sample_volume_bwd");
    return {dcan, dray, dimg};
}

}  // namespace

TORCH_LIBRARY(graspnerf, m) {
    m.def("sample_volume(Tensor imgs, Tensor img_feats, Tensor ray_feats, Tensor poses, Tensor Ks, Tensor depth_range, Tensor bbox_min, "
          "Tensor weights, int res, bool use_vis=False) -> Tensor");
    m.def("render_rays(Tensor imgs, Tensor img_feats, Tensor ray_feats, Tensor poses, Tensor Ks, Tensor depth_range, Tensor coords, "
          "Tensor que_pose, Tensor que_K, Tensor que_depth_range, Tensor? que_imgs, Tensor weights_coarse, Tensor weights_fine, int dn, "
          "int fdn, int ray_mask_view_num=2, int ray_mask_point_num=8, int ray_batch_num=0, bool fine_depth_use_all=False, "
          "bool use_vis=False) -> Tensor[]");
    m.def("sample_volume_train(Tensor imgs, Tensor img_feats, Tensor ray_feats, Tensor poses, Tensor Ks, Tensor depth_range, Tensor bbox_min, "
          "Tensor weights, int res, bool use_vis=False) -> (Tensor, Tensor, Tensor)");
    m.def("sample_volume_bwd(Tensor imgs, Tensor img_feats, Tensor ray_feats, Tensor poses, Tensor Ks, Tensor depth_range, Tensor ws, Tensor tws, "
          "Tensor dvol, Tensor weights, Tensor weights_bwd, Tensor canonical, int res, bool use_vis=False) -> (Tensor, Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(graspnerf, CUDA, m) {          // the ROCm build of PyTorch dispatches HIP tensors under the CUDA key
    m.impl("sample_volume", &sample_volume);
    m.impl("render_rays", &render_rays);
    m.impl("sample_volume_train", &sample_volume_train);
    m.impl("sample_volume_bwd", &sample_volume_bwd);
}

// Only the backend kernels above exist: an input that requires grad must not come back as an output that silently carries none.
// The Autograd key gets PyTorch's "not implemented" kernel: the forward runs, the outputs are marked, and a backward through them
// raises.  (Differentiating the path is the job of the autograd.Functions of graspnerf_amd/renderer.py, which pair
// sample_volume_train with sample_volume_bwd and hand the canonical-blob gradient to the parameters; the packed `weights` blob an
// operator sees is not a differentiable function of anything the dispatcher knows.)
TORCH_LIBRARY_IMPL(graspnerf, Autograd, m) {
    m.impl("sample_volume", torch::autograd::autogradNotImplementedFallback());
    m.impl("render_rays", torch::autograd::autogradNotImplementedFallback());
    m.impl("sample_volume_train", torch::autograd::autogradNotImplementedFallback());
    m.impl("sample_volume_bwd", torch::autograd::autogradNotImplementedFallback());
}
