// Device-side weight packer: the canonical blobs of a level (parameters gathered on the device) -> the packed forward blob
// (CHAIN / RAY sections + the fp16-pair image k_chain stages into LDS) and the transposed-fragment blob of the backward twins,
// without a device-to-host copy, a host-side pack and an upload per optimiser step (reference: the parameters live on the device,
// /root/reference/src/nr/train/trainer.py:146-158; renderer.py:14-50 nn.Modules).
// The arithmetic is gnr_pack_body.h -- the source the host packer gnr_pack.cpp instantiates -- run with grid-strided loops; the
// blobs equal the host packer's bit for bit (tests/test_pack_device.py).  The blobs are re-packed IN PLACE: the constant parts of
// a blob (structural zeros, the sinusoid position table R_PE, which the host computes with libm) are kept, so `packed_dev` must
// once have been filled from a blob of gnr_pack_weights / gnr_pack_weights_bwd (any weights).
#include <hip/hip_runtime.h>

#define GNR_HD __host__ __device__
#include "gnr_pack_body.h"
#include "../../include/gnr.h"

namespace gnr {
namespace {
__device__ inline packer::DevExec dev_exec() {
    return packer::DevExec{(int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x)};
}
__global__ __launch_bounds__(256) void k_pack_forward(const float* __restrict__ c, float* __restrict__ p) { packer::pack_forward(dev_exec(), c, p); }
__global__ __launch_bounds__(256) void k_pack_vis(const float* __restrict__ v, float* __restrict__ p) { packer::pack_vis(dev_exec(), v, p); }
__global__ __launch_bounds__(256) void k_pack_c16_copy(float* p) { packer::c16_copy_runs(dev_exec(), p); }
__global__ __launch_bounds__(256) void k_pack_c16_pairs(float* p) { packer::c16_pairs(dev_exec(), p); }
__global__ __launch_bounds__(256) void k_pack_backward(const float* __restrict__ c, float* __restrict__ p) { packer::pack_backward(dev_exec(), c, p); }
__global__ __launch_bounds__(256) void k_pack_vis_backward(const float* __restrict__ v, float* __restrict__ p) { packer::pack_vis_backward(dev_exec(), v, p); }
__global__ __launch_bounds__(256) void k_pack_backward_pairs(float* p) { packer::pack_backward_pairs(dev_exec(), p); }
__global__ __launch_bounds__(256) void k_pack_vis_backward_pairs(float* p) { packer::pack_vis_backward_pairs(dev_exec(), p); }

constexpr int PACK_BLOCKS = 64, PACK_THREADS = 256;       // 16 384 threads: the largest fragment (HOIST, 9 216 entries) in one sweep
int c16_image(float* p, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_c16_copy, dim3(PACK_BLOCKS), dim3(PACK_THREADS), 0, st, p);
    hipLaunchKernelGGL(k_pack_c16_pairs, dim3(PACK_BLOCKS), dim3(PACK_THREADS), 0, st, p);
    return hipGetLastError() == hipSuccess ? GNR_OK : GNR_ERR_HIP;
}
}  // namespace
}  // namespace gnr

extern "C" int gnr_pack_weights_device(const float* canonical_dev, float* packed_dev, void* stream) {
    if (!canonical_dev || !packed_dev) return GNR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gnr::k_pack_forward, dim3(gnr::PACK_BLOCKS), dim3(gnr::PACK_THREADS), 0, st, canonical_dev, packed_dev);
    return gnr::c16_image(packed_dev, st);
}

extern "C" int gnr_pack_vis_decoder_device(const float* vis_decoder_dev, float* packed_dev, void* stream) {
    if (!vis_decoder_dev || !packed_dev) return GNR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gnr::k_pack_vis, dim3(gnr::PACK_BLOCKS), dim3(gnr::PACK_THREADS), 0, st, vis_decoder_dev, packed_dev);
    return gnr::c16_image(packed_dev, st);
}

extern "C" int gnr_pack_weights_bwd_device(const float* canonical_dev, float* packed_bwd_dev, void* stream) {
    if (!canonical_dev || !packed_bwd_dev) return GNR_ERR_ARG;
    hipLaunchKernelGGL(gnr::k_pack_backward, dim3(gnr::PACK_BLOCKS), dim3(gnr::PACK_THREADS), 0, (hipStream_t)stream, canonical_dev, packed_bwd_dev);
    hipLaunchKernelGGL(gnr::k_pack_backward_pairs, dim3(gnr::PACK_BLOCKS), dim3(gnr::PACK_THREADS), 0, (hipStream_t)stream, packed_bwd_dev);
    return hipGetLastError() == hipSuccess ? GNR_OK : GNR_ERR_HIP;
}

extern "C" int gnr_pack_vis_decoder_bwd_device(const float* vis_decoder_dev, float* packed_bwd_dev, void* stream) {
    if (!vis_decoder_dev || !packed_bwd_dev) return GNR_ERR_ARG;
    hipLaunchKernelGGL(gnr::k_pack_vis_backward, dim3(gnr::PACK_BLOCKS), dim3(gnr::PACK_THREADS), 0, (hipStream_t)stream, vis_decoder_dev, packed_bwd_dev);
    hipLaunchKernelGGL(gnr::k_pack_vis_backward_pairs, dim3(gnr::PACK_BLOCKS), dim3(gnr::PACK_THREADS), 0, (hipStream_t)stream, packed_bwd_dev);
    return hipGetLastError() == hipSuccess ? GNR_OK : GNR_ERR_HIP;
}
