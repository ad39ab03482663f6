// Image-side streaming ops of the 2D feature extractors (reference: src/nr/network/ops.py:5-12,96-148,150-230 -- ResUNetLight
// and its blocks; init_net.py:8-35; vis_encoder.py:6-22), on the device for the training step and the full forward:
//   * gnr_instnorm_act(_bwd): nn.InstanceNorm2d(affine, no running stats) fused with what always follows it in these
//     networks -- an optional residual add and ReLU / ELU (ops.py:101-121,135-138,215) -- one read and one write of the
//     activation per direction instead of the 5 / 11 passes of the op-by-op graph;
//   * gnr_reflect_pad2d(_bwd): the F.pad(mode='reflect') that nn.Conv2d(padding_mode='reflect') runs in front of every
//     3x3 / 7x7 convolution (ops.py:8,134,163); the backward is a gather (<= 9 taps), not a scatter of atomics;
//   * gnr_upsample2x_bilinear: F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) of upconv (ops.py:147).
// All of it is HBM-bound float streaming: one workgroup per (image, channel) plane, float4 accesses, plane held in registers
// between the statistics and the apply pass; no LDS tiles, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gnr.h"

namespace gnr_img {

static thread_local char g_err[256] = "";
static int fail(int code, const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); return code; }
static int launched(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e)); return GNR_ERR_HIP; }
    return GNR_OK;
}

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum over the workgroup of up to two values at once; every thread receives the totals.  `red` holds 2 * 16 floats.
template <int T>
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    a = wave_sum(a); b = wave_sum(b);
    if constexpr (T > 64) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();                                   // previous use of `red` is over
        if (lane == 0) { red[wave] = a; red[16 + wave] = b; }
        __syncthreads();
        a = 0.f; b = 0.f;
#pragma unroll
        for (int w = 0; w < T / 64; ++w) { a += red[w]; b += red[16 + w]; }
    }
}

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == GNR_ACT_RELU) return fmaxf(v, 0.f);
    if (act == GNR_ACT_ELU) return v > 0.f ? v : expm1f(v);
    return v;
}
// derivative of the activation from its OUTPUT (ELU, alpha = 1: exp(x) = out + 1)
__device__ __forceinline__ float act_grad(float out, int act) {
    if (act == GNR_ACT_RELU) return out > 0.f ? 1.f : 0.f;
    if (act == GNR_ACT_ELU) return out > 0.f ? 1.f : out + 1.f;
    return 1.f;
}

struct InArgs {
    const float* x; const float* res; const float* weight; const float* bias;
    float* y; float* mean; float* rstd;
    int C; int HW; float eps; int act;
};

// ---- forward, plane in registers: HW = 4 * n4, n4 <= T * NV
template <int T, int NV>
__global__ __launch_bounds__(T) void k_in_fwd(InArgs a) {
    __shared__ float red[32];
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % (size_t)a.C), n4 = a.HW >> 2;
    const f4* xp = reinterpret_cast<const f4*>(a.x + plane * a.HW);
    f4 v[NV];
    float s = 0.f, dummy = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = i * T + threadIdx.x;
        v[i] = j < n4 ? xp[j] : f4{0.f, 0.f, 0.f, 0.f};
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    block_sum2<T>(s, dummy, red);
    const float inv_n = 1.f / (float)a.HW, mean = s * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = i * T + threadIdx.x;
        if (j < n4) {
            const f4 d = v[i] - mean;
            q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    }
    block_sum2<T>(q, dummy, red);
    const float rstd = 1.f / sqrtf(q * inv_n + a.eps);
    const float scale = rstd * a.weight[c], shift = a.bias[c] - mean * scale;
    f4* yp = reinterpret_cast<f4*>(a.y + plane * a.HW);
    const f4* rp = a.res ? reinterpret_cast<const f4*>(a.res + plane * a.HW) : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = i * T + threadIdx.x;
        if (j < n4) {
            f4 o = v[i] * scale + shift;
            if (rp) o += rp[j];
            o.x = act_fwd(o.x, a.act); o.y = act_fwd(o.y, a.act); o.z = act_fwd(o.z, a.act); o.w = act_fwd(o.w, a.act);
            yp[j] = o;
        }
    }
    if (threadIdx.x == 0) { a.mean[plane] = mean; a.rstd[plane] = rstd; }
}

// ---- forward, any plane size / alignment: three passes over the plane (the second and third hit the L2)
__global__ __launch_bounds__(256) void k_in_fwd_any(InArgs a) {
    __shared__ float red[32];
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % (size_t)a.C);
    const float* xp = a.x + plane * a.HW;
    float s = 0.f, dummy = 0.f;
    for (int j = threadIdx.x; j < a.HW; j += 256) s += xp[j];
    block_sum2<256>(s, dummy, red);
    const float inv_n = 1.f / (float)a.HW, mean = s * inv_n;
    float q = 0.f;
    for (int j = threadIdx.x; j < a.HW; j += 256) { const float d = xp[j] - mean; q += d * d; }
    block_sum2<256>(q, dummy, red);
    const float rstd = 1.f / sqrtf(q * inv_n + a.eps);
    const float scale = rstd * a.weight[c], shift = a.bias[c] - mean * scale;
    float* yp = a.y + plane * a.HW;
    const float* rp = a.res ? a.res + plane * a.HW : nullptr;
    for (int j = threadIdx.x; j < a.HW; j += 256) {
        float o = xp[j] * scale + shift;
        if (rp) o += rp[j];
        yp[j] = act_fwd(o, a.act);
    }
    if (threadIdx.x == 0) { a.mean[plane] = mean; a.rstd[plane] = rstd; }
}

struct InBwdArgs {
    const float* dy; const float* out; const float* x; const float* mean; const float* rstd; const float* weight;
    float* dx; float* dres; float* s1; float* s2;
    int C; int HW; int act;
};

// ---- backward: g = dy * act'(out);  dx = w rstd (g - <g> - xhat <g xhat>);  dres = g;  s1 = sum g, s2 = sum g xhat per plane
template <int T, int NV>
__global__ __launch_bounds__(T) void k_in_bwd(InBwdArgs a) {
    __shared__ float red[32];
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % (size_t)a.C), n4 = a.HW >> 2;
    const float mean = a.mean[plane], rstd = a.rstd[plane];
    const f4* gp = reinterpret_cast<const f4*>(a.dy + plane * a.HW);
    const f4* op = a.act != GNR_ACT_NONE ? reinterpret_cast<const f4*>(a.out + plane * a.HW) : nullptr;
    const f4* xp = reinterpret_cast<const f4*>(a.x + plane * a.HW);
    f4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = i * T + threadIdx.x;
        g[i] = f4{0.f, 0.f, 0.f, 0.f}; xh[i] = g[i];
        if (j < n4) {
            g[i] = gp[j];
            if (op) {
                const f4 o = op[j];
                g[i].x *= act_grad(o.x, a.act); g[i].y *= act_grad(o.y, a.act); g[i].z *= act_grad(o.z, a.act); g[i].w *= act_grad(o.w, a.act);
            }
            xh[i] = (xp[j] - mean) * rstd;
        }
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
    }
    block_sum2<T>(s1, s2, red);
    const float inv_n = 1.f / (float)a.HW, aw = a.weight[c] * rstd, m1 = s1 * inv_n, m2 = s2 * inv_n;
    f4* dxp = reinterpret_cast<f4*>(a.dx + plane * a.HW);
    f4* drp = a.dres ? reinterpret_cast<f4*>(a.dres + plane * a.HW) : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = i * T + threadIdx.x;
        if (j < n4) {
            dxp[j] = (g[i] - m1 - xh[i] * m2) * aw;
            if (drp) drp[j] = g[i];
        }
    }
    if (threadIdx.x == 0) { a.s1[plane] = s1; a.s2[plane] = s2; }
}

__global__ __launch_bounds__(256) void k_in_bwd_any(InBwdArgs a) {
    __shared__ float red[32];
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % (size_t)a.C);
    const float mean = a.mean[plane], rstd = a.rstd[plane];
    const float* gp = a.dy + plane * a.HW;
    const float* op = a.act != GNR_ACT_NONE ? a.out + plane * a.HW : nullptr;
    const float* xp = a.x + plane * a.HW;
    float s1 = 0.f, s2 = 0.f;
    for (int j = threadIdx.x; j < a.HW; j += 256) {
        const float g = gp[j] * (op ? act_grad(op[j], a.act) : 1.f);
        s1 += g; s2 += g * ((xp[j] - mean) * rstd);
    }
    block_sum2<256>(s1, s2, red);
    const float inv_n = 1.f / (float)a.HW, aw = a.weight[c] * rstd, m1 = s1 * inv_n, m2 = s2 * inv_n;
    float* dxp = a.dx + plane * a.HW;
    float* drp = a.dres ? a.dres + plane * a.HW : nullptr;
    for (int j = threadIdx.x; j < a.HW; j += 256) {
        const float g = gp[j] * (op ? act_grad(op[j], a.act) : 1.f);
        dxp[j] = (g - m1 - ((xp[j] - mean) * rstd) * m2) * aw;
        if (drp) drp[j] = g;
    }
    if (threadIdx.x == 0) { a.s1[plane] = s1; a.s2[plane] = s2; }
}

// d weight[c] = sum_n s2[n][c],  d bias[c] = sum_n s1[n][c]   (one thread per channel; N = images of the batch)
__global__ __launch_bounds__(64) void k_in_wb(const float* __restrict__ s1, const float* __restrict__ s2, float* __restrict__ dweight,
                                              float* __restrict__ dbias, int N, int C) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int n = 0; n < N; ++n) { a += s1[(size_t)n * C + c]; b += s2[(size_t)n * C + c]; }
    dbias[c] = a; dweight[c] = b;
}

// ---- reflect pad: y[oh][ow] = x[refl(oh - pad)][refl(ow - pad)],  refl(i) = i < 0 ? -i : (i >= n ? 2(n-1) - i : i)
__device__ __forceinline__ int refl(int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * (n - 1) - i : i; }

// one thread per 4 consecutive output pixels of a row (groups = ceil(OW / 4) per row)
__global__ __launch_bounds__(256) void k_reflect_pad(const float* __restrict__ x, float* __restrict__ y, int H, int W, int pad, unsigned groups,
                                                     unsigned total) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    const int OH = H + 2 * pad, OW = W + 2 * pad;
    const unsigned row = t / groups, q = t - row * groups;
    const unsigned plane = row / (unsigned)OH;
    const int oh = (int)(row - plane * (unsigned)OH), ow0 = 4 * (int)q;
    const float* xr = x + ((size_t)plane * H + refl(oh - pad, H)) * W;
    float* yr = y + (size_t)row * OW;
    if (ow0 >= pad && ow0 + 3 < W + pad) {                 // interior: a straight copy
        const float* xs = xr + (ow0 - pad);
        const float v0 = xs[0], v1 = xs[1], v2 = xs[2], v3 = xs[3];
        yr[ow0] = v0; yr[ow0 + 1] = v1; yr[ow0 + 2] = v2; yr[ow0 + 3] = v3;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ow0 + k < OW) yr[ow0 + k] = xr[refl(ow0 + k - pad, W)];
    }
}

// dx[h][w] = sum of dy over the (<= 3 x 3) padded positions that read x[h][w]; one thread per 4 consecutive pixels of a row
__global__ __launch_bounds__(256) void k_reflect_pad_bwd(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int pad, unsigned groups,
                                                         unsigned total) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    const int OW = W + 2 * pad, OH = H + 2 * pad;
    const unsigned row = t / groups, q = t - row * groups;
    const unsigned plane = row / (unsigned)H;
    const int h = (int)(row - plane * (unsigned)H), w0 = 4 * (int)q;
    int ohs[3], nh = 1;
    ohs[0] = h + pad;
    if (h >= 1 && h <= pad) ohs[nh++] = pad - h;
    if (h <= H - 2 && h >= H - 1 - pad) ohs[nh++] = pad + 2 * (H - 1) - h;
    const float* dp = dy + (size_t)plane * OH * OW;
    float* dr = dx + (size_t)row * W;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (w0 > pad && w0 + 3 < W - 1 - pad) {                // no mirrored column reads these four
        for (int i = 0; i < nh; ++i) {
            const float* r = dp + (size_t)ohs[i] * OW + w0 + pad;
            s[0] += r[0]; s[1] += r[1]; s[2] += r[2]; s[3] += r[3];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int w = w0 + k;
            if (w >= W) break;
            int ows[3], nw = 1;
            ows[0] = w + pad;
            if (w >= 1 && w <= pad) ows[nw++] = pad - w;
            if (w <= W - 2 && w >= W - 1 - pad) ows[nw++] = pad + 2 * (W - 1) - w;
            for (int i = 0; i < nh; ++i)
                for (int j = 0; j < nw; ++j) s[k] += dp[(size_t)ohs[i] * OW + ows[j]];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (w0 + k < W) dr[w0 + k] = s[k];
}

// ---- 2x bilinear upsampling, align_corners = True (the arithmetic of ATen's upsample_bilinear2d: source index = scale * dst,
// scale = (in - 1) / (out - 1); lambda1 = src - floor(src)); one thread per 4 consecutive output pixels of a row
__global__ __launch_bounds__(256) void k_upsample2x(const float* __restrict__ x, float* __restrict__ y, int H, int W, float sh, float sw, size_t total4) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int OW4 = W >> 1, OH = 2 * H;                    // 2W / 4 groups per row
    const int q = (int)(t % (size_t)OW4);
    const size_t row = t / (size_t)OW4;
    const int oh = (int)(row % (size_t)OH);
    const size_t plane = row / (size_t)OH;
    const float h1r = sh * (float)oh;
    const int h1 = min((int)h1r, H - 1), h1p = h1 < H - 1 ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const float* r0 = x + (plane * H + h1) * W;
    const float* r1 = r0 + h1p * W;
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ow = 4 * q + k;
        const float w1r = sw * (float)ow;
        const int w1 = min((int)w1r, W - 1), w1p = w1 < W - 1 ? 1 : 0;
        const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
        o[k] = h0l * (w0l * r0[w1] + w1l * r0[w1 + w1p]) + h1l * (w0l * r1[w1] + w1l * r1[w1 + w1p]);
    }
    reinterpret_cast<f4*>(y)[t] = f4{o[0], o[1], o[2], o[3]};
}

__global__ __launch_bounds__(256) void k_upsample2x_any(const float* __restrict__ x, float* __restrict__ y, int H, int W, float sh, float sw, size_t total) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int OW = 2 * W, OH = 2 * H;
    const int ow = (int)(t % (size_t)OW);
    const size_t row = t / (size_t)OW;
    const int oh = (int)(row % (size_t)OH);
    const size_t plane = row / (size_t)OH;
    const float h1r = sh * (float)oh, w1r = sw * (float)ow;
    const int h1 = min((int)h1r, H - 1), h1p = h1 < H - 1 ? 1 : 0, w1 = min((int)w1r, W - 1), w1p = w1 < W - 1 ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l, w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const float* r0 = x + (plane * H + h1) * W;
    const float* r1 = r0 + h1p * W;
    y[t] = h0l * (w0l * r0[w1] + w1l * r0[w1 + w1p]) + h1l * (w0l * r1[w1] + w1l * r1[w1 + w1p]);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace gnr_img

using namespace gnr_img;

extern "C" {

const char* gnr_img_last_error(void) { return g_err; }

int gnr_instnorm_act(const float* x, const float* res, const float* weight, const float* bias, float* y, float* mean, float* rstd,
                     long long planes, int C, int HW, float eps, int act, void* stream) {
    if (!x || !weight || !bias || !y || !mean || !rstd) return fail(GNR_ERR_ARG, "gnr_instnorm_act: null pointer");
    if (planes < 0 || C <= 0 || HW <= 0 || planes % C != 0 || act < GNR_ACT_NONE || act > GNR_ACT_ELU || planes > 0x7fffffffLL)
        return fail(GNR_ERR_SHAPE, "gnr_instnorm_act: planes must be a multiple of C > 0, HW > 0, act in {0,1,2}");
    if (planes == 0) return GNR_OK;
    hipStream_t st = (hipStream_t)stream;
    InArgs a{x, res, weight, bias, y, mean, rstd, C, HW, eps, act};
    const bool vec = HW % 4 == 0 && aligned16(x) && aligned16(y) && (!res || aligned16(res));
    const int n4 = HW / 4;
    const dim3 grid((unsigned)planes);
    if (vec && n4 <= 64 * 3) k_in_fwd<64, 3><<<grid, 64, 0, st>>>(a);
    else if (vec && n4 <= 256 * 3) k_in_fwd<256, 3><<<grid, 256, 0, st>>>(a);
    else if (vec && n4 <= 256 * 9) k_in_fwd<256, 9><<<grid, 256, 0, st>>>(a);
    else if (vec && n4 <= 1024 * 9) k_in_fwd<1024, 9><<<grid, 1024, 0, st>>>(a);
    else k_in_fwd_any<<<grid, 256, 0, st>>>(a);
    return launched("gnr_instnorm_act");
}

int gnr_instnorm_act_bwd(const float* dy, const float* out, const float* x, const float* mean, const float* rstd, const float* weight,
                         float* dx, float* dres, float* s1, float* s2, float* dweight, float* dbias, long long planes, int C, int HW, int act,
                         void* stream) {
    if (!dy || !x || !mean || !rstd || !weight || !dx || !s1 || !s2 || !dweight || !dbias || (act != GNR_ACT_NONE && !out))
        return fail(GNR_ERR_ARG, "gnr_instnorm_act_bwd: null pointer");
    if (planes < 0 || C <= 0 || HW <= 0 || planes % C != 0 || act < GNR_ACT_NONE || act > GNR_ACT_ELU || planes > 0x7fffffffLL)
        return fail(GNR_ERR_SHAPE, "gnr_instnorm_act_bwd: planes must be a multiple of C > 0, HW > 0, act in {0,1,2}");
    hipStream_t st = (hipStream_t)stream;
    if (planes == 0) return hipMemsetAsync(dweight, 0, sizeof(float) * C, st) == hipSuccess && hipMemsetAsync(dbias, 0, sizeof(float) * C, st) == hipSuccess
                                ? GNR_OK : fail(GNR_ERR_HIP, "gnr_instnorm_act_bwd: hipMemsetAsync");
    InBwdArgs a{dy, out, x, mean, rstd, weight, dx, dres, s1, s2, C, HW, act};
    const bool vec = HW % 4 == 0 && aligned16(dy) && aligned16(x) && aligned16(dx) && (!out || aligned16(out)) && (!dres || aligned16(dres));
    const int n4 = HW / 4;
    const dim3 grid((unsigned)planes);
    if (vec && n4 <= 64 * 3) k_in_bwd<64, 3><<<grid, 64, 0, st>>>(a);
    else if (vec && n4 <= 256 * 3) k_in_bwd<256, 3><<<grid, 256, 0, st>>>(a);
    else if (vec && n4 <= 256 * 9) k_in_bwd<256, 9><<<grid, 256, 0, st>>>(a);
    else if (vec && n4 <= 1024 * 9) k_in_bwd<1024, 9><<<grid, 1024, 0, st>>>(a);
    else k_in_bwd_any<<<grid, 256, 0, st>>>(a);
    k_in_wb<<<dim3((unsigned)(C + 63) / 64), 64, 0, st>>>(s1, s2, dweight, dbias, (int)(planes / C), C);
    return launched("gnr_instnorm_act_bwd");
}

static int pad_args(const void* a, const void* b, long long planes, int H, int W, int pad, const char* who) {
    if (!a || !b) return fail(GNR_ERR_ARG, who);
    if (planes < 0 || H <= 0 || W <= 0 || pad < 0 || pad >= H || pad >= W) return fail(GNR_ERR_SHAPE, who);    // F.pad: pad < size
    return GNR_OK;
}

int gnr_reflect_pad2d(const float* x, float* y, long long planes, int H, int W, int pad, void* stream) {
    if (const int rc = pad_args(x, y, planes, H, W, pad, "gnr_reflect_pad2d: null pointer or pad outside [0, min(H, W))")) return rc;
    const unsigned groups = (unsigned)(W + 2 * pad + 3) / 4;
    const unsigned long long total = (unsigned long long)planes * (H + 2 * pad) * groups;
    if (total > 0x7fffffffULL) return fail(GNR_ERR_SHAPE, "gnr_reflect_pad2d: too many pixels for one launch");
    if (total == 0) return GNR_OK;
    k_reflect_pad<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, y, H, W, pad, groups, (unsigned)total);
    return launched("gnr_reflect_pad2d");
}

int gnr_reflect_pad2d_bwd(const float* dy, float* dx, long long planes, int H, int W, int pad, void* stream) {
    if (const int rc = pad_args(dy, dx, planes, H, W, pad, "gnr_reflect_pad2d_bwd: null pointer or pad outside [0, min(H, W))")) return rc;
    const unsigned groups = (unsigned)(W + 3) / 4;
    const unsigned long long total = (unsigned long long)planes * H * groups;
    if (total > 0x7fffffffULL) return fail(GNR_ERR_SHAPE, "gnr_reflect_pad2d_bwd: too many pixels for one launch");
    if (total == 0) return GNR_OK;
    k_reflect_pad_bwd<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(dy, dx, H, W, pad, groups, (unsigned)total);
    return launched("gnr_reflect_pad2d_bwd");
}

int gnr_upsample2x_bilinear(const float* x, float* y, long long planes, int H, int W, void* stream) {
    if (!x || !y) return fail(GNR_ERR_ARG, "gnr_upsample2x_bilinear: null pointer");
    if (planes < 0 || H <= 0 || W <= 0) return fail(GNR_ERR_SHAPE, "gnr_upsample2x_bilinear: H, W > 0");
    if (planes == 0) return GNR_OK;
    const float sh = (float)(H - 1) / (float)(2 * H - 1), sw = (float)(W - 1) / (float)(2 * W - 1);
    const unsigned long long total = (unsigned long long)planes * 2 * H * 2 * W;
    hipStream_t st = (hipStream_t)stream;
    if (W % 2 == 0 && aligned16(y)) {
        const unsigned long long blocks = (total / 4 + 255) / 256;
        if (blocks > 0x7fffffffULL) return fail(GNR_ERR_SHAPE, "gnr_upsample2x_bilinear: too large for one launch");
        k_upsample2x<<<dim3((unsigned)blocks), 256, 0, st>>>(x, y, H, W, sh, sw, (size_t)(total / 4));
    } else {
        const unsigned long long blocks = (total + 255) / 256;
        if (blocks > 0x7fffffffULL) return fail(GNR_ERR_SHAPE, "gnr_upsample2x_bilinear: too large for one launch");
        k_upsample2x_any<<<dim3((unsigned)blocks), 256, 0, st>>>(x, y, H, W, sh, sw, (size_t)total);
    }
    return launched("gnr_upsample2x_bilinear");
}

}  // extern "C"
