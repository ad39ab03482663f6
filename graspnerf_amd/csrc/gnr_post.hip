// Grasp post-processing on the device: the reference planner's process() + select()  (src/nr/main.py:23-84), which
// call scipy.ndimage on the host for every plan.  64 000 voxels per scene: byte/float streaming work, one thread per
// voxel, no MFMA.  Bit-exact with scipy: the Gaussian accumulates in fp64 in scipy's order and rounds to fp32 after
// every axis (ni_filters.c, symmetric branch); dilation and the max filter are exact by nature.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gnr.h"

namespace gnr_post {

static thread_local char g_err[256] = "";
static int fail(int code, const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); return code; }

struct GaussW { double w[GNR_GAUSS_MAX_RADIUS + 1]; int radius; };

// one axis of gaussian_filter(mode='nearest'): out = fp32( w0*in[i] + sum_{k=r..1} (in[i-k] + in[i+k]) * w[k] ), fp64 inside
__global__ void k_gauss_axis(const float* __restrict__ in, float* __restrict__ out, int R, int stride, GaussW g, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i = (int)((t / (size_t)stride) % (size_t)R);
    const float* p = in + t;
    double acc = (double)p[0] * g.w[0];
    for (int k = g.radius; k >= 1; --k) {
        const int lo = max(i - k, 0) - i, hi = min(i + k, R - 1) - i;
        acc += ((double)p[(ptrdiff_t)lo * stride] + (double)p[(ptrdiff_t)hi * stride]) * g.w[k];
    }
    out[t] = (float)acc;
}

// outside = tsdf > high ; may_change = !(low < tsdf < high)      (main.py:45-49)
__global__ void k_masks(const float* __restrict__ tsdf, unsigned char* __restrict__ x, unsigned char* __restrict__ m,
                        float high, float low, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float v = tsdf[t];
    x[t] = v > high ? 1 : 0;
    m[t] = (low < v && v < high) ? 0 : 1;
}

// one iteration of binary_dilation(structure = 6-neighbourhood, mask, border_value=0)
__global__ void k_dilate(const unsigned char* __restrict__ x, const unsigned char* __restrict__ m, unsigned char* __restrict__ y,
                         int R, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int k = (int)(t % R), j = (int)((t / R) % R), i = (int)((t / ((size_t)R * R)) % R);
    unsigned char v = x[t];
    if (m[t] && !v) {
        const int R2 = R * R;
        v = (i > 0 && x[t - R2]) || (i + 1 < R && x[t + R2]) || (j > 0 && x[t - R]) || (j + 1 < R && x[t + R]) ||
            (k > 0 && x[t - 1]) || (k + 1 < R && x[t + 1]);
    }
    y[t] = v;
}

// processed quality (main.py:50-55) and its thresholded copy (main.py:61)
__global__ void k_finalize(const float* __restrict__ qs, const unsigned char* __restrict__ valid, const float* __restrict__ width,
                           float* __restrict__ qual_out, float* __restrict__ qthr, float min_w, float max_w, float thr, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float w = width[t];
    float q = qs[t];
    if (!valid[t] || w < min_w || w > max_w) q = 0.f;
    qual_out[t] = q;
    qthr[t] = q < thr ? 0.f : q;
}

// non-maximum suppression: keep q where q == maximum_filter(q, size, mode='reflect') and q != 0   (main.py:64-68)
__global__ void k_nms(const float* __restrict__ q, unsigned char* __restrict__ keep, int R, int size, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int k = (int)(t % R), j = (int)((t / R) % R), i = (int)((t / ((size_t)R * R)) % R);
    const float* vol = q + (t - ((size_t)i * R + j) * R - k);
    const float c = q[t];
    if (c == 0.f) { keep[t] = 0; return; }
    const int left = size / 2, right = size - size / 2 - 1;
    auto refl = [R](int a) { return a < 0 ? -a - 1 : (a >= R ? 2 * R - 1 - a : a); };
    float m = c;
    for (int di = -left; di <= right; ++di) {
        const int ii = refl(i + di);
        for (int dj = -left; dj <= right; ++dj) {
            const int jj = refl(j + dj);
            const float* row = vol + ((size_t)ii * R + jj) * R;
            for (int dk = -left; dk <= right; ++dk) m = fmaxf(m, row[refl(k + dk)]);
        }
    }
    keep[t] = (c == m) ? 1 : 0;
}

// ordered compaction (np.argwhere order = ascending linear index), one workgroup per volume   (main.py:70-77)
__global__ __launch_bounds__(1024) void k_compact(const unsigned char* __restrict__ keep, const float* __restrict__ q,
                                                  const float* __restrict__ rot, const float* __restrict__ width, int R, int max_n,
                                                  int* __restrict__ count, int* __restrict__ index, float* __restrict__ score,
                                                  float* __restrict__ quat, float* __restrict__ width_out) {
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n = (size_t)R * R * R;
    const unsigned char* kb = keep + (size_t)b * n;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (size_t c0 = 0; c0 < n; c0 += 1024) {
        const size_t t = c0 + threadIdx.x;
        const bool f = t < n && kb[t];
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        const int pos = off + before;
        if (f && pos < max_n) {
            const int k = (int)(t % R), j = (int)((t / R) % R), i = (int)(t / ((size_t)R * R));
            const size_t o = (size_t)b * max_n + pos;
            index[o * 3] = i; index[o * 3 + 1] = j; index[o * 3 + 2] = k;
            score[o] = q[(size_t)b * n + t];
            for (int c = 0; c < 4; ++c) quat[o * 4 + c] = rot[((size_t)b * 4 + c) * n + t];
            width_out[o] = width[(size_t)b * n + t];
        }
        __syncthreads();
        if (threadIdx.x == 0) { int s = base_s; for (int w = 0; w < 16; ++w) s += wave_tot[w]; base_s = s; }
        __syncthreads();
    }
    if (threadIdx.x == 0) count[b] = base_s;
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace gnr_post

using namespace gnr_post;

extern "C" const char* gnr_post_last_error(void) { return g_err; }

extern "C" size_t gnr_grasp_select_workspace_bytes(int B, int R) {
    if (B < 1 || R < 1) return 0;
    const size_t n = (size_t)B * R * R * R;
    return 2 * al256(n * sizeof(float)) + 3 * al256(n);
}

extern "C" int gnr_grasp_select_fwd(const float* tsdf, const float* qual, const float* rot, const float* width, int B, int R,
                                    const GnrSelectParams* p, float* qual_out, int* count, int* index, float* score, float* quat,
                                    float* width_out, int max_n, void* ws, size_t ws_bytes, void* stream) {
    if (!tsdf || !qual || !rot || !width || !p || !qual_out || !count || !index || !score || !quat || !width_out || !ws)
        return fail(GNR_ERR_ARG, "gnr_grasp_select_fwd: null pointer");
    if (B < 1 || R < 2 || R > 256 || max_n < 1) return fail(GNR_ERR_SHAPE, "gnr_grasp_select_fwd: bad B / R / max_n");
    if (p->gauss_radius < 0 || p->gauss_radius > GNR_GAUSS_MAX_RADIUS || p->dilate_iterations < 0 || p->max_filter_size < 1 ||
        p->max_filter_size > 16)
        return fail(GNR_ERR_ARG, "gnr_grasp_select_fwd: bad filter parameters");
    if (ws_bytes < gnr_grasp_select_workspace_bytes(B, R)) return fail(GNR_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * R * R * R;
    char* base = (char*)ws;
    float* fa = (float*)base;                      base += al256(n * sizeof(float));
    float* fb = (float*)base;                      base += al256(n * sizeof(float));
    unsigned char* xa = (unsigned char*)base;      base += al256(n);
    unsigned char* xb = (unsigned char*)base;      base += al256(n);
    unsigned char* mk = (unsigned char*)base;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    GaussW g;
    g.radius = p->gauss_radius;
    for (int k = 0; k <= GNR_GAUSS_MAX_RADIUS; ++k) g.w[k] = k <= p->gauss_radius ? p->gauss_w[k] : 0.0;
    // Gaussian, axes in scipy's order (0, 1, 2), fp32 rounding after each
    hipLaunchKernelGGL(k_gauss_axis, dim3(blocks), dim3(256), 0, st, qual, fa, R, R * R, g, n);
    hipLaunchKernelGGL(k_gauss_axis, dim3(blocks), dim3(256), 0, st, fa, fb, R, R, g, n);
    hipLaunchKernelGGL(k_gauss_axis, dim3(blocks), dim3(256), 0, st, fb, fa, R, 1, g, n);
    hipLaunchKernelGGL(k_masks, dim3(blocks), dim3(256), 0, st, tsdf, xa, mk, p->tsdf_thres_high, p->tsdf_thres_low, n);
    unsigned char *xin = xa, *xout = xb;
    for (int it = 0; it < p->dilate_iterations; ++it) {
        hipLaunchKernelGGL(k_dilate, dim3(blocks), dim3(256), 0, st, xin, mk, xout, R, n);
        unsigned char* tmp = xin; xin = xout; xout = tmp;
    }
    hipLaunchKernelGGL(k_finalize, dim3(blocks), dim3(256), 0, st, fa, xin, width, qual_out, fb, p->min_width, p->max_width,
                       p->threshold, n);
    hipLaunchKernelGGL(k_nms, dim3(blocks), dim3(256), 0, st, fb, xout, R, p->max_filter_size, n);
    hipLaunchKernelGGL(k_compact, dim3(B), dim3(1024), 0, st, xout, fb, rot, width, R, max_n, count, index, score, quat, width_out);
    if (hipGetLastError() != hipSuccess) return fail(GNR_ERR_HIP, "gnr_grasp_select_fwd: launch failed");
    return GNR_OK;
}
