// Micro-benchmark: does fp32-input MFMA (v_mfma_f32_16x16x4_f32) overlap with fp32 VALU work
//  (a) issued by the partner wave on the same SIMD, (b) interleaved inside one wave?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip ; run: ./mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// mode bit0: waves 0..3 run MFMA; bit1: waves 4..7 run VALU (partner waves on the same SIMDs)
__global__ __launch_bounds__(512) void k_pair(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w;
    } else {
        if (!(mode & 2)) return;
        float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
}

// one wave per SIMD (256 threads): per iteration 4 MFMAs and 4*K independent VALU fmas
template <int K>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = a + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m == 0) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            if (m == 1) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            if (m == 2) c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            if (m == 3) c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) x[j & 7] = fmaf(x[j & 7], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = c0.x + c1.y + c2.z + c3.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    for (int mode = 1; mode <= 3; ++mode) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_pair, dim3(256), dim3(512), 0, 0, out, iters, mode); });
        printf("pair mode %d (1=mfma waves only, 2=valu waves only, 3=both on the same SIMDs): %.3f ms\n", mode, ms);
    }
    printf("expected: mfma-only = iters*4*32 cyc = %.3f ms @2.4GHz ; valu-only = iters*32*~4 cyc\n", iters * 4 * 32 / 2.4e6);
#define MIX(K) { float ms = timeit([&] { hipLaunchKernelGGL(k_mix<K>, dim3(256), dim3(256), 0, 0, out, iters); }); \
                 printf("mix: 1 wave/SIMD, per MFMA %d fma: %.3f ms  (%.1f cyc per MFMA slot @2.4GHz)\n", K, ms, ms * 2.4e6 / (iters * 4)); }
    MIX(0) MIX(2) MIX(4) MIX(6) MIX(8) MIX(12) MIX(16)
    return 0;
}
