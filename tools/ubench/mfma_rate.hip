// Micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 (and 32x32x2) per SIMD with 1 / 2 waves per SIMD and
// 2..8 independent accumulators.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    f4 c[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) c[j] = f4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += c[j].x + c[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k32(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    f16v c0 = {}, c1 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1];
}
template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    for (int threads : {256, 512}) {
        const int wps = threads / 256;
        float t2 = timeit([&] { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(threads), 0, 0, out, iters * 4); });
        float t4 = timeit([&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(threads), 0, 0, out, iters * 2); });
        float t8 = timeit([&] { hipLaunchKernelGGL(k16<8>, dim3(256), dim3(threads), 0, 0, out, iters); });
        float t32 = timeit([&] { hipLaunchKernelGGL(k32, dim3(256), dim3(threads), 0, 0, out, iters * 2); });
        const double n16 = 8.0 * iters * wps, n32 = 4.0 * iters * wps;
        printf("%d waves/SIMD: 16x16x4 f32, 2/4/8 accumulators: %.2f / %.2f / %.2f cycles per MFMA per SIMD @2.4GHz (%.1f / %.1f / %.1f TFLOP/s chip);  32x32x2: %.2f cycles (%.1f TFLOP/s)\n",
               wps, t2 * 1e-3 * 2.4e9 / n16, t4 * 1e-3 * 2.4e9 / n16, t8 * 1e-3 * 2.4e9 / n16,
               n16 * 1024 * 2048 / (t2 * 1e-3) / 1e12, n16 * 1024 * 2048 / (t4 * 1e-3) / 1e12, n16 * 1024 * 2048 / (t8 * 1e-3) / 1e12,
               t32 * 1e-3 * 2.4e9 / n32, n32 * 1024 * 4096 / (t32 * 1e-3) / 1e12);
    }
    return 0;
}
