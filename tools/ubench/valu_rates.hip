// Micro-benchmark: issue cost of scalar f32 FMA vs packed v_pk_fma_f32 vs v_exp_f32 vs ds_read broadcast on gfx950,
// one / two waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = a + j;
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {            // 16 scalar fma
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(b), "v"(a));
        } else if constexpr (MODE == 1) {     // 8 packed fma (same 16 flop-pairs)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f2 v = {x[2 * j], x[2 * j + 1]};
                f2 bb = {b, b}, aa = {a, a};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(bb), "v"(aa));
                x[2 * j] = v.x; x[2 * j + 1] = v.y;
            }
        } else if constexpr (MODE == 2) {     // 16 exp
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
        } else if constexpr (MODE == 3) {     // 16 mov
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(x[j]) : "v"(x[(j + 1) & 15]));
        } else if constexpr (MODE == 4) {     // 16 med3
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(b), "v"(a));
        } else if constexpr (MODE == 5) {     // 16 rcp
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
    const char* names[] = {"v_fma_f32 x16", "v_pk_fma_f32 x8", "v_exp_f32 x16", "v_mov_b32 x16", "v_med3_f32 x16", "v_rcp_f32 x16"};
    for (int threads : {256, 512, 1024}) {
        printf("-- %d threads per CU (%d waves/SIMD), 256 blocks\n", threads, threads / 256);
        float t[6];
        t[0] = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, iters); });
        t[1] = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, iters); });
        t[2] = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, iters); });
        t[3] = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, iters); });
        t[4] = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, iters); });
        t[5] = timeit([&] { hipLaunchKernelGGL(k<5>, dim3(256), dim3(threads), 0, 0, out, iters); });
        for (int m = 0; m < 6; ++m) {
            const double insts = (m == 1 ? 8.0 : 16.0) * iters * (threads / 256);      // per SIMD
            printf("   %-18s %.3f ms  -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", names[m], t[m], t[m] * 1e-3 * 2.4e9 / insts);
        }
    }
    return 0;
}
