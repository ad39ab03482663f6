// Micro-benchmarks behind two design questions of k_chain, gfx950:
//  (1) what a register-to-register move costs in its wider forms (v_mov_b64, v_pk_mov_b32, v_swap_b32, v_accvgpr_*) next to
//      v_mov_b32 -- the per-view state of k_chain is a queue of registers that advances by moves;
//  (2) how many single-issue VALU fillers hide in the shadow of an f16 MFMA, for the 16x16x32 and the 32x32x16 shape
//      (MI355X_MICROARCH.md: <= 5 per 32x32x16 gap at one wave per SIMD), at one and two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mov_rates mov_rates.hip ; run: ./mov_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void k_mov(float* out, int iters) {
    float a = threadIdx.x * 1e-3f;
    float x[16];
    f2 y[8];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = a + j;
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = (f2){a + j, a - j};
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(x[j]) : "v"(x[(j + 1) & 15]));
        } else if constexpr (MODE == 1) {       // 8 x 64-bit moves = the same 16 dwords
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mov_b64 %0, %1" : "=v"(y[j]) : "v"(y[(j + 1) & 7]));
        } else if constexpr (MODE == 2) {       // 8 x v_pk_mov_b32 (two independent dwords per instruction)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(y[j]) : "v"(y[(j + 1) & 7]), "v"(y[(j + 2) & 7]));
        } else if constexpr (MODE == 3) {       // 8 x v_swap_b32 (16 dwords change place)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_swap_b32 %0, %1" : "+v"(x[2 * j]), "+v"(x[2 * j + 1]));
        } else if constexpr (MODE == 4) {       // 16 x v_accvgpr_write + 16 x v_accvgpr_read
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_accvgpr_write_b32 a%1, %0" :: "v"(x[j]), "n"(j));
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(x[j]) : "n"((j + 1) & 15));
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += x[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += y[j].x + y[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// SHAPE 0: v_mfma_f32_16x16x32_f16, 1: v_mfma_f32_32x32x16_f16; K independent v_fma between consecutive MFMAs (4 accumulators round robin)
template <int SHAPE, int K, int THREADS>
__global__ __launch_bounds__(THREADS) void k_mix(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b + i); }
    f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16v d[2];
    for (int i = 0; i < 16; ++i) { d[0][i] = 0.f; d[1][i] = 0.f; }
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = a + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (SHAPE == 0) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[m], 0, 0, 0);
            else d[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, d[m & 1], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(b), "v"(a));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = c[0].x + c[1].y + c[2].z + c[3].w + d[0][3] + d[1][5];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
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
    const char* names[] = {"v_mov_b32 x16", "v_mov_b64 x8", "v_pk_mov_b32 x8", "v_swap_b32 x8", "accvgpr write+read x16"};
    const double insts[] = {16, 8, 8, 8, 32};
    for (int threads : {256, 512}) {
        printf("-- moves of 16 dwords, %d waves/SIMD\n", threads / 256);
        float t[5];
        t[0] = timeit([&] { hipLaunchKernelGGL(k_mov<0>, dim3(256), dim3(threads), 0, 0, out, iters); });
        t[1] = timeit([&] { hipLaunchKernelGGL(k_mov<1>, dim3(256), dim3(threads), 0, 0, out, iters); });
        t[2] = timeit([&] { hipLaunchKernelGGL(k_mov<2>, dim3(256), dim3(threads), 0, 0, out, iters); });
        t[3] = timeit([&] { hipLaunchKernelGGL(k_mov<3>, dim3(256), dim3(threads), 0, 0, out, iters); });
        t[4] = timeit([&] { hipLaunchKernelGGL(k_mov<4>, dim3(256), dim3(threads), 0, 0, out, iters); });
        for (int m = 0; m < 5; ++m) {
            const double cyc = t[m] * 1e-3 * 2.4e9 / (iters * (threads / 256));       // SIMD cycles per 16-dword group
            printf("   %-24s %.3f ms  -> %.2f cycles per instruction, %.2f per dword moved (2.4 GHz)\n", names[m], t[m], cyc / insts[m], cyc / 16);
        }
    }
#define MIX(S, K, T) { float ms = timeit([&] { hipLaunchKernelGGL((k_mix<S, K, T>), dim3(256), dim3(T), 0, 0, out, iters); }); \
                 printf("   %s, %d wave(s)/SIMD, %2d v_fma per MFMA: %.3f ms = %.1f SIMD cycles per (MFMA + fillers)\n", S ? "32x32x16 f16" : "16x16x32 f16", T / 256, K, ms, ms * 1e-3 * 2.4e9 / (iters * 4.0 * (T / 256))); }
    printf("-- VALU fillers in the shadow of an f16 MFMA\n");
    MIX(0, 0, 256) MIX(0, 2, 256) MIX(0, 4, 256) MIX(0, 8, 256)
    MIX(0, 0, 512) MIX(0, 2, 512) MIX(0, 4, 512) MIX(0, 8, 512) MIX(0, 12, 512)
    MIX(1, 0, 256) MIX(1, 3, 256) MIX(1, 5, 256) MIX(1, 8, 256) MIX(1, 16, 256)
    MIX(1, 0, 512) MIX(1, 3, 512) MIX(1, 5, 512) MIX(1, 8, 512) MIX(1, 16, 512) MIX(1, 24, 512)
    return 0;
}
