// Probe: do the sticky IEEE exception flags of a wavefront (TRAPSTS.EXCP) record an fp32 -> fp16 conversion that overflows
// to infinity (and NaN inputs), without any trap handler / exception enable?  If so, k_chain's fp16-pair form can detect an
// activation beyond the fp16 range at zero VALU cost (one s_getreg per tile).
// Build: hipcc --offload-arch=gfx950 -O3 -o trapsts_probe trapsts_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out) {
    const unsigned before = __builtin_amdgcn_s_getreg(3 | (31 << 11));       // HW_REG_TRAPSTS, all 32 bits
    const unsigned mode = __builtin_amdgcn_s_getreg(1 | (31 << 11));         // HW_REG_MODE
    f2 x = {in[threadIdx.x], in[threadIdx.x + 64]};
    h2 h = __builtin_convertvector(x, h2);
    float y = (float)h.x + (float)h.y;
    asm volatile("s_nop 7\n s_nop 7" ::: "memory");
    const unsigned after = __builtin_amdgcn_s_getreg(3 | (31 << 11));
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = before; out[blockIdx.x * 4 + 1] = after; out[blockIdx.x * 4 + 2] = mode; }
    out[64 + blockIdx.x * 64 + threadIdx.x] = __float_as_uint(y);
}
int main() {
    float h[4][128];
    for (int i = 0; i < 128; ++i) { h[0][i] = 1.5f + i; h[1][i] = (i == 5) ? 1e6f : 1.f; h[2][i] = (i == 70) ? __builtin_nanf("") : 2.f; h[3][i] = (i == 3) ? 1e-9f : 1.f; }
    float* d; unsigned* o;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4096 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[] = {"in range", "one 1e6 (overflow to inf)", "one NaN", "one 1e-9 (underflow)"};
    for (int c = 0; c < 4; ++c) {
        hipMemset(o, 0, 4096 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d + c * 128, o);
        unsigned r[4];
        hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
        printf("%-28s TRAPSTS before %08x after %08x (EXCP bits [8:0]: %03x -> %03x)  MODE %08x\n", names[c], r[0], r[1], r[0] & 0x1ff, r[1] & 0x1ff, r[2]);
    }
    return 0;
}
