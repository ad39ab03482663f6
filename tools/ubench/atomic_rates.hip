// Micro-benchmark: throughput of wavefront-wide global atomics on gfx950 (no returned value, every lane its own address, a buffer
// that cycles through 64 MB so the lines come from the L2 / MALL like k_view1_bwd's feature-gradient scatter).  Question behind it
// (DESIGN.md 7, round 4): the scatter of k_view1_bwd runs at ONE atomic dword per cycle and CU -- is that a rate per LANE OPERATION or
// per DWORD?  If a 64-bit atomic (u64 add, f64 add) costs the same 64 cycles per wavefront instruction as a 32-bit one, two fixed-point
// channels packed into one u64 would halve the scatter (and make it deterministic); if it costs 128, nothing is gained.
// Build: hipcc --offload-arch=gfx950 -O3 -o atomic_rates atomic_rates.hip ; run: ./atomic_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(char* buf, size_t mask, int iters) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned wave = (unsigned)(tid >> 6), lane = threadIdx.x & 63;
    unsigned h = wave * 2654435761u;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;                              // a pseudo-random 512-byte line group per wavefront instruction
        const size_t line = ((size_t)(h >> 8) * 512) & mask;
        if constexpr (MODE == 0) unsafeAtomicAdd(reinterpret_cast<float*>(buf + line) + lane, 1.0f);
        else if constexpr (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(buf + line) + lane, 1u);
        else if constexpr (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(buf + line) + lane, 1ull);
        else if constexpr (MODE == 3) unsafeAtomicAdd(reinterpret_cast<double*>(buf + line) + lane, 1.0);
        else if constexpr (MODE == 4) reinterpret_cast<float*>(buf + line)[lane] = 1.0f;       // plain store, for scale
    }
}

template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main() {
    const size_t bytes = 64ull << 20;
    char* buf; hipMalloc(&buf, bytes + 4096); hipMemset(buf, 0, bytes + 4096);
    const int iters = 4096;
    const char* names[] = {"global_atomic_add_f32", "global_atomic_add (u32)", "global_atomic_add_x2 (u64)", "global_atomic_add_f64", "global_store_dword"};
    for (int wpc = 4; wpc <= 8; wpc += 4) {                          // wavefronts per CU: 4 (one per SIMD) and 8
        printf("-- %d wavefronts per CU, 256 CUs, %d wavefront instructions each, 64 MB target\n", wpc, iters);
        for (int m = 0; m < 5; ++m) {
            const int blocks = 256 * wpc / 4;
            float ms = 0;
            auto run = [&]() {
                switch (m) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, bytes - 1, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, bytes - 1, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, bytes - 1, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, buf, bytes - 1, iters); break;
                    default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, buf, bytes - 1, iters); break;
                }
            };
            ms = timeit(run);
            const double cyc = ms * 1e-3 * 2.4e9;                    // cycles of the launch
            const double per_cu = (double)wpc * iters;               // wavefront instructions per CU
            printf("   %-28s %8.3f ms  -> %6.1f cycles per wavefront instruction and CU (%4.2f lane-ops / cycle / CU)\n", names[m], ms, cyc / per_cu,
                   64.0 * per_cu / cyc);
        }
    }
    return 0;
}
