// Host helper: the first k entries of torch.randperm(n) on the CPU generator, bit-exact, without shuffling all n.
//
// The reference draws the depth-loss pixels as torch.randperm(h*w)[:8192] on the default CPU generator
// (src/nr/network/renderer.py:222-228): a forward Fisher-Yates over mt19937 (ATen randperm_cpu, the n < 2^32/20 branch:
// r[i] <-> r[i + random() % (n - i)] for i = 0 .. n-2, one 32-bit draw per step).  Entry i is final after step i, so the
// first k entries need k swaps on a sparse view of the identity permutation; the other n-1-k draws only advance the
// generator, which costs one twist per 624 draws and no tempering.  5-15 ms of random-access swaps per scene become ~0.2 ms,
// with the same coordinates and the same generator state afterwards (tests/test_host_rng.py compares both with torch).
//
// State layout = torch.get_rng_state() of the CPU generator (CPUGeneratorImplState, 5056 bytes): seed u64 @0, left i32 @8,
// seeded i32 @12, next u64 @16, state[624] as u64 @24, then the normal-distribution caches (untouched).
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/gnr.h"

namespace {

constexpr int N = 624, M = 397;

struct Mt {
    uint32_t s[N];
    int left;        // ATen mt19937: draws left before the next twist (+1)
    int next;

    static uint32_t mix(uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); }
    void twist() {
        int k = 0;
        for (; k < N - M; ++k) s[k] = s[k + M] ^ mix(s[k], s[k + 1]);
        for (; k < N - 1; ++k) s[k] = s[k + M - N] ^ mix(s[k], s[k + 1]);
        s[N - 1] = s[M - 1] ^ mix(s[N - 1], s[0]);
        left = N;
        next = 0;
    }
    uint32_t draw() {
        if (--left == 0) twist();
        uint32_t y = s[next++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

// sparse view of the identity permutation: open addressing, keys are positions
struct Sparse {
    std::vector<long long> key, val;
    size_t mask;
    explicit Sparse(size_t cap) : key(cap, -1), val(cap, 0), mask(cap - 1) {}
    size_t slot(long long p) const {
        size_t h = ((uint64_t)p * 0x9E3779B97F4A7C15ull) >> 20 & mask;
        while (key[h] != -1 && key[h] != p) h = (h + 1) & mask;
        return h;
    }
    long long get(long long p) const { const size_t h = slot(p); return key[h] == p ? val[h] : p; }
    void set(long long p, long long v) { const size_t h = slot(p); key[h] = p; val[h] = v; }
};

}  // namespace

extern "C" int gnr_host_randperm_prefix(unsigned char* torch_cpu_rng_state, long long state_bytes, long long n, int k,
                                        long long* out) {
    if (!torch_cpu_rng_state || (!out && k > 0)) return GNR_ERR_ARG;
    if (state_bytes != 5056 || n < 1 || k < 0 || k > n || n >= (long long)(0xffffffffu / 20)) return GNR_ERR_SHAPE;
    Mt g;
    int32_t left, seeded;
    uint64_t next;
    std::memcpy(&left, torch_cpu_rng_state + 8, 4);
    std::memcpy(&seeded, torch_cpu_rng_state + 12, 4);
    std::memcpy(&next, torch_cpu_rng_state + 16, 8);
    if (!seeded || left < 1 || left > N || next > (uint64_t)N) return GNR_ERR_SHAPE;
    for (int i = 0; i < N; ++i) {
        uint64_t w;
        std::memcpy(&w, torch_cpu_rng_state + 24 + 8 * i, 8);
        g.s[i] = (uint32_t)w;
    }
    g.left = left;
    g.next = (int)next;
    size_t cap = 1024;
    while (cap < (size_t)(4 * (k + 1))) cap <<= 1;
    Sparse perm(cap);
    const long long steps = n - 1;                 // randperm_cpu draws n-1 numbers
    long long i = 0;
    for (; i < steps && i < k; ++i) {
        const long long z = (long long)(g.draw() % (uint32_t)(n - i));
        const long long a = perm.get(i), b = perm.get(i + z);
        perm.set(i, b);
        perm.set(i + z, a);
        out[i] = b;
    }
    for (long long j = i; j < k; ++j) out[j] = perm.get(j);          // k == n: the last entry
    for (; i < steps; ++i) {                        // advance the generator: values unused
        if (--g.left == 0) g.twist();
        ++g.next;
    }
    left = g.left;
    next = (uint64_t)g.next;
    std::memcpy(torch_cpu_rng_state + 8, &left, 4);
    std::memcpy(torch_cpu_rng_state + 16, &next, 8);
    for (int q = 0; q < N; ++q) {
        const uint64_t w = g.s[q];
        std::memcpy(torch_cpu_rng_state + 24 + 8 * q, &w, 8);
    }
    return GNR_OK;
}
