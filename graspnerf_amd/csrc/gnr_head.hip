// VGN grasp head (ref: src/gd/networks.py:39-97) as fp32 implicit-GEMM 3D
// convolutions on v_mfma_f32_16x16x4_f32.  "MFMA only for the 3D conv's im2col GEMM where it is a true
// dense contraction" (BASELINE.json north_star) -- this is that contraction.
//
//   enc: conv(1->16,k5,s2)+ReLU, conv(16->32,k3,s2)+ReLU, conv(32->64,k3,s2)+ReLU
//   dec: conv(64->64,k3)+ReLU, nearest->10, conv(64->32,k3)+ReLU, nearest->20, conv(32->16,k5)+ReLU, nearest->40
//   heads (k5): qual = sigmoid(conv 16->1), rot = normalize(conv 16->4), width = conv 16->1
//
// Kernel k_conv3d: one wavefront = one 4x4x4 brick of output voxels (4 MFMA column tiles of 4x4 voxels, one
// per z) x all output channels.  D[cout][voxel] = sum_{tap,cin} W[cout][cin][tap] * In[cin][src(voxel*s+tap-p)]
// with the weight fragment as the MFMA A operand and the gathered input as B; the k order is tap-major /
// cin-minor so a k-step (4 input channels of ONE tap) shares its spatial address and bounds test, and the
// nearest-neighbour upsampling in front of a conv is folded into src() (never materialised).  Inputs are
// a few hundred KB per scene and stay cache resident, so the B operand is gathered straight from global
// memory; A fragments are pre-packed [tap][cin/4][cout/16][64 lanes] and read coalesced.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <atomic>
#include <vector>

#include "../../include/gnr.h"

// per-launch timing brackets of the library (gnr_capi.inc; active between gnr_timing_begin / gnr_timing_end)
extern "C" int gnr_internal_timing_open(const char* label, void* stream);
extern "C" void gnr_internal_timing_close(int idx, void* stream);
struct HeadScope {
    void* st; int idx;
    HeadScope(const char* label, void* s) : st(s), idx(gnr_internal_timing_open(label, s)) {}
    ~HeadScope() { gnr_internal_timing_close(idx, st); }
};

namespace gnrh {

typedef float f4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
DEV f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int NLAYER = 7;
struct LayerDesc { int cin, cout, k, stride, off_w, off_b; };   // canonical offsets (floats)
// canonical blob = ConvNet state-dict order (networks.py:39-47): encoder.conv{1,2,3}, decoder.conv{1,2,3},
// conv_qual, conv_rot, conv_width, each .weight then .bias
constexpr int w_sz(int co, int ci, int k) { return co * ci * k * k * k; }
constexpr int C_E1W = 0, C_E1B = C_E1W + w_sz(16, 1, 5), C_E2W = C_E1B + 16, C_E2B = C_E2W + w_sz(32, 16, 3);
constexpr int C_E3W = C_E2B + 32, C_E3B = C_E3W + w_sz(64, 32, 3), C_D1W = C_E3B + 64, C_D1B = C_D1W + w_sz(64, 64, 3);
constexpr int C_D2W = C_D1B + 64, C_D2B = C_D2W + w_sz(32, 64, 3), C_D3W = C_D2B + 32, C_D3B = C_D3W + w_sz(16, 32, 5);
constexpr int C_QW = C_D3B + 16, C_QB = C_QW + w_sz(1, 16, 5), C_RW = C_QB + 1, C_RB = C_RW + w_sz(4, 16, 5);
constexpr int C_WW = C_RB + 4, C_WB = C_WW + w_sz(1, 16, 5), C_TOTAL = C_WB + 1;

// packed blob: layer 0 (cin = 1) keeps its canonical weights (VALU kernel); layers 1..6 are A fragments
// [tap][cin/4][cout_blocks][64] followed by a bias block [cout_blocks*16]
constexpr int frag_sz(int co, int ci, int k) { return k * k * k * (ci / 4) * ((co + 15) / 16) * 64; }
constexpr int P_E1 = 0, P_E2 = P_E1 + w_sz(16, 1, 5) + 16, P_E3 = P_E2 + frag_sz(32, 16, 3) + 32;
constexpr int P_D1 = P_E3 + frag_sz(64, 32, 3) + 64, P_D2 = P_D1 + frag_sz(64, 64, 3) + 64;
// decoder.conv3 and the heads always see an input that was nearest-upsampled by exactly 2 (10->20, 20->40:
// networks.py:91-96).  A k5 conv on a x2-replicated grid is, per output parity (pz,py,px), a k3 conv on the
// un-replicated grid whose weights are sums of the original taps (per axis, parity 0: {-2,-1}|{0,1}|{2},
// parity 1: {-2}|{-1,0}|{1,2}).  Folding is exact algebra (weights summed in fp64 on the host) and cuts the
// MACs of these two layers (92 % of the head) by 125/27 = 4.6x.  Layout: 8 parity classes x [27 taps] fragments.
constexpr int P_D3 = P_D2 + frag_sz(32, 64, 3) + 32, P_HD = P_D3 + 8 * frag_sz(16, 32, 3) + 16;
constexpr int P_TOTAL = P_HD + 8 * frag_sz(16, 16, 3) + 16;

// ---- layer 0: 1 -> 16 channels, k5, stride 2, ReLU; one thread per output voxel (16 MMAC per scene) -------
__global__ __launch_bounds__(256) void k_conv_first(const float* __restrict__ vol, const float* __restrict__ wb,
                                                    float* __restrict__ out, int Din, int Dout, int B) {
    const int n = Dout * Dout * Dout;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * n) return;
    const int b = i / n, v = i - b * n;
    const int z = v / (Dout * Dout), y = (v / Dout) % Dout, x = v % Dout;
    const float* in = vol + (size_t)b * Din * Din * Din;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = wb[16 * 125 + c];
    for (int tz = 0; tz < 5; ++tz) {
        const int iz = 2 * z + tz - 2;
        for (int ty = 0; ty < 5; ++ty) {
            const int iy = 2 * y + ty - 2;
#pragma unroll
            for (int tx = 0; tx < 5; ++tx) {
                const int ix = 2 * x + tx - 2;
                const bool ok = (unsigned)iz < (unsigned)Din && (unsigned)iy < (unsigned)Din && (unsigned)ix < (unsigned)Din;
                const float val = ok ? in[(iz * Din + iy) * Din + ix] : 0.f;
                const int tap = (tz * 5 + ty) * 5 + tx;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(wb[c * 125 + tap], val, acc[c]);
            }
        }
    }
    float* o = out + (size_t)b * 16 * n + v;
#pragma unroll
    for (int c = 0; c < 16; ++c) o[(size_t)c * n] = fmaxf(acc[c], 0.f);
}

// nearest-neighbour source index of F.interpolate(x, size): min(floor(dst * in/out), in-1)   (networks.py:88-96)
__global__ void k_umaps(int* __restrict__ um, int d3) {
    const int i = threadIdx.x;
    if (i < 10) um[i] = min((int)floorf(i * ((float)d3 / 10.f)), d3 - 1);
    else if (i < 30) um[i] = min((int)floorf((i - 10) * (10.f / 20.f)), 9);
    else if (i < 70) um[i] = min((int)floorf((i - 30) * (20.f / 40.f)), 19);
}

struct ConvArgs {
    const float* in;      // [B][CIN][Din^3]
    const float* wfrag;   // [taps][CIN/4][NB][64]
    const float* bias;    // [NB*16]
    const int* umap;      // [Deff] source index of each (virtually upsampled) input coordinate, or null
    float* out;           // [B][COUT][Dout^3]           (EPI 0)
    float *qual, *rot, *width;   // heads               (EPI 1)
    int B, Din, Deff, Dout, cout;
    int nb_total;         // output-channel blocks of the layer; blockIdx.y selects NB of them (small layers: more waves)
};

// epilogue shared by both kernels: lane (voxel r of tile t, group g) holds output channels 16*nb + 4*g + {0..3}
template <int NB, int EPI>
DEV void conv_epilogue(const ConvArgs& a, const f4 (&acc)[4][NB], int b, int bz, int oy, int ox, int g, int parity = -1,
                       int nb0 = 0) {
    // parity >= 0 (folded x2 upsampling): source-grid voxel (z,y,x) of class (pz,py,px) is output voxel (2z+pz, ...)
    const int Ds = a.Dout, D = parity >= 0 ? 2 * Ds : Ds, n = D * D * D;
    const int pz = parity >= 0 ? (parity >> 2) & 1 : 0, py = parity >= 0 ? (parity >> 1) & 1 : 0, px = parity >= 0 ? parity & 1 : 0;
    const int sc = parity >= 0 ? 2 : 1;
    const bool xyok = ox < Ds && oy < Ds && b < a.B;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int oz = bz * 4 + t;
        if (!xyok || oz >= Ds) continue;
        const int v = ((oz * sc + pz) * D + oy * sc + py) * D + ox * sc + px;
        if constexpr (EPI == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float* o = a.out + ((size_t)b * a.cout + 16 * (nb0 + nb) + 4 * g) * n + v;
                o[0] = fmaxf(acc[t][nb].x, 0.f); o[(size_t)n] = fmaxf(acc[t][nb].y, 0.f);
                o[(size_t)2 * n] = fmaxf(acc[t][nb].z, 0.f); o[(size_t)3 * n] = fmaxf(acc[t][nb].w, 0.f);
            }
        } else {                                              // heads, packed channel order rot0..3, qual, width
            const f4 x = acc[t][0];
            if (g == 0) {                                     // rot: F.normalize(dim=1), eps 1e-12 (networks.py:52)
                const float nr = fmaxf(sqrtf(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w), 1e-12f);
                float* o = a.rot + (size_t)b * 4 * n + v;
                o[0] = x.x / nr; o[(size_t)n] = x.y / nr; o[(size_t)2 * n] = x.z / nr; o[(size_t)3 * n] = x.w / nr;
            } else if (g == 1) {
                a.qual[(size_t)b * n + v] = 1.f / (1.f + __expf(-x.x));   // networks.py:51
                a.width[(size_t)b * n + v] = x.y;                         // networks.py:53
            }
        }
    }
}

// ---- strided encoder layers (large input footprint per brick, ~2 % of the FLOPs): one wavefront per 4x4x4
// output brick, B operand gathered straight from global memory, A fragments read coalesced from global.
template <int CIN, int NB, int KS, int STRIDE, int EPI>
__global__ __launch_bounds__(256) void k_conv3d_direct(ConvArgs a) {
    constexpr int PAD = KS / 2, C4 = CIN / 4;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int Din = a.Din, Din3 = Din * Din * Din;
    const int nbr = (a.Dout + 3) >> 2;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= a.B * nbr * nbr * nbr) return;
    const int b = task / (nbr * nbr * nbr), br = task - b * nbr * nbr * nbr;
    const int bz = br / (nbr * nbr);
    const int ox = (br % nbr) * 4 + (r & 3), oy = ((br / nbr) % nbr) * 4 + (r >> 2);
    const float* ing = a.in + (size_t)b * CIN * Din3 + (size_t)g * Din3;   // lane group g = input channel 4c+g
    const int nb0 = blockIdx.y * NB, NBT = a.nb_total;
    f4 acc[4][NB];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = reinterpret_cast<const f4*>(a.bias)[(nb0 + nb) * 4 + g];
    const float* wf = a.wfrag + nb0 * 64 + lane;
    for (int tz = 0; tz < KS; ++tz) {
        int zoff[4];
        bool zok[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int iz = (bz * 4 + t) * STRIDE + tz - PAD;
            zok[t] = (unsigned)iz < (unsigned)a.Deff;
            const int izc = zok[t] ? iz : 0;
            zoff[t] = (a.umap ? a.umap[izc] : izc) * Din * Din;
        }
        for (int ty = 0; ty < KS; ++ty) {
            const int iy = oy * STRIDE + ty - PAD;
            const bool yok = (unsigned)iy < (unsigned)a.Deff;
            const int iyc = yok ? iy : 0;
            const int yoff = (a.umap ? a.umap[iyc] : iyc) * Din;
#pragma unroll 1
            for (int tx = 0; tx < KS; ++tx) {
                const int ix = ox * STRIDE + tx - PAD;
                const bool xin = (unsigned)ix < (unsigned)a.Deff;
                const bool xok = yok && xin;
                const int ixc = xin ? ix : 0;
                const int xyoff = yoff + (a.umap ? a.umap[ixc] : ixc);
                const float* wt = wf + (size_t)((tz * KS + ty) * KS + tx) * C4 * NBT * 64;
#pragma unroll
                for (int c = 0; c < C4; ++c) {
                    float av[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) av[nb] = wt[(c * NBT + nb) * 64];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float v = ing[(size_t)(4 * c) * Din3 + zoff[t] + xyoff];
                        const float bv = (xok && zok[t]) ? v : 0.f;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma16(av[nb], bv, acc[t][nb]);
                    }
                }
            }
        }
    }
    conv_epilogue<NB, EPI>(a, acc, b, bz, oy, ox, g, -1, nb0);
}

// ---- stride-1 decoder / head layers (98 % of the FLOPs): one workgroup (4 wavefronts) per 8x8x4 output brick.
//   * the brick's input halo, INCLUDING an explicit zero border, is copied once into LDS ([CIN][HALO_MAX]); the
//     nearest-neighbour upsampling is folded into the source index, so the halo is at most 6x6x4 cells + border;
//     the inner loop then needs no bounds tests: B operand = one ds_read at base + constant offset
//   * the A fragments of TS consecutive taps are staged in LDS per step (read from HBM/L2 once per workgroup
//     instead of once per wavefront, and off the latency-critical path)
//   The first version (4x4 voxel patch gathered from global: 16 cache lines per load) was L1-bound at 20 % MFMA
//   utilisation.
// floats per channel in LDS: upsampled-input layers on a <=5^3 source grid need 7x7x6 = 294 (grid + zero border);
// the folded k3 layers read the full (8+2)x(8+2)x(4+2) = 600-cell neighbourhood of their 8x8x4 brick
constexpr int halo_max(bool fold) { return fold ? 608 : 320; }

template <int CIN, int NB, int KS, int TS, int EPI, bool FOLD>
__global__ __launch_bounds__(256) void k_conv3d_staged(ConvArgs a) {
    constexpr int PAD = KS / 2, C4 = CIN / 4, TAPS = KS * KS * KS, AFL = C4 * NB * 64;   // A floats per tap
    constexpr int HALO_MAX = halo_max(FOLD);
    static_assert(TAPS % TS == 0, "tap staging step must divide the tap count");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                                   // [CIN][HALO_MAX]
    float* asl = smem + CIN * HALO_MAX;                   // [TS][C4][NB][64]
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int Din = a.Din, Din3 = Din * Din * Din, Deff = a.Deff;
    const int nbx = (a.Dout + 7) >> 3, nbz = (a.Dout + 3) >> 2;
    const int blk = blockIdx.x;
    const int b = blk / (nbx * nbx * nbz), br = blk - b * nbx * nbx * nbz;
    const int bz = br / (nbx * nbx), by8 = ((br / nbx) % nbx) * 8, bx8 = (br % nbx) * 8;
    const int ox = bx8 + (wave & 1) * 4 + (r & 3), oy = by8 + (wave >> 1) * 4 + (r >> 2);
    const float* in = a.in + (size_t)b * CIN * Din3;
    const int nb0 = blockIdx.y * NB, NBT = a.nb_total;                    // this block's output-channel blocks
    constexpr int NPAR = FOLD ? 8 : 1;                                    // parity classes share the halo (loaded once)
    const float* wfrag = a.wfrag;
    // source cell of an effective (virtually upsampled) coordinate; -1 / Din are the zero border
    auto src = [&](int e) { return e < 0 ? -1 : (e >= Deff ? Din : (a.umap ? a.umap[e] : e)); };
    const int sx0 = src(bx8 - PAD), sy0 = src(by8 - PAD), sz0 = src(bz * 4 - PAD);
    const int hx = src(bx8 + 7 + KS - 1 - PAD) - sx0 + 1, hy = src(by8 + 7 + KS - 1 - PAD) - sy0 + 1;
    const int hz = src(bz * 4 + 3 + KS - 1 - PAD) - sz0 + 1, hv = hx * hy * hz;          // host guarantees hv <= HALO_MAX
    for (int i = threadIdx.x; i < CIN * hv; i += 256) {
        const int c = i / hv, rem = i - c * hv;
        const int z = sz0 + rem / (hx * hy), y = sy0 + (rem / hx) % hy, x = sx0 + rem % hx;
        const bool ok = (unsigned)z < (unsigned)Din && (unsigned)y < (unsigned)Din && (unsigned)x < (unsigned)Din;
        halo[c * HALO_MAX + rem] = ok ? in[(size_t)c * Din3 + (z * Din + y) * Din + x] : 0.f;
    }
    // per-lane halo offsets of the KS taps along each axis (stride 1)
    int xo[KS], yo[KS], zo[4][KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        xo[k] = src(ox + k - PAD) - sx0;
        yo[k] = (src(oy + k - PAD) - sy0) * hx;
#pragma unroll
        for (int t = 0; t < 4; ++t) zo[t][k] = (src(bz * 4 + t + k - PAD) - sz0) * hx * hy;
    }
    f4 acc[4][NB];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = reinterpret_cast<const f4*>(a.bias)[(nb0 + nb) * 4 + g];

    const float* hg = halo + g * HALO_MAX;                // lane group g = input channel 4c+g
    static_assert(TS % KS == 0 && C4 % 4 == 0, "a staging step is a whole number of x-rows; cin multiple of 16");
    // A-fragment slices are prefetched into registers one step ahead (global latency hides under the MFMAs of
    // the current slice) and written to LDS between two barriers once everybody has finished reading.
    constexpr int PF = TS * AFL / 256;                    // floats per thread per slice
    static_assert((TS * AFL) % 256 == 0, "slice must split evenly over the workgroup");
    float pf[PF];
    // element i of a slice = (tap tp, c, nb, lane); in global memory the nb axis has NBT entries
    auto gidx = [&](int i) { const int ln = i & 63, q = i >> 6, nb = q % NB, c = (q / NB) % C4, tp = q / (NB * C4);
                             return ((size_t)(tp * C4 + c) * NBT + nb0 + nb) * 64 + ln; };
#pragma unroll
    for (int k = 0; k < PF; ++k) pf[k] = wfrag[gidx(threadIdx.x + 256 * k)];
    for (int s0 = 0; s0 < TAPS * NPAR; s0 += TS) {
        __syncthreads();                                  // previous slice consumed (and halo written, first trip)
#pragma unroll
        for (int k = 0; k < PF; ++k) {                    // global [tap][c][nb][lane] -> LDS [tap][nb][lane][c]:
            const int i = threadIdx.x + 256 * k;          // one ds_read_b128 then feeds 4 k-steps
            const int ln = i & 63, q = i >> 6, nb = q % NB, c = (q / NB) % C4, tp = q / (NB * C4);
            asl[((tp * NB + nb) * 64 + ln) * C4 + c] = pf[k];
        }
        __syncthreads();
        if (s0 + TS < TAPS * NPAR) {
            const float* srcp = wfrag + (size_t)(s0 + TS) * C4 * NBT * 64;
#pragma unroll
            for (int k = 0; k < PF; ++k) pf[k] = srcp[gidx(threadIdx.x + 256 * k)];
        }
#pragma unroll 1
        for (int row = 0; row < TS / KS; ++row) {         // one (tz, ty) row of KS taps, x-taps unrolled
            const int rowi = (s0 % TAPS) / KS + row, tz = rowi / KS, ty = rowi % KS;
            int yz[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < KS; ++k) {                // static-index selects keep yo/zo in registers
                if (ty == k) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) yz[t] += yo[k];
                }
                if (tz == k) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) yz[t] += zo[t][k];
                }
            }
#pragma unroll
            for (int tx = 0; tx < KS; ++tx) {
                const f4* ap = reinterpret_cast<const f4*>(asl + ((row * KS + tx) * NB * 64 + lane) * C4);
#pragma unroll
                for (int c4 = 0; c4 < C4 / 4; ++c4) {
                    f4 av[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) av[nb] = ap[nb * 64 * (C4 / 4) + c4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float bv = hg[4 * (4 * c4 + cc) * HALO_MAX + yz[t] + xo[tx]];
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma16(av[nb][cc], bv, acc[t][nb]);
                        }
                    }
                }
            }
        }
        if ((s0 + TS) % TAPS == 0) {                      // a parity class (or the whole layer) is complete
            conv_epilogue<NB, EPI>(a, acc, b, bz, oy, ox, g, FOLD ? s0 / TAPS : -1, nb0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[t][nb] = reinterpret_cast<const f4*>(a.bias)[(nb0 + nb) * 4 + g];
        }
    }
}

}  // namespace gnrh

using namespace gnrh;

static thread_local char h_err[256] = "";
// hipFuncSetAttribute (dynamic LDS above 64 KiB) is per device: one bit per device id and kernel
static bool head_attr_needed(std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load() & bit) return false;
    done.fetch_or(bit);
    return true;
}
extern "C" const char* gnr_head_last_error(void) { return h_err; }

extern "C" int gnr_head_canonical_floats(void) { return C_TOTAL; }
extern "C" int gnr_head_packed_floats(void) { return P_TOTAL; }

static void pack_conv(float* dst, const float* W, const float* bias, int cout, int cin, int k, int cout_pad_map(int)) {
    const int taps = k * k * k, C4 = cin / 4, NB = (cout + 15) / 16;
    for (int tap = 0; tap < taps; ++tap)
        for (int c = 0; c < C4; ++c)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane) {
                    const int o = cout_pad_map(16 * nb + (lane & 15)), ci = 4 * c + (lane >> 4);
                    dst[((tap * C4 + c) * NB + nb) * 64 + lane] = (o >= 0 && o < cout) ? W[((size_t)o * cin + ci) * taps + tap] : 0.f;
                }
    float* bd = dst + (size_t)taps * C4 * NB * 64;
    for (int i = 0; i < NB * 16; ++i) { const int o = cout_pad_map(i); bd[i] = (o >= 0 && o < cout) ? bias[o] : 0.f; }
}

// fold a k5 conv that follows a x2 nearest upsampling into 8 parity-class k3 convs (see the layout comment)
static void pack_folded(float* dst, const float* W, const float* bias, int cout, int cin) {
    static const int cell[2][5] = {{-1, -1, 0, 0, 1}, {-1, 0, 0, 1, 1}};      // source cell of tap d-2 for parity 0 / 1
    std::vector<float> wf((size_t)cout * cin * 27);
    const int fs = 27 * (cin / 4) * ((cout + 15) / 16) * 64;
    std::vector<float> tmp((size_t)fs + 64);
    for (int par = 0; par < 8; ++par) {
        const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1;
        for (int o = 0; o < cout; ++o)
            for (int ci = 0; ci < cin; ++ci) {
                double acc[27] = {0};
                for (int dz = 0; dz < 5; ++dz)
                    for (int dy = 0; dy < 5; ++dy)
                        for (int dx = 0; dx < 5; ++dx)
                            acc[((cell[pz][dz] + 1) * 3 + cell[py][dy] + 1) * 3 + cell[px][dx] + 1] +=
                                (double)W[((size_t)o * cin + ci) * 125 + (dz * 5 + dy) * 5 + dx];
                for (int t = 0; t < 27; ++t) wf[((size_t)o * cin + ci) * 27 + t] = (float)acc[t];
            }
        pack_conv(tmp.data(), wf.data(), bias, cout, cin, 3, [](int o) { return o; });
        for (int i = 0; i < fs; ++i) dst[(size_t)par * fs + i] = tmp[i];
        if (par == 7) for (int i = 0; i < ((cout + 15) / 16) * 16; ++i) dst[(size_t)8 * fs + i] = tmp[fs + i];
    }
}

extern "C" int gnr_pack_grasp_head(const float* c, float* p) {
    if (!c || !p) return GNR_ERR_ARG;
    for (int i = 0; i < P_TOTAL; ++i) p[i] = 0.f;
    for (int i = 0; i < w_sz(16, 1, 5); ++i) p[P_E1 + i] = c[C_E1W + i];
    for (int i = 0; i < 16; ++i) p[P_E1 + w_sz(16, 1, 5) + i] = c[C_E1B + i];
    auto ident = [](int o) { return o; };
    pack_conv(p + P_E2, c + C_E2W, c + C_E2B, 32, 16, 3, ident);
    pack_conv(p + P_E3, c + C_E3W, c + C_E3B, 64, 32, 3, ident);
    pack_conv(p + P_D1, c + C_D1W, c + C_D1B, 64, 64, 3, ident);
    pack_conv(p + P_D2, c + C_D2W, c + C_D2B, 32, 64, 3, ident);
    pack_folded(p + P_D3, c + C_D3W, c + C_D3B, 16, 32);
    // heads fused into one 16->6 conv; packed channel order: rot0..3, qual, width
    static float hw[6 * 16 * 125], hb[6];
    for (int o = 0; o < 4; ++o) { for (int i = 0; i < 2000; ++i) hw[o * 2000 + i] = c[C_RW + o * 2000 + i]; hb[o] = c[C_RB + o]; }
    for (int i = 0; i < 2000; ++i) { hw[4 * 2000 + i] = c[C_QW + i]; hw[5 * 2000 + i] = c[C_WW + i]; }
    hb[4] = c[C_QB]; hb[5] = c[C_WB];
    pack_folded(p + P_HD, hw, hb, 6, 16);
    return GNR_OK;
}

#define HCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf(h_err, sizeof(h_err), "%s: %s", #call, hipGetErrorString(e_)); return GNR_ERR_HIP; } } while (0)

extern "C" size_t gnr_grasp_head_workspace_bytes(int B, int R) {
    const int d1 = (R - 1) / 2 + 1, d2 = (d1 - 1) / 2 + 1, d3 = (d2 - 1) / 2 + 1;
    size_t fl = (size_t)16 * d1 * d1 * d1 + (size_t)32 * d2 * d2 * d2 + 2 * (size_t)64 * d3 * d3 * d3 + 32 * 1000 + 16 * 8000;
    return (size_t)B * fl * sizeof(float) + 4096;
}

template <int CIN, int NB, int KS, int STRIDE, int EPI>
static int launch_direct(const ConvArgs& a, hipStream_t st) {
    const int nbr = (a.Dout + 3) / 4;
    const long blocks = ((long)a.B * nbr * nbr * nbr + 3) / 4;
    hipLaunchKernelGGL((k_conv3d_direct<CIN, NB, KS, STRIDE, EPI>), dim3((unsigned)blocks, a.nb_total / NB), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(h_err, sizeof(h_err), "k_conv3d_direct launch: %s", hipGetErrorString(e)); return GNR_ERR_HIP; }
    return GNR_OK;
}

template <int CIN, int NB, int KS, int TS, int EPI, bool FOLD>
static int launch_staged(const ConvArgs& a, hipStream_t st) {
    const int nbx = (a.Dout + 7) / 8, nbz = (a.Dout + 3) / 4;
    const long blocks = (long)a.B * nbx * nbx * nbz;
    const size_t lds = ((size_t)CIN * halo_max(FOLD) + (size_t)TS * (CIN / 4) * NB * 64) * sizeof(float);
    static std::atomic<unsigned long long> attr{0};
    if (head_attr_needed(attr)) {
        hipError_t e = hipFuncSetAttribute((const void*)k_conv3d_staged<CIN, NB, KS, TS, EPI, FOLD>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { snprintf(h_err, sizeof(h_err), "hipFuncSetAttribute: %s", hipGetErrorString(e)); return GNR_ERR_HIP; }
    }
    hipLaunchKernelGGL((k_conv3d_staged<CIN, NB, KS, TS, EPI, FOLD>), dim3((unsigned)blocks, a.nb_total / NB), dim3(256), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(h_err, sizeof(h_err), "k_conv3d_staged launch: %s", hipGetErrorString(e)); return GNR_ERR_HIP; }
    return GNR_OK;
}

// ConvNet.forward (networks.py:48-54).  volume [B,1,R,R,R]; outputs qual [B,1,40,40,40], rot [B,4,40^3], width [B,1,40^3]
// (the reference interpolates to the fixed sizes 10/20/40 whatever R is, networks.py:88-96).
extern "C" int gnr_grasp_head_fwd(int B, int R, const float* volume, const float* packed, float* qual, float* rot, float* width,
                                  void* ws, size_t ws_bytes, void* stream) {
    if (!volume || !packed || !qual || !rot || !width || !ws) { snprintf(h_err, sizeof(h_err), "gnr_grasp_head_fwd: null pointer"); return GNR_ERR_ARG; }
    if (B < 1 || R < 8 || R > 64) { snprintf(h_err, sizeof(h_err), "gnr_grasp_head_fwd: bad B/R"); return GNR_ERR_SHAPE; }
    if (ws_bytes < gnr_grasp_head_workspace_bytes(B, R)) { snprintf(h_err, sizeof(h_err), "workspace too small"); return GNR_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    HeadScope hs("grasp_head_fwd(all kernels)@gnr_grasp_head_fwd", stream);
    const int d1 = (R - 1) / 2 + 1, d2 = (d1 - 1) / 2 + 1, d3 = (d2 - 1) / 2 + 1;      // conv k, s2, pad k/2
    float* a1 = (float*)ws;                                   // [B][16][d1^3]
    float* a2 = a1 + (size_t)B * 16 * d1 * d1 * d1;           // [B][32][d2^3]
    float* a3 = a2 + (size_t)B * 32 * d2 * d2 * d2;           // [B][64][d3^3]
    float* a4 = a3 + (size_t)B * 64 * d3 * d3 * d3;           // [B][64][d3^3]
    float* a5 = a4 + (size_t)B * 64 * d3 * d3 * d3;           // [B][32][10^3]
    float* a6 = a5 + (size_t)B * 32 * 1000;                   // [B][16][20^3]
    int* umaps = (int*)(a6 + (size_t)B * 16 * 8000);          // 3 nearest-neighbour index maps
    hipLaunchKernelGGL(k_umaps, dim3(1), dim3(128), 0, st, umaps, d3);
    HCHK(hipGetLastError());
    {
        const int n = B * d1 * d1 * d1;
        hipLaunchKernelGGL(k_conv_first, dim3((n + 255) / 256), dim3(256), 0, st, volume, packed + P_E1, a1, R, d1, B);
        HCHK(hipGetLastError());
    }
    ConvArgs c{};
    c.B = B;
    int rc;
    c.in = a1; c.wfrag = packed + P_E2; c.bias = c.wfrag + frag_sz(32, 16, 3); c.umap = nullptr; c.out = a2; c.Din = d1; c.Deff = d1; c.Dout = d2; c.cout = 32; c.nb_total = 2;
    if ((rc = launch_direct<16, 1, 3, 2, 0>(c, st))) return rc;
    c.in = a2; c.wfrag = packed + P_E3; c.bias = c.wfrag + frag_sz(64, 32, 3); c.out = a3; c.Din = d2; c.Deff = d2; c.Dout = d3; c.cout = 64; c.nb_total = 4;
    if ((rc = launch_direct<32, 1, 3, 2, 0>(c, st))) return rc;
    c.in = a3; c.wfrag = packed + P_D1; c.bias = c.wfrag + frag_sz(64, 64, 3); c.out = a4; c.Din = d3; c.Deff = d3; c.Dout = d3; c.cout = 64; c.nb_total = 4;
    // LDS staging needs the brick's source halo (+ zero border) to fit HALO_MAX floats per channel: always true for
    // the fixed 10^3 / 20^3 decoder grids (<= 6x6x4), and for the 64-channel layers when the bottleneck grid is
    // <= 5^3 (R <= 40: at most 7x7x6 = 294 cells with border)
    const bool small = d3 <= 5;
    if ((rc = small ? launch_staged<64, 1, 3, 9, 0, false>(c, st) : launch_direct<64, 1, 3, 1, 0>(c, st))) return rc;
    c.in = a4; c.wfrag = packed + P_D2; c.bias = c.wfrag + frag_sz(32, 64, 3); c.umap = umaps; c.out = a5; c.Din = d3; c.Deff = 10; c.Dout = 10; c.cout = 32; c.nb_total = 2;
    if ((rc = small ? launch_staged<64, 1, 3, 9, 0, false>(c, st) : launch_direct<64, 1, 3, 1, 0>(c, st))) return rc;
    // decoder.conv3 and the heads: folded x2 upsampling -> k3 convs on the 10^3 / 20^3 source grids, 8 parity classes
    c.in = a5; c.wfrag = packed + P_D3; c.bias = c.wfrag + 8 * frag_sz(16, 32, 3); c.umap = nullptr; c.out = a6; c.Din = 10; c.Deff = 10; c.Dout = 10; c.cout = 16; c.nb_total = 1;
    if ((rc = launch_staged<32, 1, 3, 9, 0, true>(c, st))) return rc;
    c.in = a6; c.wfrag = packed + P_HD; c.bias = c.wfrag + 8 * frag_sz(16, 16, 3); c.out = nullptr; c.Din = 20; c.Deff = 20; c.Dout = 20; c.cout = 6; c.nb_total = 1;
    c.qual = qual; c.rot = rot; c.width = width;
    return launch_staged<16, 1, 3, 9, 1, true>(c, st);
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of a stride-1 "same" 3D convolution (gd.networks.ConvNet under autograd):
//   dW[o][i][tap] += sum_{b, voxel} dy[b][o][voxel] * x[b][i][voxel + offset(tap)]        (zero padding)
// MIOpen's path for this (a CK batched-GEMM) takes 75 ms for the fused 16 -> 6 k5 head at 40^3, batch 8.
// One wavefront owns one (tap, 16 x 16 block of [Cout x Cin], chunk of voxels): the voxel axis is the K of
// v_mfma_f32_16x16x4_f32 (A = dy [o][voxel], B = x [voxel][i]); partial blocks are added with atomics.
// ---------------------------------------------------------------------------------------------------------
namespace gnr_head {

constexpr int BW_CHUNK = 4096;          // voxels per wavefront

__global__ __launch_bounds__(256) void k_conv3d_bwd_weight(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dw, int B, int Cin, int Cout, int D, int H, int W,
                                                           int K, int nchunk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kg = lane >> 4;
    const int V3 = D * H * W, K3 = K * K * K, pad = K / 2;
    const int nbi = (Cin + 15) / 16, nbo = (Cout + 15) / 16;
    long job = (long)blockIdx.x * 4 + wave;                       // (chunk, tap, bo, bi), chunk fastest
    const long njobs = (long)nchunk * K3 * nbo * nbi;
    if (job >= njobs) return;
    const int chunk = (int)(job % nchunk); job /= nchunk;
    const int tap = (int)(job % K3); job /= K3;
    const int bo = (int)(job % nbo), bi = (int)(job / nbo);
    const int td = tap / (K * K) - pad, th = (tap / K) % K - pad, tw = tap % K - pad;
    const int o = 16 * bo + c, i = 16 * bi + c;
    const bool o_ok = o < Cout, i_ok = i < Cin;
    const long v_begin = (long)chunk * BW_CHUNK, v_end = min((long)B * V3, v_begin + BW_CHUNK);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long v0 = v_begin; v0 < v_end; v0 += 16) {
        float a[4], bb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long v = v0 + 4 * kg + j;
            float av = 0.f, bv = 0.f;
            if (v < v_end) {
                const int b = (int)(v / V3), r = (int)(v - (long)b * V3);
                const int zd = r / (H * W), zh = (r / W) % H, zw = r % W;
                if (o_ok) av = dy[((long)b * Cout + o) * V3 + r];
                const int sd = zd + td, sh = zh + th, sw = zw + tw;
                if (i_ok && sd >= 0 && sd < D && sh >= 0 && sh < H && sw >= 0 && sw < W)
                    bv = x[((long)b * Cin + i) * V3 + ((long)sd * H + sh) * W + sw];
            }
            a[j] = av; bb[j] = bv;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bb[j], acc, 0, 0, 0);
    }
    const float e[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int oo = 16 * bo + 4 * kg + q, ii = 16 * bi + c;
        if (oo < Cout && ii < Cin) unsafeAtomicAdd(dw + ((long)oo * Cin + ii) * K3 + tap, e[q]);
    }
}

}  // namespace gnr_head

// dw [Cout][Cin][K][K][K] is ACCUMULATED (zero it first).  x [B][Cin][D][H][W], dy [B][Cout][D][H][W], stride 1, padding K/2.
extern "C" int gnr_conv3d_bwd_weight(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int D, int H, int W, int K,
                                     void* stream) {
    if (!x || !dy || !dw || B < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || K < 1 || !(K & 1)) return GNR_ERR_ARG;
    const long total = (long)B * D * H * W;
    const int nchunk = (int)((total + gnr_head::BW_CHUNK - 1) / gnr_head::BW_CHUNK);
    const long njobs = (long)nchunk * K * K * K * ((Cin + 15) / 16) * ((Cout + 15) / 16);
    HeadScope hs("k_conv3d_bwd_weight@gnr_conv3d_bwd_weight", stream);
    hipLaunchKernelGGL(gnr_head::k_conv3d_bwd_weight, dim3((unsigned)((njobs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, dy, dw, B,
                       Cin, Cout, D, H, W, K, nchunk);
    return hipGetLastError() == hipSuccess ? GNR_OK : GNR_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------
// Stride-1, same-padding 3D convolution for the grasp head UNDER AUTOGRAD (decoder.conv3 and the fused heads: the two
// k5 layers at 20^3 / 40^3 that hold 92 % of the head's MACs).  One kernel serves the forward and the backward-data pass:
//   mode 0:  y [b,co,p] = bias[co] + sum_{ci,tap} w[co,ci,tap]  x[b,ci,p + tap - K/2]
//   mode 1:  dx[b,ci,p] =            sum_{co,tap} w[co,ci,tap] dy[b,co,p - (tap - K/2)]   (= mode 0 with the weights
//            transposed and the taps flipped)
// The weights change every optimiser step, so the MFMA A fragments are packed ON THE DEVICE from the canonical
// [Cout][Cin][K^3] tensor (k_pack_conv3d_frag) right before the convolution.
// Implicit GEMM like k_conv3d_staged: a workgroup (4 wavefronts) owns an 8x8x4 output brick and one block of 16 output
// channels; the input channels are swept in chunks of 16 whose halo (+ zero border) is staged in LDS, the A fragments of one
// x-row of taps at a time.  73.7 KB (halo of a k5 chunk) + 5 KB of fragments: two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------
namespace gnr_head {

typedef float f4 __attribute__((ext_vector_type(4)));

// frag [chunk][tap][c = 0..3][nb][64 lanes]: A operand of lane l = W_eff[out = 16 nb + l % 16][in = 16 chunk + 4 c + l / 16][tap]
__global__ void k_pack_conv3d_frag(const float* __restrict__ w, float* __restrict__ frag, int Cin, int Cout, int K, int mode,
                                   int nchunks, int nbt) {
    const int K3 = K * K * K;
    const long n = (long)nchunks * K3 * 4 * nbt * 64;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int lane = (int)(i & 63);
    long q = i >> 6;
    const int nb = (int)(q % nbt); q /= nbt;
    const int c = (int)(q & 3); q >>= 2;
    const int tap = (int)(q % K3), chunk = (int)(q / K3);
    const int o = 16 * nb + (lane & 15), in = 16 * chunk + 4 * c + (lane >> 4);
    float v = 0.f;
    if (mode == 0) { if (o < Cout && in < Cin) v = w[((long)o * Cin + in) * K3 + tap]; }
    else           { if (o < Cin && in < Cout) v = w[((long)in * Cin + o) * K3 + (K3 - 1 - tap)]; }
    frag[i] = v;
}

struct Conv1Args {
    const float* in;       // [B][cin][D][H][W]
    const float* frag;     // [chunks][taps][4][nbt][64]
    const float* bias;     // [cout] or null
    float* out;            // [B][cout][D][H][W]
    int B, cin, cout, D, H, W, nchunks, nbt;
    const unsigned* mask;  // tap mask [forward Cin block][forward Cout block][4 words] (k_conv3d_tap_mask) or null = dense
    int mode;              // 0: chunks = forward Cin blocks, nb = forward Cout block;  1 (backward data): the other way round, taps flipped
};

// Which taps of a (16 input channels, 16 output channels) block hold a structural non-zero: bit `tap` of mask[bi][bo][tap / 32].  The
// pattern is a [Cout][Cin][K^3] tensor of the layer's shape whose non-zeros mark the weights that exist (a stride-2 layer run as a
// stride-1 k3 convolution over its space-to-depth input has 27 of its 8 x 27 (parity, tap) slots per input channel; backbone.conv3d_stride2).
__global__ void k_conv3d_tap_mask(const float* __restrict__ pattern, unsigned* __restrict__ mask, int Cin, int Cout, int K3, int nbo) {
    const int bi = blockIdx.x / nbo, bo = blockIdx.x - bi * nbo, lane = threadIdx.x;
    unsigned words[4] = {0u, 0u, 0u, 0u};
    for (int tap = 0; tap < K3; ++tap) {
        bool nz = false;
        for (int q = 0; q < 4; ++q) {
            const int e = lane * 4 + q, o = 16 * bo + (e >> 4), i = 16 * bi + (e & 15);
            if (o < Cout && i < Cin) nz = nz || pattern[((size_t)o * Cin + i) * K3 + tap] != 0.f;
        }
        if (__ballot(nz)) words[tap >> 5] |= 1u << (tap & 31);
    }
    if (lane == 0)
        for (int q = 0; q < 4; ++q) mask[(size_t)blockIdx.x * 4 + q] = words[q];
}

template <int KS>
__global__ __launch_bounds__(256, 2) void k_conv3d_s1(Conv1Args a) {
    constexpr int PAD = KS / 2, TAPS = KS * KS * KS;
    constexpr int HX = 8 + KS - 1, HY = 8 + KS - 1, HZ = 4 + KS - 1, HALO = HX * HY * HZ;
    constexpr int AFL = 4 * 64;                           // A floats per tap (16 input channels x 16 outputs)
    constexpr int TS = KS;                                // taps staged per step: one x-row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                                   // [16][HALO]
    float* asl = smem + 16 * HALO;                        // [TS][64 lanes][4 c]
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int nbx = (a.W + 7) >> 3, nby = (a.H + 7) >> 3, nbz = (a.D + 3) >> 2;
    const int blk = blockIdx.x;
    const int b = blk / (nbx * nby * nbz), br = blk - b * nbx * nby * nbz;
    const int bz4 = (br / (nbx * nby)) * 4, by8 = ((br / nbx) % nby) * 8, bx8 = (br % nbx) * 8;
    const int ox = bx8 + (wave & 1) * 4 + (r & 3), oy = by8 + (wave >> 1) * 4 + (r >> 2);
    const int nb = blockIdx.y;
    const size_t V3 = (size_t)a.D * a.H * a.W;
    f4 acc[4];
    {
        f4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const int o = 16 * nb + 4 * g;
            b4.x = o < a.cout ? a.bias[o] : 0.f; b4.y = o + 1 < a.cout ? a.bias[o + 1] : 0.f;
            b4.z = o + 2 < a.cout ? a.bias[o + 2] : 0.f; b4.w = o + 3 < a.cout ? a.bias[o + 3] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = b4;
    }
    // per-lane halo offsets: output (ox, oy, bz4 + t) reads halo cell (ox - bx8 + tx, oy - by8 + ty, t + tz)
    const int lx = ox - bx8, ly = (oy - by8) * HX;
    const float* hg = halo + g * HALO;                    // lane group g = input channel 4c + g of the chunk
    for (int ch = 0; ch < a.nchunks; ++ch) {
        unsigned mw[4] = {~0u, ~0u, ~0u, ~0u};
        if (a.mask) {                                     // wave-uniform: scalar loads
            const unsigned* mp = a.mask + (size_t)(a.mode ? nb * a.nchunks + ch : ch * a.nbt + nb) * 4;
            mw[0] = mp[0]; mw[1] = mp[1]; mw[2] = mp[2]; mw[3] = mp[3];
            if (!(mw[0] | mw[1] | mw[2] | mw[3])) continue;   // no weight joins this chunk to this output block
        }
        auto tap_on = [&](int tap) { const int t = a.mode ? TAPS - 1 - tap : tap; return (mw[t >> 5] >> (t & 31)) & 1u; };
        __syncthreads();                                  // everybody is done with the previous chunk's halo
        const float* in = a.in + ((size_t)b * a.cin + 16 * ch) * V3;
        for (int i0 = threadIdx.x; i0 < 16 * HALO; i0 += 256 * 8) {     // eight loads in flight per lane, then the LDS stores
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u, c = i / HALO, rem = i - c * HALO;
                const int z = bz4 - PAD + rem / (HX * HY), y = by8 - PAD + (rem / HX) % HY, x = bx8 - PAD + rem % HX;
                const bool ok = i < 16 * HALO && 16 * ch + c < a.cin && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                v[u] = ok ? in[(size_t)c * V3 + ((size_t)z * a.H + y) * a.W + x] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 256 * u < 16 * HALO) halo[i0 + 256 * u] = v[u];
        }
        const float* fr = a.frag + ((size_t)ch * TAPS * 4 * a.nbt + nb) * 64;      // + ((tap * 4 + c) * nbt) * 64 + lane
        for (int row = 0; row < KS * KS; ++row) {         // (tz, ty) row of KS x-taps
            unsigned rowbits = 0;
#pragma unroll
            for (int tx = 0; tx < KS; ++tx) rowbits |= tap_on(row * KS + tx) << tx;
            if (!rowbits) continue;                       // structurally empty row (uniform over the workgroup)
            __syncthreads();                              // previous slice consumed / halo written
            for (int i = threadIdx.x; i < TS * AFL; i += 256) {
                const int ln = i & 63, c = (i >> 6) & 3, tp = i >> 8;
                asl[(tp * 64 + ln) * 4 + c] = fr[((size_t)((row * KS + tp) * 4 + c) * a.nbt) * 64 + ln];
            }
            __syncthreads();
            const int tz = row / KS, ty = row - tz * KS;
            const int base = (tz * HY + ty) * HX + ly + lx;
#pragma unroll
            for (int tx = 0; tx < KS; ++tx) {
                if (!((rowbits >> tx) & 1u)) continue;
                const f4 av = *reinterpret_cast<const f4*>(asl + (tx * 64 + lane) * 4);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float bv = hg[4 * cc * HALO + base + t * HX * HY + tx];
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cc], bv, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (ox < a.W && oy < a.H && b < a.B) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int oz = bz4 + t;
            if (oz >= a.D) continue;
            const size_t v = ((size_t)oz * a.H + oy) * a.W + ox;
            const float e[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = 16 * nb + 4 * g + q;
                if (o < a.cout) a.out[((size_t)b * a.cout + o) * V3 + v] = e[q];
            }
        }
    }
}


// K = 3 form of k_conv3d_s1 for the small grids of the head under training (5^3 .. 20^3, where every layer is a k3 convolution since
// the x2 upsamplings and the stride-2 layers were folded into k3 weights): a chunk's WHOLE weight slice (27 taps x 16 x 16) goes to
// LDS with its halo, so a chunk costs two barriers instead of 2 x 9, and the next chunk's halo + weights are requested into registers
// (38 + 27 loads per lane, all in flight) before the current chunk's MFMAs -- the first form spent its time on 7+ dependent global-load
// phases per chunk (enc.conv3, 16 chunks on 64 workgroups: 230 us for 55 MMAC).  Same tiling, same fragment layout, same mask semantics.
__global__ __launch_bounds__(256, 2) void k_conv3d_s1_k3(Conv1Args a) {
    constexpr int KS = 3, PAD = 1, TAPS = 27;
    constexpr int HX = 10, HY = 10, HZ = 6, HALO = HX * HY * HZ;
    constexpr int NH = (16 * HALO + 255) / 256;           // halo floats per lane (38)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                                   // [16][HALO]
    float* asl = smem + 16 * HALO;                        // [27 taps][64 lanes][4 c]
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int nbx = (a.W + 7) >> 3, nby = (a.H + 7) >> 3, nbz = (a.D + 3) >> 2;
    const int blk = blockIdx.x;
    const int b = blk / (nbx * nby * nbz), br = blk - b * nbx * nby * nbz;
    const int bz4 = (br / (nbx * nby)) * 4, by8 = ((br / nbx) % nby) * 8, bx8 = (br % nbx) * 8;
    const int ox = bx8 + (wave & 1) * 4 + (r & 3), oy = by8 + (wave >> 1) * 4 + (r >> 2);
    const int nb = blockIdx.y;
    const size_t V3 = (size_t)a.D * a.H * a.W;
    f4 acc[4];
    {
        f4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const int o = 16 * nb + 4 * g;
            b4.x = o < a.cout ? a.bias[o] : 0.f; b4.y = o + 1 < a.cout ? a.bias[o + 1] : 0.f;
            b4.z = o + 2 < a.cout ? a.bias[o + 2] : 0.f; b4.w = o + 3 < a.cout ? a.bias[o + 3] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = b4;
    }
    // per-lane halo cell of each of this lane's NH staging slots (the same for every chunk): global offset or -1
    unsigned hoff[NH];                                    // BYTE offset from the chunk's first channel (wave-uniform base + 32-bit lane offset)
#pragma unroll
    for (int u = 0; u < NH; ++u) {
        const int i = threadIdx.x + 256 * u, c = i / HALO, rem = i - c * HALO;
        const int z = bz4 - PAD + rem / (HX * HY), y = by8 - PAD + (rem / HX) % HY, x = bx8 - PAD + rem % HX;
        const bool ok = i < 16 * HALO && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        hoff[u] = ok ? (unsigned)((c * V3 + ((size_t)z * a.H + y) * a.W + x) * sizeof(float)) : 0xffffffffu;   // all ones: zero border
    }
    auto chunk_mask = [&](int ch) -> unsigned {           // 27 bits, in THIS launch's tap order (backward data: flipped)
        if (!a.mask) return (1u << TAPS) - 1u;
        const unsigned m = a.mask[(size_t)(a.mode ? nb * a.nchunks + ch : ch * a.nbt + nb) * 4];
        return a.mode ? __builtin_bitreverse32(m) >> (32 - TAPS) : m;
    };
    auto next_chunk = [&](int ch) { while (ch < a.nchunks && !chunk_mask(ch)) ++ch; return ch; };
    float hv[NH], av[TAPS];
    auto request = [&](int ch, unsigned m) {              // all loads of a chunk in flight
        // buffer loads: offsets past the chunk's existing channels and the all-ones border offset come back as 0.0 from the range check
        const int cleft = a.cin - 16 * ch;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + ((size_t)b * a.cin + 16 * ch) * V3), 0,
                                                          (int)((cleft < 16 ? cleft : 16) * V3 * sizeof(float)), 0x00020000);
#pragma unroll
        for (int u = 0; u < NH; ++u) hv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)hoff[u], 0, 0));
        const auto rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.frag + ((size_t)ch * TAPS * 4 * a.nbt + nb) * 64), 0,
                                                          (int)((TAPS * 4 - 1) * a.nbt * 64 + 64) * (int)sizeof(float), 0x00020000);
        const int fo = (int)(((threadIdx.x >> 6) * a.nbt * 64 + (threadIdx.x & 63)) * sizeof(float));
        const int fstep = (int)(4 * a.nbt * 64 * sizeof(float));
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
            av[tp] = ((m >> tp) & 1u) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, fo + tp * fstep, 0, 0)) : 0.f;
    };
    const int lx = ox - bx8, ly = (oy - by8) * HX;
    const float* hg = halo + g * HALO;
    int ch = next_chunk(0);
    unsigned m = ch < a.nchunks ? chunk_mask(ch) : 0u;
    if (ch < a.nchunks) request(ch, m);
    while (ch < a.nchunks) {
        __syncthreads();                                  // the previous chunk's MFMAs are done with the tiles
#pragma unroll
        for (int u = 0; u < NH; ++u)
            if (threadIdx.x + 256 * u < 16 * HALO) halo[threadIdx.x + 256 * u] = hv[u];
        {
            const int ln = threadIdx.x & 63, c = threadIdx.x >> 6;
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) asl[(tp * 64 + ln) * 4 + c] = av[tp];
        }
        __syncthreads();
        const unsigned mc = m;
        const int nx = next_chunk(ch + 1);
        if (nx < a.nchunks) { m = chunk_mask(nx); request(nx, m); }
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
            if (!((mc >> tp) & 1u)) continue;             // wave-uniform
            const int tz = tp / 9, ty = (tp / 3) % 3, tx = tp % 3;
            const int base = (tz * HY + ty) * HX + ly + lx + tx;
            const f4 aw = *reinterpret_cast<const f4*>(asl + (tp * 64 + lane) * 4);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[cc], hg[4 * cc * HALO + base + t * HX * HY], acc[t], 0, 0, 0);
            }
        }
        ch = nx;
    }
    if (ox < a.W && oy < a.H && b < a.B) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int oz = bz4 + t;
            if (oz >= a.D) continue;
            const size_t v = ((size_t)oz * a.H + oy) * a.W + ox;
            const float e[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = 16 * nb + 4 * g + q;
                if (o < a.cout) a.out[((size_t)b * a.cout + o) * V3 + v] = e[q];
            }
        }
    }
}

}  // namespace gnr_head

extern "C" size_t gnr_conv3d_same_workspace_bytes(int Cin, int Cout, int K) {
    if (Cin < 1 || Cout < 1 || K < 1) return 0;
    const int m = Cin > Cout ? Cin : Cout;
    const size_t nch = (m + 15) / 16, nbt = (m + 15) / 16;
    return nch * K * K * K * 4 * nbt * 64 * sizeof(float);
}

// mode 0: x [B][Cin][D][H][W] -> y [B][Cout][D][H][W] (+ bias [Cout] or NULL);  mode 1: x = dy [B][Cout][..] -> y = dx [B][Cin][..]
// (bias ignored).  w = the layer's canonical weights [Cout][Cin][K][K][K] on the device.  K = 3 or 5.
// GNR_CONV3D_FIRST_GEN (tests; ORed into `mode` / `K`, include/gnr.h): route a K = 3 call through the first-generation kernels (the path
// taken for volumes beyond the buffer loads' 32-bit offsets).  A per-call flag: the library keeps no switch.

extern "C" size_t gnr_conv3d_tap_mask_words(int Cin, int Cout) {
    if (Cin < 1 || Cout < 1) return 0;
    return (size_t)((Cin + 15) / 16) * ((Cout + 15) / 16) * 4;
}

// mask [Cin blocks][Cout blocks][4 words] <- which taps of each 16 x 16 weight block exist in `pattern` [Cout][Cin][K^3] (device, any
// non-zero = the weight exists).  Made once per layer shape; the masked entry points below skip the absent taps in all three directions.
extern "C" int gnr_conv3d_tap_mask(const float* pattern, unsigned* mask, int Cin, int Cout, int K, void* stream) {
    if (!pattern || !mask || Cin < 1 || Cout < 1 || (K != 3 && K != 5)) return GNR_ERR_ARG;
    const int nbi = (Cin + 15) / 16, nbo = (Cout + 15) / 16;
    hipLaunchKernelGGL(gnr_head::k_conv3d_tap_mask, dim3(nbi * nbo), dim3(64), 0, (hipStream_t)stream, pattern, mask, Cin, Cout, K * K * K, nbo);
    HCHK(hipGetLastError());
    return GNR_OK;
}

extern "C" int gnr_conv3d_same_masked(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int D, int H,
                                      int W, int K, int mode, const unsigned* mask, void* ws, size_t ws_bytes, void* stream);

extern "C" int gnr_conv3d_same(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int D, int H,
                               int W, int K, int mode, void* ws, size_t ws_bytes, void* stream) {
    return gnr_conv3d_same_masked(x, w, bias, y, B, Cin, Cout, D, H, W, K, mode, nullptr, ws, ws_bytes, stream);
}

// `mask` = gnr_conv3d_tap_mask of the layer's weight pattern (or NULL: dense): taps without a weight are skipped.
extern "C" int gnr_conv3d_same_masked(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int D, int H,
                                      int W, int K, int mode, const unsigned* mask, void* ws, size_t ws_bytes, void* stream) {
    const bool first_gen = (mode & GNR_CONV3D_FIRST_GEN) != 0;
    mode &= ~GNR_CONV3D_FIRST_GEN;
    if (!x || !w || !y || !ws) { snprintf(h_err, sizeof(h_err), "gnr_conv3d_same: null pointer"); return GNR_ERR_ARG; }
    if (B < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || (K != 3 && K != 5) || (mode != 0 && mode != 1)) {
        snprintf(h_err, sizeof(h_err), "gnr_conv3d_same: bad shape / mode (K must be 3 or 5)"); return GNR_ERR_SHAPE; }
    if (ws_bytes < gnr_conv3d_same_workspace_bytes(Cin, Cout, K)) { snprintf(h_err, sizeof(h_err), "gnr_conv3d_same: workspace too small"); return GNR_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const int cin = mode ? Cout : Cin, cout = mode ? Cin : Cout;
    const int nchunks = (cin + 15) / 16, nbt = (cout + 15) / 16, K3 = K * K * K;
    float* frag = (float*)ws;
    {
        const long n = (long)nchunks * K3 * 4 * nbt * 64;
        HeadScope hs("k_pack_conv3d_frag@gnr_conv3d_same", stream);
        hipLaunchKernelGGL(gnr_head::k_pack_conv3d_frag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, frag, Cin, Cout, K, mode, nchunks, nbt);
        HCHK(hipGetLastError());
    }
    gnr_head::Conv1Args a{x, frag, mode ? nullptr : bias, y, B, cin, cout, D, H, W, nchunks, nbt, mask, mode};
    const long blocks = (long)B * ((W + 7) / 8) * ((H + 7) / 8) * ((D + 3) / 4);
    HeadScope hs(mode ? "k_conv3d_s1.bwd_data@gnr_conv3d_same" : "k_conv3d_s1.fwd@gnr_conv3d_same", stream);
    if (K == 5) {
        const size_t lds = (16 * (12 * 12 * 8) + 5 * 256) * sizeof(float);
        static std::atomic<unsigned long long> attr{0};
        if (head_attr_needed(attr)) { HCHK(hipFuncSetAttribute((const void*)gnr_head::k_conv3d_s1<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
        hipLaunchKernelGGL(gnr_head::k_conv3d_s1<5>, dim3((unsigned)blocks, nbt), dim3(256), lds, st, a);
    } else if (first_gen || (size_t)16 * D * H * W * sizeof(float) >= ((size_t)1 << 31) || (size_t)27 * 4 * nbt * 64 * sizeof(float) >= ((size_t)1 << 31)) {
        // a chunk's 16 channels do not fit the 32-bit byte offsets of the buffer loads (> 32 M voxels): the first-generation kernel
        const size_t lds = (16 * (10 * 10 * 6) + 3 * 256) * sizeof(float);
        static std::atomic<unsigned long long> attr{0};
        if (head_attr_needed(attr)) { HCHK(hipFuncSetAttribute((const void*)gnr_head::k_conv3d_s1<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
        hipLaunchKernelGGL(gnr_head::k_conv3d_s1<3>, dim3((unsigned)blocks, nbt), dim3(256), lds, st, a);
    } else {
        const size_t lds = (16 * (10 * 10 * 6) + 27 * 256) * sizeof(float);
        static std::atomic<unsigned long long> attr{0};
        if (head_attr_needed(attr)) { HCHK(hipFuncSetAttribute((const void*)gnr_head::k_conv3d_s1_k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
        hipLaunchKernelGGL(gnr_head::k_conv3d_s1_k3, dim3((unsigned)blocks, nbt), dim3(256), lds, st, a);
    }
    HCHK(hipGetLastError());
    return GNR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the same stride-1 convolutions, LDS-staged (replaces the one-wavefront-per-(tap, block, voxel chunk)
// kernel above for K = 3 / 5: that one re-derives voxel coordinates per element and fetches both operands from global
// memory per k-step, 9 TFLOP/s):
//   dW[o][i][tap] = sum_{b, voxel} dy[b][o][voxel] * x[b][i][voxel + tap - K/2]
// A workgroup owns one (16 input channels, 16 output channels) pair and sweeps 8x8x4 voxel bricks: the brick's x halo
// (channel-minor, + zero border) and its dy tile go to LDS once, then the four wavefronts split the K^3 taps and run
// 64 k-steps (4 voxels each) per tap with the voxel axis as the MFMA K: A = dy [o][voxel], B = x [voxel + tap][i], both
// read conflict-free (64 consecutive floats per wavefront instruction).  The 16x16 accumulators of a wavefront's <= 32 taps
// stay in registers across all bricks of the workgroup; every workgroup stores its partial blocks, k_conv3d_wgrad_reduce
// sums them into dw.
// ---------------------------------------------------------------------------------------------------------
namespace gnr_head {

template <int KS>
__global__ __launch_bounds__(256, 1) void k_conv3d_wgrad_s1(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                            int B, int Cin, int Cout, int D, int H, int W, int nbi, const unsigned* __restrict__ mask) {
    constexpr int PAD = KS / 2, TAPS = KS * KS * KS, TPW = (TAPS + 3) / 4;      // taps per wavefront
    constexpr int HX = 8 + KS - 1, HY = 8 + KS - 1, HZ = 4 + KS - 1, HALO = HX * HY * HZ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                                   // [HALO cells][16 input channels]
    float* dyt = smem + HALO * 16;                        // [256 voxels][16 output channels]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.y, bi = pair % nbi, bo = pair / nbi;
    const int nbx = (W + 7) >> 3, nby = (H + 7) >> 3, nbz = (D + 3) >> 2;
    const int nbricks = B * nbx * nby * nbz;
    const size_t V3 = (size_t)D * H * W;
    f4 acc[TPW];
    int off[TPW];
    unsigned act = 0;                                     // bit j: this wavefront's tap wave + 4 j has weights in block (bi, bo)
    const int nbo = gridDim.y / nbi;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        acc[j] = f4{0.f, 0.f, 0.f, 0.f};
        const int tap = __builtin_amdgcn_readfirstlane(wave) + 4 * j, t = tap < TAPS ? tap : 0;
        off[j] = ((t / (KS * KS)) * HY + (t / KS) % KS) * HX + t % KS;
        const unsigned on = mask ? (tap < TAPS ? (mask[((size_t)bi * nbo + bo) * 4 + (t >> 5)] >> (t & 31)) & 1u : 0u) : 1u;
        act |= on << j;
    }
    if (mask && !__builtin_amdgcn_readfirstlane(mask[((size_t)bi * nbo + bo) * 4] | mask[((size_t)bi * nbo + bo) * 4 + 1] |
                                                mask[((size_t)bi * nbo + bo) * 4 + 2] | mask[((size_t)bi * nbo + bo) * 4 + 3])) {
        float* pp0 = part + (((size_t)pair * gridDim.x + blockIdx.x) * TAPS) * 256;       // nothing joins the two blocks: zeros
        for (int i = threadIdx.x; i < TAPS * 256; i += 256) pp0[i] = 0.f;
        return;
    }
    for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
        const int b = brick / (nbx * nby * nbz), br = brick - b * nbx * nby * nbz;
        const int bz4 = (br / (nbx * nby)) * 4, by8 = ((br / nbx) % nby) * 8, bx8 = (br % nbx) * 8;
        __syncthreads();                                  // the previous brick's MFMAs are done with the tiles
        {
            const float* xin = x + ((size_t)b * Cin + 16 * bi) * V3;
            for (int i0 = threadIdx.x; i0 < 16 * HALO; i0 += 256 * 8) {     // eight loads in flight per lane, then the LDS stores
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 256 * u, c = i / HALO, cell = i - c * HALO;
                    const int z = bz4 - PAD + cell / (HX * HY), y = by8 - PAD + (cell / HX) % HY, xx = bx8 - PAD + cell % HX;
                    const bool ok = i < 16 * HALO && 16 * bi + c < Cin && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
                    v[u] = ok ? xin[(size_t)c * V3 + ((size_t)z * H + y) * W + xx] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 256 * u, c = i / HALO, cell = i - c * HALO;
                    if (i < 16 * HALO) halo[cell * 16 + c] = v[u];
                }
            }
            const float* din = dy + ((size_t)b * Cout + 16 * bo) * V3;
            {
                float v[16];
#pragma unroll
                for (int o = 0; o < 16; ++o) {                // lane = voxel of the brick, sixteen output channels in flight
                    const int vx = threadIdx.x;
                    const int z = bz4 + (vx >> 6), y = by8 + ((vx >> 3) & 7), xx = bx8 + (vx & 7);
                    const bool ok = 16 * bo + o < Cout && z < D && y < H && xx < W;
                    v[o] = ok ? din[(size_t)o * V3 + ((size_t)z * H + y) * W + xx] : 0.f;
                }
#pragma unroll
                for (int o = 0; o < 16; ++o) dyt[threadIdx.x * 16 + o] = v[o];
            }
        }
        __syncthreads();
        if (act == (TPW >= 32 ? ~0u : (1u << (TPW & 31)) - 1u)) {                    // dense block: every tap of this wavefront (straight-line MFMA chain)
#pragma unroll 2
            for (int s = 0; s < 64; ++s) {                // k-step s = voxels 4s .. 4s+3 (consecutive x)
                const float av = dyt[64 * s + lane];      // A: row o = lane % 16, k = lane / 16  <->  dyt[(4s + k) * 16 + o]
                const int cellbase = ((s >> 4) * HY + ((s >> 1) & 7)) * HX + 4 * (s & 1);
                const float* hb = halo + cellbase * 16 + lane;    // B: col i = lane % 16, k = lane / 16  <->  halo[(cell + k) * 16 + i]
#pragma unroll
                for (int j = 0; j < TPW; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, hb[off[j] * 16], acc[j], 0, 0, 0);
            }
        } else if (act) {                                 // structurally sparse block: only the taps that hold a weight
            for (int s = 0; s < 64; ++s) {
                const float av = dyt[64 * s + lane];
                const int cellbase = ((s >> 4) * HY + ((s >> 1) & 7)) * HX + 4 * (s & 1);
                const float* hb = halo + cellbase * 16 + lane;
#pragma unroll
                for (int j = 0; j < TPW; ++j)
                    if ((act >> j) & 1u) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, hb[off[j] * 16], acc[j], 0, 0, 0);
            }
        }
    }
    // partial sums of this workgroup: part[pair][workgroup][tap][o = 4 kg + q][i = lane % 16] (plain stores; hundreds of
    // workgroups adding atomically into the same few thousand weights serialise at the memory side: 1.2 ms of a 1.5 ms kernel)
    float* pp = part + (((size_t)pair * gridDim.x + blockIdx.x) * TAPS) * 256;
    const int kg = lane >> 4;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tap = wave + 4 * j;
        if (tap >= TAPS) continue;
        const float e[4] = {acc[j].x, acc[j].y, acc[j].z, acc[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) pp[(size_t)tap * 256 + (4 * kg + q) * 16 + (lane & 15)] = e[q];
    }
}

// K = 3 form of the weight-gradient kernel (every layer of the head under training is a k3 convolution on a 5^3 .. 20^3 grid since the
// upsamplings / strides were folded): the brick's halo and dy tile are REQUESTED into registers (38 + 16 buffer loads per lane, all in
// flight; the zero border and the channels past Cin / Cout come back as 0.0 from the descriptor's range check -- no predicated loads)
// while the previous brick's MFMAs run, lanes take the channel as their fast index so that the LDS stores are linear, and a structurally
// sparse block (tap mask) runs tap-outer over its few present taps.  Same partial-block layout as k_conv3d_wgrad_s1<3>.
__global__ __launch_bounds__(256, 1) void k_conv3d_wgrad_k3(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                            int B, int Cin, int Cout, int D, int H, int W, int nbi, const unsigned* __restrict__ mask) {
    constexpr int KS = 3, PAD = 1, TAPS = 27, TPW = 7;
    constexpr int HX = 10, HY = 10, HZ = 6, HALO = HX * HY * HZ;
    constexpr int NH = (HALO + 15) / 16;                  // halo cells per lane (38): lane = (cell group, channel)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                                   // [HALO cells][16 input channels]
    float* dyt = smem + HALO * 16;                        // [256 voxels][16 output channels]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = blockIdx.y, bi = pair % nbi, bo = pair / nbi, nbo = gridDim.y / nbi;
    const int nbx = (W + 7) >> 3, nby = (H + 7) >> 3, nbz = (D + 3) >> 2;
    const int nbricks = B * nbx * nby * nbz;
    const size_t V3 = (size_t)D * H * W;
    float* pp = part + (((size_t)pair * gridDim.x + blockIdx.x) * TAPS) * 256;
    const unsigned mword = mask ? mask[((size_t)bi * nbo + bo) * 4] : (1u << TAPS) - 1u;
    if (!mword) {                                         // nothing joins the two channel blocks: zeros
        for (int i = threadIdx.x; i < TAPS * 256; i += 256) pp[i] = 0.f;
        return;
    }
    f4 acc[TPW];
    int off[TPW];
    unsigned act = 0;                                     // bit j: tap wave + 4 j exists in block (bi, bo)
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        acc[j] = f4{0.f, 0.f, 0.f, 0.f};
        const int tap = wave + 4 * j, t = tap < TAPS ? tap : 0;
        off[j] = (((t / 9) * HY + (t / 3) % 3) * HX + t % 3) * 16;
        act |= (tap < TAPS ? (mword >> t) & 1u : (mask ? 0u : 1u)) << j;
    }
    const int hc = threadIdx.x & 15, hg = threadIdx.x >> 4;               // staging: channel, first cell
    const int cin_left = Cin - 16 * bi, cout_left = Cout - 16 * bo;
    float hv[NH], dv[16];
    auto request = [&](int brick) {
        const int b = brick / (nbx * nby * nbz), br = brick - b * nbx * nby * nbz;
        const int bz4 = (br / (nbx * nby)) * 4, by8 = ((br / nbx) % nby) * 8, bx8 = (br % nbx) * 8;
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + ((size_t)b * Cin + 16 * bi) * V3), 0,
                                                          (int)((cin_left < 16 ? cin_left : 16) * V3 * sizeof(float)), 0x00020000);
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            const int cell = hg + 16 * u;
            const int z = bz4 - PAD + cell / (HX * HY), y = by8 - PAD + (cell / HX) % HY, xx = bx8 - PAD + cell % HX;
            const bool ok = cell < HALO && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
            const int o = ok ? (int)((hc * V3 + ((size_t)z * H + y) * W + xx) * sizeof(float)) : -1;
            hv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, 0));
        }
        const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + ((size_t)b * Cout + 16 * bo) * V3), 0,
                                                          (int)((cout_left < 16 ? cout_left : 16) * V3 * sizeof(float)), 0x00020000);
        const int vx = threadIdx.x;
        const int z = bz4 + (vx >> 6), y = by8 + ((vx >> 3) & 7), xx = bx8 + (vx & 7);
        const bool ok = z < D && y < H && xx < W;
        const int vo = (int)((((size_t)z * H + y) * W + xx) * sizeof(float));
#pragma unroll
        for (int o = 0; o < 16; ++o)
            dv[o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, ok ? vo + (int)(o * V3 * sizeof(float)) : -1, 0, 0));
    };
    int brick = blockIdx.x;
    if (brick < nbricks) request(brick);
    for (; brick < nbricks; brick += gridDim.x) {
        __syncthreads();                                  // the previous brick's MFMAs are done with the tiles
#pragma unroll
        for (int u = 0; u < NH; ++u)
            if (hg + 16 * u < HALO) halo[(hg + 16 * u) * 16 + hc] = hv[u];          // linear in the lane index
#pragma unroll
        for (int o = 0; o < 16; o += 4)
            *reinterpret_cast<f4*>(dyt + threadIdx.x * 16 + o) = f4{dv[o], dv[o + 1], dv[o + 2], dv[o + 3]};
        __syncthreads();
        if (brick + (int)gridDim.x < nbricks) request(brick + gridDim.x);
        if (act == (1u << TPW) - 1u) {                    // dense block: k-step outer, the wavefront's seven taps share the dy operand
#pragma unroll 2
            for (int s = 0; s < 64; ++s) {                // k-step s = voxels 4s .. 4s+3 (consecutive x)
                const float av = dyt[64 * s + lane];      // A: row o = lane % 16, k = lane / 16  <->  dyt[(4s + k) * 16 + o]
                const int cellbase = ((s >> 4) * HY + ((s >> 1) & 7)) * HX + 4 * (s & 1);
                const float* hb = halo + cellbase * 16 + lane;    // B: col i = lane % 16, k = lane / 16  <->  halo[(cell + k) * 16 + i]
#pragma unroll
                for (int j = 0; j < TPW; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, hb[off[j]], acc[j], 0, 0, 0);
            }
        } else {                                          // structurally sparse block: tap outer, only the taps that hold a weight
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (!((act >> j) & 1u)) continue;
                const float* hj = halo + off[j] + lane;
                f4 a0 = acc[j], a1 = f4{0.f, 0.f, 0.f, 0.f};      // two chains: the dependent-MFMA latency is not hidden by other taps here
#pragma unroll 4
                for (int s = 0; s < 64; s += 2) {
                    const int c0 = (((s >> 4) * HY + ((s >> 1) & 7)) * HX) * 16;      // s even: 4 * (s & 1) = 0; s + 1: + 4 cells
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dyt[64 * s + lane], hj[c0], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dyt[64 * s + 64 + lane], hj[c0 + 64], a1, 0, 0, 0);
                }
                acc[j] = a0 + a1;
            }
        }
    }
    const int kg = lane >> 4;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tap = wave + 4 * j;
        if (tap >= TAPS) continue;
        const float e[4] = {acc[j].x, acc[j].y, acc[j].z, acc[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) pp[(size_t)tap * 256 + (4 * kg + q) * 16 + (lane & 15)] = e[q];
    }
}

// dw[o][i][tap] += sum over the workgroups' partial blocks
__global__ void k_conv3d_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int Cin, int Cout, int taps, int nbi,
                                      int npairs, int nwg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;              // (pair, tap, o16, i16)
    if (idx >= npairs * taps * 256) return;
    const int pair = idx / (taps * 256), rem = idx - pair * taps * 256;
    const int tap = rem >> 8, o16 = (rem >> 4) & 15, i16 = rem & 15;
    const int bi = pair % nbi, bo = pair / nbi;
    const int o = 16 * bo + o16, i = 16 * bi + i16;
    if (o >= Cout || i >= Cin) return;
    const float* p = part + ((size_t)pair * nwg * taps) * 256 + rem;
    const size_t st = (size_t)taps * 256;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};                   // four interleaved sums in a fixed order: the loads of a lane overlap (up to 256 blocks)
    int w = 0;
    for (; w + 8 <= nwg; w += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(w + u) * st];
#pragma unroll
        for (int u = 0; u < 8; ++u) s4[u & 3] += v[u];
    }
    for (; w < nwg; ++w) s4[w & 3] += p[(size_t)w * st];
    dw[((size_t)o * Cin + i) * taps + tap] += (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

}  // namespace gnr_head

static long wgrad_grid(int B, int Cin, int Cout, int D, int H, int W) {
    const int nbi = (Cin + 15) / 16, nbo = (Cout + 15) / 16;
    const long nbricks = (long)B * ((W + 7) / 8) * ((H + 7) / 8) * ((D + 3) / 4);
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    long gx = (cus + nbi * nbo - 1) / (nbi * nbo);
    if (gx > nbricks) gx = nbricks;
    return gx < 1 ? 1 : gx;
}

extern "C" size_t gnr_conv3d_same_bwd_weight_workspace_bytes(int B, int Cin, int Cout, int D, int H, int W, int K) {
    if (B < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || K < 1) return 0;
    const size_t npairs = (size_t)((Cin + 15) / 16) * ((Cout + 15) / 16);
    return npairs * (size_t)wgrad_grid(B, Cin, Cout, D, H, W) * K * K * K * 256 * sizeof(float);
}

// dw [Cout][Cin][K][K][K] is ACCUMULATED (zero it first); K = 3 or 5; workspace = gnr_conv3d_same_bwd_weight_workspace_bytes(...).
extern "C" int gnr_conv3d_same_bwd_weight_masked(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int D, int H, int W,
                                                 int K, const unsigned* mask, void* ws, size_t ws_bytes, void* stream);

extern "C" int gnr_conv3d_same_bwd_weight(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int D, int H, int W, int K,
                                          void* ws, size_t ws_bytes, void* stream) {
    return gnr_conv3d_same_bwd_weight_masked(x, dy, dw, B, Cin, Cout, D, H, W, K, nullptr, ws, ws_bytes, stream);
}

// `mask` = gnr_conv3d_tap_mask of the layer's weight pattern (or NULL): the gradient of an absent weight is not computed (stays as it was in dw).
extern "C" int gnr_conv3d_same_bwd_weight_masked(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int D, int H, int W,
                                                 int K, const unsigned* mask, void* ws, size_t ws_bytes, void* stream) {
    const bool first_gen = (K & GNR_CONV3D_FIRST_GEN) != 0;
    K &= ~GNR_CONV3D_FIRST_GEN;
    if (!x || !dy || !dw || !ws || B < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || (K != 3 && K != 5)) return GNR_ERR_ARG;
    if (ws_bytes < gnr_conv3d_same_bwd_weight_workspace_bytes(B, Cin, Cout, D, H, W, K)) return GNR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nbi = (Cin + 15) / 16, nbo = (Cout + 15) / 16;
    const long gx = wgrad_grid(B, Cin, Cout, D, H, W);
    float* part = (float*)ws;
    {
        HeadScope hs("k_conv3d_wgrad_s1@gnr_conv3d_same_bwd_weight", stream);
        if (K == 5) {
            const size_t lds = (16 * (12 * 12 * 8) + 16 * 256) * sizeof(float);
            static std::atomic<unsigned long long> attr{0};
            if (head_attr_needed(attr)) { HCHK(hipFuncSetAttribute((const void*)gnr_head::k_conv3d_wgrad_s1<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
            hipLaunchKernelGGL(gnr_head::k_conv3d_wgrad_s1<5>, dim3((unsigned)gx, nbi * nbo), dim3(256), lds, st, x, dy, part, B, Cin, Cout, D, H, W, nbi, mask);
        } else if (first_gen || (size_t)16 * D * H * W * sizeof(float) >= ((size_t)1 << 31)) {      // beyond the buffer loads' 32-bit byte offsets
            const size_t lds = (16 * (10 * 10 * 6) + 16 * 256) * sizeof(float);
            static std::atomic<unsigned long long> attr{0};
            if (head_attr_needed(attr)) { HCHK(hipFuncSetAttribute((const void*)gnr_head::k_conv3d_wgrad_s1<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
            hipLaunchKernelGGL(gnr_head::k_conv3d_wgrad_s1<3>, dim3((unsigned)gx, nbi * nbo), dim3(256), lds, st, x, dy, part, B, Cin, Cout, D, H, W, nbi, mask);
        } else {
            const size_t lds = (16 * (10 * 10 * 6) + 16 * 256) * sizeof(float);
            static std::atomic<unsigned long long> attr{0};
            if (head_attr_needed(attr)) { HCHK(hipFuncSetAttribute((const void*)gnr_head::k_conv3d_wgrad_k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
            hipLaunchKernelGGL(gnr_head::k_conv3d_wgrad_k3, dim3((unsigned)gx, nbi * nbo), dim3(256), lds, st, x, dy, part, B, Cin, Cout, D, H, W, nbi, mask);
        }
        HCHK(hipGetLastError());
    }
    {
        HeadScope hs("k_conv3d_wgrad_reduce@gnr_conv3d_same_bwd_weight", stream);
        const int K3 = K * K * K, n = nbi * nbo * K3 * 256;
        hipLaunchKernelGGL(gnr_head::k_conv3d_wgrad_reduce, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)part, dw, Cin, Cout, K3, nbi, nbi * nbo, (int)gx);
        HCHK(hipGetLastError());
    }
    return GNR_OK;
}
