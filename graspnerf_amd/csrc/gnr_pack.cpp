// Host-side weight packer: reference state-dict order (canonical blob) -> MFMA A-fragment
// blob consumed by the gfx950 kernels.  See gnr_layout.h for the execution model.
// Parameter shapes/order: ref dist_decoder.py:64-88, aggregate_net.py:29-33, ibrnet.py:382-423.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

#include "gnr_layout.h"
#include "../../include/gnr.h"

namespace {
using namespace gnr;
using IdxFn = std::function<int(int, int)>;

inline int nat_in(int j, int g) { return 16 * (j / 4) + 4 * g + (j % 4); }
inline int nat_out(int nb, int i) { return 16 * nb + i; }
// layout of the 35-wide colour feature x = [r,g,b, img_feats(32)] in 9 slots
inline int xfeat(int j, int g) { return j < 8 ? 3 + 8 * g + j : (g < 3 ? g : -1); }

// Scaled-ELU convention (saves one VALU multiply per activation in k_chain): a layer that feeds an
// ELU emits x' = log2(e) * (W x + b), the kernel computes u~ = med3(x', log2e*(2^x' - 1), 0) = log2e * ELU(x),
// and every consumer of u~ has the factor divided out of its weight columns.  `oscale` multiplies the
// rows (and bias), `iscale(i)` is the factor carried by logical input i (weights are divided by it).
using ScaleFn = std::function<double(int)>;
constexpr double LOG2E = 1.4426950408889634;
const ScaleFn kTrue = [](int) { return 1.0; };
const ScaleFn kTilde = [](int) { return LOG2E; };

// frag[(j,nb,lane)] = oscale * W[psi(nb, lane&15)][phi(j, lane>>4)] / iscale(phi)
void pack_frag(float* dst, const float* W, int ldw, int J, int NB, const IdxFn& phi, const IdxFn& psi,
               double oscale = 1.0, const ScaleFn& iscale = kTrue) {
    std::memset(dst, 0, sizeof(float) * frag_floats(J, NB));
    for (int j = 0; j < J; ++j)
        for (int nb = 0; nb < NB; ++nb)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = psi(nb, lane & 15), i = phi(j, lane >> 4);
                const float v = (o >= 0 && i >= 0) ? (float)((double)W[o * ldw + i] * oscale / iscale(i)) : 0.f;
                int idx;
                if (NB == 1) idx = ((j / 4) * 64 + lane) * 4 + (j % 4);
                else if (NB == 3) idx = (j * 64 + lane) * 4 + nb;
                else idx = (j * 64 + lane) * NB + nb;
                dst[idx] = v;
            }
}

void pack_bias(float* dst, const float* b, int NB, const IdxFn& psi, double oscale = 1.0) {
    for (int nb = 0; nb < NB; ++nb)
        for (int i = 0; i < 16; ++i) {
            const int o = psi(nb, i);
            dst[nb * 16 + i] = o >= 0 ? (float)((double)b[o] * oscale) : 0.f;   // i = 4g + reg
        }
}

// ---- fp16 pairs (C16 section, gnr_layout.h) ---------------------------------------------------------------------
inline uint16_t f32_to_f16(float f) {                       // round to nearest even, subnormals kept, overflow -> inf
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, ef = (x >> 23) & 0xffu, mant = x & 0x7fffffu;
    if (ef == 0xffu) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0u));
    const int e = (int)ef - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        const uint32_t m = mant | 0x800000u;
        const int shift = 14 - e;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}
inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) return std::ldexp((float)m, -24) * (sign ? -1.f : 1.f);
    if (e == 31) return sign ? -INFINITY : INFINITY;
    uint32_t x = sign | ((e - 15 + 127) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

inline float frag_at(const float* frag, int NB, int j, int nb, int lane) {     // the element pack_frag wrote for (j, nb, lane)
    if (NB == 1) return frag[((j / 4) * 64 + lane) * 4 + (j % 4)];
    if (NB == 3) return frag[(j * 64 + lane) * 4 + nb];
    return frag[(j * 64 + lane) * NB + nb];
}

// which k-steps of a layer become K32 / K16 pair blocks and which stay fp32 (k_chain's call sites use the same split)
// which k-steps of a layer become K32 pair blocks (lists of <= 8 k-steps; a short list is zero-padded) and which stay fp32
// fragments (k_chain's call sites use the same split)
struct C16Plan { int off, J, NB; std::vector<std::vector<int>> k32; std::vector<int> rest; };

// fp32 fragment of one layer (src) -> its C16 form at dst.  false: a weight is outside the fp16 range.
bool to_pairs(float* dst, const float* src, const C16Plan& pl) {
    const int NB = pl.NB, Jr = (int)pl.rest.size();
    std::vector<float> out(pl.k32.size() * pk::k32_floats(NB) + (Jr ? frag_floats(Jr, NB) : 0), 0.f);
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out.data());
    bool ok = true;
    size_t pos = 0;                                                              // floats
    for (const std::vector<int>& ks : pl.k32) {
        for (int nb = 0; nb < NB; ++nb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < (int)ks.size(); ++i) {
                    const float w = frag_at(src, NB, ks[i], nb, lane);
                    const uint16_t h = f32_to_f16(w);
                    const float hf = f16_to_f32(h);
                    if (!std::isfinite(hf)) { ok = false; continue; }
                    uint16_t* base = o16 + 2 * pos;
                    base[(((size_t)nb * 2 + 0) * 64 + lane) * 8 + i] = h;
                    base[(((size_t)nb * 2 + 1) * 64 + lane) * 8 + i] = f32_to_f16((w - hf) * 2048.f);
                }
        pos += pk::k32_floats(NB);
    }
    for (int jr = 0; jr < Jr; ++jr)
        for (int nb = 0; nb < NB; ++nb)
            for (int lane = 0; lane < 64; ++lane) {
                const float v = frag_at(src, NB, pl.rest[jr], nb, lane);
                size_t idx;
                if (NB == 1) idx = ((jr / 4) * 64 + lane) * 4 + (jr % 4);
                else if (NB == 3) idx = (jr * 64 + lane) * 4 + nb;
                else idx = ((size_t)jr * 64 + lane) * NB + nb;
                out[pos + idx] = v;
            }
    std::memcpy(dst, out.data(), sizeof(float) * out.size());
    return ok;
}

std::vector<int> ksteps(int k0, int n) { std::vector<int> v(n); for (int i = 0; i < n; ++i) v[i] = k0 + i; return v; }

std::vector<C16Plan> c16_plan() {
    using namespace gnr::pk;
    std::vector<C16Plan> v;
    for (int br = 0; br < 3; ++br) {
        v.push_back({DEC1 + br * frag_floats(8, 2), 8, 2, {ksteps(0, 8)}, {}});
        v.push_back({DEC2 + br * frag_floats(8, 2), 8, 2, {ksteps(0, 8)}, {}});
    }
    v.push_back({PE1, 9, 2, {ksteps(0, 8)}, {8}});                               // ray features | (hit, vis)
    v.push_back({NR1, 8, 1, {ksteps(0, 8)}, {}});
    v.push_back({BASE1, 17, 4, {ksteps(0, 8), ksteps(9, 8)}, {8}});              // x[0..7] | e1[0..7] | x[8] (rgb)
    v.push_back({BASE2, 16, 2, {ksteps(0, 8), ksteps(8, 8)}, {}});
    v.push_back({VIS1, 8, 2, {ksteps(0, 8)}, {}});
    v.push_back({VIS2, 8, 2, {ksteps(0, 8)}, {}});
    v.push_back({VISB1, 8, 2, {ksteps(0, 8)}, {}});
    v.push_back({RGB1, 10, 1, {ksteps(0, 8)}, {8, 9}});
    v.push_back({HOIST, 36, 4, {ksteps(0, 8), ksteps(8, 8), ksteps(16, 8), ksteps(24, 8), ksteps(32, 4)}, {}});      // 4-k-step tail zero-padded
    v.push_back({GEO1, 23, 4, {ksteps(0, 8), ksteps(8, 8), ksteps(16, 7)}, {}});                                     // 7-k-step tail zero-padded
    v.push_back({GEO2, 16, 1, {ksteps(0, 8), ksteps(8, 8)}, {}});
    v.push_back({DECV1, 8, 2, {ksteps(0, 8)}, {}});
    v.push_back({DECV2, 8, 2, {ksteps(0, 8)}, {}});
    return v;                                                // RDF1 (one k-step), RDF2 and RGB2 (4 k-steps) stay fp32
}

// C16 section of a packed blob from its CHAIN section: the same slots (two of them grown, gnr_layout.h c16_off) with the wide
// layers' fragments as fp16 pairs: what k_chain stages into LDS
int build_c16(float* p) {
    using namespace gnr;
    std::memset(p + pk::C16, 0, sizeof(float) * pk::C16_END);
    const int cuts[4] = {0, pk::GEO1, pk::GEO2, pk::CHAIN_END};                   // the slot behind each grown layer starts a new run
    for (int r = 0; r < 3; ++r)
        std::memcpy(p + pk::C16 + pk::c16_off(cuts[r]), p + cuts[r], sizeof(float) * (cuts[r + 1] - cuts[r]));
    // A weight beyond the fp16 range (|w| >= 65 520) has no pair: the blob says so (T_VIS + 2, in both images) and k_chain's
    // pair kernels hand every launch with this blob to their fp32-MFMA twins (range guard, gnr_kernels.hip).  The flag is
    // sticky over gnr_pack_weights -> gnr_pack_vis_decoder (which re-runs this on the same blob).
    bool ok = true;
    for (const C16Plan& pl : c16_plan())
        if (!to_pairs(p + pk::C16 + pk::c16_off(pl.off), p + pl.off, pl)) ok = false;
    if (!ok) p[pk::T_VIS + 2] = 1.f;
    p[pk::C16 + pk::c16_off(pk::T_VIS) + 2] = p[pk::T_VIS + 2];
    return GNR_OK;
}

// per-lane-group table of one output row over a natural-layout input of J slots: T[g][j]
void pack_row(float* dst, const float* wrow, int J, double iscale = 1.0) {
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < J; ++j) dst[g * J + j] = (float)((double)wrow[nat_in(j, g)] / iscale);
}
}  // namespace

// name -> float offset inside the packed blob (tests / tooling); -1 if unknown
extern "C" int gnr_layout_offset(const char* name) {
    using namespace gnr::pk;
    struct E { const char* n; int o; };
    static const E tab[] = {
        {"DEC1", DEC1}, {"DEC2", DEC2}, {"PE1", PE1}, {"RDF1", RDF1}, {"RDF2", RDF2}, {"NR1", NR1},
        {"BASE1", BASE1}, {"BASE2", BASE2}, {"VIS1", VIS1}, {"VIS2", VIS2}, {"VISB1", VISB1}, {"RGB1", RGB1},
        {"RGB2", RGB2}, {"HOIST", HOIST}, {"GEO1", GEO1}, {"GEO2", GEO2}, {"DECV1", DECV1}, {"DECV2", DECV2}, {"FRAG_END", FRAG_END},
        {"B_DEC1", B_DEC1}, {"B_DEC2", B_DEC2}, {"B_PE1", B_PE1}, {"B_RDF1", B_RDF1},
        {"B_RDF2", B_RDF2}, {"B_NR1", B_NR1}, {"B_HOIST", B_HOIST}, {"B_BASE2", B_BASE2}, {"B_VIS1", B_VIS1},
        {"B_VIS2", B_VIS2}, {"B_VISB1", B_VISB1}, {"B_RGB1", B_RGB1}, {"B_RGB2", B_RGB2}, {"B_GEO1", B_GEO1},
        {"B_GEO2", B_GEO2}, {"B_DECV1", B_DECV1}, {"B_DECV2", B_DECV2}, {"T_DECV3", T_DECV3}, {"T_VIS", T_VIS}, {"T_DEC3", T_DEC3}, {"T_DEC3_B", T_DEC3_B}, {"T_NR2", T_NR2}, {"T_VIS2R", T_VIS2R},
        {"T_VISB2", T_VISB2}, {"T_RGB3", T_RGB3}, {"T_SCAL", T_SCAL}, {"CHAIN_END", CHAIN_END},
        {"R_WQ", R_WQ}, {"R_WK", R_WK}, {"R_WV", R_WV}, {"R_WFC", R_WFC}, {"R_LNW", R_LNW}, {"R_LNB", R_LNB},
        {"R_OUT0W", R_OUT0W}, {"R_OUT0B", R_OUT0B}, {"R_OUT1W", R_OUT1W}, {"R_OUT1B", R_OUT1B},
        {"R_GEO2W", R_GEO2W}, {"R_GEO1E", R_GEO1E}, {"R_VARIANCE", R_VARIANCE}, {"R_PE", R_PE}, {"R_WQT", R_WQT},
        {"R_WKT", R_WKT}, {"R_WVT", R_WVT}, {"R_WFCT", R_WFCT}, {"R_GEO2WT", R_GEO2WT}, {"R_OUTVJP", R_OUTVJP}, {"R_OUTB", R_OUTB}, {"C16", C16}, {"C16_END", C16_END}, {"TOTAL", TOTAL}};
    if (!std::strncmp(name, "C16.", 4)) {                     // where a CHAIN-section name sits in the blob's C16 image
        for (const E& e : tab)
            if (!std::strcmp(e.n, name + 4) && e.o <= CHAIN_END) return C16 + c16_off(e.o);
        return -1;
    }
    for (const E& e : tab)
        if (!std::strcmp(e.n, name)) return e.o;
    return -1;
}

extern "C" int gnr_packed_bwd_floats(void) { return gnr::pkb::TOTAL; }

// Transposed fragments for the backward twins (see gnr_layout.h, namespace pkb)
extern "C" int gnr_pack_weights_bwd(const float* c, float* p) {
    if (!c || !p) return GNR_ERR_ARG;
    using namespace gnr;
    std::memset(p, 0, sizeof(float) * pkb::TOTAL);
    std::vector<float> T(1024);
    auto transpose32 = [&](const float* W) { for (int o = 0; o < 32; ++o) for (int i = 0; i < 32; ++i) T[i * 32 + o] = W[o * 32 + i]; };
    const IdxFn natI = nat_in, natO = nat_out;
    // output rows land in the gather layout: lane group g, register 4*nb + t  <->  ray channel 8g + 4nb + t
    const IdxFn gatherO = [](int nb, int i) { return 8 * (i / 4) + 4 * nb + (i % 4); };
    transpose32(c + can::MEAN2_W);
    pack_frag(p + pkb::DM_W2T, T.data(), 32, 8, 2, natI, natO);
    transpose32(c + can::MEAN0_W);
    pack_frag(p + pkb::DM_W1T, T.data(), 32, 8, 2, natI, gatherO);
    auto transposed = [&](const float* W, int rows, int cols) {       // W [rows][cols] -> [cols][rows]
        std::vector<float> t((size_t)rows * cols);
        for (int o = 0; o < rows; ++o) for (int i = 0; i < cols; ++i) t[(size_t)i * rows + o] = W[(size_t)o * cols + i];
        return t;
    };
    {   // geometry_fc.2^T [64][16], geometry_fc.0^T [86][64]; output rows in the Z-slot layout of the forward (gnr_pack_weights)
        const std::vector<float> g2t = transposed(c + can::GEO2_W, 16, 64), g1t = transposed(c + can::GEO0_W, 64, 86);
        pack_frag(p + pkb::GEO2T, g2t.data(), 16, 4, 4, natI, natO);
        const auto zslot = [](int j, int g) {
            if (j < 8) return nat_in(j, g);
            if (j < 16) return 32 + nat_in(j - 8, g);
            const int k = j - 16;
            if (g == 0) return k == 0 ? 64 : -1;
            return 65 + 3 * k + (g - 1);
        };
        pack_frag(p + pkb::GEO1T_A, g1t.data(), 64, 16, 4, natI, [&](int nb, int i) { return zslot(4 * nb + (i & 3), i >> 2); });
        pack_frag(p + pkb::GEO1T_B, g1t.data(), 64, 16, 1, natI, [&](int, int i) { return zslot(16 + (i & 3), i >> 2); });
    }
    {   // second view loop
        pack_frag(p + pkb::PE2F, c + can::PE2_W, 32, 8, 2, natI, natO);
        pack_bias(p + pkb::B_PE2, c + can::PE2_B, 2, natO);
        const std::vector<float> vb1t = transposed(c + can::VISB0_W, 32, 32), v2t = transposed(c + can::VIS2_W, 32, 32),
                                 v1t = transposed(c + can::VIS0_W, 32, 32), b2t = transposed(c + can::BASE2_W, 32, 64);
        pack_frag(p + pkb::VISB1T, vb1t.data(), 32, 8, 2, natI, natO);
        pack_frag(p + pkb::VIS2T, v2t.data(), 32, 8, 2, natI, natO);          // rows 0..31 of the 33 (first 32x32 block)
        pack_frag(p + pkb::VIS1T, v1t.data(), 32, 8, 2, natI, natO);
        pack_frag(p + pkb::BASE2T, b2t.data(), 32, 8, 4, natI, natO);
        std::vector<float> xt(35 * 64), et(32 * 64);
        for (int o = 0; o < 64; ++o) {
            for (int i = 0; i < 35; ++i) xt[i * 64 + o] = c[can::BASE0_W + o * 207 + 140 + i];
            for (int i = 0; i < 32; ++i) et[i * 64 + o] = c[can::BASE0_W + o * 207 + 175 + i];
        }
        const IdxFn xout = [](int nb, int i) {
            const int g = i >> 2, r = i & 3;
            if (nb == 0) return 3 + 8 * g + r;
            if (nb == 1) return 3 + 8 * g + 4 + r;
            return (r == 0 && g < 3) ? g : -1;
        };
        pack_frag(p + pkb::BASE1XT, xt.data(), 64, 16, 3, natI, xout);
        pack_frag(p + pkb::BASE1ET, et.data(), 64, 16, 2, natI, natO);
        // hoisted columns: statistic slot s (0..35) of lane group g <-> column 35*(s/9) + xfeat(s%9, g)
        std::vector<float> ht(140 * 64);
        for (int o = 0; o < 64; ++o)
            for (int i = 0; i < 140; ++i) ht[i * 64 + o] = c[can::BASE0_W + o * 207 + i];
        const auto sslot = [](int s, int g) { const int x = xfeat(s % 9, g); return x < 0 ? -1 : 35 * (s / 9) + x; };
        pack_frag(p + pkb::HOISTT_A, ht.data(), 64, 16, 4, natI, [&](int nb, int i) { return sslot(4 * nb + (i & 3), i >> 2); });
        pack_frag(p + pkb::HOISTT_B, ht.data(), 64, 16, 4, natI, [&](int nb, int i) { return sslot(16 + 4 * nb + (i & 3), i >> 2); });
        pack_frag(p + pkb::HOISTT_C, ht.data(), 64, 16, 1, natI, [&](int, int i) { return sslot(32 + (i & 3), i >> 2); });
    }
    {   // first view loop
        const int d0w[3] = {can::MEAN0_W, can::VAR0_W, can::AW0_W}, d2w[3] = {can::MEAN2_W, can::VAR2_W, can::AW2_W};
        for (int br = 0; br < 3; ++br) {
            const std::vector<float> t2 = transposed(c + d2w[br], 32, 32), t1 = transposed(c + d0w[br], 32, 32);
            pack_frag(p + pkb::DEC2T + br * 1024, t2.data(), 32, 8, 2, natI, natO);
            pack_frag(p + pkb::DEC1T + br * 1024, t1.data(), 32, 8, 2, natI, gatherO);
        }
        pack_frag(p + pkb::V1_PE2F, c + can::PE2_W, 32, 8, 2, natI, natO);
        pack_bias(p + pkb::V1_B_PE2, c + can::PE2_B, 2, natO);
        const std::vector<float> pe2t = transposed(c + can::PE2_W, 32, 32);
        pack_frag(p + pkb::PE2T, pe2t.data(), 32, 8, 2, natI, natO);
        std::vector<float> pe0t(32 * 32), hrow(32), vrow(32);
        for (int o = 0; o < 32; ++o) {
            for (int i = 0; i < 32; ++i) pe0t[i * 32 + o] = c[can::PE0_W + o * 34 + i];
            hrow[o] = c[can::PE0_W + o * 34 + 32];
            vrow[o] = c[can::PE0_W + o * 34 + 33];
        }
        pack_frag(p + pkb::PE0T, pe0t.data(), 32, 8, 2, natI, gatherO);
        pack_row(p + pkb::T_PE0HV, hrow.data(), 8);
        pack_row(p + pkb::T_PE0HV + 32, vrow.data(), 8);
        const std::vector<float> nr0t = transposed(c + can::NR0_W, 8, 32);            // [32][8]
        pack_frag(p + pkb::NR0T, nr0t.data(), 8, 4, 2, [](int j, int g) { const int f = 4 * g + j; return f < 8 ? f : -1; }, natO);
        const std::vector<float> r2t = transposed(c + can::RDF2_W, 35, 16);           // [16][35]
        pack_frag(p + pkb::RDF2T, r2t.data(), 35, 9, 1, [](int j, int g) { return xfeat(j, g); }, natO);
    }
    {   // colour head
        const std::vector<float> c2t = transposed(c + can::RGB2_W, 8, 16);            // [16][8]
        pack_frag(p + pkb::RGB2T, c2t.data(), 8, 4, 1, [](int j, int g) { const int f = 4 * g + j; return f < 8 ? f : -1; }, natO);
        std::vector<float> c0ht(32 * 16);
        for (int o = 0; o < 16; ++o)
            for (int i = 0; i < 32; ++i) c0ht[i * 16 + o] = c[can::RGB0_W + o * 37 + i];
        pack_frag(p + pkb::RGB0HT, c0ht.data(), 16, 4, 2, natI, natO);
        for (int g = 0; g < 4; ++g)
            for (int r = 0; r < 4; ++r) p[pkb::T_RGB0V + g * 4 + r] = c[can::RGB0_W + (4 * g + r) * 37 + 32];
    }
    return GNR_OK;
}

// The fourth decoder branch's transposed fragments (use_vis training) into a blob of gnr_pack_weights_bwd.  v as in
// gnr_pack_vis_decoder (state-dict order, 2 145 floats); output rows of vis_decoder.0^T land in the gather layout like DEC1T.
extern "C" int gnr_pack_vis_decoder_bwd(const float* v, float* p) {
    if (!v || !p) return GNR_ERR_ARG;
    using namespace gnr;
    const IdxFn natI = nat_in, natO = nat_out;
    const IdxFn gatherO = [](int nb, int i) { return 8 * (i / 4) + 4 * nb + (i % 4); };
    std::vector<float> t(1024);
    for (int o = 0; o < 32; ++o) for (int i = 0; i < 32; ++i) t[i * 32 + o] = v[1056 + o * 32 + i];
    pack_frag(p + pkb::DECV2T, t.data(), 32, 8, 2, natI, natO);
    for (int o = 0; o < 32; ++o) for (int i = 0; i < 32; ++i) t[i * 32 + o] = v[o * 32 + i];
    pack_frag(p + pkb::DECV1T, t.data(), 32, 8, 2, natI, gatherO);
    return GNR_OK;
}

extern "C" int gnr_canonical_vis_floats(void) { return gnr::can::TOTAL_VIS - gnr::can::TOTAL; }
extern "C" int gnr_canonical_weights_floats(void) { return gnr::can::TOTAL; }
extern "C" int gnr_packed_weights_floats(void) { return gnr::pk::TOTAL; }

extern "C" int gnr_pack_weights(const float* c, float* p) {
    if (!c || !p) return GNR_ERR_ARG;
    using namespace gnr;
    std::memset(p, 0, sizeof(float) * pk::TOTAL);
    const IdxFn natI = nat_in, natO = nat_out;
    const IdxFn ray8 = [](int j, int g) { return 8 * g + j; };
    const IdxFn first8 = [](int, int i) { return i < 8 ? i : -1; };

    // --- decoder: three branches, layers .0 (input = ray feature channels 8g+j) and .2
    const int d0w[3] = {can::MEAN0_W, can::VAR0_W, can::AW0_W}, d0b[3] = {can::MEAN0_B, can::VAR0_B, can::AW0_B};
    const int d2w[3] = {can::MEAN2_W, can::VAR2_W, can::AW2_W}, d2b[3] = {can::MEAN2_B, can::VAR2_B, can::AW2_B};
    for (int br = 0; br < 3; ++br) {
        pack_frag(p + pk::DEC1 + br * frag_floats(8, 2), c + d0w[br], 32, 8, 2, ray8, natO, LOG2E, kTrue);
        pack_bias(p + pk::B_DEC1 + br * 32, c + d0b[br], 2, natO, LOG2E);
        pack_frag(p + pk::DEC2 + br * frag_floats(8, 2), c + d2w[br], 32, 8, 2, natI, natO, LOG2E, kTilde);
        pack_bias(p + pk::B_DEC2 + br * 32, c + d2b[br], 2, natO, LOG2E);
    }
    // decoder .4 rows on the VALU: mean0 mean1 var0 var1 aw
    pack_row(p + pk::T_DEC3 + 0 * 32, c + can::MEAN4_W, 8, LOG2E);
    pack_row(p + pk::T_DEC3 + 1 * 32, c + can::MEAN4_W + 32, 8, LOG2E);
    pack_row(p + pk::T_DEC3 + 2 * 32, c + can::VAR4_W, 8, LOG2E);
    pack_row(p + pk::T_DEC3 + 3 * 32, c + can::VAR4_W + 32, 8, LOG2E);
    pack_row(p + pk::T_DEC3 + 4 * 32, c + can::AW4_W, 8, LOG2E);
    p[pk::T_DEC3_B + 0] = c[can::MEAN4_B]; p[pk::T_DEC3_B + 1] = c[can::MEAN4_B + 1];
    p[pk::T_DEC3_B + 2] = c[can::VAR4_B]; p[pk::T_DEC3_B + 3] = c[can::VAR4_B + 1];
    p[pk::T_DEC3_B + 4] = c[can::AW4_B];

    // --- prob_embed: 34 -> 32 -> 32 ; slot 8 carries (hit', vis') on groups 0,1
    pack_frag(p + pk::PE1, c + can::PE0_W, 34, 9, 2,
              [](int j, int g) { return j < 8 ? 8 * g + j : (g == 0 ? 32 : (g == 1 ? 33 : -1)); }, natO);
    pack_bias(p + pk::B_PE1, c + can::PE0_B, 2, natO);
    // prob_embed.2 (32x32 + bias, no activation) is folded into its two linear consumers (see gnr_layout.h):
    // products in double, then rounded once.
    const float* Wp2 = c + can::PE2_W;      // [32][32]
    const float* bp2 = c + can::PE2_B;

    // --- ray_dir_fc: 4 -> 16 -> 35, output laid out like x (see xfeat)
    pack_frag(p + pk::RDF1, c + can::RDF0_W, 4, 1, 1, [](int, int g) { return g; }, natO, LOG2E, kTrue);
    pack_bias(p + pk::B_RDF1, c + can::RDF0_B, 1, natO, LOG2E);
    const IdxFn xout = [](int nb, int i) {
        const int g = i >> 2, r = i & 3;
        if (nb == 0) return 3 + 8 * g + r;
        if (nb == 1) return 3 + 8 * g + 4 + r;
        return (r == 0 && g < 3) ? g : -1;
    };
    pack_frag(p + pk::RDF2, c + can::RDF2_W, 16, 4, 3, natI, xout, LOG2E, kTilde);
    pack_bias(p + pk::B_RDF2, c + can::RDF2_B, 3, xout, LOG2E);

    // --- neuray_fc: 32 -> 8 (MFMA) -> 1 (VALU)
    {   // neuray_fc.0 o prob_embed.2 : [8][32], input = ReLU output of prob_embed.0
        std::vector<float> Wc(8 * 32), bc(8);
        for (int o = 0; o < 8; ++o) {
            double bb = c[can::NR0_B + o];
            for (int k = 0; k < 32; ++k) bb += (double)c[can::NR0_W + o * 32 + k] * bp2[k];
            bc[o] = (float)bb;
            for (int i = 0; i < 32; ++i) {
                double a = 0;
                for (int k = 0; k < 32; ++k) a += (double)c[can::NR0_W + o * 32 + k] * Wp2[k * 32 + i];
                Wc[o * 32 + i] = (float)a;
            }
        }
        pack_frag(p + pk::NR1, Wc.data(), 32, 8, 1, natI, first8, LOG2E, kTrue);
        pack_bias(p + pk::B_NR1, bc.data(), 1, first8, LOG2E);
    }
    for (int g = 0; g < 4; ++g)
        for (int r = 0; r < 4; ++r) p[pk::T_NR2 + g * 4 + r] = (4 * g + r < 8) ? (float)(c[can::NR2_W + 4 * g + r] / LOG2E) : 0.f;
    p[pk::T_SCAL + 0] = c[can::NR2_B];

    // --- base_fc.0 split: view-invariant 140 columns (HOIST) + per-view 67 columns (BASE1)
    pack_frag(p + pk::HOIST, c + can::BASE0_W, 207, 36, 4,
              [](int j, int g) { const int x = xfeat(j % 9, g); return x < 0 ? -1 : 35 * (j / 9) + x; }, natO, LOG2E, kTrue);
    {   // base_fc.0 with its prob-embedding columns (175..206) multiplied by prob_embed.2
        std::vector<float> Wc(64 * 207), bc(64);
        for (int o = 0; o < 64; ++o) {
            double bb = c[can::BASE0_B + o];
            for (int k = 0; k < 32; ++k) bb += (double)c[can::BASE0_W + o * 207 + 175 + k] * bp2[k];
            bc[o] = (float)bb;
            for (int i = 0; i < 175; ++i) Wc[o * 207 + i] = c[can::BASE0_W + o * 207 + i];
            for (int i = 0; i < 32; ++i) {
                double a = 0;
                for (int k = 0; k < 32; ++k) a += (double)c[can::BASE0_W + o * 207 + 175 + k] * Wp2[k * 32 + i];
                Wc[o * 207 + 175 + i] = (float)a;
            }
        }
        pack_bias(p + pk::B_HOIST, bc.data(), 4, natO, LOG2E);
        pack_frag(p + pk::BASE1, Wc.data(), 207, 17, 4,
                  [](int j, int g) {
                      if (j < 9) { const int x = xfeat(j, g); return x < 0 ? -1 : 140 + x; }
                      return 175 + nat_in(j - 9, g);
                  }, natO, LOG2E, kTrue);
    }
    pack_frag(p + pk::BASE2, c + can::BASE2_W, 64, 16, 2, natI, natO, LOG2E, kTilde);
    pack_bias(p + pk::B_BASE2, c + can::BASE2_B, 2, natO, LOG2E);

    // --- vis_fc (32 -> 32 -> 32+1) and vis_fc2 (32 -> 32 -> 1)
    pack_frag(p + pk::VIS1, c + can::VIS0_W, 32, 8, 2, natI, natO, LOG2E, kTilde);
    pack_bias(p + pk::B_VIS1, c + can::VIS0_B, 2, natO, LOG2E);
    pack_frag(p + pk::VIS2, c + can::VIS2_W, 32, 8, 2, natI, natO, LOG2E, kTilde);
    pack_bias(p + pk::B_VIS2, c + can::VIS2_B, 2, natO, LOG2E);
    pack_row(p + pk::T_VIS2R, c + can::VIS2_W + 32 * 32, 8, LOG2E);
    p[pk::T_SCAL + 1] = c[can::VIS2_B + 32];
    pack_frag(p + pk::VISB1, c + can::VISB0_W, 32, 8, 2, natI, natO, LOG2E, kTilde);
    pack_bias(p + pk::B_VISB1, c + can::VISB0_B, 2, natO, LOG2E);
    pack_row(p + pk::T_VISB2, c + can::VISB2_W, 8, LOG2E);
    p[pk::T_SCAL + 2] = c[can::VISB2_B];

    // --- rgb_fc: [h(32), vis(1), dir_diff(4)] -> 16 -> 8 -> 1
    pack_frag(p + pk::RGB1, c + can::RGB0_W, 37, 10, 1,
              [](int j, int g) {
                  if (j < 8) return nat_in(j, g);
                  if (j == 8) return g == 0 ? 32 : 33 + (g - 1);
                  return g == 0 ? 36 : -1;
              }, natO, LOG2E, [](int i) { return i < 32 ? LOG2E : 1.0; });
    pack_bias(p + pk::B_RGB1, c + can::RGB0_B, 1, natO, LOG2E);
    pack_frag(p + pk::RGB2, c + can::RGB2_W, 16, 4, 1, natI, first8, LOG2E, kTilde);
    pack_bias(p + pk::B_RGB2, c + can::RGB2_B, 1, first8, LOG2E);
    for (int g = 0; g < 4; ++g)
        for (int r = 0; r < 4; ++r) p[pk::T_RGB3 + g * 4 + r] = (4 * g + r < 8) ? (float)(c[can::RGB4_W + 4 * g + r] / LOG2E) : 0.f;
    p[pk::T_SCAL + 3] = c[can::RGB4_B];

    // --- geometry_fc: [mean(32), var(32), wbar, embed(21)] -> 64 -> 16
    //     slots 16..22: group 0 carries wbar in slot 16; group g>=1 carries coordinate g-1,
    //     kind k = slot-16 of [p, sin p, cos p, sin 2p, cos 2p, sin 4p, cos 4p]  (neus.py:37-45)
    pack_frag(p + pk::GEO1, c + can::GEO0_W, 86, 23, 4,
              [](int j, int g) {
                  if (j < 8) return nat_in(j, g);
                  if (j < 16) return 32 + nat_in(j - 8, g);
                  const int k = j - 16;
                  if (g == 0) return k == 0 ? 64 : -1;
                  return 65 + 3 * k + (g - 1);
              }, natO, LOG2E, [](int i) { return i < 32 ? LOG2E : (i < 64 ? LOG2E * LOG2E : 1.0); });
    pack_bias(p + pk::B_GEO1, c + can::GEO0_B, 4, natO, LOG2E);
    pack_frag(p + pk::GEO2, c + can::GEO2_W, 64, 16, 1, natI, natO, LOG2E, kTilde);
    pack_bias(p + pk::B_GEO2, c + can::GEO2_B, 1, natO, LOG2E);

    // --- RAY section
    std::memcpy(p + pk::R_WQ, c + can::WQ, sizeof(float) * 256);
    std::memcpy(p + pk::R_WK, c + can::WK, sizeof(float) * 256);
    std::memcpy(p + pk::R_WV, c + can::WV, sizeof(float) * 256);
    std::memcpy(p + pk::R_WFC, c + can::WFC, sizeof(float) * 256);
    std::memcpy(p + pk::R_LNW, c + can::LN_W, sizeof(float) * 16);
    std::memcpy(p + pk::R_LNB, c + can::LN_B, sizeof(float) * 16);
    std::memcpy(p + pk::R_OUT0W, c + can::OUT0_W, sizeof(float) * 256);
    std::memcpy(p + pk::R_OUT0B, c + can::OUT0_B, sizeof(float) * 16);
    std::memcpy(p + pk::R_OUT1W, c + can::OUT1_W, sizeof(float) * 16);
    p[pk::R_OUT1B] = c[can::OUT1_B];
    std::memcpy(p + pk::R_GEO2W, c + can::GEO2_W, sizeof(float) * 1024);
    for (int h = 0; h < 64; ++h)
        for (int e = 0; e < 21; ++e) p[pk::R_GEO1E + h * 24 + e] = c[can::GEO0_W + h * 86 + 65 + e];
    p[pk::R_VARIANCE] = c[can::VARIANCE];
    // positional table, float64 then cast (ref: ibrnet.py:437-445)
    for (int pos = 0; pos < 128; ++pos)
        for (int k = 0; k < 16; ++k) {
            const double ang = (double)pos / std::pow(10000.0, 2.0 * (k / 2) / 16.0);
            p[pk::R_PE + pos * 16 + k] = (float)((k % 2 == 0) ? std::sin(ang) : std::cos(ang));
        }
    for (int i = 0; i < 16; ++i)
        for (int o = 0; o < 16; ++o) {
            p[pk::R_WQT + i * 16 + o] = c[can::WQ + o * 16 + i];
            p[pk::R_WKT + i * 16 + o] = c[can::WK + o * 16 + i];
            p[pk::R_WVT + i * 16 + o] = c[can::WV + o * 16 + i];
            p[pk::R_WFCT + i * 16 + o] = c[can::WFC + o * 16 + i];
        }
    for (int h = 0; h < 64; ++h)
        for (int o = 0; o < 16; ++o) p[pk::R_GEO2WT + h * 16 + o] = c[can::GEO2_W + o * 64 + h];
    for (int i = 0; i < 16; ++i) {
        double acc = 0;
        for (int f = 0; f < 16; ++f) acc += (double)c[can::OUT0_W + f * 16 + i] * (double)c[can::OUT1_W + f];
        p[pk::R_OUTVJP + i] = (float)acc;
    }
    {
        double acc = c[can::OUT1_B];
        for (int f = 0; f < 16; ++f) acc += (double)c[can::OUT1_W + f] * (double)c[can::OUT0_B + f];
        p[pk::R_OUTB] = (float)acc;
    }
    return build_c16(p);
}

// The optional fourth decoder branch of a level (dist_decoder_cfg.use_vis: true, dist_decoder.py:89-97,103-104,133-134) into an
// already packed blob.  v = vis_decoder.{0.weight [32][32], 0.bias [32], 2.weight [32][32], 2.bias [32], 4.weight [1][32],
// 4.bias [1]} in state-dict order (2 145 floats).  Sets the flag k_chain tests.
extern "C" int gnr_pack_vis_decoder(const float* v, float* p) {
    if (!v || !p) return GNR_ERR_ARG;
    using namespace gnr;
    const IdxFn natI = nat_in, natO = nat_out;
    const IdxFn ray8 = [](int j, int g) { return 8 * g + j; };
    pack_frag(p + pk::DECV1, v, 32, 8, 2, ray8, natO, LOG2E, kTrue);
    pack_bias(p + pk::B_DECV1, v + 1024, 2, natO, LOG2E);
    pack_frag(p + pk::DECV2, v + 1056, 32, 8, 2, natI, natO, LOG2E, kTilde);
    pack_bias(p + pk::B_DECV2, v + 2080, 2, natO, LOG2E);
    pack_row(p + pk::T_DECV3, v + 2112, 8, LOG2E);
    p[pk::T_VIS] = v[2144];
    p[pk::T_VIS + 1] = 1.f;
    return build_c16(p);
}
