// Host-side weight packer: reference state-dict order (canonical blob) -> MFMA A-fragment
// blob consumed by the gfx950 kernels.  See gnr_layout.h for the execution model.  The arithmetic lives in gnr_pack_body.h
// (shared, bit for bit, with the device-side packer gnr_pack_dev.hip); this file instantiates it with plain loops.
// Parameter shapes/order: ref dist_decoder.py:64-88, aggregate_net.py:29-33, ibrnet.py:382-423.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "gnr_pack_body.h"
#include "../../include/gnr.h"

namespace {
using namespace gnr;
// C16 section of a packed blob from its CHAIN section: the same slots (two of them grown, gnr_layout.h c16_off) with the wide
// layers' fragments as fp16 pairs: what k_chain stages into LDS.  The "a weight has no pair" flag (T_VIS + 2) is sticky over
// gnr_pack_weights -> gnr_pack_vis_decoder (which re-runs this on the same blob).
int build_c16(float* p) {
    std::memset(p + pk::C16, 0, sizeof(float) * pk::C16_END);
    const packer::HostExec ex;
    packer::c16_copy_runs(ex, p);
    packer::c16_pairs(ex, p);
    p[pk::C16 + pk::c16_off(pk::T_VIS) + 2] = p[pk::T_VIS + 2];
    return GNR_OK;
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
        {"R_WKT", R_WKT}, {"R_WVT", R_WVT}, {"R_WFCT", R_WFCT}, {"R_GEO2WT", R_GEO2WT}, {"R_OUTVJP", R_OUTVJP}, {"R_OUTB", R_OUTB}, {"RM_DC", RM_DC}, {"RM_GEOA", RM_GEOA}, {"RM_GEOB", RM_GEOB}, {"RM_END", RM_END}, {"C16", C16}, {"C16_END", C16_END}, {"TOTAL", TOTAL}};
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
    std::memset(p, 0, sizeof(float) * gnr::pkb::TOTAL);
    gnr::packer::pack_backward(gnr::packer::HostExec(), c, p);
    gnr::packer::pack_backward_pairs(gnr::packer::HostExec(), p);
    return GNR_OK;
}

// The fourth decoder branch's transposed fragments (use_vis training) into a blob of gnr_pack_weights_bwd.  v as in
// gnr_pack_vis_decoder (state-dict order, 2 145 floats).
extern "C" int gnr_pack_vis_decoder_bwd(const float* v, float* p) {
    if (!v || !p) return GNR_ERR_ARG;
    gnr::packer::pack_vis_backward(gnr::packer::HostExec(), v, p);
    gnr::packer::pack_vis_backward_pairs(gnr::packer::HostExec(), p);
    return GNR_OK;
}

extern "C" int gnr_canonical_vis_floats(void) { return gnr::can::TOTAL_VIS - gnr::can::TOTAL; }
extern "C" int gnr_canonical_weights_floats(void) { return gnr::can::TOTAL; }
extern "C" int gnr_packed_weights_floats(void) { return gnr::pk::TOTAL; }

extern "C" int gnr_pack_weights(const float* c, float* p) {
    if (!c || !p) return GNR_ERR_ARG;
    using namespace gnr;
    std::memset(p, 0, sizeof(float) * pk::TOTAL);
    packer::pack_forward(packer::HostExec(), c, p);
    // positional table, float64 then cast (ref: ibrnet.py:437-445): a constant of the blob (the device-side packer keeps it)
    for (int pos = 0; pos < 128; ++pos)
        for (int k = 0; k < 16; ++k) {
            const double ang = (double)pos / std::pow(10000.0, 2.0 * (k / 2) / 16.0);
            p[pk::R_PE + pos * 16 + k] = (float)((k % 2 == 0) ? std::sin(ang) : std::cos(ang));
        }
    return build_c16(p);
}

// The optional fourth decoder branch of a level (dist_decoder_cfg.use_vis: true, dist_decoder.py:89-97,103-104,133-134) into an
// already packed blob.  v = vis_decoder.{0.weight [32][32], 0.bias [32], 2.weight [32][32], 2.bias [32], 4.weight [1][32],
// 4.bias [1]} in state-dict order (2 145 floats).  Sets the flag k_chain tests.
extern "C" int gnr_pack_vis_decoder(const float* v, float* p) {
    if (!v || !p) return GNR_ERR_ARG;
    gnr::packer::pack_vis(gnr::packer::HostExec(), v, p);
    return build_c16(p);
}
