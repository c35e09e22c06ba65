// hm_tail_p_geo.h -- geometry of the strip tail kernel (hm_tail_p.hip, hm_tail_p16.hip): LDS plan and conv5's operand addresses.
#pragma once
#include "hm_convp.h"

// operand ring depth (half8 registers) and look-ahead (k-blocks) of conv5's / conv6's streams in the strip kernel
#ifndef HM_TAILP_C5_NS
#define HM_TAILP_C5_NS 8
#define HM_TAILP_C5_LA 1
#endif
#ifndef HM_TAILP_C6_NS
#define HM_TAILP_C6_NS 8
#define HM_TAILP_C6_LA 1
#endif

namespace hm {

namespace {

struct PGeo {
    static constexpr int S = 16, NW = 4;
    static constexpr int L4 = C4_LEN, L5 = 13, L6 = 7, L7 = 4, L8 = 2;
    static constexpr int RS = 104, RS64 = 72;               // halves
    // fc1's fp32 output for the VALU fc2: a site's 256 values in 8 parts of 32, parts 36 floats apart (fc2's 16 lanes of a site read 8 distinct
    // parts: 144-byte steps fall into different banks), sites 288 apart; fc2's weights likewise, the two outputs 324 apart
    static constexpr int HPS = 36, HRS = 8 * HPS, F2S = 324;
    static constexpr int R = TAILP_STRIP;                  // strip rows (lattice rows 16 map rows apart)
    static constexpr int SPAN = R - (L4 - 2);              // a pass's sites start at most this many lattice rows apart (0 .. SPAN)
    // plane A (halves from the plane pointer)
    static constexpr int STRIP = 0, EDGE0 = R * RS, EDGE24 = (R + S) * RS, PA = (R + 2 * S) * RS;
    // plane B
    static constexpr int P5 = L5 * S * RS;
    static constexpr int C7 = 0, C8 = L7 * S * RS64, HFC = C8 + L8 * S * RS64;   // conv7's / conv8's planes, fc1's fp32 output, inside B's hi | lo plane
    static constexpr int LDS_HALVES = 2 * PA + 2 * P5;
    static_assert(L6 * S * RS <= R * RS, "conv6's output overlays the strip");
    static_assert(HFC * 2 % 16 == 0 && HFC + S * HRS * 2 <= P5, "fc1's output fits behind conv8's, inside the hi plane");
    static constexpr int LATE_ROWS = L6 * S;               // strip rows conv6's output overlays: fetched once conv7 has read it
    static constexpr int QROWS = 4, QBYTES = QROWS * RS * 2, NQ = R / QROWS, LATEQ = LATE_ROWS / QROWS;
    static_assert(R % QROWS == 0 && LATE_ROWS % QROWS == 0 && RS * 2 == 13 * 16, "row-aligned pieces of 13 sixteen-byte units");
};

// conv5's operands: data row d of a site = its edge row 0 (d = 0), strip rows s + d - 1 (d = 1 .. 23), its edge row 24 (d = 24);
// output position p reads data rows 2p - 1 .. 2p + 1
template <class C>
struct PInStrip {
    using T = PGeo;
    int sb;  // s * RS + 8 * lk: this lane's site's first strip row
    int eb;  // li * RS + 8 * lk
    static constexpr int drow(int tile, int kb) { return 2 * tile - 1 + C::tap(kb); }
    static constexpr bool skip(int tile, int kb) { return drow(tile, kb) < 0 || drow(tile, kb) >= T::L4; }
    template <int TILE, int KB>
    __device__ __forceinline__ int off() const {
        constexpr int d = drow(TILE, KB), ch = C::ch0(KB);
        if constexpr (d == 0) return eb + (T::EDGE0 + ch);
        else if constexpr (d == T::L4 - 1) return eb + (T::EDGE24 + ch);
        else return sb + (T::STRIP + (d - 1) * T::RS + ch);
    }
};

}  // namespace

#define HM_TAILP_PARAMS SiteRange sr, CtxWeights W, float* __restrict__ logits, float* __restrict__ prob, uint8_t* __restrict__ ml, \
                        const half_t* __restrict__ e4, const half_t* __restrict__ edge4, const int32_t* __restrict__ order,      \
                        const int32_t* __restrict__ okey, int n_rows, int32_t* __restrict__ pass_count, half_t* __restrict__ x8, int w16

}  // namespace hm
