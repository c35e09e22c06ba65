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

// conv5 of the strip kernel: 1 = DENSE over the strip's rows (an m-tile = 16 consecutive centre rows, every row once for all the sites that
// read it) + two site-major tiles for the positions that touch a site's own edge rows; 0 = site-major throughout (m-tile p = position p of
// the 16 sites), the form of the first strip kernel
#ifndef HM_TAILP_DENSE5
#define HM_TAILP_DENSE5 1
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
    // plane B: conv5's output.  Dense form: [E5 of strip centre row c at row c, c = 1 .. R - 2 | row R: unused | position 0 of the 16 sites at
    // rows R + 1 .. | position 12 at rows R + 1 + S ..]; site-major form: [13 positions][16 sites]
    static constexpr int D5T = R / 16;                     // dense tiles: centre rows 16 t + 1 .. 16 t + 16 (the last two of the last tile are not strip centres: never read)
    static constexpr int E5P0 = R + 1, E5P12 = R + 1 + S;  // rows of the two site-major tiles
    static constexpr int P5 = HM_TAILP_DENSE5 ? (R + 1 + 2 * S) * RS : L5 * S * RS;
    static_assert(R % 16 == 0, "whole dense tiles");
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

// conv5, dense form.  Tile t < D5T: lane li computes E5 at strip centre row c = 16 t + li + 1 from strip rows c - 1, c, c + 1 -- consecutive
// lanes read consecutive rows (13 sixteen-byte units apart: 16 different bank quads), whatever the sites' rows are.  A site's position p = 1 .. 11
// is the centre row s + 2p - 1 (s = its first strip row): every row is computed once however many of the pass's sites read it -- 9 tiles for
// the strip's 142 centres instead of 11 tiles of (position, site) pairs.  Tiles D5T / D5T + 1: positions 0 / 12 of the 16 sites, site-major as
// before (taps: padding -- skipped --, the site's own edge row, one strip row).
template <class C>
struct PInStripDense {
    using T = PGeo;
    int rb;  // li * RS + 8 * lk: row li of a dense tile; also this lane's site's edge rows
    int sb;  // s * RS + 8 * lk: this lane's site's first strip row
    static constexpr bool skip(int tile, int kb) { return (tile == T::D5T && C::tap(kb) == 0) || (tile == T::D5T + 1 && C::tap(kb) == 2); }
    template <int TILE, int KB>
    __device__ __forceinline__ int off() const {
        constexpr int tap = C::tap(KB), ch = C::ch0(KB);
        if constexpr (TILE < T::D5T) return rb + (T::STRIP + (16 * TILE + tap) * T::RS + ch);
        else if constexpr (TILE == T::D5T) return tap == 1 ? rb + (T::EDGE0 + ch) : sb + (T::STRIP + ch);                      // data rows 0 (edge), 1 (strip row s)
        else return tap == 1 ? rb + (T::EDGE24 + ch) : sb + (T::STRIP + (T::L4 - 3) * T::RS + ch);                              // data rows 23 (strip row s + 22), 24 (edge)
    }
};
// conv6 over the dense form's output: position q reads conv5 positions 2q - 1 .. 2q + 1 of the lane's site -- rows s + 2p - 1 of the dense part
// (p = 1 .. 11), the site-major rows for p = 0 / 12, padding (skipped) outside
template <class C>
struct PInE5 {
    using T = PGeo;
    int rb;  // li * RS + 8 * lk
    int sb;  // s * RS + 8 * lk
    static constexpr int pos(int tile, int kb) { return 2 * tile - 1 + C::tap(kb); }
    static constexpr bool skip(int tile, int kb) { return pos(tile, kb) < 0 || pos(tile, kb) >= T::L5; }
    template <int TILE, int KB>
    __device__ __forceinline__ int off() const {
        constexpr int p = pos(TILE, KB), ch = C::ch0(KB);
        if constexpr (p == 0) return rb + (T::E5P0 * T::RS + ch);
        else if constexpr (p == T::L5 - 1) return rb + (T::E5P12 * T::RS + ch);
        else return sb + ((2 * p - 1) * T::RS + ch);
    }
};

}  // namespace

#define HM_TAILP_PARAMS SiteRange sr, CtxWeights W, float* __restrict__ logits, float* __restrict__ prob, uint8_t* __restrict__ ml, \
                        const half_t* __restrict__ e4, const half_t* __restrict__ edge4, const int32_t* __restrict__ order,      \
                        const int32_t* __restrict__ okey, int n_rows, int32_t* __restrict__ pass_count, half_t* __restrict__ x8, int w16

}  // namespace hm
