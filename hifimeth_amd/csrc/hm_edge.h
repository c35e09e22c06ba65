// hm_edge.h -- what the two edge kernels (edge_kernel in hm_trunk.hip, edge2_kernel in hm_edge2.hip) and the dense trunk share:
// the LDS row geometry, the feature-row decoder, the geometry of the edge chains and their epilogues.
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98).
#pragma once
#include "hm_convh.h"

namespace hm {

namespace {

constexpr int TR_RS = 136;     // halves per LDS row: 128 channels + 16 B pad (conflict-free ds_read_b128 at row stride 1)
constexpr int TR_WRS = 8;      // feature row = 8 exact halves

// feature row of view position x of a read: 8 EXACT halves (one-hot base, decoded frame counts / 32; bn0 lives in conv1's
// folded weights, hm_weights.cpp); all zeros outside the read
__device__ __forceinline__ uint4 feature_row(int b, uint32_t k, int view) {
    uint4 row = make_uint4(0u, 0u, 0u, 0u);
    if (b < 0) return row;
    if (view) {  // the read seen from its reverse strand: complemented base, the strands' kinetics swapped
        if (b < 4) b = 3 - b;
        k = (k >> 16) | (k << 16);
    }
    row.x = b == 0 ? 0x3c00u : b == 1 ? 0x3c000000u : 0u;
    row.y = b == 2 ? 0x3c00u : b == 3 ? 0x3c000000u : 0u;
    // codev1 byte t -> frames = (((t & 63) + 64) << (t >> 6)) - 64 (bam_info.cpp:562-570); frames / 32 is exact in fp16
    half_t f[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t tt = (k >> (8 * c)) & 255u;
        f[c] = (half_t)((float)((int)(((tt & 63u) + 64u) << (tt >> 6)) - 64) * 0.03125f);
    }
    row.z = (uint32_t)__builtin_bit_cast(uint16_t, f[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, f[1]) << 16);
    row.w = (uint32_t)__builtin_bit_cast(uint16_t, f[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, f[3]) << 16);
    return row;
}

// Which map rows do the edge chains read?  A site at view position `off` reads (K1 = 11 | 13):
//   left chain : E1, E2, E3 at off - 199
//   right chain: E1 at off + 189 | off + 185, off + 187 ; E2 at off + 185 | off + 177, off + 181 ; E3 at off + 169, off + 177 | off + 169
template <int K1>
struct EdgeGeo {
    static constexpr int L1 = (KMER + 2 - K1) / 2 + 1, L2 = (L1 - 1) / 2 + 1, L3 = (L2 - 1) / 2 + 1;
    // the last output of a layer over Lin rows takes rows 2(Lout-1)-1 .. +1: (shared, specific, pad) if that ends at Lin,
    // (shared, shared, specific) if it ends at Lin - 1
    static constexpr bool PAD2 = 2 * (L2 - 1) + 1 == L1, PAD3 = 2 * (L3 - 1) + 1 == L2, PAD4 = 2 * (C4_LEN - 1) + 1 == L3;
    // view-position deltas of the shared rows of the right chain (first shared tap; the second, if any, is +step)
    static constexpr int R1 = -201 + 2 * (2 * (L2 - 1) - 1), R2 = -203 + 4 * (2 * (L3 - 1) - 1), R3 = -207 + 8 * (2 * (C4_LEN - 1) - 1);
    static constexpr int LEFT = -199;
    static constexpr int X_LEFT = -201, X_RIGHT = -201 + 2 * (L1 - 1);  // first feature row of conv1's first / last output
};
static_assert(EdgeGeo<11>::PAD2 && EdgeGeo<11>::PAD3 && !EdgeGeo<11>::PAD4 && EdgeGeo<11>::R1 == 189 && EdgeGeo<11>::R2 == 185 && EdgeGeo<11>::R3 == 169, "k1 = 11 edge geometry");
static_assert(!EdgeGeo<13>::PAD2 && !EdgeGeo<13>::PAD3 && EdgeGeo<13>::PAD4 && EdgeGeo<13>::R1 == 185 && EdgeGeo<13>::R2 == 177 && EdgeGeo<13>::R3 == 169, "k1 = 13 edge geometry");


constexpr int EG_S = 32, EG_M = 2 * EG_S;
constexpr int EG_XROWS = 16;  // feature rows per pseudo-row (K1 <= 13 used)
constexpr int EG_OP = EG_M * 3 * TR_RS;  // halves per operand plane
constexpr int EG_SP = EG_M * TR_RS;      // halves per "specific" plane
constexpr int EG_XH = EG_M * EG_XROWS * TR_WRS;  // halves of the feature-row buffer

struct EdgeSite {
    int64_t bo;
    int64_t vrow;  // map row of view position 0 (incl. the view's plane offset)
    int L, off, view, valid;
};

enum { SRC_ZERO = 0, SRC_SPEC = 1, SRC_MAP = 2 };
struct TapSrc {
    int kind, delta;
};

template <int K1>
__device__ __forceinline__ TapSrc tap_source(int layer, int side, int tap) {
    using G = EdgeGeo<K1>;
    if (side == 0) return tap == 0 ? TapSrc{SRC_ZERO, 0} : tap == 1 ? TapSrc{SRC_SPEC, 0} : TapSrc{SRC_MAP, G::LEFT};
    const bool pad = layer == 2 ? G::PAD2 : layer == 3 ? G::PAD3 : G::PAD4;
    const int r = layer == 2 ? G::R1 : layer == 3 ? G::R2 : G::R3, step = layer == 2 ? 2 : layer == 3 ? 4 : 8;
    if (pad) return tap == 0 ? TapSrc{SRC_MAP, r} : tap == 1 ? TapSrc{SRC_SPEC, 0} : TapSrc{SRC_ZERO, 0};
    return tap == 0 ? TapSrc{SRC_MAP, r} : tap == 1 ? TapSrc{SRC_MAP, r + step} : TapSrc{SRC_SPEC, 0};
}

// ReLU + split -> the "specific" planes, row m
struct EpiSpec {
    half_t* hi;
    half_t* lo;
    const float* __restrict__ bias;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        half4 h, l;
        split4(acc, h, l);
        *reinterpret_cast<half4*>(hi + m * TR_RS + col) = h;
        *reinterpret_cast<half4*>(lo + m * TR_RS + col) = l;
    }
};

// conv1's edge outputs: the folded bn0 constant must not count for the tap on the zero padding (c1f_corr, hm_weights.cpp);
// a lane's four channels of the two correction rows are loaded before the layer (in the epilogue the load would queue
// behind the map-row requests)
struct EpiSpecC1 {
    half_t* hi;
    half_t* lo;
    const float* __restrict__ bias;
    float4 c0, c1;  // first output row (left chains), last output row (right chains)
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        const float4 c = m >= EG_S ? c1 : c0;
        half4 h, l;
        split4(f32x4{acc[0] - c.x, acc[1] - c.y, acc[2] - c.z, acc[3] - c.w}, h, l);
        *reinterpret_cast<half4*>(hi + m * TR_RS + col) = h;
        *reinterpret_cast<half4*>(lo + m * TR_RS + col) = l;
    }
};

struct EpiEdgeOut {
    half_t* __restrict__ out;  // [site][side][hi 96 | lo 96] of this pass
    const float* __restrict__ bias;
    int nvalid;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        const int side = m >= EG_S, site = m - side * EG_S;
        if (site < nvalid) {
            half4 h, l;
            split4(acc, h, l);
            half_t* o = out + (size_t)site * (4 * C4_CH) + side * (2 * C4_CH) + col;
            *reinterpret_cast<half4*>(o) = h;
            *reinterpret_cast<half4*>(o + C4_CH) = l;
        }
    }
};


}  // namespace

}  // namespace hm
