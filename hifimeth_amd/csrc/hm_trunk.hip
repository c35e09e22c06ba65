// hm_trunk.hip -- the dense trunk: conv1..conv4 of the CNN evaluated once per (read, strand view) POSITION.
//
// The reference runs the whole network on every site's 401-row window (mod_batch.cpp:66-75).  Neighbouring sites of a
// read are a few bases apart, so their windows are the same rows shifted, and with stride-2 convolutions the
// activations of a window are SAMPLES of dense maps over the read (hm_device.h, "dense trunk"):
//     c1[p] = E1[off - 201 + 2p]   c2[q] = E2[off - 203 + 4q]   c3[r] = E3[off - 207 + 8r]   c4[s] = E4[off - 215 + 16s]
// with E(l) the layer evaluated at EVERY position with its taps 2^(l-1) rows apart (dilation instead of stride).  This
// holds for every output position except the first and the last of each layer (they see the window's zero padding, or
// the neighbour that does).  So per strand view a read needs each layer once per base -- 146 k MAC per position -- instead
// of 197 / 99 / 50 / 25 positions per site: ~5x fewer MACs at the site density of an all-context run, with identical
// products (same weights, same inputs, fp32 accumulation; only the summation order inside the MFMA differs).
//
//   trunk2_kernel: one workgroup per tile of 112 view positions: feature rows -> E1 -> E2 -> E3 in LDS (split fp16
//                  planes, a-trous implicit GEMM on v_mfma_f32_16x16x32_f16, M = 144 / 144 / 128 / 112 rows: no ragged
//                  tiles) -> E4 to HBM; the E1..E3 rows that some site's edge chain needs are written out as well.
//                  Streaming form (hm_convs.h): 4 waves, a layer's weights resident in registers, the positions
//                  streamed past them in tile groups, the epilogue between the next group's MFMAs (the default).
//   trunk_kernel : the same tile in the 8-wave ConvH form (engine option trunk_impl = 0; byte-identical results).
//   rowlist_kernel: per tile and layer, the rows an edge chain will read (trunk2_kernel copies exactly those out).
//   edge_kernel  : the first and last conv4 row of every site (the only two that are not samples of E4): per side a
//                  chain of four one-row layers over rows gathered from the maps, 32 sites stacked along M.
//   The tail kernel (hm_front_h.hip, GATHER) then picks a site's 23 interior conv4 rows out of E4 + its two edge rows.
//
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98).
#include "hm_convh.h"
#ifdef HM_TRUNK_STAMP  // diagnostic build (make stamp): per-wave shader-clock phase sums of workgroup 0, read by tools/trunk_stamps.py
#include "hm_stamp.h"
namespace hm { __device__ unsigned long long g_trunk_stamp[8][24]; __device__ unsigned long long g_edge_stamp[8][12]; }
extern "C" int hm_debug_edge_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(hm::g_edge_stamp), sizeof(hm::g_edge_stamp)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[8][12];
        if (hipMemcpyToSymbol(HIP_SYMBOL(hm::g_edge_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int hm_debug_trunk_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(hm::g_trunk_stamp), sizeof(hm::g_trunk_stamp)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[8][24];
        if (hipMemcpyToSymbol(HIP_SYMBOL(hm::g_trunk_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#include "hm_convs.h"
#include "hm_edge.h"

namespace hm {

namespace {

constexpr int TR_M1 = 144, TR_M2 = 144, TR_M3 = 128, TR_M4 = TR_OWN;  // rows computed per layer (needed: 140 / 136 / 128 / 112)
constexpr int TR_XROWS = 160;  // feature rows of a tile: 144 + K1 - 1 <= 156
constexpr int TR_AROWS = 148, TR_BROWS = 144;
static_assert(TR_M1 + 4 <= TR_AROWS && TR_M3 + 8 <= TR_BROWS && TR_M4 + 16 <= TR_M3, "halo plan");
constexpr int TR_LDS_HALVES = 2 * TR_AROWS * TR_RS + 2 * TR_BROWS * TR_RS;
constexpr int TR_RL = 3 * TR_OWN;   // bytes of a tile's row lists (rowlist_kernel)
static_assert(TR_OWN % 16 == 0 && TR_RL % 4 == 0, "row lists; copy slots of two rows over 4 or 8 waves");

// ReLU + split -> LDS planes (row m, no padding rows: the dense form has none); the rows an edge chain will read also
// go to the HBM map as [hi 128 | lo 128].  (Measured: all map stores together cost 11 % of the kernel.  Staging them in
// LDS and copying whole rows out in 16-byte chunks behind the next layer's first weight loads, with the stores hidden
// from the compiler's wait counting, did not change that: +-1 % in same-box A/Bs.  The simple form stays.)
template <int BIT>
struct EpiTrunk {
    half_t* hi;
    half_t* lo;
    const float* __restrict__ bias;
    const uint8_t* flags;
    half_t* __restrict__ g;  // map row of tile row 0
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        half4 h, l;
        split4(acc, h, l);
        *reinterpret_cast<half4*>(hi + m * TR_RS + col) = h;
        *reinterpret_cast<half4*>(lo + m * TR_RS + col) = l;
        if (m < TR_OWN && ((flags[m] >> BIT) & 1)) {
            *reinterpret_cast<half4*>(g + (size_t)m * 256 + col) = h;
            *reinterpret_cast<half4*>(g + (size_t)m * 256 + 128 + col) = l;
        }
    }
};

// conv4 rows leave the trunk already split, [hi 96 | lo 96]: the tail copies them into its input planes as they are
struct EpiE4 {
    half_t* __restrict__ g;
    const float* __restrict__ bias;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        half4 h, l;
        split4(acc, h, l);
        *reinterpret_cast<half4*>(g + (size_t)m * (2 * C4_CH) + col) = h;
        *reinterpret_cast<half4*>(g + (size_t)m * (2 * C4_CH) + C4_CH + col) = l;
    }
};


}  // namespace

template <int K1, bool W16>
__global__ __launch_bounds__(512) void trunk_kernel(const TrunkTile* __restrict__ tiles, int n_tiles, int n_views, int ctx,
                                                     const RInfo* __restrict__ rinfo, const uint8_t* __restrict__ bases,
                                                     const uint32_t* __restrict__ kin, const uint8_t* __restrict__ sctx,
                                                     CtxWeights W, TrunkMaps mp) {
    using G = EdgeGeo<K1>;
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) half_t smem[TR_LDS_HALVES];
    __shared__ uint8_t flags[TR_XROWS];
    __shared__ int64_t s_grow0;
    half_t* a_hi = smem;
    half_t* a_lo = a_hi + TR_AROWS * TR_RS;
    half_t* b_hi = a_lo + TR_AROWS * TR_RS;
    half_t* b_lo = b_hi + TR_BROWS * TR_RS;
    const int n_work = n_tiles * n_views;

    // rows 144..147 of planes A are read by conv2's last position tile (results never used) but never written: keep them finite
    for (int i = threadIdx.x; i < (TR_AROWS - TR_M1) * TR_RS; i += NW * 64) {
        a_hi[TR_M1 * TR_RS + i] = (half_t)0.f;
        a_lo[TR_M1 * TR_RS + i] = (half_t)0.f;
    }

    // feature rows + map-row flags of work item w -> planes B / flags, by threads [0, nt)
    auto build = [&](const int w, const int t, const int nt) __attribute__((always_inline)) {
        const int view = w >= n_tiles;
        const TrunkTile tl = tiles[view ? w - n_tiles : w];
        const RInfo ri = rinfo[tl.read_idx];
        const int L = ri.len;
        const int64_t bo = ri.base_off;
        if (t == 0) s_grow0 = (int64_t)view * mp.view_rows + ri.map_off + (tl.u0 + TR_PAD);
        const int want_base = view ? 2 : 1;  // a site sits on a C of its own strand: forward C, or forward G seen from the reverse strand
        auto site_at = [&](int y) __attribute__((always_inline)) {  // is view position y a site of this context?
            if (y < 0 || y >= L) return 0;
            const int64_t j = bo + (view ? L - 1 - y : y);
            return (int)(sctx[j] == (ctx | (want_base == 2 ? 4 : 0)));   // context | strand << 2 (hm_kernels.h)
        };
        for (int r = t; r < TR_XROWS; r += nt) {
            const int x = tl.u0 + r;
            int b = -1;
            uint32_t k = 0;
            if (x >= 0 && x < L) {
                const int64_t j = bo + (view ? L - 1 - x : x);
                b = bases[j];
                k = kin[j];
            }
            *reinterpret_cast<uint4*>(b_hi + r * TR_WRS) = feature_row(b, k, view);
            int f = 0;
            if (r < TR_OWN) {
                const int left = site_at(x - G::LEFT);
                f |= (left | site_at(x - G::R1) | (G::PAD2 ? 0 : site_at(x - G::R1 - 2))) << 0;
                f |= (left | site_at(x - G::R2) | (G::PAD3 ? 0 : site_at(x - G::R2 - 4))) << 1;
                f |= (left | site_at(x - G::R3) | (G::PAD4 ? 0 : site_at(x - G::R3 - 8))) << 2;
            }
            flags[r] = (uint8_t)f;
        }
    };

    // conv1: folded bn0, exact fp16 operand, weights' hi / lo halves stacked along K; dense: every position, taps 1 row apart
    using C1 = ConvH<NW, 8, (2 * K1 + 3) / 4 * 4, 128, TR_M1, TR_WRS, 1, 8, 4, 1, 0, 0, true, false, false, K1, true, 0, 1, 1>;
#ifndef TRK_WM
#define TRK_WM 1
#define TRK_WN 8
#define TRK_BR 3
#define TRK_BR4 3
#endif
    using C2 = ConvH<NW, 128, 3, 128, TR_M2, TR_RS, TRK_WM, TRK_WN, TRK_BR, 1, 0, 0, !W16, true, false, 0, true, 1, 1, 2>;
    using C3 = ConvH<NW, 128, 3, 128, TR_M3, TR_RS, TRK_WM, TRK_WN, TRK_BR, 1, 0, 0, !W16, true, false, 0, true, 1, 1, 4>;
    using C4 = ConvH<NW, 128, 3, C4_CH, TR_M4, TR_RS, 1, 6, TRK_BR4, 1, 0, 0, !W16, true, false, 0, true, 1, 1, 8>;
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };

    if ((int)blockIdx.x < n_work) build(blockIdx.x, threadIdx.x, NW * 64);
#ifdef HM_TRUNK_STAMP
    // phases per layer: [0] barrier -> prologue loads issued, [1] k-loop, [2] epilogue, [3] wait at the next barrier
    unsigned long long ts[20], acc_t[16] = {};
    unsigned long long n_it = 0;
    const bool st_on = blockIdx.x == 0;
#define TS(i) do { if (st_on) ts[i] = hm_stamp(); } while (0)
#define MK(l) [&](int i) __attribute__((always_inline)) { if (st_on) ts[4 * (l) + 1 + i] = hm_stamp(); }
#else
#define TS(i)
#define MK(l) typename C2::NoMark{}
#endif
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        __syncthreads();  // feature rows, flags and the map row of this tile are in LDS; the previous conv4 is done with planes A
        TS(0);
        const int64_t grow0 = s_grow0;
        half_t* g1 = reinterpret_cast<half_t*>(mp.e[0]) + (size_t)grow0 * 256;
        half_t* g2 = reinterpret_cast<half_t*>(mp.e[1]) + (size_t)grow0 * 256;
        half_t* g3 = reinterpret_cast<half_t*>(mp.e[2]) + (size_t)grow0 * 256;
        C1::run(b_hi, b_hi, reinterpret_cast<const half_t*>(W.c1f), EpiTrunk<0>{a_hi, a_lo, W.c1f_bias, flags, g1}, MK(0));
        TS(3);
        __syncthreads();
        TS(4);
        C2::run(a_hi, a_lo, wf(1), EpiTrunk<1>{b_hi, b_lo, W.bias[1], flags, g2}, MK(1));
        TS(7);
        __syncthreads();
        TS(8);
        C3::run(b_hi, b_lo, wf(2), EpiTrunk<2>{a_hi, a_lo, W.bias[2], flags, g3}, MK(2));
        TS(11);
        __syncthreads();
        TS(12);
        // conv4 on 6 waves (96 channels); the other two build the next tile's feature rows in planes B meanwhile
        C4::run(a_hi, a_lo, wf(3), EpiE4{reinterpret_cast<half_t*>(mp.e4) + (size_t)grow0 * (2 * C4_CH), W.bias[3]}, MK(3));
        const int wn = w + gridDim.x;
        if (wn < n_work && (int)threadIdx.x >= 384) build(wn, threadIdx.x - 384, 128);
        TS(15);
#ifdef HM_TRUNK_STAMP
        if (st_on) {
            __syncthreads();
            ts[16] = hm_stamp();
            const bool c4w = threadIdx.x < 384;
            for (int l = 0; l < 4; ++l) {
                if (l == 3 && !c4w) { acc_t[12] += ts[15] - ts[12]; acc_t[15] += ts[16] - ts[15]; continue; }
                acc_t[4 * l + 0] += ts[4 * l + 1] - ts[4 * l];
                acc_t[4 * l + 1] += ts[4 * l + 2] - ts[4 * l + 1];
                acc_t[4 * l + 2] += ts[4 * l + 3] - ts[4 * l + 2];
                acc_t[4 * l + 3] += ts[4 * l + 4] - ts[4 * l + 3];
            }
            ++n_it;
        }
#endif
    }
#ifdef HM_TRUNK_STAMP
    if (st_on && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 16; ++i) atomicAdd(&g_trunk_stamp[threadIdx.x >> 6][i], acc_t[i]);
        atomicAdd(&g_trunk_stamp[threadIdx.x >> 6][16], n_it);
    }
#endif
#undef TS
#undef MK
}

// ------------------------------------------------------------------------------------------------------------------------
// Which map rows do the edge chains read?  Per tile and view, for each of E1..E3 a list of TR_OWN tile rows: the flagged ones
// first, the rest filled with row 0 (storing a valid row once more is harmless), so that the streaming trunk copies
// "every list entry" with no count and no branch.
template <int K1>
__global__ __launch_bounds__(128) void rowlist_kernel(const TrunkTile* __restrict__ tiles, int n_tiles, int ctx,
                                                       const RInfo* __restrict__ rinfo, const uint8_t* __restrict__ bases,
                                                       const uint8_t* __restrict__ sctx, uint8_t* __restrict__ rowlist) {
    using G = EdgeGeo<K1>;
    __shared__ int cnt0[3];
    const int w = blockIdx.x, view = w >= n_tiles, r = threadIdx.x;
    const TrunkTile tl = tiles[view ? w - n_tiles : w];
    const RInfo ri = rinfo[tl.read_idx];
    const int L = ri.len, x = tl.u0 + r, want_base = view ? 2 : 1;
    auto site_at = [&](int y) __attribute__((always_inline)) {
        if (y < 0 || y >= L) return 0;
        const int64_t j = ri.base_off + (view ? L - 1 - y : y);
        return (int)(sctx[j] == (ctx | (want_base == 2 ? 4 : 0)));   // context | strand << 2 (hm_kernels.h)
    };
    int f = 0;
    if (r < TR_OWN) {
        const int left = site_at(x - G::LEFT);
        f |= (left | site_at(x - G::R1) | (G::PAD2 ? 0 : site_at(x - G::R1 - 2))) << 0;
        f |= (left | site_at(x - G::R2) | (G::PAD3 ? 0 : site_at(x - G::R2 - 4))) << 1;
        f |= (left | site_at(x - G::R3) | (G::PAD4 ? 0 : site_at(x - G::R3 - 8))) << 2;
    }
    uint8_t* out = rowlist + (size_t)w * TR_RL;
    for (int i = r; i < TR_RL; i += 128) out[i] = 0;
    uint64_t bal[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) bal[l] = __ballot((f >> l) & 1);
    if (r < 3) cnt0[r] = 0;
    __syncthreads();  // also orders the zero fill before the entries below (same block, global memory)
    if (r == 0) {
#pragma unroll
        for (int l = 0; l < 3; ++l) cnt0[l] = __popcll(bal[l]);
    }
    __syncthreads();
    const int lane = r & 63;
#pragma unroll
    for (int l = 0; l < 3; ++l)
        if ((f >> l) & 1) out[l * TR_OWN + (r >= 64 ? cnt0[l] : 0) + __popcll(bal[l] & ((1ull << lane) - 1))] = (uint8_t)r;
}

// ------------------------------------------------------------------------------------------------------------------------
// trunk_kernel in the streaming form of hm_convs.h: 4 waves (one per SIMD), a wave owns 32 channels of a layer with its
// weights resident in registers, the positions stream through in groups of tiles.  Same tiles, same LDS planes, same maps.
namespace {

struct EpiTrunkS {
    half_t* hi;
    half_t* lo;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        half4 h, l;
        split4(acc, h, l);
        *reinterpret_cast<half4*>(hi + m * TR_RS + col) = h;
        *reinterpret_cast<half4*>(lo + m * TR_RS + col) = l;
    }
};

struct EpiE4S {
    half_t* __restrict__ g;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        half4 h, l;
        split4(acc, h, l);

        *reinterpret_cast<half4*>(g + (size_t)m * (2 * C4_CH) + col) = h;
        *reinterpret_cast<half4*>(g + (size_t)m * (2 * C4_CH) + C4_CH + col) = l;
    }
};

template <int NWV_>
struct CopyRows {
    static constexpr int NWV = NWV_, CS = TR_OWN / (2 * NWV_);  // NWV waves x 2 rows x CS slots = every own row of the tile
    const uint8_t* rows;  // LDS: this layer's list of TR_OWN row numbers
    half_t* g;            // map row of tile row 0
};

}  // namespace

// NW = 4: one wave per SIMD, 32 channels per wave (the default).  NW = 8: two waves per SIMD, 16 channels per wave -- twice
// the LDS operand reads, but two waves can issue MFMAs 14 % faster than one (tools/micro/mfma_rate.hip).
template <int K1, bool W16, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
void trunk2_kernel(const TrunkTile* __restrict__ tiles, int n_tiles, int n_views, int ctx, const RInfo* __restrict__ rinfo,
                   const uint8_t* __restrict__ bases, const uint32_t* __restrict__ kin, const uint8_t* __restrict__ sctx,
                   CtxWeights W, TrunkMaps mp) {
    constexpr int NTW = 8 / NW;  // channel tiles per wave in the 128-channel layers
    __shared__ __attribute__((aligned(16))) half_t smem[TR_LDS_HALVES + TR_XROWS * TR_WRS];
    __shared__ uint32_t rlist[2][3 * 32];  // the tile's row lists ([3][128] bytes, entries past TR_OWN stay 0); two buffers: conv4
                                          // still copies E3 rows while the next tile's arrive
    __shared__ int64_t s_grow0;
    half_t* a_hi = smem;
    half_t* a_lo = a_hi + TR_AROWS * TR_RS;
    half_t* b_hi = a_lo + TR_AROWS * TR_RS;
    half_t* b_lo = b_hi + TR_BROWS * TR_RS;
    half_t* xb = b_lo + TR_BROWS * TR_RS;  // feature rows of the tile (their own buffer: built while conv4 runs)
    const int n_work = n_tiles * n_views;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    for (int i = threadIdx.x; i < (TR_AROWS - TR_M1) * TR_RS; i += NW * 64) {
        a_hi[TR_M1 * TR_RS + i] = (half_t)0.f;
        a_lo[TR_M1 * TR_RS + i] = (half_t)0.f;
    }
    if (threadIdx.x < 2 * 3 * 32) rlist[0][threadIdx.x] = 0;
    __syncthreads();

    // The next tile's feature rows and map-row flags, one row per thread, in three steps that sit between the layers so
    // that no load's latency is exposed: (1) the tile and its read's descriptor, during conv1; (2) the row's base,
    // kinetics and four bytes of its row lists (rowlist_kernel), before conv3 -- branch-free, from clamped addresses; (3) decode + LDS
    // writes at the start of conv4.
    struct Build {
        TrunkTile tl;
        RInfo ri;
        int w, view, b;
        uint32_t k, rl;
    } bd;
    // the descriptors are two dependent loads of wave-uniform values: each is issued one layer before its result is made
    // scalar (readfirstlane), so that neither wait lands inside a layer's instruction stream
    auto build_desc1 = [&](const int w) __attribute__((always_inline)) {
        bd.w = w;
        bd.view = w >= n_tiles;
        bd.tl = tiles[bd.view ? w - n_tiles : w];
    };
    auto build_desc2 = [&]() __attribute__((always_inline)) {
        bd.tl.read_idx = __builtin_amdgcn_readfirstlane(bd.tl.read_idx);
        bd.tl.u0 = __builtin_amdgcn_readfirstlane(bd.tl.u0);
        bd.ri = rinfo[bd.tl.read_idx];
    };
    auto build_desc3 = [&]() __attribute__((always_inline)) {
        bd.ri.len = __builtin_amdgcn_readfirstlane(bd.ri.len);
        bd.ri.map_off = __builtin_amdgcn_readfirstlane(bd.ri.map_off);
        bd.ri.base_off = ((int64_t)__builtin_amdgcn_readfirstlane((int)(bd.ri.base_off >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)bd.ri.base_off);
    };
    auto grow_of = [&]() __attribute__((always_inline)) { return (int64_t)bd.view * mp.view_rows + bd.ri.map_off + (bd.tl.u0 + TR_PAD); };
    auto build_loads = [&]() __attribute__((always_inline)) {
        const int r = threadIdx.x, L = bd.ri.len, x = bd.tl.u0 + r;
        const int xc = min(max(x, 0), L - 1);
        const int64_t j = bd.ri.base_off + (bd.view ? L - 1 - xc : xc);
        bd.b = bases[j];
        bd.k = kin[j];
        bd.rl = reinterpret_cast<const uint32_t*>(mp.rowlist + (size_t)bd.w * TR_RL)[min(r, TR_RL / 4 - 1)];
    };
    auto build_store = [&](const int buf) __attribute__((always_inline)) {
        const int r = threadIdx.x, x = bd.tl.u0 + r;
        if (r == 0) s_grow0 = grow_of();
        if (r < TR_XROWS) {
            const bool in = x >= 0 && x < bd.ri.len;
            *reinterpret_cast<uint4*>(xb + r * TR_WRS) = feature_row(in ? bd.b : -1, bd.k, bd.view);
        }
        if (r < TR_RL / 4) rlist[buf][r / (TR_OWN / 4) * 32 + r % (TR_OWN / 4)] = bd.rl;
    };

    using C1 = SCfg<8, (2 * K1 + 3) / 4 * 4, TR_WRS, 1, true, false, K1, 2, NTW>;
    using C2 = SCfg<128, 3, TR_RS, 2, !W16, true, 0, 1, NTW>;
    using C3 = SCfg<128, 3, TR_RS, 4, !W16, true, 0, 1, NTW>;
    // conv4: 96 channels = 6 n-tiles x 7 position tiles = 42 tile pairs.  Four waves: a wave holds TWO n-tiles (a, b) -- 192
    // weight registers, like every other layer -- and runs the pair on one range of position tiles and a alone on the other,
    // b being shared by two waves that take complementary ranges: 11 + 10 + 11 + 10 tile pairs.  (Until round 3 three waves
    // took 14 pairs each and the fourth copied the E3 rows out, waiting a fifth of the tile at the barrier; three n-tiles
    // x half the positions on four waves needs 288 weight registers and spills.)  The E3 copy slots ride in all four waves.
    // Eight waves (trunk_impl = 2): six of them one n-tile each, as before.
    constexpr bool SPLIT4 = NW == 4;
    using C4 = SCfg<128, 3, TR_RS, 8, !W16, true, 0, 1, NTW>;
    using C4s = SCfg<128, 3, TR_RS, 8, !W16, true, 0, 1, 1>;   // one of the wave's two resident n-tiles alone
    using L4 = SConv<C4, C1, 0, 4, 3>;
    constexpr int NW4 = SPLIT4 ? NW : 6 / NTW;  // waves with channels in conv4
    // resident n-tiles nt04, nt04 + 1: waves 0 / 2 hold (a, b) = (0, 1) / (3, 4), waves 1 / 3 hold (b, a) = (1, 2) / (4, 5)
    const int nt04 = SPLIT4 ? (wave == 0 ? 0 : wave == 1 ? 1 : wave == 2 ? 3 : 4) : (wave < NW4 ? NTW * wave : 6 - NTW);
    using L1 = SConv<C1, C2, 0, 3, 3, 3>;
    using L2 = SConv<C2, C3, 0, 2, 3, 4>;  // the longest group last: it is the window in which conv3's weights can be fetched
    using L3 = SConv<C3, C4, 0, 4, 4>;
    static_assert(TR_M1 == 144 && TR_M2 == 144 && TR_M3 == 128 && TR_M4 == 112, "tile groups of the streaming layers");
    static_assert(TR_XROWS <= NW * 64, "one feature row per thread");
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };
    const half_t* c1f = reinterpret_cast<const half_t*>(W.c1f);
    const int nt0 = NTW * wave;

    WRegs wr;
    sconv_load_w<C1, 0, C1::KB>(c1f, nt0, lane, wr);
    sconv_load_bias<C1>(W.c1f_bias, nt0, lane, wr);
    if ((int)blockIdx.x < n_work) {
        build_desc1(blockIdx.x);
        build_desc2();
        build_desc3();
        build_loads();
        build_store(0);
    }
    int buf = 0;
#ifdef HM_TRUNK_STAMP
    // per layer: [0] barrier -> run returns, [3] wait at the next barrier
    unsigned long long ts[10], acc_t[16] = {};
    unsigned long long n_it = 0;
    const bool st_on = blockIdx.x == 0;
#define TS(i) do { if (st_on) ts[i] = hm_stamp(); } while (0)
#else
#define TS(i)
#endif
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        __syncthreads();
        TS(0);
        const int64_t grow0 = s_grow0;
        const int wn = w + gridDim.x;
        half_t* g1 = reinterpret_cast<half_t*>(mp.e[0]) + (size_t)grow0 * 256;
        half_t* g2 = reinterpret_cast<half_t*>(mp.e[1]) + (size_t)grow0 * 256;
        half_t* g3 = reinterpret_cast<half_t*>(mp.e[2]) + (size_t)grow0 * 256;
        const uint8_t* rl = reinterpret_cast<const uint8_t*>(rlist[buf]);
        L1::run(xb, xb, wr, EpiTrunkS{a_hi, a_lo}, wf(1), W.bias[1], nt0, nt0);
        build_desc1(min(wn, n_work - 1));  // unconditional (as are the loads below): the compiler's wait counts stay exact
        TS(1);
        __syncthreads();
        TS(2);
        build_desc2();
        L2::run(a_hi, a_lo, wr, EpiTrunkS{b_hi, b_lo}, wf(2), W.bias[2], nt0, nt0, CopyRows<NW>{rl, g1});
        TS(3);
        __syncthreads();
        TS(4);
        build_desc3();
        build_loads();
        L3::run(b_hi, b_lo, wr, EpiTrunkS{a_hi, a_lo}, wf(3), W.bias[3], nt0, nt04, CopyRows<NW>{rl + 128, g2});
        TS(5);
        __syncthreads();
        TS(6);
        const EpiE4S e4{reinterpret_cast<half_t*>(mp.e4) + (size_t)grow0 * (2 * C4_CH)};
        if constexpr (SPLIT4) {
            // the four-tile part first (it carries the E3 copy slots), the three-tile part last (behind it the next tile's conv1
            // weights are fetched into the registers it frees)
            if (wave & 1) {  // a = resident tile 1 alone on position tiles 0 .. 3, the pair on tiles 4 .. 6
                SConv<C4s, void, 0, 2, 2>::template run<1>(a_hi, a_lo, wr, e4, c1f, W.c1f_bias, nt04, nt0, CopyRows<NW>{rl + 256, g3});
                SConv<C4, C1, 4, 3>::run(a_hi, a_lo, wr, e4, c1f, W.c1f_bias, nt04, nt0);
            } else {         // the pair on position tiles 0 .. 3, a = resident tile 0 alone on tiles 4 .. 6
                SConv<C4, void, 0, 2, 2>::run(a_hi, a_lo, wr, e4, c1f, W.c1f_bias, nt04, nt0, CopyRows<NW>{rl + 256, g3});
                SConv<C4s, C1, 4, 3>::template run<0>(a_hi, a_lo, wr, e4, c1f, W.c1f_bias, nt04, nt0);
            }
        } else if (wave < NW4) {
            L4::run(a_hi, a_lo, wr, e4, c1f, W.c1f_bias, nt04, nt0);
        } else {  // a quarter of the waves has no channels in conv4: they copy the flagged E3 rows out (56 two-row slots)
            sconv_load_bias<C1>(W.c1f_bias, nt0, lane, wr);
            sconv_load_w<C1, 0, C1::KB>(c1f, nt0, lane, wr);
            constexpr int NCW = NW - NW4, QW = TR_OWN / 2 / NCW;  // copying waves, slots per wave
            static_assert(QW % 4 == 0, "copy slots in fours");
            const uint8_t* cpr = rl + 256 + 2 * QW * (wave - NW4) + (lane >> 5);
            half8 cd[4];
#pragma unroll
            for (int q0 = 0; q0 < QW; q0 += 4) {
                int row[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) row[q] = cpr[2 * (q0 + q)];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    cd[q] = *reinterpret_cast<const half8*>(((lane & 16) ? a_lo : a_hi) + row[q] * TR_RS + (lane & 15) * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<half8*>(g3 + (size_t)row[q] * 256 + (lane & 31) * 8) = cd[q];
            }
        }
        build_store(buf ^ 1);
        buf ^= 1;
        TS(7);
#ifdef HM_TRUNK_STAMP
        if (st_on) {
            __syncthreads();
            ts[8] = hm_stamp();
            for (int l = 0; l < 4; ++l) {
                acc_t[4 * l + 0] += ts[2 * l + 1] - ts[2 * l];
                acc_t[4 * l + 3] += ts[2 * l + 2] - ts[2 * l + 1];
            }
            ++n_it;
        }
#endif
    }
#ifdef HM_TRUNK_STAMP
    if (st_on && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 16; ++i) atomicAdd(&g_trunk_stamp[threadIdx.x >> 6][i], acc_t[i]);
        atomicAdd(&g_trunk_stamp[threadIdx.x >> 6][16], n_it);
    }
#endif
#undef TS
}

// ------------------------------------------------------------------------------------------------------------------------
// edge kernel: conv4 rows 0 and 24 of every site.  S sites per pass, two pseudo-rows per site (left edge, right edge):
//   conv1 (the output that touches the window's zero padding; folded constant taken back for the pad tap)
//   conv2..conv4: one output each over three taps that are either the previous layer's edge output ("specific"), a row
//   of the dense map, or the zero padding.
// ------------------------------------------------------------------------------------------------------------------------

template <int K1, bool W16>
__global__ __launch_bounds__(512) void edge_kernel(SiteRange sr, const RInfo* __restrict__ rinfo,
                                                    const uint8_t* __restrict__ bases, const uint32_t* __restrict__ kin,
                                                    CtxWeights W, TrunkMaps mp, uint16_t* __restrict__ edge4,
                                                    int32_t* __restrict__ e4row) {
    using G = EdgeGeo<K1>;
    constexpr int NW = 8;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    __shared__ __attribute__((aligned(16))) half_t smem[2 * EG_OP + 2 * EG_SP + EG_XH];
    __shared__ EdgeSite sinfo2[2][EG_S];
    half_t* op_hi = smem;
    half_t* op_lo = op_hi + EG_OP;
    half_t* sp_hi = op_lo + EG_OP;
    half_t* sp_lo = sp_hi + EG_SP;
    half_t* xb = sp_lo + EG_SP;  // feature rows of conv1's edge outputs: [pseudo-row][16 rows][8 halves]
    EdgeSite* sinfo = sinfo2[0];
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };

    // one 3-tap layer over the staged operand rows [pseudo-row][tap][128]
    using CE128 = ConvH<NW, 128, 3, 128, 1, TR_RS, 1, 8, 3, EG_M, 3 * TR_RS, 0, !W16, true, false, 0, false, 0, 1, 1>;
    using CE96 = ConvH<NW, 128, 3, C4_CH, 1, TR_RS, 1, 6, 3, EG_M, 3 * TR_RS, 0, !W16, true, false, 0, false, 0, 1, 1>;
    // all of conv1's weight blocks are requested in its prologue (ring depth = blocks + 1): the map rows can then be
    // requested right behind them, before the first MFMA, without standing in front of any later weight load
    constexpr int KB1E = (2 * K1 + 3) / 4 * 4 * 8 / 32;
    using C1E = ConvH<NW, 8, (2 * K1 + 3) / 4 * 4, 128, 1, TR_WRS, 1, 8, KB1E + 1, EG_M, EG_XROWS * TR_WRS, 0, true, false, false, K1, false, 0, 1, 1>;

    // Site descriptors + feature rows of the pass that starts at site s0, by threads [0, nt) (t < 0: not taking part).
    // A dependent chain of loads (site -> read -> base) followed by 1024 row builds: done for the NEXT pass by the two
    // waves that have no tile in the last layer (96 channels = 6 waves), so that a pass starts with everything in LDS.
    // The descriptors are a chain of three dependent loads (site -> read -> base at the site).  For the NEXT pass every
    // thread walks it for site (t mod 32), one link per layer of the current pass, so that each link's latency has a whole
    // layer to hide behind; all threads issue the same few loads unconditionally -- a load behind a branch would make the
    // compiler's wait counts for everything after it conservative.
    struct Desc {
        Site st;
        RInfo ri;
        int bs, valid;
    } nd;
    auto desc_a = [&](const int s0) __attribute__((always_inline)) {
        const int sc = min(s0, max(n_sites - 1, 0));                      // past the end: the last pass's sites again, dropped below
        const int nvalid = s0 < n_sites ? min(EG_S, n_sites - s0) : 0;
        const int t = threadIdx.x & (EG_S - 1);
        nd.valid = t < nvalid;
        nd.st = sites[min(sc + t, max(n_sites - 1, 0))];  // pad slots repeat the last site; their results are dropped
    };
    auto desc_b = [&]() __attribute__((always_inline)) { nd.ri = rinfo[nd.st.read_idx]; };
    auto desc_c = [&]() __attribute__((always_inline)) { nd.bs = bases[nd.ri.base_off + nd.st.qoff]; };
    auto desc_d = [&](const int s0, EdgeSite* si) __attribute__((always_inline)) {
        if ((int)threadIdx.x < EG_S) {
            EdgeSite es;
            es.bo = nd.ri.base_off;
            es.L = nd.ri.len;
            es.view = nd.bs == 2;
            es.off = es.view ? nd.ri.len - 1 - nd.st.qoff : nd.st.qoff;
            es.vrow = (int64_t)es.view * mp.view_rows + nd.ri.map_off + TR_PAD;
            es.valid = nd.valid;
            si[threadIdx.x] = es;
            if (es.valid) e4row[s0 + threadIdx.x] = (int32_t)(es.vrow + es.off - 215);
        }
    };
    // feature rows of conv1's first / last output: K1 rows per pseudo-row, the one on the zero padding all zeros.
    // NR rows per thread, all their loads issued before the first is used (clamped addresses, no branches).
    auto build_rows = [&](const EdgeSite* si, const int t, auto nt_) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_)::value, NR = EG_M * EG_XROWS / NT;
        int b[NR];
        uint32_t k[NR];
        bool live[NR];
        int view[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = t + u * NT;
            const int r = i / EG_XROWS, tt = i - r * EG_XROWS;
            const int side = r >= EG_S, site = r - side * EG_S;
            const EdgeSite& es = si[site];
            const int x = es.off + (side ? G::X_RIGHT : G::X_LEFT) + tt;
            const bool is_pad = side ? tt == K1 - 1 : tt == 0;
            live[u] = tt < K1 && !is_pad && x >= 0 && x < es.L;
            view[u] = es.view;
            const int xc = min(max(x, 0), es.L - 1);
            const int64_t j = es.bo + (es.view ? es.L - 1 - xc : xc);
            b[u] = bases[j];
            k[u] = kin[j];
        }
#pragma unroll
        for (int u = 0; u < NR; ++u)
            *reinterpret_cast<uint4*>(xb + (t + u * NT) * TR_WRS) = feature_row(live[u] ? b[u] : -1, k[u], view[u]);
    };

    // Shared (map) rows of layer `LAYER` (2..4) of all pseudo-rows: left chains read one row, right chains one or two.
    // They depend on the sites only, not on anything computed here, so a pass REQUESTS all of them (three layers) right
    // after its site descriptors are known and parks them in registers; the HBM latency then hides behind the feature
    // rows and conv1 instead of standing in front of every layer.
    auto n_map_chunks = [](int layer) constexpr {
        const bool pad = layer == 2 ? G::PAD2 : layer == 3 ? G::PAD3 : G::PAD4;
        return EG_S * (pad ? 2 : 3) * 32 / (NW * 64);  // rows x 32 sixteen-byte chunks / threads
    };
    constexpr int NC2 = n_map_chunks(2), NC3 = n_map_chunks(3), NC4 = n_map_chunks(4);
    uint4 m2[NC2], m3[NC3], m4[NC4];
    auto map_chunk = [&](const int layer, const int j, int& r, int& tap, int& plane, int& q, int& delta) __attribute__((always_inline)) {
        const int c = threadIdx.x + NW * 64 * j, mr = c >> 5;
        plane = (c >> 4) & 1;
        q = c & 15;
        const int grp = mr / EG_S, site = mr - grp * EG_S;  // 0: left chain, 1 / 2: first / second shared tap of the right chain
        const int rr = layer == 2 ? G::R1 : layer == 3 ? G::R2 : G::R3, step = layer == 2 ? 2 : layer == 3 ? 4 : 8;
        r = grp == 0 ? site : EG_S + site;
        tap = grp == 0 ? 2 : grp - 1;
        delta = grp == 0 ? G::LEFT : rr + (grp - 1) * step;
        return site;
    };
    auto request = [&](auto ltag) __attribute__((always_inline)) {
        constexpr int LAYER = decltype(ltag)::value, N = n_map_chunks(LAYER);
        const half_t* __restrict__ map = reinterpret_cast<const half_t*>(mp.e[LAYER - 2]);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            int r, tap, plane, q, delta;
            const int site = map_chunk(LAYER, j, r, tap, plane, q, delta);
            const EdgeSite& es = sinfo[site];
            const uint4 v = *reinterpret_cast<const uint4*>(map + (size_t)(es.vrow + es.off + delta) * 256 + plane * 128 + q * 8);
            if constexpr (LAYER == 2) m2[j] = v;
            else if constexpr (LAYER == 3) m3[j] = v;
            else m4[j] = v;
        }
    };
    // operand rows of layer LAYER: the parked map rows, the previous layer's edge outputs ("specific"), zero padding
    auto stage = [&](auto ltag) __attribute__((always_inline)) {
        constexpr int LAYER = decltype(ltag)::value, N = n_map_chunks(LAYER);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            int r, tap, plane, q, delta;
            map_chunk(LAYER, j, r, tap, plane, q, delta);
            uint4 v;
            if constexpr (LAYER == 2) v = m2[j];
            else if constexpr (LAYER == 3) v = m3[j];
            else v = m4[j];
            *reinterpret_cast<uint4*>((plane ? op_lo : op_hi) + (r * 3 + tap) * TR_RS + q * 8) = v;
        }
        // the previous layer's edge outputs ("specific" tap) and the zero tap, side by side: per side the tap of each kind is
        // a compile-time constant, so the two copies are plain strided loops (pseudo-row x plane x 16 sixteen-byte chunks)
        constexpr int HALF = EG_S * 2 * 16;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            int spec_tap = -1, zero_tap = -1;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const TapSrc src = tap_source<K1>(LAYER, side, tap);
                if (src.kind == SRC_SPEC) spec_tap = tap;
                if (src.kind == SRC_ZERO) zero_tap = tap;
            }
            for (int i = threadIdx.x; i < HALF; i += NW * 64) {
                const int q = i & 15, plane = (i >> 4) & 1, r = side * EG_S + (i >> 5);
                const uint4 v = *reinterpret_cast<const uint4*>((plane ? sp_lo : sp_hi) + r * TR_RS + q * 8);
                half_t* o = (plane ? op_lo : op_hi) + (r * 3) * TR_RS + q * 8;
                *reinterpret_cast<uint4*>(o + spec_tap * TR_RS) = v;
                if (zero_tap >= 0) *reinterpret_cast<uint4*>(o + zero_tap * TR_RS) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    using L2t = std::integral_constant<int, 2>;
    using L3t = std::integral_constant<int, 3>;
    using L4t = std::integral_constant<int, 4>;

    if ((int)blockIdx.x * EG_S < n_sites) {  // first pass of this workgroup: prepared by everybody
        desc_a(blockIdx.x * EG_S);
        desc_b();
        desc_c();
        desc_d(blockIdx.x * EG_S, sinfo2[0]);
        __syncthreads();
        build_rows(sinfo2[0], threadIdx.x, std::integral_constant<int, NW * 64>{});
    }
    int cur = 0;
#ifdef HM_TRUNK_STAMP
    unsigned long long ets[12], eacc[12] = {};
    unsigned long long en_it = 0;
    const bool est_on = blockIdx.x == 0;
#define ETS(i) do { if (est_on) ets[i] = hm_stamp(); } while (0)
#else
#define ETS(i)
#endif
    for (int s0 = blockIdx.x * EG_S; s0 < n_sites; s0 += gridDim.x * EG_S) {
        const int nvalid = min(EG_S, n_sites - s0);
        sinfo = sinfo2[cur];
        __syncthreads();  // descriptors + feature rows of this pass are in LDS; the previous pass is done with the planes
        ETS(0);
        const int sn = s0 + gridDim.x * EG_S;
        desc_a(sn);
        {
            const int col = (threadIdx.x >> 6) * 16 + 4 * ((threadIdx.x & 63) >> 4);  // this lane's channels in conv1 (1 x 8 wave grid)
            const EpiSpecC1 epi1{sp_hi, sp_lo, W.c1f_bias, *reinterpret_cast<const float4*>(W.c1f_corr + col),
                                 *reinterpret_cast<const float4*>(W.c1f_corr + 128 + col)};
            C1E::run(xb, xb, reinterpret_cast<const half_t*>(W.c1f), epi1,
                     [&](int i) __attribute__((always_inline)) { if (i == 0) request(L2t{}); });
        }
        request(L3t{});
        request(L4t{});
        ETS(1);
        __syncthreads();
        ETS(2);
        stage(L2t{});
        __syncthreads();
        ETS(3);
        desc_b();
        CE128::run(op_hi, op_lo, wf(1), EpiSpec{sp_hi, sp_lo, W.bias[1]});
        ETS(4);
        __syncthreads();
        ETS(5);
        stage(L3t{});
        __syncthreads();
        ETS(6);
        desc_c();
        CE128::run(op_hi, op_lo, wf(2), EpiSpec{sp_hi, sp_lo, W.bias[2]});
        __syncthreads();
        ETS(7);
        desc_d(sn, sinfo2[cur ^ 1]);
        stage(L4t{});
        __syncthreads();
        ETS(8);
        if ((int)threadIdx.x >= 384) {  // waves 6, 7 have no tile in the 96-channel layer: they build the next pass's feature rows
            build_rows(sinfo2[cur ^ 1], threadIdx.x - 384, std::integral_constant<int, 128>{});
        } else {
            CE96::run(op_hi, op_lo, wf(3), EpiEdgeOut{reinterpret_cast<half_t*>(edge4) + (size_t)s0 * (4 * C4_CH), W.bias[3], nvalid});
        }
        ETS(9);
        cur ^= 1;
#ifdef HM_TRUNK_STAMP
        if (est_on) {
            for (int i = 0; i < 9; ++i) eacc[i] += ets[i + 1] - ets[i];
            ++en_it;
        }
#endif
    }
#ifdef HM_TRUNK_STAMP
    if (est_on && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 9; ++i) atomicAdd(&g_edge_stamp[threadIdx.x >> 6][i], eacc[i]);
        atomicAdd(&g_edge_stamp[threadIdx.x >> 6][9], en_it);
    }
#endif
#undef ETS
}

// ------------------------------------------------------------------------------------------------------------------------
void launch_trunk(hipStream_t st, int k1, const TrunkTile* tiles, int n_tiles, int n_views, int ctx, const RInfo* rinfo,
                  const uint8_t* bases, const uint32_t* kin, const uint8_t* sctx, const CtxWeights& w,
                  const TrunkMaps& maps, int grid, bool w16) {
    if (n_tiles <= 0) return;
    const dim3 g(min(n_tiles * n_views, grid)), b(512);
#define HM_TRUNK(K1, W16) \
    hipLaunchKernelGGL((trunk_kernel<K1, W16>), g, b, 0, st, tiles, n_tiles, n_views, ctx, rinfo, bases, kin, sctx, w, maps)
    (void)w16;  // (plain-fp16-weight variants: closed in round 3, no longer instantiated)
    if (k1 == 11) HM_TRUNK(11, false); else HM_TRUNK(13, false);
#undef HM_TRUNK
}

void launch_trunk2(hipStream_t st, int k1, const TrunkTile* tiles, int n_tiles, int n_views, int ctx, const RInfo* rinfo,
                   const uint8_t* bases, const uint32_t* kin, const uint8_t* sctx, const CtxWeights& w,
                   const TrunkMaps& maps, int grid, bool w16, bool waves8) {
    if (n_tiles <= 0) return;
    const dim3 g(min(n_tiles * n_views, grid)), b(waves8 ? 512 : 256);
    if (k1 == 11) hipLaunchKernelGGL(rowlist_kernel<11>, dim3(n_tiles * n_views), dim3(128), 0, st, tiles, n_tiles, ctx, rinfo, bases, sctx, maps.rowlist);
    else hipLaunchKernelGGL(rowlist_kernel<13>, dim3(n_tiles * n_views), dim3(128), 0, st, tiles, n_tiles, ctx, rinfo, bases, sctx, maps.rowlist);
#define HM_TRUNK(K1, W16) \
    do { if (waves8) hipLaunchKernelGGL((trunk2_kernel<K1, W16, 8>), g, b, 0, st, tiles, n_tiles, n_views, ctx, rinfo, bases, kin, sctx, w, maps); \
         else hipLaunchKernelGGL((trunk2_kernel<K1, W16, 4>), g, b, 0, st, tiles, n_tiles, n_views, ctx, rinfo, bases, kin, sctx, w, maps); } while (0)
    (void)w16;
    if (k1 == 11) HM_TRUNK(11, false); else HM_TRUNK(13, false);
#undef HM_TRUNK
}

void launch_edge(hipStream_t st, int k1, const SiteRange& sr, const RInfo* rinfo, const uint8_t* bases,
                 const uint32_t* kin, const CtxWeights& w, const TrunkMaps& maps, uint16_t* edge4, int32_t* e4row, int grid,
                 bool w16) {
    if (sr.cap <= 0) return;
    const dim3 g(sr.totals ? grid : max(1, min((sr.cap + EG_S - 1) / EG_S, grid))), b(512);
#define HM_EDGE(K1, W16) hipLaunchKernelGGL((edge_kernel<K1, W16>), g, b, 0, st, sr, rinfo, bases, kin, w, maps, edge4, e4row)
    (void)w16;
    if (k1 == 11) HM_EDGE(11, false); else HM_EDGE(13, false);
#undef HM_EDGE
}

size_t trunk_lds_bytes() { return sizeof(half_t) * TR_LDS_HALVES; }

}  // namespace hm
