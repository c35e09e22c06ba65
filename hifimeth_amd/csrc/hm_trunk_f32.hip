// hm_trunk_f32.hip -- the dense trunk and the edge kernel in strict fp32 (engine option precision = 0): the same dataflow as
// hm_trunk.hip -- conv1..conv4 once per (read, strand view) position as dilated maps E1..E4, the two conv4 rows per site
// that touch the window padding from a chain of one-row layers -- on v_mfma_f32_16x16x4_f32: exact fp32 products, fp32
// accumulation, no operand splitting.  Differences from the split-half kernels: activations are plain fp32 rows (528-byte
// LDS rows, the same footprint as hi + lo planes), bn0 is applied when the feature rows are decoded (lookup tables computed
// on the host with the ONNX formula: bit-identical to the oracle's bn0), maps and edge rows are fp32.
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98).
#include "hm_conv32.h"

namespace hm {

namespace {

constexpr int F_M1 = 144, F_M2 = 144, F_M3 = 128, F_M4 = TR_OWN;  // rows computed per layer (needed: 140 / 136 / 128 / 112)
constexpr int F_XROWS = 160;
constexpr int F_RS = 132;   // floats per LDS row: 128 channels + 16 B pad
constexpr int F_WRS = 12;   // floats per feature row (8 used)
constexpr int F_AROWS = 148, F_BROWS = 144;
static_assert(F_XROWS * F_WRS <= F_BROWS * F_RS, "the feature rows live in buffer B");

template <int K1>
struct EdgeGeoF {  // see EdgeGeo in hm_trunk.hip
    static constexpr int L1 = (KMER + 2 - K1) / 2 + 1, L2 = (L1 - 1) / 2 + 1, L3 = (L2 - 1) / 2 + 1;
    static constexpr bool PAD2 = 2 * (L2 - 1) + 1 == L1, PAD3 = 2 * (L3 - 1) + 1 == L2, PAD4 = 2 * (C4_LEN - 1) + 1 == L3;
    static constexpr int R1 = -201 + 2 * (2 * (L2 - 1) - 1), R2 = -203 + 4 * (2 * (L3 - 1) - 1), R3 = -207 + 8 * (2 * (C4_LEN - 1) - 1);
    static constexpr int LEFT = -199;
    static constexpr int X_LEFT = -201, X_RIGHT = -201 + 2 * (L1 - 1);
    static constexpr int KT1 = (K1 * FEATS + 15) / 16 * 2;  // conv1 taps incl. zero-weight K padding: 12 / 14
};

// bn0-normalised feature row of a read position (the reference zero-fills rows outside the read BEFORE bn0:
// eval_kmer_features.cpp:36-40, so those rows are bn0(0), not 0)
__device__ __forceinline__ void feature_row_f32(const BnTables* __restrict__ bn, int b, uint32_t k, int view, float4& lo, float4& hi) {
    if (b < 0) {
        lo = make_float4(bn->zero[0], bn->zero[1], bn->zero[2], bn->zero[3]);
        hi = make_float4(bn->zero[4], bn->zero[5], bn->zero[6], bn->zero[7]);
        return;
    }
    if (view) {
        if (b < 4) b = 3 - b;
        k = (k >> 16) | (k << 16);
    }
    lo = make_float4(b == 0 ? bn->hot[0] : bn->zero[0], b == 1 ? bn->hot[1] : bn->zero[1], b == 2 ? bn->hot[2] : bn->zero[2],
                     b == 3 ? bn->hot[3] : bn->zero[3]);
    hi = make_float4(bn->lut[0][k & 255], bn->lut[1][(k >> 8) & 255], bn->lut[2][(k >> 16) & 255], bn->lut[3][k >> 24]);
}

template <int BIT>
struct EpiT32 {  // ReLU -> LDS row m; flagged rows of the tile's own 112 also to the map
    float* out;
    const float* __restrict__ bias;
    const uint8_t* flags;
    float* __restrict__ g;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4_t& acc) const {
        const float4 v = make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
        *reinterpret_cast<float4*>(out + m * F_RS + col) = v;
        if (m < TR_OWN && ((flags[m] >> BIT) & 1)) *reinterpret_cast<float4*>(g + (size_t)m * 128 + col) = v;
    }
    __device__ __forceinline__ void store1(int, int, float) const {}
};

struct EpiE4F {
    float* __restrict__ g;
    const float* __restrict__ bias;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4_t& acc) const {
        *reinterpret_cast<float4*>(g + (size_t)m * C4_CH + col) =
            make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    }
    __device__ __forceinline__ void store1(int, int, float) const {}
};

}  // namespace

template <int K1>
__global__ __launch_bounds__(512) void trunk_kernel_f32(const TrunkTile* __restrict__ tiles, int n_tiles, int n_views, int ctx,
                                                         const RInfo* __restrict__ rinfo, const uint8_t* __restrict__ bases,
                                                         const uint32_t* __restrict__ kin, const uint8_t* __restrict__ sctx,
                                                         CtxWeights W, TrunkMaps mp) {
    using G = EdgeGeoF<K1>;
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) float smem[(F_AROWS + F_BROWS) * F_RS];
    __shared__ uint8_t flags[F_XROWS];
    __shared__ int64_t s_grow0;
    float* bufA = smem;
    float* bufB = smem + F_AROWS * F_RS;
    const BnTables* __restrict__ bn = W.bn;
    const int n_work = n_tiles * n_views;
    for (int i = threadIdx.x; i < (F_AROWS - F_M1) * F_RS; i += NW * 64) bufA[F_M1 * F_RS + i] = 0.f;  // read by conv2's last tile only

    auto build = [&](const int w, const int t, const int nt) __attribute__((always_inline)) {
        const int view = w >= n_tiles;
        const TrunkTile tl = tiles[view ? w - n_tiles : w];
        const RInfo ri = rinfo[tl.read_idx];
        const int L = ri.len;
        const int64_t bo = ri.base_off;
        if (t == 0) s_grow0 = (int64_t)view * mp.view_rows + ri.map_off + (tl.u0 + TR_PAD);
        const int want_base = view ? 2 : 1;
        auto site_at = [&](int y) __attribute__((always_inline)) {
            if (y < 0 || y >= L) return 0;
            const int64_t j = bo + (view ? L - 1 - y : y);
            return (int)(sctx[j] == (ctx | (want_base == 2 ? 4 : 0)));   // context | strand << 2 (hm_kernels.h)
        };
        for (int r = t; r < F_XROWS; r += nt) {
            const int x = tl.u0 + r;
            int b = -1;
            uint32_t k = 0;
            if (x >= 0 && x < L) {
                const int64_t j = bo + (view ? L - 1 - x : x);
                b = bases[j];
                k = kin[j];
            }
            float4 lo, hi;
            feature_row_f32(bn, b, k, view, lo, hi);
            *reinterpret_cast<float4*>(bufB + r * F_WRS) = lo;
            *reinterpret_cast<float4*>(bufB + r * F_WRS + 4) = hi;
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

    if ((int)blockIdx.x < n_work) build(blockIdx.x, threadIdx.x, NW * 64);
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        __syncthreads();
        const int64_t grow0 = s_grow0;
        float* g1 = reinterpret_cast<float*>(mp.e[0]) + (size_t)grow0 * 128;
        float* g2 = reinterpret_cast<float*>(mp.e[1]) + (size_t)grow0 * 128;
        float* g3 = reinterpret_cast<float*>(mp.e[2]) + (size_t)grow0 * 128;
        Conv<NW, 1, 8, G::KT1, 128, F_M1, F_WRS, 0, 0, 1, 8, 2, 0, 1, 1>::run(bufB, W.wfrag[0], EpiT32<0>{bufA, W.bias[0], flags, g1});
        __syncthreads();
        Conv<NW, 1, 128, 3, 128, F_M2, F_RS, 0, 0, 1, 8, 2, 0, 1, 2>::run(bufA, W.wfrag[1], EpiT32<1>{bufB, W.bias[1], flags, g2});
        __syncthreads();
        Conv<NW, 1, 128, 3, 128, F_M3, F_RS, 0, 0, 1, 8, 3, 0, 1, 4>::run(bufB, W.wfrag[2], EpiT32<2>{bufA, W.bias[2], flags, g3});
        __syncthreads();
        Conv<NW, 1, 128, 3, C4_CH, F_M4, F_RS, 0, 0, 1, 6, 3, 0, 1, 8>::run(
            bufA, W.wfrag[3], EpiE4F{reinterpret_cast<float*>(mp.e4) + (size_t)grow0 * C4_CH, W.bias[3]});
        const int wn = w + gridDim.x;
        if (wn < n_work && (int)threadIdx.x >= 384) build(wn, threadIdx.x - 384, 128);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int FG_S = 32, FG_M = 2 * FG_S;
constexpr int FG_XROWS = 16;
constexpr int FG_OP = FG_M * 3 * F_RS;  // floats of the operand buffer
constexpr int FG_SP = FG_M * F_RS;
static_assert(FG_M * FG_XROWS * F_WRS <= FG_OP, "the feature rows alias the operand buffer");

struct EdgeSiteF {
    int64_t bo, vrow;
    int L, off, view, pad;
};

template <int K1>
__device__ __forceinline__ void tap_source_f(int layer, int side, int tap, int& kind, int& delta) {  // 0 zero, 1 specific, 2 map
    using G = EdgeGeoF<K1>;
    if (side == 0) {
        kind = tap == 0 ? 0 : tap == 1 ? 1 : 2;
        delta = G::LEFT;
        return;
    }
    const bool pad = layer == 2 ? G::PAD2 : layer == 3 ? G::PAD3 : G::PAD4;
    const int r = layer == 2 ? G::R1 : layer == 3 ? G::R2 : G::R3, step = layer == 2 ? 2 : layer == 3 ? 4 : 8;
    if (pad) {
        kind = tap == 0 ? 2 : tap == 1 ? 1 : 0;
        delta = r;
    } else {
        kind = tap == 2 ? 1 : 2;
        delta = tap == 0 ? r : r + step;
    }
}

struct EpiSpecF {
    float* out;
    const float* __restrict__ bias;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4_t& acc) const {
        *reinterpret_cast<float4*>(out + m * F_RS + col) =
            make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    }
    __device__ __forceinline__ void store1(int, int, float) const {}
};

struct EpiEdgeOutF {
    float* __restrict__ out;  // [site][2][96] of this pass
    const float* __restrict__ bias;
    int nvalid;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4_t& acc) const {
        const int side = m >= FG_S, site = m - side * FG_S;
        if (site < nvalid)
            *reinterpret_cast<float4*>(out + (size_t)site * (2 * C4_CH) + side * C4_CH + col) =
                make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    }
    __device__ __forceinline__ void store1(int, int, float) const {}
};

}  // namespace

template <int K1>
__global__ __launch_bounds__(512) void edge_kernel_f32(SiteRange sr, const RInfo* __restrict__ rinfo,
                                                        const uint8_t* __restrict__ bases, const uint32_t* __restrict__ kin,
                                                        CtxWeights W, TrunkMaps mp, float* __restrict__ edge4,
                                                        int32_t* __restrict__ e4row) {
    using G = EdgeGeoF<K1>;
    constexpr int NW = 8;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    __shared__ __attribute__((aligned(16))) float smem[FG_OP + FG_SP];
    __shared__ EdgeSiteF sinfo[FG_S];
    float* op = smem;
    float* sp = smem + FG_OP;
    const BnTables* __restrict__ bn = W.bn;

    auto stage = [&](const int layer) __attribute__((always_inline)) {
        const float* __restrict__ map = reinterpret_cast<const float*>(mp.e[layer - 2]);
        constexpr int CHUNKS = FG_M * 3 * 32;  // 16-byte chunks: pseudo-row x tap x 32
        for (int i = threadIdx.x; i < CHUNKS; i += NW * 64) {
            const int q = i & 31, rt = i >> 5;
            const int r = rt / 3, tap = rt - 3 * r;
            const int side = r >= FG_S, site = r - side * FG_S;
            int kind, delta;
            tap_source_f<K1>(layer, side, tap, kind, delta);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kind == 1) {
                v = *reinterpret_cast<const float4*>(sp + r * F_RS + q * 4);
            } else if (kind == 2) {
                const EdgeSiteF& es = sinfo[site];
                v = *reinterpret_cast<const float4*>(map + (size_t)(es.vrow + es.off + delta) * 128 + q * 4);
            }
            *reinterpret_cast<float4*>(op + (r * 3 + tap) * F_RS + q * 4) = v;
        }
    };
    using CE128 = Conv<NW, FG_M, 128, 3, 128, 1, F_RS, 3 * F_RS, 0, 1, 8, 2, 0, 1, 1>;
    using CE96 = Conv<NW, FG_M, 128, 3, C4_CH, 1, F_RS, 3 * F_RS, 0, 1, 6, 2, 0, 1, 1>;
    using C1E = Conv<NW, FG_M, 8, G::KT1, 128, 1, F_WRS, FG_XROWS * F_WRS, 0, 1, 8, 2, 0, 1, 1>;

    for (int s0 = blockIdx.x * FG_S; s0 < n_sites; s0 += gridDim.x * FG_S) {
        const int nvalid = min(FG_S, n_sites - s0);
        __syncthreads();
        if (threadIdx.x < FG_S) {
            const int i = min((int)threadIdx.x, nvalid - 1);
            const Site st = sites[s0 + i];
            const RInfo ri = rinfo[st.read_idx];
            EdgeSiteF es;
            es.bo = ri.base_off;
            es.L = ri.len;
            es.view = bases[ri.base_off + st.qoff] == 2;
            es.off = es.view ? ri.len - 1 - st.qoff : st.qoff;
            es.vrow = (int64_t)es.view * mp.view_rows + ri.map_off + TR_PAD;
            es.pad = 0;
            sinfo[threadIdx.x] = es;
            if ((int)threadIdx.x < nvalid) e4row[s0 + threadIdx.x] = (int32_t)(es.vrow + es.off - 215);
        }
        __syncthreads();
        // feature rows of conv1's first / last output; the row on the conv's zero padding is 0 AFTER bn0, i.e. plain zeros
        for (int i = threadIdx.x; i < FG_M * FG_XROWS; i += NW * 64) {
            const int r = i / FG_XROWS, t = i - r * FG_XROWS;
            const int side = r >= FG_S, site = r - side * FG_S;
            const EdgeSiteF& es = sinfo[site];
            const int x = es.off + (side ? G::X_RIGHT : G::X_LEFT) + t;
            const bool is_pad = side ? t == K1 - 1 : t == 0;
            float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
            if (t < K1 && !is_pad) {
                int b = -1;
                uint32_t k = 0;
                if (x >= 0 && x < es.L) {
                    const int64_t j = es.bo + (es.view ? es.L - 1 - x : x);
                    b = bases[j];
                    k = kin[j];
                }
                feature_row_f32(bn, b, k, es.view, lo, hi);
            }
            *reinterpret_cast<float4*>(op + i * F_WRS) = lo;
            *reinterpret_cast<float4*>(op + i * F_WRS + 4) = hi;
        }
        __syncthreads();
        C1E::run(op, W.wfrag[0], EpiSpecF{sp, W.bias[0]});
        __syncthreads();
        stage(2);
        __syncthreads();
        CE128::run(op, W.wfrag[1], EpiSpecF{sp, W.bias[1]});
        __syncthreads();
        stage(3);
        __syncthreads();
        CE128::run(op, W.wfrag[2], EpiSpecF{sp, W.bias[2]});
        __syncthreads();
        stage(4);
        __syncthreads();
        CE96::run(op, W.wfrag[3], EpiEdgeOutF{edge4 + (size_t)s0 * (2 * C4_CH), W.bias[3], nvalid});
    }
}

// ------------------------------------------------------------------------------------------------------------------------
void launch_trunk_f32(hipStream_t st, int k1, const TrunkTile* tiles, int n_tiles, int n_views, int ctx, const RInfo* rinfo,
                      const uint8_t* bases, const uint32_t* kin, const uint8_t* sctx, const CtxWeights& w,
                      const TrunkMaps& maps, int grid) {
    if (n_tiles <= 0) return;
    const dim3 g(min(n_tiles * n_views, grid)), b(512);
    if (k1 == 11) hipLaunchKernelGGL(trunk_kernel_f32<11>, g, b, 0, st, tiles, n_tiles, n_views, ctx, rinfo, bases, kin, sctx, w, maps);
    else hipLaunchKernelGGL(trunk_kernel_f32<13>, g, b, 0, st, tiles, n_tiles, n_views, ctx, rinfo, bases, kin, sctx, w, maps);
}

void launch_edge_f32(hipStream_t st, int k1, const SiteRange& sr, const RInfo* rinfo, const uint8_t* bases, const uint32_t* kin,
                     const CtxWeights& w, const TrunkMaps& maps, float* edge4, int32_t* e4row, int grid) {
    if (sr.cap <= 0) return;
    const dim3 g(sr.totals ? grid : max(1, min((sr.cap + FG_S - 1) / FG_S, grid))), b(512);
    if (k1 == 11) hipLaunchKernelGGL(edge_kernel_f32<11>, g, b, 0, st, sr, rinfo, bases, kin, w, maps, edge4, e4row);
    else hipLaunchKernelGGL(edge_kernel_f32<13>, g, b, 0, st, sr, rinfo, bases, kin, w, maps, edge4, e4row);
}

}  // namespace hm
