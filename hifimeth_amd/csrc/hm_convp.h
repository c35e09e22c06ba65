// hm_convp.h -- the split-half implicit-GEMM convolution of the STRIP tail kernel (hm_tail_p.hip): conv5 .. conv8 and fc1 over
// 16 sites per pass with the sites along the MFMA tile's ROWS ("p-major"): m-tile p of a layer = output position p of all 16
// sites, lane (li = site, lk).  hm_convt.h's TConv stacks the sites' rows back to back along M (m -> site = m / LOUT), which
//   * mixes positions inside a tile, so the taps that fall on a site's zero padding (the first tap of position 0, the last tap of
//     the last position) are multiplied out like any other: 5 % of conv5's products, 10 % of conv6's, 17 % of conv7's / conv8's;
//   * leaves ragged last tiles (8 x 13 = 104 rows in 7 tiles, 8 x 7 = 56 in 4);
//   * needs a row offset per (lane, tile) for the operand reads and another for the stores.
// Here a tile's position is a compile-time constant: its padding taps are skipped altogether (`IA::skip`; the products are exact
// zeros, so every accumulator still sees bias, then its k-blocks in ascending order with w_hi x_hi, w_hi x_lo, w_lo x_hi each:
// the same fp32 sums as TConv / ConvH, bit for bit), 16 sites fill every tile, and every LDS address of the stream is ONE
// per-lane base (the lane's site) plus an immediate.
//
// Planes are [row][16 sites][RS halves]: a site's rows are 16 RS apart, neighbouring lanes RS apart (RS = 104 | 72 halves: 52 | 36
// dwords, so the 16 sites of a ds_read_b128 fall into different banks).
//
// Reference for what is computed: training/model_cnn.py:8-85 (conv5 .. fc1), models/*.onnx as loaded at mod_main.cpp:32-98.
#pragma once
#include "hm_convt.h"

namespace hm {

// layer geometry: CIN channels per tap, KT taps
// (WLO = false: plain fp16 weights, the w_lo x_hi product dropped -- hm_convt.h's TCfg)
template <int CIN_, int KT_, bool WLO_ = true>
struct PCfg {
    static constexpr int CIN = CIN_, KT = KT_, KB = KT_ * CIN_ / 32;
    static constexpr bool WLO = WLO_;
    static constexpr int NPR = WLO_ ? 3 : 2;
    static_assert((KT_ * CIN_) % 32 == 0 && CIN_ % 32 == 0, "a k-block never straddles taps");
    static constexpr int tap(int kb) { return kb * 32 / CIN; }
    static constexpr int ch0(int kb) { return kb * 32 - tap(kb) * CIN; }
};

// Operand addresses of a layer whose input is a p-major plane pair of LIN data rows starting BASE halves behind the plane
// pointers: output tile t reads data rows MSTR t + ROW0 + tap (stride-2 conv: MSTR = 2, ROW0 = -1; fc1 over conv8's two rows:
// tile 0, ROW0 = 0).  Rows outside [0, LIN) are the layer's zero padding: skipped.
template <class C, int RS, int LIN, int BASE, int MSTR = 2, int ROW0 = -1>
struct PInRows {
    int lb;  // li * RS + 8 * lk: this lane's site and K offset
    static constexpr int row(int tile, int kb) { return MSTR * tile + ROW0 + C::tap(kb); }
    static constexpr bool skip(int tile, int kb) { return row(tile, kb) < 0 || row(tile, kb) >= LIN; }
    template <int TILE, int KB>
    __device__ __forceinline__ int off() const { return lb + (BASE + row(TILE, KB) * 16 * RS + C::ch0(KB)); }
};

// Epilogues: ReLU + hi / lo split -> p-major planes.  The pointers are this lane's: plane + li * ORS + 4 * lk.
template <int ORS>
struct EpiP {
    static constexpr int NV0 = 6, NV1 = 1, NW = 2, WMASK = 0x200;
    struct St { half4 h, l; };
    half_t* hi;
    half_t* lo;
    __device__ __forceinline__ void s0(const f32x4& acc, St& s) const { split4(acc, s.h, s.l); }
    template <int TILE>
    __device__ __forceinline__ void s1(int col, const St& s) const {
        *reinterpret_cast<half4*>(hi + TILE * 16 * ORS + col) = s.h;
        *reinterpret_cast<half4*>(lo + TILE * 16 * ORS + col) = s.l;
    }
};
// conv8's rows on their way to tail_fc_kernel (hm_tail_fc.hip): ReLU + split -> global, a site's 512 bytes = [hi: position * 64 + channel | lo];
// `g` is this lane's: x8 + (the site's list position) * TAIL_X8_HALVES + 4 * lk
struct EpiX8 {
    static constexpr int NV0 = 6, NV1 = 1, NW = 2, WMASK = 0x040;
    struct St { half4 h, l; };
    half_t* g;
    __device__ __forceinline__ void s0(const f32x4& acc, St& s) const { split4(acc, s.h, s.l); }
    template <int TILE>
    __device__ __forceinline__ void s1(int col, const St& s) const {
        *reinterpret_cast<half4*>(g + TILE * 64 + col) = s.h;
        *reinterpret_cast<half4*>(g + TAIL_X8_HALVES / 2 + TILE * 64 + col) = s.l;
    }
};
// fc1: ReLU, fp32 h[site][8 parts of 32, HPS floats apart] for the VALU fc2; `out` is this lane's: h + li * (8 HPS) + 4 * lk
template <int HPS>
struct EpiFc1P {
    static constexpr int NV0 = 4, NV1 = 1, NW = 1, WMASK = 0x200;
    struct St { float4 v; };
    float* out;
    __device__ __forceinline__ void s0(const f32x4& acc, St& s) const { s.v = make_float4(relu1(acc[0]), relu1(acc[1]), relu1(acc[2]), relu1(acc[3])); }
    template <int TILE>
    __device__ __forceinline__ void s1(int col, const St& s) const { *reinterpret_cast<float4*>(out + col + (col >> 5) * (HPS - 32)) = s.v; }
};

// The stream: groups of tiles (hm_convt.h's TG) run back to back as (group, k-block) blocks; per block the MFMAs of the tiles that
// do not skip this k-block, a share of the previous group's epilogue (two stages per accumulator) and hook(block).  Operands come
// through a ring of NS half8 registers, read at most LA blocks ahead.
template <class C, class IA, int NS, int LA, class... GR>
struct PConv {
    static constexpr int NG = sizeof...(GR);
    static constexpr int nts[NG] = {GR::NT...}, nas[NG] = {GR::NA...};
    static constexpr int KB = C::KB, NB = NG * KB;
    static constexpr int amax() { int m = 0; for (int g = 0; g < NG; ++g) m = nas[g] > m ? nas[g] : m; return m; }
    static constexpr int tmax() { int m = 0; for (int g = 0; g < NG; ++g) m = nts[g] > m ? nts[g] : m; return m; }
    static constexpr int AMAX = amax(), TMAX = tmax();
    static_assert(NS >= 2 * TMAX, "the ring holds at least one block's operands");
    template <int g>
    using Grp = std::tuple_element_t<g, std::tuple<GR...>>;
    // i-th tile of group g / tile index of accumulator a, for the compile-time bookkeeping that walks over blocks
    static constexpr int tile_of(int g, int i) { int r = 0, gi = 0; ((gi++ == g ? void(r = GR::tile(i)) : void()), ...); return r; }
    static constexpr int acc_tile_of(int g, int a) { int r = 0, gi = 0; ((gi++ == g ? void(r = GR::acc_tile(a)) : void()), ...); return r; }
    static constexpr bool skips(int g, int i, int kb) { return IA::skip(tile_of(g, i), kb); }
    static constexpr int nreads(int c) { int n = 0; for (int i = 0; i < nts[c / KB]; ++i) n += skips(c / KB, i, c % KB) ? 0 : 2; return n; }
    static constexpr int rslot(int c, int i) { int n = 0; for (int q = 0; q < i; ++q) n += skips(c / KB, q, c % KB) ? 0 : 2; return n; }
    static constexpr int nmfma(int c) { int n = 0; for (int a = 0; a < nas[c / KB]; ++a) n += skips(c / KB, acc_tile_of(c / KB, a), c % KB) ? 0 : C::NPR; return n; }
    static constexpr int mfmas() { int n = 0; for (int c = 0; c < NB; ++c) n += nmfma(c); return n; }
    static constexpr int slot_base(int c) { int n = 0; for (int b = 0; b < c; ++b) n += nreads(b); return n % NS; }
    static constexpr int issued(int c) {
        int p = 0;
        for (int cc = 0; cc <= c; ++cc) {
            int held = 0;
            for (int b = cc; b < p; ++b) held += nreads(b);
            while (p < NB && p <= cc + LA && held + nreads(p) <= NS) held += nreads(p), ++p;
        }
        return p;
    }

    // bias: this layer's biases in LDS (const float*), or a callable bias(j) -> float4 of this lane's four channels of n-tile j
    // ncol[j]: first channel of resident n-tile j.  HDS: LDS reads the hook issues in a block where it is active.
    template <int HDS = 0, class WT, class Bias, class Epi, class Hook = TNoHook>
    static __device__ __forceinline__ void run(const half_t* __restrict__ in_hi, const half_t* __restrict__ in_lo, const WT& W,
                                               const Bias& bias, const int (&ncol)[WT::NTR], const IA& ia, Epi epi, Hook hook = Hook{}) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lk = (tid & 63) >> 4;
        f32x4 acc[2][AMAX];
        half8 x[NS];
        // this lane's biases, once per call (read per group, the LDS read and its wait stood in front of every group's first MFMA)
        float4 bz[WT::NTR];
#pragma unroll
        for (int j = 0; j < WT::NTR; ++j) {
            if constexpr (std::is_pointer_v<Bias>) bz[j] = *reinterpret_cast<const float4*>(bias + ncol[j] + 4 * lk);
            else bz[j] = bias(j);
        }

        auto reads = [&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value, g = b / KB, kb = b % KB, s0 = slot_base(b);
            using G = Grp<g>;
            tstatic_for<0, G::NT>([&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value, tile = G::tile(i);
                if constexpr (!IA::skip(tile, kb)) {
                    constexpr int sl = (s0 + rslot(b, i)) % NS;
                    const int off = ia.template off<tile, kb>();
#ifdef HM_XP_NO_READS   // timing ablation (make xp / stampxp): only the first blocks' operands are read, the stream multiplies stale registers
                    if constexpr (b >= issued(0)) return;
#endif
                    x[sl] = *reinterpret_cast<const half8*>(in_hi + off);
                    x[(sl + 1) % NS] = *reinterpret_cast<const half8*>(in_lo + off);
                }
            });
        };
        // an accumulator's epilogue: compute (ReLU, split) and store back to back -- in this layout it is 7 vector instructions and two
        // LDS stores with an immediate address, so that one k-block hides it; nothing is staged across blocks (12 registers fewer)
        auto stage = [&](auto g_, auto a_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value, a = decltype(a_)::value;
            using G = Grp<g>;
            constexpr int ti = G::acc_tile(a), tile = G::tile(ti), j = G::acc_j(a);
#if defined(HM_XP_NO_EPI)      // timing ablation: the accumulator is kept alive, nothing is computed from it or stored
            const f32x4 av = acc[g & 1][a];
            const float a0_ = av[0], a1_ = av[1], a2_ = av[2], a3_ = av[3];
            asm volatile("" ::"v"(a0_), "v"(a1_), "v"(a2_), "v"(a3_));
#elif defined(HM_XP_NO_STORE)  // timing ablation: ReLU + split, no store
            typename Epi::St est;
            epi.s0(acc[g & 1][a], est);
            uint32_t wv[sizeof(est) / 4];
            __builtin_memcpy(wv, &est, sizeof(est));
#pragma unroll
            for (int q = 0; q < (int)(sizeof(est) / 4); ++q) asm volatile("" ::"v"(wv[q]));
#else
            typename Epi::St est;
            epi.s0(acc[g & 1][a], est);
            epi.template s1<tile>(ncol[j], est);
#endif
        };
        tstatic_for<0, issued(0)>(reads);

        tstatic_for<0, NB>([&](auto c_) __attribute__((always_inline)) {
            constexpr int c = decltype(c_)::value, g = c / KB, kb = c % KB, s0 = slot_base(c);
            using G = Grp<g>;
            if constexpr (kb == 0) {
                tstatic_for<0, G::NA>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value, j = G::acc_j(a);
                    acc[g & 1][a] = f32x4{bz[j].x, bz[j].y, bz[j].z, bz[j].w};
                });
            }
            constexpr int P0 = issued(c > 0 ? c - 1 : 0), P1 = issued(c);
            if constexpr (c > 0) tstatic_for<P0, P1>(reads);
            tstatic_for<0, C::NPR>([&](auto pr_) __attribute__((always_inline)) {
                constexpr int pr = decltype(pr_)::value;
                tstatic_for<0, G::NA>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value, i = G::acc_tile(a), j = G::acc_j(a), tile = G::tile(i);
                    if constexpr (!IA::skip(tile, kb)) {
                        constexpr int sl = (s0 + rslot(c, i) + (pr == 1 ? 1 : 0)) % NS;
                        acc[g & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.w[kb][j][pr == 2 ? 1 : 0], x[sl], acc[g & 1][a], 0, 0, 0);
                    }
                });
            });
            // the previous group's accumulators leave between this group's MFMAs: 2 stages each, dealt over its k-blocks
            constexpr int NAP = g > 0 ? nas[g > 0 ? g - 1 : 0] : 0;
            constexpr int E0 = kb * NAP / KB, E1 = (kb + 1) * NAP / KB;
            if constexpr (g > 0)
                tstatic_for<E0, E1>([&](auto e_) __attribute__((always_inline)) {
                    stage(std::integral_constant<int, (g > 0 ? g - 1 : 0)>{}, e_);
                });
            hook(c_);
            {
                constexpr int NRD = [&]() constexpr { int n = 0; for (int b = (c > 0 ? P0 : P1); b < P1; ++b) n += nreads(b); return n; }();
                constexpr int ND = NRD + HDS;
                constexpr int NM = nmfma(c);
                constexpr int NV = g > 0 ? (E1 - E0) * (Epi::NV0 + Epi::NV1) : 0;
                constexpr int NWR = g > 0 ? (E1 - E0) * Epi::NW : 0;
                tstatic_for<0, NM>([&](auto q_) __attribute__((always_inline)) {
                    constexpr int q = decltype(q_)::value;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (q < ND) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    constexpr int nv = (q + 1) * NV / NM - q * NV / NM;
                    if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
                    if constexpr (q >= NM - NWR) __builtin_amdgcn_sched_group_barrier(Epi::WMASK, 1, 0);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        tstatic_for<0, nas[NG - 1]>([&](auto a_) __attribute__((always_inline)) { stage(std::integral_constant<int, NG - 1>{}, a_); });
    }
};

}  // namespace hm
