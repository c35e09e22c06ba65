// hm_convt.h -- the split-half implicit-GEMM convolution for the RESIDENT tail kernel (hm_tail_r.hip): conv5..conv8 of the
// model over S sites stacked along M, on 4 waves = one per SIMD with up to 512 registers each, every wave holding ITS share
// of a layer's weights in registers for the whole launch.
//
// Why: tail_kernel_h (hm_front_h.hip) streams 375 KB of weights from L2 for every pass of 8 sites -- 47 KB per site
// against 9.6 KB of activations -- through the CU's one 64 B/clk vector-memory path, layer after layer, each layer's first
// MFMA behind a fresh L2 round trip.  Its matrix pipe is busy a third of the time (profiles/r02_pmc_derived.txt).  The tail's
// weights (342 KB for conv5..conv8) are smaller than the CU's register file (512 KB); kept there, a pass needs no weight
// traffic at all and M can stay small.
//
// Work split.  A layer with NT n-tiles (16 output channels each) and MT m-tiles (16 stacked positions each) is NT x MT
// (n, m) tile pairs.  6 n-tiles do not divide by 4 waves, so a wave holds TWO n-tiles (a, b) and runs
//     pair   (a, b) on one range of m-tiles  -- both n-tiles share the activation reads --
//     single (a)    on the other range,
// with b shared between two waves that take complementary m ranges: conv5 (6 x 7 = 42 pairs) = 11 + 10 + 11 + 10, conv6
// (6 x 4 = 24) = 6 each.  Every SIMD's matrix pipe gets the same number of MFMAs; conv7 / conv8 (4 n-tiles) are one
// n-tile per wave.
//
// Per accumulator the products are issued in ConvH's order (bias, then per k-block w_hi x_hi, w_hi x_lo, w_lo x_hi), so the
// results are bit-identical to tail_kernel_h's.
#pragma once
#include <tuple>

#include "hm_convh.h"

namespace hm {

// layer geometry: CIN channels per tap, KT taps (stride-2 conv over a site's rows: taps one row apart), IRS halves per row
template <int CIN_, int KT_, int IRS_>
struct TCfg {
    static constexpr int CIN = CIN_, KT = KT_, IRS = IRS_;
    static constexpr int KB = KT * CIN / 32;
    static_assert((KT * CIN) % 32 == 0 && CIN % 32 == 0, "a k-block never straddles taps");
    // halves from a row's first element to this lane's 8 K elements of k-block kb
    static __device__ __forceinline__ int lane_off(int kb, int lk) {
        const int kk = kb * 32, tap = kk / CIN;
        return tap * IRS + (kk - tap * CIN) + 8 * lk;
    }
};

// S sites stacked along M: m -> (site = m / LOUT, p = m % LOUT); output p reads rows 2p, 2p+1, 2p+2 of the site's rows
// (row 0 and the row behind the last are zero padding).  Rows past M re-read the last valid row (their results are dropped).
template <int LOUT_, int ISS_, int M_>
struct TRows {
    static constexpr int LOUT = LOUT_, ISS = ISS_, M = M_;
    template <class C>
    static __device__ __forceinline__ int off(int m) {
        const int mc = m < M ? m : M - 1, site = mc / LOUT, p = mc - site * LOUT;
        return site * ISS + 2 * p * C::IRS;
    }
};

// a wave's resident weights of one layer: NTR n-tiles x KB k-blocks x (hi, lo)
template <int KB_, int NTR_>
struct TW {
    static constexpr int KB = KB_, NTR = NTR_;
    half8 w[KB_][NTR_][2];
};

// fragments [n-tile][k-block][plane][lane] half8 of n-tiles nt[0..NTR) -> registers
template <int KB, int NTR>
__device__ __forceinline__ void tw_load(const half_t* __restrict__ wfrag, const int (&nt)[NTR], int lane, TW<KB, NTR>& W) {
    const half8* wp = reinterpret_cast<const half8*>(wfrag) + lane;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int j = 0; j < NTR; ++j) {
            W.w[kb][j][0] = wp[(size_t)(nt[j] * KB + kb) * 128];
            W.w[kb][j][1] = wp[(size_t)(nt[j] * KB + kb) * 128 + 64];
        }
}

// One group of a wave's schedule: GP m-tiles from TP on which BOTH resident n-tiles run (a pair: the two share the activation
// reads) and GS m-tiles from TS on which only resident n-tile 0 runs.  A group's 2 GP + GS accumulators are visited round-robin,
// product by product.  (How many accumulators a group has does not set its rate: a chain of v_mfma_f32_16x16x32_f16 onto ONE
// accumulator issues every 16 cycles -- tools/micro/mfma_chain.hip -- and tools/micro/tconv_rate.hip measures 23 .. 28 ticks per
// MFMA for every shape: the stream is bound by the ~2.5 other instructions that ride with each MFMA.)
template <int TP_, int GP_, int TS_, int GS_>
struct TG {
    static constexpr int TP = TP_, GP = GP_, TS = TS_, GS = GS_;
    static constexpr int NT = GP_ + GS_, NA = 2 * GP_ + GS_;
    static constexpr int tile(int i) { return i < GP_ ? TP_ + i : TS_ + (i - GP_); }   // i-th tile of the group
    static constexpr int acc_tile(int a) { return a < 2 * GP_ ? a / 2 : GP_ + (a - 2 * GP_); }   // tile index (in the group) of accumulator a
    static constexpr int acc_j(int a) { return a < 2 * GP_ ? a % 2 : 0; }                        // resident n-tile of accumulator a
};

template <int I, int N, class F>
__device__ __forceinline__ void tstatic_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        tstatic_for<I + 1, N>(f);
    }
}

struct TNoHook {
    template <int C>
    __device__ __forceinline__ void operator()(std::integral_constant<int, C>) const {}
};

// The groups run back to back as ONE stream of (group, k-block) blocks: per block 3 MFMAs per accumulator, a share of the
// previous group's epilogue (ReLU, split, LDS stores) and `hook(block)` -- the caller's slot for work that should ride
// between the MFMAs (the LDS-DMA gather of the next group of sites).
// Activation operands come through a RING of NS half8 registers: a block's reads are issued as early as the ring has room
// for them, at most LA blocks ahead.
// Epi: epi(m, col, acc).  ncol[j]: first channel of the wave's resident n-tile j.
template <class C, class RM, int NS, int LA, class... GR>
struct TConv {
    static constexpr int NG = sizeof...(GR);
    static constexpr int nts[NG] = {GR::NT...}, nas[NG] = {GR::NA...};
    static constexpr int KB = C::KB, NB = NG * KB;
    static constexpr int amax() { int m = 0; for (int g = 0; g < NG; ++g) m = nas[g] > m ? nas[g] : m; return m; }
    static constexpr int tmax() { int m = 0; for (int g = 0; g < NG; ++g) m = nts[g] > m ? nts[g] : m; return m; }
    static constexpr int tile_base(int g) { int t = 0; for (int i = 0; i < g; ++i) t += nts[i]; return t; }
    static constexpr int AMAX = amax(), TMAX = tmax(), NTILES = tile_base(NG);
    static_assert(NS >= 2 * TMAX, "the ring holds at least one block's operands");
    static constexpr int mfmas() { int n = 0; for (int g = 0; g < NG; ++g) n += nas[g] * 3 * KB; return n; }
    // ring bookkeeping, all at compile time
    static constexpr int nreads(int c) { return 2 * nts[c / KB]; }  // (hi, lo) per tile of block c
    static constexpr int slot_base(int c) { int n = 0; for (int b = 0; b < c; ++b) n += nreads(b); return n % NS; }
    // blocks [0, issued(c)) have had their reads issued once block c's prefetch step has run: the step of block c sees the
    // slots of every block before c free again and issues as many further blocks as then fit
    static constexpr int issued(int c) {
        int p = 0;
        for (int cc = 0; cc <= c; ++cc) {
            int held = 0;
            for (int b = cc; b < p; ++b) held += nreads(b);
            while (p < NB && p <= cc + LA && held + nreads(p) <= NS) held += nreads(p), ++p;
        }
        return p;
    }
    template <int g>
    using Grp = std::tuple_element_t<g, std::tuple<GR...>>;

    // HDS: LDS reads the hook issues in a block where it is active (for the instruction-interleave plan)
    // bias: this layer's biases in LDS (const float*), or a callable bias(j) -> float4 of this lane's four channels of n-tile j
    template <int HDS = 0, class WT, class Bias, class Epi, class Hook = TNoHook>
    static __device__ __forceinline__ void run(const half_t* __restrict__ in_hi, const half_t* __restrict__ in_lo, const WT& W,
                                               const Bias& bias, const int (&ncol)[WT::NTR], Epi epi, Hook hook = Hook{}) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // keep this layer's address arithmetic inside the pass loop
        const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
        int aoff[NTILES];  // per tile of the schedule: this lane's first row in the input planes
        tstatic_for<0, NG>([&](auto g_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            using G = Grp<g>;
            tstatic_for<0, G::NT>([&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value;
                aoff[tile_base(g) + i] = RM::template off<C>(G::tile(i) * 16 + li);
            });
        });
        f32x4 acc[2][AMAX];
        half8 x[NS];

        auto reads = [&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value, g = b / KB, kb = b % KB, s0 = slot_base(b);
            const int off = C::lane_off(kb, lk);
#ifdef HM_ABL_NOREAD   // tools/micro ablation: operands stay what the first block read
            if constexpr (b >= NG * 0 + issued(0)) return;
#endif
#pragma unroll
            for (int i = 0; i < nts[g]; ++i) {
                x[(s0 + 2 * i) % NS] = *reinterpret_cast<const half8*>(in_hi + aoff[tile_base(g) + i] + off);
                x[(s0 + 2 * i + 1) % NS] = *reinterpret_cast<const half8*>(in_lo + aoff[tile_base(g) + i] + off);
            }
        };
        auto epilogue = [&](auto g_, auto a_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value, a = decltype(a_)::value;
            using G = Grp<g>;
            constexpr int tile = G::tile(G::acc_tile(a)), j = G::acc_j(a);
            int m = tile * 16 + li, col = ncol[j] + 4 * lk;
            if constexpr ((tile + 1) * 16 > RM::M) {
                // ragged last tile: rows past M write their (meaningless) values into the 16-byte pad behind the channels
                // of the last valid row instead of being branched around -- a branch here would cut the block in two
                col = m < RM::M ? col : Epi::PADCOL;
                m = m < RM::M ? m : RM::M - 1;
            }
#ifdef HM_ABL_NOEPI    // tools/micro ablation: the accumulator is kept alive, nothing is computed or stored from it
            { const f32x4 keep = acc[g & 1][a]; asm volatile("" ::"v"(keep)); }
#else
            epi(m, col, acc[g & 1][a]);
#endif
        };
        tstatic_for<0, issued(0)>(reads);

        tstatic_for<0, NB>([&](auto c_) __attribute__((always_inline)) {
            constexpr int c = decltype(c_)::value, g = c / KB, kb = c % KB, s0 = slot_base(c);
            using G = Grp<g>;
            if constexpr (kb == 0) {
                float4 bz[WT::NTR];
                if constexpr (std::is_pointer_v<Bias>) {
                    bz[0] = *reinterpret_cast<const float4*>(bias + ncol[0] + 4 * lk);
                    if constexpr (G::GP > 0) bz[WT::NTR - 1] = *reinterpret_cast<const float4*>(bias + ncol[WT::NTR - 1] + 4 * lk);
                } else {
                    bz[0] = bias(0);
                    if constexpr (G::GP > 0) bz[WT::NTR - 1] = bias(WT::NTR - 1);
                }
                tstatic_for<0, G::NA>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value, j = G::acc_j(a);
                    acc[g & 1][a] = f32x4{bz[j].x, bz[j].y, bz[j].z, bz[j].w};
                });
            }
            constexpr int P0 = issued(c > 0 ? c - 1 : 0), P1 = issued(c);
            if constexpr (c > 0) tstatic_for<P0, P1>(reads);
            // product-major over the group's accumulators: an accumulator is revisited NA MFMAs later
            tstatic_for<0, 3>([&](auto pr_) __attribute__((always_inline)) {
                constexpr int pr = decltype(pr_)::value;
                tstatic_for<0, G::NA>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value, i = G::acc_tile(a), j = G::acc_j(a);
                    acc[g & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.w[kb][j][pr == 2 ? 1 : 0], x[(s0 + 2 * i + (pr == 1 ? 1 : 0)) % NS],
                                                                           acc[g & 1][a], 0, 0, 0);
                });
            });
            // the previous group's accumulators leave between this group's MFMAs, spread over its k-blocks
            constexpr int NAP = g > 0 ? nas[g > 0 ? g - 1 : 0] : 0;
            constexpr int A0 = kb * NAP / KB, A1 = (kb + 1) * NAP / KB;
            if constexpr (g > 0)
                tstatic_for<A0, A1>([&](auto a_) __attribute__((always_inline)) { epilogue(std::integral_constant<int, (g > 0 ? g - 1 : 0)>{}, a_); });
            hook(c_);
            {   // interleave plan: this step's LDS reads behind the first MFMAs, epilogue VALU + its LDS stores behind the rest
                constexpr int NRD = [&]() constexpr { int n = 0; for (int b = (c > 0 ? P0 : P1); b < P1; ++b) n += nreads(b); return n; }();
                constexpr int ND = NRD + (kb == 0 && std::is_pointer_v<Bias> ? (G::GP > 0 ? 2 : 1) : 0) + HDS;
                constexpr int NM = 3 * G::NA;
                constexpr int NR = NM > ND ? NM - ND : 1;
                constexpr int NV = ((A1 - A0) * 26 + NR - 1) / NR;
#pragma unroll
                for (int q = 0; q < (ND < NM ? ND : NM); ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int q = 0; q < NM - ND; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                    if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        tstatic_for<0, nas[NG - 1]>([&](auto a_) __attribute__((always_inline)) { epilogue(std::integral_constant<int, NG - 1>{}, a_); });
    }
};

}  // namespace hm
