// hm_convs.h -- the split-half implicit-GEMM convolution in its STREAMING form (dense trunk, hm_trunk.hip):
//   * one wave per SIMD (4-wave workgroups, up to 512 registers per lane), each wave owns 32 output channels (two MFMA
//     n-tiles) of the layer and ALL of its positions;
//   * the wave's weights of the whole layer -- 12 k-blocks x 2 n-tiles x (hi, lo) = 192 registers -- are RESIDENT in
//     registers (WRegs): they are fetched once per layer, progressively, while the previous layer's last positions are
//     still in the matrix pipe (a register is refilled as soon as its last MFMA has issued);
//   * the positions stream through in groups of G 16-row tiles: per k-block 2G ds_read_b128 feed 6G MFMAs, and the
//     epilogue (ReLU, split, LDS / global stores) of group g rides between the MFMAs of group g + 1.
// Against ConvH's form (8 waves, each all positions x 16 channels, weights streamed per k-block, epilogue after the
// k-loop) this halves the LDS operand reads, takes every weight fetch and all but the last group's epilogue off the
// critical path, and leaves no second wave per SIMD to be starved.  Same products in the same order per accumulator
// (bias, then per k-block: w_hi x_hi, w_hi x_lo, w_lo x_hi), so results are bit-identical to ConvH's.
#pragma once
#include "hm_convh.h"

namespace hm {

struct WRegs {
    half8 w[12][4][2];  // [k-block][n-tile][plane hi / lo]; a layer uses its first NTW n-tiles
    float4 bz[4];
};

// Row maps: where output row m of a layer finds its first tap in the input planes (halves).
struct DenseRows {  // the dense trunk: row m of the tile, taps DIL rows apart
    static constexpr bool DENSE = true;
    static constexpr int M = 1 << 30;
};
// S sites stacked along M (tail kernel): m -> (site = m / LOUT, p = m % LOUT), a stride-MSTR conv inside the site's rows;
// rows past M (the ragged last tile) re-read the last valid row and are not written
template <int LOUT, int ISS, int MSTR, int ROW0, int M_>
struct SiteRows {
    static constexpr bool DENSE = false;
    static constexpr int M = M_;
    template <class C>
    static __device__ __forceinline__ int off(int m) {
        const int mc = m < M ? m : M - 1, site = mc / LOUT, p = mc - site * LOUT;
        return site * ISS + (MSTR * p + ROW0) * C::IRS;
    }
};

template <int CIN_, int KT_, int IRS_, int DIL_, bool WLO_, bool XLO_, int KSTACK_, int XD_, int NTW_ = 2>
struct SCfg {
    static constexpr int CIN = CIN_, KT = KT_, IRS = IRS_, DIL = DIL_, KSTACK = KSTACK_, XD = XD_, NTW = NTW_;
    static constexpr bool WLO = WLO_ && KSTACK_ == 0, XLO = XLO_;
    static constexpr int KB = KT * CIN / 32;
    static constexpr int WSTR = KSTACK ? 64 : 128;  // half8 per (n-tile, k-block)
    static constexpr int NTERM = 1 + (XLO ? 1 : 0) + (WLO ? 1 : 0);
    static_assert(KB <= 12 && (KT * CIN) % 32 == 0 && (CIN % 32 == 0 || KSTACK > 0), "bad streaming conv geometry");
    // halves from a tile row's first element to this lane's 8 K elements of k-block kb
    static __device__ __forceinline__ int lane_off(int kb, int lk) {
        if constexpr (KSTACK > 0) {  // tap slot 4 kb + lk; slots >= K1 walk the same rows again (lo halves of the weights)
            int t = 4 * kb + lk;
            t = t >= 2 * KSTACK ? 0 : t >= KSTACK ? t - KSTACK : t;
            return t * IRS;
        } else {
            const int kk = kb * 32, tap = kk / CIN;
            return tap * DIL * IRS + (kk - tap * CIN) + 8 * lk;
        }
    }
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// issue the loads of k-blocks [K0, K1) of n-tiles nt0 .. nt0 + NTW - 1 into W
template <class C, int K0, int K1>
__device__ __forceinline__ void sconv_load_w(const half_t* __restrict__ wfrag, int nt0, int lane, WRegs& W) {
    const half8* wp = reinterpret_cast<const half8*>(wfrag) + (size_t)nt0 * C::KB * C::WSTR + lane;
    static_for<K0, K1>([&](auto kb_) __attribute__((always_inline)) {
        constexpr int kb = decltype(kb_)::value;
#pragma unroll
        for (int j = 0; j < C::NTW; ++j) {
            W.w[kb][j][0] = wp[(size_t)(j * C::KB + kb) * C::WSTR];
            if (C::WLO) W.w[kb][j][1] = wp[(size_t)(j * C::KB + kb) * C::WSTR + 64];
        }
    });
}
template <class C>
__device__ __forceinline__ void sconv_load_bias(const float* __restrict__ bias, int nt0, int lane, WRegs& W) {
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) W.bz[j] = *reinterpret_cast<const float4*>(bias + (nt0 + j) * 16 + 4 * (lane >> 4));
}

// C: this layer; CN: the layer whose weights are loaded behind this one's last group (void: none); GS...: tiles per group.
// Epi: epi(m, col, acc) -- ReLU + split + stores; nothing in a layer is conditional: a store behind a branch makes the
// compiler's wait counts for the resident-weight loads conservative (it cannot tell how many stores are in flight behind a
// load), which puts whole HBM round trips in front of the next layer.
// Copy: the rows of THIS layer's input planes that must also reach HBM (the map rows an edge chain reads) leave from here,
// as whole 512-byte rows: CS fixed slots per wave, two rows per slot (one ds_read_b128 + one global_store_dwordx4 per lane,
// the store one k-block behind its read), row numbers from a list in LDS that is padded with a row that is always valid.
// T0: the first 16-row tile of this wave's share of the positions.  RM: row map (DenseRows / SiteRows, or a map with state --
// a row list in LDS, hm_trunk3.hip -- passed as `rm`).
template <class C, class CN, class RM, int T0, int... GS>
struct SConvR {
    static constexpr int NG = sizeof...(GS);
    static constexpr int gs[NG] = {GS...};
    static constexpr int gmax() { int m = 0; for (int g = 0; g < NG; ++g) m = gs[g] > m ? gs[g] : m; return m; }
    static constexpr int gs_at(int g) { return g < NG ? gs[g < NG ? g : 0] : 0; }
    static constexpr int NTW = C::NTW;
    static constexpr int tile0(int g) { int t = T0; for (int i = 0; i < g; ++i) t += gs[i]; return t; }
    static constexpr int GMAX = gmax(), KB = C::KB, NB = NG * KB, XS = C::XD + 1;

    struct NoCopy {
        static constexpr int CS = 0, NWV = 4;
        const uint8_t* rows = nullptr;
        half_t* g = nullptr;
    };
    // J0: the first of this layer's NTW n-tiles among the wave's resident ones (W.w[.][J0 ..]); nt0 is resident tile 0's
    // n-tile.  A wave that holds two n-tiles may run one of them alone over some positions (NTW = 1, J0 = 0 | 1): conv4's
    // 6 x 7 tile pairs are dealt to four waves that way (hm_trunk.hip).
    template <int J0 = 0, class Epi, class Copy = NoCopy>
    static __device__ __forceinline__ void run(const half_t* __restrict__ in_hi, const half_t* __restrict__ in_lo, WRegs& W,
                                               Epi epi, const half_t* __restrict__ wnext, const float* __restrict__ bnext,
                                               int nt0, int nt0n, Copy cp = Copy{}, RM rm = RM{}) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const int a0 = li * C::IRS;
        constexpr int NTILES = tile0(NG) - T0;
        int aoff[RM::DENSE ? 1 : NTILES];  // per tile: this lane's row in the input planes (site-stacked layers)
        if constexpr (!RM::DENSE) {
#pragma unroll
            for (int t = 0; t < NTILES; ++t) aoff[t] = rm.template off<C>((T0 + t) * 16 + li);
        }
        f32x4 acc[2][GMAX][NTW];
        half8 x[XS][GMAX][2];
        constexpr int CS = Copy::CS;
        static_assert(CS + 2 <= NB, "a copy slot per k-block");
        // copy pipeline over k-blocks: block c fetches the row number of slot c, reads the row of slot c - 1 out of the
        // planes and stores the row of slot c - 2
        int crow[3];
        half8 cdat[2];
        const uint8_t* cpr = CS > 0 ? cp.rows + 2 * __builtin_amdgcn_readfirstlane(tid >> 6) + (lane >> 5) : nullptr;

        auto reads = [&](auto c_) __attribute__((always_inline)) {
            constexpr int c = decltype(c_)::value, g = c / KB, kb = c % KB;
            const int off = C::lane_off(kb, lk);
#pragma unroll
            for (int i = 0; i < gs[g]; ++i) {
                int o;
                if constexpr (RM::DENSE) o = a0 + off + (tile0(g) + i) * 16 * C::IRS;
                else o = aoff[RM::DENSE ? 0 : tile0(g) + i - T0] + off;
                x[c % XS][i][0] = *reinterpret_cast<const half8*>(in_hi + o);
                if (C::XLO) x[c % XS][i][1] = *reinterpret_cast<const half8*>(in_lo + o);
            }
        };
        static_for<0, (C::XD < NB ? C::XD : NB)>(reads);

        static_for<0, NB>([&](auto c_) __attribute__((always_inline)) {
            constexpr int c = decltype(c_)::value, g = c / KB, kb = c % KB, G = gs[g];
            if constexpr (kb == 0) {
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) acc[g & 1][i][j] = f32x4{W.bz[J0 + j].x, W.bz[J0 + j].y, W.bz[J0 + j].z, W.bz[J0 + j].w};
            }
            // the next layer's bias goes first of its loads (vector memory returns in order: the first MFMA of the next layer
            // needs the bias, and would otherwise wait for every weight fragment issued before it)
            if constexpr (!std::is_void_v<CN> && g == NG - 1 && kb == 0)
                sconv_load_bias<std::conditional_t<std::is_void_v<CN>, C, CN>>(bnext, nt0n, lane, W);
            if constexpr (c + C::XD < NB) reads(std::integral_constant<int, c + C::XD>{});
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {
                if ((pr == 1 && !C::XLO) || (pr == 2 && !C::WLO)) continue;
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[g & 1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.w[kb][J0 + j][pr == 2 ? 1 : 0], x[c % XS][i][pr == 1 ? 1 : 0],
                                                                                 acc[g & 1][i][j], 0, 0, 0);
            }
            constexpr int NA = g > 0 ? gs[g > 0 ? g - 1 : 0] * NTW : 0;  // accumulators of the previous group
            constexpr int A0 = kb * NA / KB, A1 = (kb + 1) * NA / KB;
            if constexpr (g > 0) {
#pragma unroll
                for (int a = A0; a < A1; ++a) {
                    const int m = (tile0(g - 1) + a / NTW) * 16 + li;
                    if ((tile0(g - 1) + a / NTW + 1) * 16 <= RM::M || m < RM::M) epi(m, (nt0 + J0 + a % NTW) * 16 + 4 * lk, acc[(g - 1) & 1][a / NTW][a % NTW]);
                }
            }
            if constexpr (c >= 2 && c < CS + 2)
                *reinterpret_cast<half8*>(cp.g + (size_t)crow[(c - 2) % 3] * 256 + (lane & 31) * 8) = cdat[(c - 2) & 1];
            if constexpr (c >= 1 && c < CS + 1)
                cdat[(c - 1) & 1] = *reinterpret_cast<const half8*>(((lane & 16) ? in_lo : in_hi) + crow[(c - 1) % 3] * C::IRS + (lane & 15) * 8);
            if constexpr (c < CS) crow[c % 3] = cpr[c * 2 * Copy::NWV];
            if constexpr (!std::is_void_v<CN>) {
                // the next layer's weights: a register is refilled as soon as its last MFMA of this layer has issued (last
                // group, k-block kb); registers this layer does not use at all (other k-blocks, the lo plane of a layer with
                // exact operands) are filled from the first group on -- spread out, because 4 waves x 48 KB through the
                // CU's 64 B/clk vector-memory path take 3 000 cycles whatever the latency
                using CNN = std::conditional_t<std::is_void_v<CN>, C, CN>;
                constexpr int NPN = CNN::WLO ? 2 : 1, EN = CNN::KB * CNN::NTW * NPN;
                auto is_free = [](int e) constexpr {
                    const int kb2 = e / (CNN::NTW * NPN), j2 = e / NPN % CNN::NTW, p2 = e % NPN;
                    return kb2 >= KB || j2 < J0 || j2 >= J0 + NTW || p2 >= (C::WLO ? 2 : 1);
                };
                constexpr int NFREE = [&]() constexpr { int n = 0; for (int e = 0; e < EN; ++e) n += is_free(e); return n; }();
                constexpr int NEARLY = NB - KB;  // blocks before the last group
                const half8* wp = reinterpret_cast<const half8*>(wnext) + (size_t)nt0n * CNN::KB * CNN::WSTR + lane;
                static_for<0, EN>([&](auto e_) __attribute__((always_inline)) {
                    constexpr int e = decltype(e_)::value;
                    constexpr int kb2 = e / (CNN::NTW * NPN), j2 = e / NPN % CNN::NTW, p2 = e % NPN;
                    constexpr bool fr = is_free(e);
                    constexpr int rank = [&]() constexpr { int n = 0; for (int q = 0; q < e; ++q) n += is_free(q); return n; }();
                    constexpr bool now = fr ? (NEARLY > 0 ? (c < NEARLY && rank * NEARLY / (NFREE > 0 ? NFREE : 1) == c) : (c == NB - KB))
                                            : (g == NG - 1 && kb2 == kb);
                    if constexpr (now) W.w[kb2][j2][p2] = wp[(size_t)(j2 * CNN::KB + kb2) * CNN::WSTR + 64 * p2];
                });
            }
            {   // the block's LDS reads ride between its first MFMAs, the previous group's epilogue between the others
                constexpr int ND = gs_at((c + C::XD) / KB) * (C::XLO ? 2 : 1);
                constexpr int NM = C::NTERM * G * NTW;
                constexpr int NR = NM > ND ? NM - ND : 1;
                constexpr int NV = g > 0 ? ((A1 - A0) * 24 + NR - 1) / NR : 0;
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
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (c <= CS + 1) __builtin_amdgcn_sched_group_barrier(0x140, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        {
            constexpr int g = NG - 1;
#pragma unroll
            for (int a = 0; a < gs[g] * NTW; ++a) {
                const int m = (tile0(g) + a / NTW) * 16 + li;
                if ((tile0(g) + a / NTW + 1) * 16 <= RM::M || m < RM::M) epi(m, (nt0 + J0 + a % NTW) * 16 + 4 * lk, acc[g & 1][a / NTW][a % NTW]);
            }
        }
    }
};

template <class C, class CN, int T0, int... GS>
using SConv = SConvR<C, CN, DenseRows, T0, GS...>;

}  // namespace hm
