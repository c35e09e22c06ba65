// hm_convh.h -- the split-half ("f16x3") implicit-GEMM convolution on v_mfma_f32_16x16x32_f16 shared by the per-site
// front / tail kernels (hm_front_h.hip) and the dense trunk / edge kernels (hm_trunk.hip).  See hm_front_h.hip for the
// arithmetic (x = hi + lo halves, three fp16 products per MAC, fp32 accumulation) and the scheduling notes.
#pragma once
#include <type_traits>
#include <utility>

#include "hm_kernels.h"

namespace hm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half_t;

typedef __fp16 pk2 __attribute__((ext_vector_type(2)));

// max(x, 0) as ONE instruction (v_med3_f32 x, 0, 3e38): fmaxf() makes the compiler quiet a possible signalling NaN first
// (an extra v_max_f32 x, x, x per value) -- and with +inf as the upper bound it rewrites the med3 into that same pair, hence
// the finite bound, far above any activation.  A compiler builtin, NOT inline asm: x is an MFMA result, and the wait states an
// MFMA result needs before a VALU read are only inserted for instructions the compiler can see -- an asm v_max_f32 placed
// right behind the MFMAs of its tile read stale accumulators.
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 3.0e38f); }

// lo halves of two values: fp16(x - hi) with v_fma_mix{lo,hi}_f16 -- the f16 half is widened inside the instruction,
// subtracted from the fp32 x exactly and the residual rounded once (2 instructions instead of 2 cvt + 2 sub + 1 cvt_pk)
__device__ __forceinline__ uint32_t split_lo2(uint32_t h01, float x0, float x1) {
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, -%1, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l)
        : "v"(h01), "v"(x0), "v"(x1));
#ifdef HM_XP_LO_BITS   // experiment: the lo halves rounded to 10 - HM_XP_LO_BITS mantissa bits (operand toggling vs the clock the chip holds)
    l = (l + ((1u << (HM_XP_LO_BITS - 1)) * 0x00010001u)) & ((0xffffu << HM_XP_LO_BITS & 0xffffu) * 0x00010001u);
#endif
    return l;
}

// ReLU, then x = hi + lo: hi = fp16(x) by v_cvt_pkrtz_f16_f32 (top 11 bits), lo = fp16(x - hi)
// -> |x - (hi + lo)| <= 2^-21 |x|.  (-40 % epilogue VALU, -0.9 % kernel time in a same-box A/B.)
__device__ __forceinline__ void split4(const f32x4& v, half4& hi, half4& lo) {
    const float x0 = relu1(v[0]), x1 = relu1(v[1]), x2 = relu1(v[2]), x3 = relu1(v[3]);
    union { pk2 h; uint32_t u; } h01, h23;
    h01.h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    h23.h = __builtin_amdgcn_cvt_pkrtz(x2, x3);
    union { uint32_t u[2]; half4 h; } H, L;
    H.u[0] = h01.u; H.u[1] = h23.u;
    L.u[0] = split_lo2(h01.u, x0, x1);
    L.u[1] = split_lo2(h23.u, x2, x3);
    hi = H.h;
    lo = L.h;
}

// K is processed in blocks of 32 (one MFMA).  Lane (li = l&15, lk = l>>4) owns 8 consecutive K elements:
//   CIN >= 32 : channels c0 + 8*lk .. +7 of tap (32*kb)/CIN          (a block never straddles taps)
//   CIN == 8  : all 8 channels of tap 4*kb + lk
// S sites may be stacked along M (tail kernel): row m -> (site = m / LOUT, p = m % LOUT), site stride ISS halves.
// WLO = false drops the w_lo*x_hi pass: the layer then runs with plain fp16 WEIGHTS (BASELINE.json configs[4]).
// XLO = false: the activations are exact fp16 (no lo plane; the w_hi*x_lo pass is dropped).  EDGE: the first and the last
// position tile are written through epi.edge() (conv1 with bn0 folded into the weights, see hm_weights.cpp).
// KSTACK = K1 > 0 (conv1 with an exact operand): the hi and the lo halves of the weights are stacked along K as 2*K1
// "taps" over the SAME window rows -- tap slot t < K1 holds w_hi of tap t, slot K1 + t holds w_lo of tap t -- so one MFMA
// product per block covers both and only ceil(2*K1 / 4) blocks are needed (7 instead of 4 x 2 for K1 = 13).  KT_ is then
// the number of tap slots (a multiple of 4) and the weights have no plane dimension.
template <int NW_, int CIN_, int KT_, int COUT_, int LOUT_, int IRS_, int WM_, int WN_, int BR_ = 2, int S_ = 1,
          int ISS_ = 0, int ROW0_ = 0, bool WLO = true, bool XLO = true, bool EDGE = false, int KSTACK = 0, bool ILV = false, int LASTT = 0,
          int MSTR_ = 2, int DIL_ = 1>
struct ConvH {
    // MSTR: LDS rows between consecutive output positions (2 = the model's stride-2 conv over one site's rows, 1 = dense
    // evaluation at every position); DIL: rows between consecutive taps (1, or the dilation of the dense a-trous form)
    static constexpr int MSTR = MSTR_, DIL = DIL_;
    static constexpr int NW = NW_, CIN = CIN_, KT = KT_, COUT = COUT_, LOUT = LOUT_, IRS = IRS_, WM = WM_, WN = WN_, BR = BR_;
    static constexpr int S = S_, ISS = ISS_, ROW0 = ROW0_;
    static constexpr int M = S * LOUT;
    static constexpr int MT = (M + 15) / 16;
    static constexpr int NT = COUT / 16;
    static constexpr int MTW = (MT + WM - 1) / WM;
    static constexpr int NTW = NT / WN;
    static constexpr int K = KT * CIN;
    static constexpr int KB = K / 32;
    static_assert(K % 32 == 0 && NT % WN == 0 && WM * WN <= NW && (CIN % 32 == 0 || CIN == 8), "bad conv geometry");
    static_assert(IRS % 8 == 0 && BR >= 2 && KB >= BR - 1, "bad layout / pipeline depth");

    static __device__ __forceinline__ int block_off(int kb) {
        if (CIN == 8) return 4 * kb * IRS;
        const int kk = kb * 32;
        const int tap = kk / CIN;
        return tap * DIL * IRS + (kk - tap * CIN);
    }

    struct NoMark {
        __device__ __forceinline__ void operator()(int) const {}
    };

    // The first HB k-blocks of a wave's weights, kept in registers for the whole launch (they are the same for every site):
    // a layer then starts its k-loop straight after the barrier instead of waiting one L2 round trip for them.
    template <int HB>
    struct Head {
        half8 w[HB > 0 ? HB : 1][NTW][2];
        float4 bz[NTW];
    };
    template <int HB>
    static __device__ __forceinline__ void load_head(const half_t* __restrict__ wfrag, const float* __restrict__ bias, Head<HB>& h) {
        constexpr int WSTR = KSTACK ? 64 : 128;
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (WM * WN < NW && wave >= WM * WN) return;
        const int wn = WM == 1 ? wave : wave % WN;
        const half8* wp = reinterpret_cast<const half8*>(wfrag) + (size_t)(wn * NTW) * KB * WSTR + lane;
#pragma unroll
        for (int j = 0; j < NTW; ++j) h.bz[j] = *reinterpret_cast<const float4*>(bias + (wn * NTW + j) * 16 + 4 * (lane >> 4));
#pragma unroll
        for (int r = 0; r < HB; ++r)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                h.w[r][j][0] = wp[(size_t)(j * KB + r) * WSTR];
                if (WLO && !KSTACK) h.w[r][j][1] = wp[(size_t)(j * KB + r) * WSTR + 64];
            }
    }

    // `mark(0)` after the prologue loads are issued, `mark(1)` after the k-loop (diagnostic stamps only)
    template <class Epi, class Mark = NoMark, int HB = 0>
    static __device__ __forceinline__ void run(const half_t* __restrict__ in_hi, const half_t* __restrict__ in_lo,
                                               const half_t* __restrict__ wfrag, Epi epi, Mark mark = Mark{},
                                               const Head<HB>* head = nullptr) {
        static_assert(HB <= BR - 1, "the head cannot be deeper than the prologue's share of the ring");
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // keep this layer's address arithmetic out of the persistent site loop
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (WM * WN < NW && wave >= WM * WN) return;
        const int wm = WM == 1 ? 0 : wave / WN, wn = WM == 1 ? wave : wave % WN;
        const int li = lane & 15, lk = lane >> 4;
        const int lk_off = KSTACK ? 0 : CIN == 8 ? lk * IRS : 8 * lk;
        constexpr int WSTR = KSTACK ? 64 : 128;  // half8 per (n-tile, k-block): one plane when stacked
        // window row of tap slot 4*kb + lk when stacked: slots >= K1 walk the same rows again with the lo halves
        auto stack_off = [&](int kb) __attribute__((always_inline)) {
            int t = 4 * kb + lk;
            t = t >= 2 * KSTACK ? 0 : t >= KSTACK ? t - KSTACK : t;
            return t * IRS;
        };

        int aoff[MTW];
        if constexpr (WM == 1 && S == 1) {
            // one base register: tiles 0..MTW-2 are full, so their rows sit at compile-time strides from tile 0 (the LDS
            // reads then carry the tile as an immediate offset); only the ragged last tile clamps its row
            const int a0 = (MSTR * li + ROW0) * IRS + lk_off;
#pragma unroll
            for (int i = 0; i < MTW - 1; ++i) aoff[i] = a0 + i * (16 * MSTR * IRS);
            int m = (MTW - 1) * 16 + li;
            m = m < M ? m : M - 1;
            aoff[MTW - 1] = (MSTR * m + ROW0) * IRS + lk_off;
        } else {
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                int m = (wm * MTW + i) * 16 + li;
                m = m < M ? m : M - 1;
                const int site = S == 1 ? 0 : m / LOUT, p = S == 1 ? m : m - site * LOUT;
                aoff[i] = site * ISS + (MSTR * p + ROW0) * IRS + lk_off;
            }
        }
        f32x4 acc[MTW][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const float4 bz = HB > 0 ? head->bz[j] : *reinterpret_cast<const float4*>(epi.bias + (wn * NTW + j) * 16 + 4 * lk);
#pragma unroll
            for (int i = 0; i < MTW; ++i) acc[i][j] = f32x4{bz.x, bz.y, bz.z, bz.w};
            if constexpr (EDGE) {
                // the folded constant must not count where the first / last output row reaches the zero padding: it comes
                // off those two rows' accumulators here, from registers the caller filled before the barrier (in the
                // epilogue the two loads sat in a divergent branch with their latency exposed)
                static_assert(WM == 1 && S == 1 && NTW == 1, "EDGE assumes one channel tile per wave over all positions");
                const float f0 = li == 0 ? 1.f : 0.f, f1 = (MTW - 1) * 16 + li == M - 1 ? 1.f : 0.f;
                acc[0][j][0] -= f0 * epi.c0.x; acc[0][j][1] -= f0 * epi.c0.y; acc[0][j][2] -= f0 * epi.c0.z; acc[0][j][3] -= f0 * epi.c0.w;
                acc[MTW - 1][j][0] -= f1 * epi.c1.x; acc[MTW - 1][j][1] -= f1 * epi.c1.y;
                acc[MTW - 1][j][2] -= f1 * epi.c1.z; acc[MTW - 1][j][3] -= f1 * epi.c1.w;
            }
        }

        // weights: [n-tile][k-block][plane hi/lo][lane] half8
        const half8* wp = reinterpret_cast<const half8*>(wfrag) + (size_t)(wn * NTW) * KB * WSTR + lane;
        half8 wq[BR][NTW][2];
        half8 x[2][MTW][2];
#pragma unroll
        for (int r = 0; r < BR - 1; ++r)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (r < HB) {
                    wq[r][j][0] = head->w[r < HB ? r : 0][j][0];
                    if (WLO && !KSTACK) wq[r][j][1] = head->w[r < HB ? r : 0][j][1];
                    continue;
                }
                wq[r][j][0] = wp[(size_t)(j * KB + r) * WSTR];
                if (WLO && !KSTACK) wq[r][j][1] = wp[(size_t)(j * KB + r) * WSTR + 64];
            }
        {
            const int bo0 = KSTACK ? stack_off(0) : 0;
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                x[0][i][0] = *reinterpret_cast<const half8*>(in_hi + bo0 + aoff[i]);
                if (XLO) x[0][i][1] = *reinterpret_cast<const half8*>(in_lo + bo0 + aoff[i]);
            }
        }

        auto block = [&](auto rb_tag, auto ra_tag, const int kb) __attribute__((always_inline)) {
            constexpr int RB = decltype(rb_tag)::value;
            constexpr int RA = decltype(ra_tag)::value;
            {
                const int kw = kb + BR - 1 < KB ? kb + BR - 1 : KB - 1;
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    wq[(RB + BR - 1) % BR][j][0] = wp[(size_t)(j * KB + kw) * WSTR];
                    if (WLO && !KSTACK) wq[(RB + BR - 1) % BR][j][1] = wp[(size_t)(j * KB + kw) * WSTR + 64];
                }
            }
            {
                const int kn = kb + 1 < KB ? kb + 1 : KB - 1;
                const int bo = KSTACK ? stack_off(kn) : block_off(kn);
#pragma unroll
                for (int i = 0; i < MTW; ++i) {
                    x[RA ^ 1][i][0] = *reinterpret_cast<const half8*>(in_hi + bo + aoff[i]);
                    if (XLO) x[RA ^ 1][i][1] = *reinterpret_cast<const half8*>(in_lo + bo + aoff[i]);
                }
            }
            if (!ILV) __builtin_amdgcn_sched_barrier(0);
            // the three partial products, outermost so that an accumulator is revisited only after
            // MTW*NTW other MFMAs (no back-to-back dependent MFMAs)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {
                if ((pr == 1 && !XLO) || (pr == 2 && (!WLO || KSTACK))) continue;  // (w_hi, x_hi), (w_hi, x_lo), (w_lo, x_hi)
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[RB][j][pr == 2 ? 1 : 0], x[RA][i][pr == 1 ? 1 : 0],
                                                                           acc[i][j], 0, 0, 0);
            }
            if (ILV) {  // the block's own LDS reads ride between its MFMAs instead of in front of them
                constexpr int NP = 1 + (XLO ? 1 : 0) + ((WLO && !KSTACK) ? 1 : 0);
                constexpr int NM = NP * MTW * NTW, ND = MTW * (XLO ? 2 : 1), NV = NTW * ((WLO && !KSTACK) ? 2 : 1);
#pragma unroll
                for (int q = 0; q < ND; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, NV, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NM - ND - 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        mark(0);
        constexpr int UN = BR % 2 == 0 ? BR : 2 * BR;
        // with LASTT the final k-block is peeled: its MFMAs go tile by tile (all products of tile 0, then tile 1, ...) and no
        // scheduling fence follows, so the epilogue of the first tiles can issue while the MFMAs of the last ones still run
        constexpr int KBL = KB - LASTT;
        int kb = 0;
#pragma nounroll
        for (; kb + UN <= KBL; kb += UN) {
            [&]<int... R>(std::integer_sequence<int, R...>) __attribute__((always_inline)) {
                (block(std::integral_constant<int, R % BR>{}, std::integral_constant<int, R % 2>{}, kb + R), ...);
            }(std::make_integer_sequence<int, UN>{});
        }
        [&]<int... R>(std::integer_sequence<int, R...>) __attribute__((always_inline)) {
            ((R < KBL % UN ? block(std::integral_constant<int, R % BR>{}, std::integral_constant<int, R % 2>{}, KBL - KBL % UN + R)
                           : (void)0), ...);
        }(std::make_integer_sequence<int, UN>{});
        if constexpr (LASTT > 0) {
            static_assert(LASTT <= 2 && LASTT <= BR - 1, "the peeled blocks must all be resident in the rings");
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int b = KB - LASTT; b < KB; ++b)
#pragma unroll
                        for (int pr = 0; pr < 3; ++pr) {
                            if ((pr == 1 && !XLO) || (pr == 2 && (!WLO || KSTACK))) continue;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[b % BR][j][pr == 2 ? 1 : 0], x[b % 2][i][pr == 1 ? 1 : 0],
                                                                               acc[i][j], 0, 0, 0);
                        }
        }
        mark(1);

#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int tile = wm * MTW + i;  // wave-uniform
            const int m = tile * 16 + li;
            const bool full = WM == 1 ? (i + 1) * 16 <= M : false;
            if (full || m < M) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) epi(m, (wn * NTW + j) * 16 + 4 * lk, acc[i][j]);
            }
        }
    }
};

}  // namespace hm
