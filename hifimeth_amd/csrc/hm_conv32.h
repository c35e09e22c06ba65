// hm_conv32.h -- the fp32 implicit-GEMM convolution on v_mfma_f32_16x16x4_f32 (exact fp32 products and accumulation) shared
// by the per-site fp32 front / tail kernels (hm_kernels.hip) and the fp32 dense trunk / edge kernels (hm_trunk_f32.hip).
#pragma once
#include <type_traits>
#include <utility>

#include "hm_kernels.h"

namespace hm {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// =================================================================================================
// CNN: implicit-GEMM conv1d(stride 2, pad 1) on v_mfma_f32_16x16x4_f32
// =================================================================================================
//
// One layer = GEMM  Out[m][co] = sum_kk A[m][kk] * Wk[kk][co],  m = (site, p) stacked over S sites,
// kk = tap*CIN + c,  A[m][kk] = in[site][2p - 1 + tap][c].  Activations live in LDS channels-last
// with one zero row in front of and behind every site (the conv's own padding), so the im2col row
// of output position p is the contiguous slice starting at physical row 2p:
//     A[m][kk] = lds[site*ISS + (2p + ROW0 + tap)*IRS + c]
// Row strides are 4 x odd floats (12 / 132 / 100 / 68): rows stay 16-byte aligned and the ds_read_b128
// pattern of the A fragment (lane l: row l&15, 4 consecutive K elements chosen by l>>4) is bank-conflict free.
// Weights are pre-packed on the host in fragment order [n-tile][k-group of 16][lane][4]: lane l of
// n-tile nt holds Wk[kg*16 + 4*(l>>4) + s][nt*16 + (l&15)] in component s (the same k permutation as the
// A side), so every lane fetches the B operands of four k-steps with one 16-byte load straight from L2.
// The 4 waves of a workgroup tile the output as WM x WN blocks of (MTW x NTW) 16x16 tiles.

template <int NW_, int S_, int CIN_, int KT_, int COUT_, int LOUT_, int IRS_, int ISS_, int ROW0_, int WM_, int WN_,
          int BR_ = 2, int VR_ = 0, int MSTR_ = 2, int DIL_ = 1>
struct Conv {
    // MSTR: LDS rows between consecutive output positions (2 = the model's stride-2 conv over one site's rows, 1 = dense
    // evaluation at every position); DIL: rows between consecutive taps (the dilation of the dense a-trous form)
    static constexpr int MSTR = MSTR_, DIL = DIL_;
    static constexpr int NW = NW_, S = S_, CIN = CIN_, KT = KT_, COUT = COUT_, LOUT = LOUT_, IRS = IRS_, ISS = ISS_;
    static constexpr int ROW0 = ROW0_, WM = WM_, WN = WN_;
    static constexpr int BR = BR_;  // B-fragment register ring: BR-1 k-groups in flight from L2
    static constexpr int VR = VR_;  // ragged rows (M mod 16) computed on the VALU beside the MFMAs instead of
                                    // padding one more 16-row MFMA tile with them
    static constexpr int M = S * LOUT;
    static constexpr int MT = (M - VR + 15) / 16;
    static constexpr int NT = COUT / 16;
    static constexpr int MTW = (MT + WM - 1) / WM;
    static constexpr int NTW = NT / WN;
    static constexpr int K = KT * CIN;
    static constexpr int KG = K / 16;
    static_assert(K % 16 == 0 && COUT % 16 == 0 && NT % WN == 0 && WM * WN <= NW && CIN % 4 == 0, "bad conv geometry");
    static_assert(CIN % 16 == 0 || 16 % CIN == 0, "a 16-wide k-group must not straddle taps unevenly");
    static_assert(IRS % 4 == 0 && ISS % 4 == 0, "A fragments are fetched with 16-byte LDS reads");
    static_assert(BR >= 2 && KG >= BR - 1 && KG >= 1, "bad pipeline depth");
    static_assert(VR == 0 || (WM == 1 && S == 1 && (M - VR) % 16 == 0 && VR < 16), "VALU rows need a 1 x N wave grid");

    // LDS offset (floats) of the first element of k-group kg relative to physical row 2p+ROW0, channel 0
    static __device__ __forceinline__ int group_off(int kg) {
        const int kk = kg * 16;
        const int tap = kk / CIN;
        return tap * DIL * IRS + (kk - tap * CIN);
    }

    struct NoMark {
        __device__ __forceinline__ void operator()(int) const {}
    };

    template <class Epi, class Mark = NoMark>
    static __device__ __forceinline__ void run(const float* __restrict__ in, const float* __restrict__ wfrag, Epi epi,
                                               Mark mark = Mark{}) {
        // launder the thread id so that hipcc does not hoist this layer's address arithmetic out of the
        // persistent site loop (it otherwise keeps every layer's invariants live and spills)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (WM * WN < NW && wave >= WM * WN) return;  // spare waves of a small layer (no barrier inside run)
        const int wm = WM == 1 ? 0 : wave / WN, wn = WM == 1 ? wave : wave % WN;
        const int li = lane & 15, lk = lane >> 4;
        // k permutation inside a 16-wide k-group: MFMA k-step s, k-index lk  <->  K element 4*lk + s.
        // Lane (li, lk) therefore owns 4 consecutive K elements = one 16-byte LDS read per k-group.
        const int lk_off = CIN >= 16 ? 4 * lk : ((4 * lk) / CIN) * DIL * IRS + (4 * lk) % CIN;

        int aoff[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            int m = (wm * MTW + i) * 16 + li;
            m = m < M ? m : M - 1;  // rows of the ragged last tile re-read the last valid row (discarded later)
            const int site = m / LOUT, p = m - site * LOUT;
            aoff[i] = site * ISS + (MSTR * p + ROW0) * IRS + lk_off;
        }
        // Operands are swapped (weights as the MFMA A operand, activations as B) so that the C/D layout
        // gives every lane 4 CONSECUTIVE output channels of one position: D[row = cout = 4*lk + r][col = pos li].
        // The bias is folded into the accumulator init; the epilogue stores 16 bytes per tile.
        f32x4_t acc[MTW][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const float4 bz = *reinterpret_cast<const float4*>(epi.bias + (wn * NTW + j) * 16 + 4 * lk);
#pragma unroll
            for (int i = 0; i < MTW; ++i) acc[i][j] = f32x4_t{bz.x, bz.y, bz.z, bz.w};
        }

        // VALU rows: lane (li, lk) accumulates the partial dot product over its own 4 K elements per group for
        // output channel nt*16 + li; the 4 lk lane-groups are summed once after the k-loop.
        constexpr int VRN = VR > 0 ? VR : 1;
        int roff[VRN];
        float psum[VRN][NTW];
        float4 ar[VRN];
#pragma unroll
        for (int r = 0; r < VRN; ++r) {
            roff[r] = (MSTR * (MT * 16 + r) + ROW0) * IRS + lk_off;
#pragma unroll
            for (int j = 0; j < NTW; ++j) psum[r][j] = 0.f;
        }

        float bcol[NTW];  // bias of this lane's VALU-row output channel, fetched up front
        if (VR > 0) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) bcol[j] = epi.bias[(wn * NTW + j) * 16 + li];
        }

        const float4* wp = reinterpret_cast<const float4*>(wfrag) + (size_t)(wn * NTW) * KG * 64 + lane;
        float4 bq[BR][NTW];
        float4 a[2][MTW];
        // prologue: BR-1 k-groups of B and one k-group of A in flight
#pragma unroll
        for (int r = 0; r < BR - 1; ++r)
#pragma unroll
            for (int j = 0; j < NTW; ++j) bq[r][j] = wp[(size_t)(j * KG + r) * 64];
#pragma unroll
        for (int i = 0; i < MTW; ++i) a[0][i] = *reinterpret_cast<const float4*>(in + aoff[i]);

        // One k-group = 4 MFMA k-steps.  Software pipeline pinned with sched_barrier so that hipcc keeps
        // it: B fragments BR-1 groups ahead (global/L2), A fragments one group ahead (LDS, ds_read_b128).
        auto group = [&](auto rb_tag, auto ra_tag, const int kg) __attribute__((always_inline)) {
            constexpr int RB = decltype(rb_tag)::value;
            constexpr int RA = decltype(ra_tag)::value;
            // loads are unconditional (the last groups re-fetch the final group) so that the body stays one
            // basic block = one scheduling region
            if (VR > 0) {  // the VALU rows of THIS group
                const float* gc = in + group_off(kg);
#pragma unroll
                for (int r = 0; r < VRN; ++r) ar[r] = *reinterpret_cast<const float4*>(gc + roff[r]);
            }
            {
                const int kb = kg + BR - 1 < KG ? kg + BR - 1 : KG - 1;
#pragma unroll
                for (int j = 0; j < NTW; ++j) bq[(RB + BR - 1) % BR][j] = wp[(size_t)(j * KG + kb) * 64];
            }
            {
                const float* gn = in + group_off(kg + 1 < KG ? kg + 1 : KG - 1);
#pragma unroll
                for (int i = 0; i < MTW; ++i) a[RA ^ 1][i] = *reinterpret_cast<const float4*>(gn + aoff[i]);
            }
            __builtin_amdgcn_sched_barrier(0);  // loads first, then the MFMA block (measured faster than interleaving)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const float av = s == 0 ? a[RA][i].x : s == 1 ? a[RA][i].y : s == 2 ? a[RA][i].z : a[RA][i].w;
                        const float bv = s == 0 ? bq[RB][j].x : s == 1 ? bq[RB][j].y : s == 2 ? bq[RB][j].z : bq[RB][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);
                    }
            if (VR > 0) {
#pragma unroll
                for (int r = 0; r < VRN; ++r)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        float t = psum[r][j];
                        t = fmaf(ar[r].x, bq[RB][j].x, t);
                        t = fmaf(ar[r].y, bq[RB][j].y, t);
                        t = fmaf(ar[r].z, bq[RB][j].z, t);
                        t = fmaf(ar[r].w, bq[RB][j].w, t);
                        psum[r][j] = t;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        mark(0);
        // main loop unrolled by lcm(BR, 2) so that every ring index is a compile-time constant
        constexpr int UN = BR % 2 == 0 ? BR : 2 * BR;
        int kg = 0;
#pragma nounroll
        for (; kg + UN <= KG; kg += UN) {
            [&]<int... R>(std::integer_sequence<int, R...>) __attribute__((always_inline)) {
                (group(std::integral_constant<int, R % BR>{}, std::integral_constant<int, R % 2>{}, kg + R), ...);
            }(std::make_integer_sequence<int, UN>{});
        }
        [&]<int... R>(std::integer_sequence<int, R...>) __attribute__((always_inline)) {
            ((R < KG % UN ? group(std::integral_constant<int, R % BR>{}, std::integral_constant<int, R % 2>{}, KG - KG % UN + R)
                          : (void)0), ...);
        }(std::make_integer_sequence<int, UN>{});
        mark(1);

#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int m = (wm * MTW + i) * 16 + li;  // this lane's output position (row of the GEMM)
            // tiles that lie completely inside M need no predicate (compile-time when WM == 1)
            const bool full = WM == 1 ? (i + 1) * 16 <= M : false;
            if (full || m < M) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) epi(m, (wn * NTW + j) * 16 + 4 * lk, acc[i][j]);
            }
        }
        if (VR > 0) {
#pragma unroll
            for (int r = 0; r < VRN; ++r)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    float v = psum[r][j];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    const int col = (wn * NTW + j) * 16 + li;
                    if (lk == (r & 3)) epi.store1(MT * 16 + r, col, v + bcol[j]);
                }
        }
    }
};



// ReLU (bias already in the accumulator); 4 consecutive channels of position m -> LDS channels-last with
// row stride ORS / site stride OSS (physical row p+1)
template <int LOUT, int ORS, int OSS>
struct EpiLds {
    float* out;
    const float* __restrict__ bias;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4_t& acc) const {
        const int site = OSS == 0 ? 0 : m / LOUT;
        const int p = OSS == 0 ? m : m - site * LOUT;
        *reinterpret_cast<float4*>(out + site * OSS + (p + 1) * ORS + col) =
            make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    }
    __device__ __forceinline__ void store1(int m, int col, float v) const {
        const int site = OSS == 0 ? 0 : m / LOUT;
        const int p = OSS == 0 ? m : m - site * LOUT;
        out[site * OSS + (p + 1) * ORS + col] = fmaxf(v, 0.f);
    }
};

// ReLU, result to global channels-last [m][COUT]
template <int COUT>
struct EpiGlobal {
    float* __restrict__ out;
    const float* __restrict__ bias;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4_t& acc) const {
        *reinterpret_cast<float4*>(out + (size_t)m * COUT + col) =
            make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    }
    __device__ __forceinline__ void store1(int m, int col, float v) const { out[(size_t)m * COUT + col] = fmaxf(v, 0.f); }
};

// zero the padding rows (physical rows 0 and LOUT+1) of S stacked sites
template <int S, int LOUT, int COUT, int ORS, int OSS>
__device__ __forceinline__ void zero_pad_rows(float* out) {
    for (int i = threadIdx.x; i < S * 2 * COUT; i += blockDim.x) {
        const int site = i / (2 * COUT), rem = i - site * 2 * COUT;
        const int which = rem / COUT, c = rem - which * COUT;
        out[site * OSS + (which ? (LOUT + 1) : 0) * ORS + c] = 0.f;
    }
}

template <int L, int C, int RS>
__device__ __forceinline__ void dump_lds(const float* buf, float* __restrict__ dbg) {
    for (int i = threadIdx.x; i < L * C; i += blockDim.x) dbg[i] = buf[(i / C + 1) * RS + (i % C)];
}

}  // namespace hm
