// hm_tail_head.hip -- conv7, conv8, fc1, fc2 and the softmax of the dense-trunk path's CHH sites as a kernel of their own (round 5),
// behind the strip tail kernel (hm_tail_p.hip), which stops at conv6.
//
// Inside the strip kernel conv7 and conv8 are LDS-latency chains: 120 MFMAs per wave and pass in 6.5 k cycles, six MFMAs per k-block
// with ONE block of look-ahead -- conv5's and conv6's resident weights leave no registers for a deeper operand ring, nor for conv7's
// and conv8's own weights, which cross the CU's vector-memory path once per pass of 14 sites (122 KB) -- and conv6's output overlays the
// strip, so that most of the next pass's rows can only be fetched once conv7 has read it.  Cut behind conv6 the strip kernel's pass
// takes 20.6 k cycles instead of 26.5 k (profiles/r05_strip_main_only_timing_dense5.txt).  Here conv7's, conv8's and fc1's weights of a
// wave (144 + 96 + 128 registers) are RESIDENT for the launch, the operand rings are 16 deep with three blocks of look-ahead, and the
// sites arrive as full tiles of 16 in the strip kernel's list order.
//
// Hand-over: conv6's output, 7 positions x 96 channels, split: [hi: p * 96 + c | lo] = 1 344 halves = 2 688 B per site (TAIL_XH_HALVES),
// at the site's position in the class-sorted list; dst[pos] = the site's slot in the batch's result arrays (written by the class sort).
// A tile's 43 KB come in by LDS-DMA (four LDS rows = one position of four sites per instruction), the next tile's while this one is
// computed (two input buffers).
//
// Same products in the same order per accumulator as tail_kernel_r (bias, then the live k-blocks in ascending order with w_hi x_hi,
// w_hi x_lo, w_lo x_hi; the zero-padding taps skipped: exact zeros), fc2's sums in the same order: byte-identical calls
// (tests/test_gpu_parity.py).  W16: engine option precision = 2 -- conv8 and fc1 with plain fp16 weights.
//
// Reference for what is computed: training/model_cnn.py:50-85 (conv7 .. fc2), softmax -> ML byte: mod_batch.cpp:46-64.
#include "hm_tail_p_geo.h"
#ifdef HM_TRUNK_STAMP   // diagnostic build (make stamp): shader-clock phase sums of workgroup 0, read by tools/tailhead_stamps.py
#include "hm_stamp.h"
namespace hm { __device__ unsigned long long g_tailhead_stamp[4][16]; }
extern "C" int hm_debug_tailhead_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(hm::g_tailhead_stamp), sizeof(hm::g_tailhead_stamp)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[4][16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(hm::g_tailhead_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifndef HM_HEAD_GRID_PER_CU
#define HM_HEAD_GRID_PER_CU 4
#endif

namespace hm {

namespace {

struct HGeo {
    static constexpr int S = 16, NW = 4;
    static constexpr int L6 = PGeo::L6, L7 = PGeo::L7, L8 = PGeo::L8;
    static constexpr int RS = PGeo::RS, RS64 = PGeo::RS64, HPS = PGeo::HPS, F2S = PGeo::F2S;
    static constexpr int HRS = 8 * HPS + 4;        // fc1's output rows 292 floats apart: conflict-free float4 stores (hm_tail_fc.hip)
    static constexpr int IN = L6 * S * RS;         // halves of one plane of one input buffer: [7 positions][16 sites][RS]
    static constexpr int C7 = L7 * S * RS64, C8 = L8 * S * RS64;
    static constexpr int XH = TAIL_XH_HALVES;
    static_assert(XH == 2 * L6 * 96 && RS * 2 == 13 * 16, "a site's hand-over = two planes of 7 rows of 12 sixteen-byte units; an LDS row = 13");
};

}  // namespace

template <bool W16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tail_head_p_kernel(SiteRange sr, CtxWeights W, float* __restrict__ logits, float* __restrict__ prob, uint8_t* __restrict__ ml,
                        const half_t* __restrict__ xh, const int32_t* __restrict__ dst) {
    using T = HGeo;
    const Site* sites;   // (not read here: the sites' batch slots come through dst)
    const int n_sites = resolve_sites(sr, sites);
    const int n_tiles = (n_sites + T::S - 1) / T::S;
    if ((int)blockIdx.x >= n_tiles) return;

    struct Lds {
        half_t in[2][2][T::IN];           // [buffer][hi | lo]
        half_t c7[2][T::C7], c8[2][T::C8];
        float hfc[T::S * T::HRS];
        float fc2w[T::F2S + 8 * T::HPS + 4];
        float bias_l[64 + 64];            // conv7's, conv8's
    };
    __shared__ __attribute__((aligned(16))) Lds lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 256 + 2; i += T::NW * 64) {
        if (i < 512) lds.fc2w[(i >> 8) * T::F2S + ((i & 255) >> 5) * T::HPS + (i & 31)] = W.fc2_w[i];
        else lds.fc2w[T::F2S + 8 * T::HPS + (i - 512)] = W.fc2_b[i - 512];
    }
    for (int i = tid; i < 128; i += T::NW * 64) lds.bias_l[i] = i < 64 ? W.bias[6][i] : W.bias[7][i - 64];
    // (the planes' pad columns -- 96 .. 103 of the input rows, 64 .. 71 of conv7's / conv8's -- are never read: a k-block ends at the last channel)

    // ---- resident weights.  conv7 (4 position tiles x 4 n-tiles) and conv8 (2 x 4) are dealt 2 x 2: wave (mh, nh) = (wave >> 1, wave & 1) runs
    // n-tiles 2 nh, 2 nh + 1 on conv7's tiles 2 mh, 2 mh + 1 and on conv8's tile mh -- a wave reads HALF of a layer's input rows for TWICE the
    // MFMAs per read (with one n-tile per wave on all tiles, as inside the strip kernel, the four waves' operand reads alone keep the LDS busy
    // for 1.9 k of conv7's 1.4 k MFMA cycles).  fc1: n-tiles 4 wave .. 4 wave + 3 ------------------------------------------
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };
    const int mh = wave >> 1, nh = wave & 1;
    const int nt78[2] = {2 * nh, 2 * nh + 1}, col78[2] = {32 * nh, 32 * nh + 16};
    const int ntf[4] = {4 * wave, 4 * wave + 1, 4 * wave + 2, 4 * wave + 3}, colf[4] = {64 * wave, 64 * wave + 16, 64 * wave + 32, 64 * wave + 48};
    TW<9, 2> W7;
    TW<6, 2> W8;
    TW<4, 4> WF;
    tw_load(wf(6), nt78, lane, W7);
    tw_load<!W16>(wf(7), nt78, lane, W8);
    tw_load<!W16>(wf(8), ntf, lane, WF);
    float4 bzf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bzf[j] = *reinterpret_cast<const float4*>(W.bias[8] + colf[j] + 4 * (lane >> 4));

    using C96 = PCfg<96, 3>;
    using C64 = PCfg<64, 3, !W16>;
    using CF = PCfg<64, 2, !W16>;
    using I8 = PInRows<C64, T::RS64, T::L7, 0>;
    using IF = PInRows<CF, T::RS64, T::L8, 0, 0, 0>;

    // ---- a tile's rows: 28 pieces of (position p, sites 4 j .. 4 j + 3) per plane, 7 of them per wave; lane l < 52 brings chunk l % 13 of
    // row l / 13 (chunk 12 is the LDS row's pad: it reads the 16 bytes behind the 192 it needs) --------------------------------------------
    const unsigned long long xhb = (unsigned long long)(uintptr_t)xh;
    const unsigned long long lanes52 = 0x000FFFFFFFFFFFFFull;
    const int G = (int)gridDim.x, last = n_tiles - 1;
    auto dma_tile = [&](const int tile, const uint32_t lds_hi, const uint32_t lds_lo) __attribute__((always_inline)) {
        const int ln = threadIdx.x & 63, r = min(ln / 13, 3), c16 = (ln % 13) * 16;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int j = wave + T::NW * k, p = j >> 2, s0 = (j & 3) * 4;   // wave-uniform
            const int site = min(tile * T::S + s0 + r, n_sites - 1);       // (a ragged last tile re-reads the last site; its extra results are stored to the same slot: same values)
            const unsigned long long src = xhb + (unsigned long long)site * (T::XH * 2) + (unsigned)(p * 192 + c16);
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(lds_hi + (uint32_t)((p * T::S + s0) * T::RS * 2));
            const uint32_t d1 = __builtin_amdgcn_readfirstlane(lds_lo + (uint32_t)((p * T::S + s0) * T::RS * 2) - (uint32_t)T::XH);
            unsigned long long sv;
            uint32_t km;
            asm volatile(
                "s_mov_b64 %0, exec\n\t"
                "s_mov_b32 %1, m0\n\t"
                "s_mov_b64 exec, %2\n\t"
                "s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %5, off\n\t"
                "s_mov_b32 m0, %4\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %5, off offset:1344\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_mov_b64 exec, %0"
                : "=&s"(sv), "=&s"(km)
                : "s"(lanes52), "s"(d0), "s"(d1), "v"(src));
        }
    };
    static_assert(HGeo::XH == 1344, "the lo plane's immediate offset above");
    const uint32_t lds_in[2][2] = {{(uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)lds.in[0][0],
                                    (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)lds.in[0][1]},
                                   {(uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)lds.in[1][0],
                                    (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)lds.in[1][1]}};
    const float* const b7 = lds.bias_l;
    const float* const b8 = lds.bias_l + 64;
    auto load_dst = [&](const int tile) __attribute__((always_inline)) { return dst[min(tile * T::S + (tid >> 4), n_sites - 1)]; };

#ifdef HM_TRUNK_STAMP
    unsigned long long hts[10], hacc[10] = {}, hn = 0;
    const bool hst = blockIdx.x == 0;
    const unsigned long long hk0 = hm_stamp(), hr0 = __builtin_amdgcn_s_memrealtime();
#define HTS(i) do { if (hst) hts[i] = hm_stamp(); } while (0)
#else
#define HTS(i)
#endif
    // a tile's turn on input buffer BUF: the next tile's rows are requested into the other buffer first (its last reader, conv7 of the
    // tile before this one, is behind the barrier that opened this turn)
    auto turn = [&](auto buf_, const int tile, int& dreg) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_)::value;
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int li = tl & 15, lk = (tl & 63) >> 4;
        const int d = dreg;
        HTS(0);
        dma_tile(min(tile + G, last), lds_in[BUF ^ 1][0], lds_in[BUF ^ 1][1]);
        dreg = load_dst(min(tile + 2 * G, last));
        // ---- conv7: input rows -> c7 ----
        {
            using I7 = PInRows<C96, T::RS, T::L6, 0>;
            const I7 ia{li * T::RS + 8 * lk};
            const EpiP<T::RS64> e7{lds.c7[0] + li * T::RS64 + 4 * lk, lds.c7[1] + li * T::RS64 + 4 * lk};
            if (mh) PConv<C96, I7, 8, 1, TG<2, 2, 0, 0>>::run(lds.in[BUF][0], lds.in[BUF][1], W7, b7, col78, ia, e7);
            else PConv<C96, I7, 8, 1, TG<0, 2, 0, 0>>::run(lds.in[BUF][0], lds.in[BUF][1], W7, b7, col78, ia, e7);
        }
        HTS(1);
        lds_barrier();
        HTS(2);
        // ---- conv8: c7 -> c8 ----
        {
            const I8 ia{li * T::RS64 + 8 * lk};
            const EpiP<T::RS64> e8{lds.c8[0] + li * T::RS64 + 4 * lk, lds.c8[1] + li * T::RS64 + 4 * lk};
            if (mh) PConv<C64, I8, 8, 1, TG<1, 1, 0, 0>>::run(lds.c7[0], lds.c7[1], W8, b8, col78, ia, e8);
            else PConv<C64, I8, 8, 1, TG<0, 1, 0, 0>>::run(lds.c7[0], lds.c7[1], W8, b8, col78, ia, e8);
        }
        HTS(3);
        lds_barrier();
        HTS(4);
        // ---- fc1: c8 -> hfc (one stream of the wave's four n-tiles, as in tail_fc_kernel) ----
        {
            const IF ia{li * T::RS64 + 8 * lk};
            const EpiFc1P<T::HPS> ef{lds.hfc + li * T::HRS + 4 * lk};
            PConv<CF, IF, 8, 3, TG<0, 1, 0, 0, 4>>::run(lds.c8[0], lds.c8[1], WF, [&](int j) __attribute__((always_inline)) { return bzf[j]; }, colf, ia, ef);
        }
        HTS(5);
        vm_drain();      // this wave's share of the next tile's rows has landed -- requested a conv7, a conv8 and an fc1 ago; the stores still in flight
                         // are the previous tile's.  (Waiting behind fc2 instead would wait for ITS stores: a full memory round trip per tile.)
        HTS(6);
        lds_barrier();   // ... every wave's share; fc1's output is complete
        HTS(7);
        // ---- fc2 + softmax (mod_batch.cpp:46-64) in fp32: 16 lanes per site = 2 outputs x 8 partial sums; every lane of a site stores the
        // site's results (same values, same addresses: no branch around the stores -- hm_tail_fc.hip) ----
        {
            const int bsite = tl >> 4, o = (tl >> 3) & 1, part = tl & 7;
            const float4* h = reinterpret_cast<const float4*>(lds.hfc + bsite * T::HRS + part * T::HPS);
            const float4* w2 = reinterpret_cast<const float4*>(lds.fc2w + o * T::F2S + part * T::HPS);
            float4 hv[8], wv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                hv[k] = h[k];
                wv[k] = w2[k];
            }
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sum = fmaf(hv[k].x, wv[k].x, sum);
                sum = fmaf(hv[k].y, wv[k].y, sum);
                sum = fmaf(hv[k].z, wv[k].z, sum);
                sum = fmaf(hv[k].w, wv[k].w, sum);
            }
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 1, 64);
            sum += lds.fc2w[T::F2S + 8 * T::HPS + o];
            const float other = __shfl_xor(sum, 8, 64);
            const float v0 = o ? other : sum, v1 = o ? sum : other;
            const float mx = fmaxf(v0, v1);
            const float e0 = expf(v0 - mx), e1 = expf(v1 - mx);
            const float p1 = e1 / (e0 + e1);
            int q = (int)(255 * p1);
            q = q > 255 ? 255 : q;
            *reinterpret_cast<float2*>(logits + 2 * (size_t)d) = make_float2(v0, v1);
            prob[d] = p1;
            ml[d] = (uint8_t)q;
        }
        HTS(8);
#ifdef HM_TRUNK_STAMP
        if (hst) { for (int i = 0; i < 8; ++i) hacc[i] += hts[i + 1] - hts[i]; ++hn; }
#endif
        // (no barrier here: the next turn's first writer of hfc is its fc1, two barriers on; of c7 its conv7, whose last reader -- this turn's
        //  conv8 -- is two barriers back; the buffer the next turn's DMA fills was last read by this turn's conv7)
    };
    int tile = blockIdx.x;
    int d0 = load_dst(tile), d1 = load_dst(min(tile + G, last));
    dma_tile(tile, lds_in[0][0], lds_in[0][1]);
    vm_drain();
    lds_barrier();   // the first tile's rows, fc2's weights and the biases are in LDS
    while (true) {
        turn(std::integral_constant<int, 0>{}, tile, d0);
        if (tile + G >= n_tiles) break;
        turn(std::integral_constant<int, 1>{}, tile + G, d1);
        if (tile + 2 * G >= n_tiles) break;
        tile += 2 * G;
    }
#ifdef HM_TRUNK_STAMP
    if (hst && lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&g_tailhead_stamp[wave][i], hacc[i]);
        atomicAdd(&g_tailhead_stamp[wave][11], hn);
        atomicAdd(&g_tailhead_stamp[wave][13], hm_stamp() - hk0);
        atomicAdd(&g_tailhead_stamp[wave][14], __builtin_amdgcn_s_memrealtime() - hr0);
    }
#endif
#undef HTS
}

size_t tail_head_xh_bytes(int64_t sites) { return (size_t)std::max<int64_t>(sites, 1) * TAIL_XH_HALVES * sizeof(uint16_t); }

void launch_tail_head_p(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const uint16_t* xh, const int32_t* dst, float* logits, float* p,
                        uint8_t* ml, int grid, bool w16) {
    if (sr.cap <= 0) return;
    // one workgroup fits a CU (142 KB of LDS); a few times as many are launched so that the dispatcher evens the CUs out (each loads its
    // 250 KB of weights once); with a host-known count no more workgroups than there are tiles
    const int want = HM_HEAD_GRID_PER_CU * grid;
    const dim3 g(sr.totals ? want : max(1, min((sr.cap + HGeo::S - 1) / HGeo::S, want)));
    if (w16) hipLaunchKernelGGL(tail_head_p_kernel<true>, g, dim3(256), 0, st, sr, w, logits, p, ml, reinterpret_cast<const half_t*>(xh), dst);
    else hipLaunchKernelGGL(tail_head_p_kernel<false>, g, dim3(256), 0, st, sr, w, logits, p, ml, reinterpret_cast<const half_t*>(xh), dst);
}

}  // namespace hm
