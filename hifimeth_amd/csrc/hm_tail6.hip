// hm_tail6.hip -- the per-site layers behind the dense trunk, split in two kernels:
//
//   conv5_ws_kernel : conv5 (25 -> 13 rows, 96 -> 96 channels) WEIGHT-STATIONARY.  The layer's 110 KB of split weights sit
//                     in LDS for the whole launch; a wave takes 32 GEMM rows (sites x 13 output rows, stacked densely), reads
//                     its activations straight from the E4 map / the edge rows as MFMA operands (16-byte loads, no LDS
//                     staging) and owns all 96 output channels.  There is no barrier after the weights are in: eight waves
//                     per CU run free of each other, one's load latency under the others' MFMAs.  In the fused tail the same
//                     layer streamed its weights from L2 once per 8 sites and waited at a barrier before and after.
//   tail6_kernel    : conv6 .. conv8, fc1, fc2, softmax for 8 sites per workgroup pass, from conv5's rows (13 per site,
//                     contiguous, already split): half the input rows of the conv4-based tail, so the whole next group is
//                     staged by LDS-DMA in one go beside conv7.
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98); mod_batch.cpp:46-64.
#include "hm_convh.h"

namespace hm {

namespace {

constexpr int L4 = 25, L5 = 13, L6 = 7, L7 = 4, L8 = 2;
constexpr int C5_ROW = 2 * C4_CH;  // halves per conv4 / conv5 row in HBM: [hi 96 | lo 96]

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
template <bool WLO>
__global__ __launch_bounds__(512) void conv5_ws_kernel(SiteRange sr, CtxWeights W, const half_t* __restrict__ e4,
                                                        const half_t* __restrict__ edge4, const int32_t* __restrict__ e4row,
                                                        const half_t* __restrict__ zeros, half_t* __restrict__ c5) {
    constexpr int NT = 6, KB = 9, MTW = 2;  // 96 channels, K = 3 taps x 96 = 9 blocks of 32, 32 rows per wave pass
    __shared__ __attribute__((aligned(16))) half_t wl[NT * KB * 2 * 64 * 8];  // [n-tile][k-block][plane][lane][8] = 110 592 B
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    {
        const uint4* src = reinterpret_cast<const uint4*>(W.wfrag_h[4]);
        uint4* dst = reinterpret_cast<uint4*>(wl);
        for (int i = threadIdx.x; i < NT * KB * 2 * 64; i += 512) dst[i] = src[i];
    }
    __syncthreads();  // the only barrier of the kernel
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_rows = n_sites * L5;
    const half8* wl8 = reinterpret_cast<const half8*>(wl) + lane;
    // A workgroup takes a CONTIGUOUS range of row blocks (its waves interleave inside it): neighbouring sites read the same
    // E4 rows (a row serves the ~3 sites whose windows sample it), so a CU re-reads them from its own L1 / its XCD's L2 a few
    // blocks later.  Dealt round-robin over the chip, the same rows were fetched into all eight L2s: 38 ms instead of 13.
    const int n_blk = (n_rows + 16 * MTW - 1) / (16 * MTW);
    const int per_wg = (n_blk + (int)gridDim.x - 1) / (int)gridDim.x;
    const int blk_end = min(n_blk, ((int)blockIdx.x + 1) * per_wg);
    for (int blk = blockIdx.x * per_wg + wave; blk < blk_end; blk += 8) {
        // this lane's activation rows: output row R = site * 13 + p reads conv4 rows 2p - 1 + tap of its site
        const half_t* rp[MTW][3];
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int R = blk * (16 * MTW) + i * 16 + li;
            const int site = R / L5, p = R - site * L5;
            const bool valid = R < n_rows;
            int32_t row0 = 0;
            if (valid) row0 = e4row[site];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int q = 2 * p - 1 + t;
                const half_t* ptr = zeros;  // the conv's zero padding, and rows past the end
                if (valid && q >= 0 && q < L4)
                    ptr = q == 0 ? edge4 + (size_t)site * (2 * C5_ROW) : q == L4 - 1 ? edge4 + (size_t)site * (2 * C5_ROW) + C5_ROW
                                                                                    : e4 + ((size_t)row0 + 16 * q) * C5_ROW;
#if defined(HM_C5_EXP) && HM_C5_EXP == 1
                ptr = zeros;  // experiment: no gather (wrong results)
#elif defined(HM_C5_EXP) && HM_C5_EXP == 2
                ptr = e4 + (size_t)((R * 3 + t) % 4096) * C5_ROW;  // experiment: gather inside a 1.5 MB region
#endif
                rp[i][t] = ptr + 8 * lk;
            }
        }
        f32x4 acc[MTW][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float4 bz = *reinterpret_cast<const float4*>(W.bias[4] + j * 16 + 4 * lk);
#pragma unroll
            for (int i = 0; i < MTW; ++i) acc[i][j] = f32x4{bz.x, bz.y, bz.z, bz.w};
        }
        // Activations one TAP ahead (L2 / HBM), weights of the block from LDS.  A tap's three k-blocks read the three
        // 32-channel thirds of the same rows: requested together, the second touch of each 128-byte line still hits the L1
        // (one k-block at a time it had been evicted by the other waves' rows: twice the L2 requests).
        half8 x[2][3][MTW][2];
        auto load_tap = [&](auto slot_tag, const int tap) __attribute__((always_inline)) {
            constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const half_t* p = rp[i][tap] + 32 * j;
                    x[SLOT][j][i][0] = *reinterpret_cast<const half8*>(p);
                    x[SLOT][j][i][1] = *reinterpret_cast<const half8*>(p + C4_CH);
                }
        };
        // one k-block: this block's weights from LDS, 36 MFMAs.  The scheduling fence keeps a block's weight fragments from
        // being hoisted over the previous block (48 VGPRs each).
        auto block = [&](auto slot_tag, auto j_tag, const int kb) __attribute__((always_inline)) {
            constexpr int SLOT = decltype(slot_tag)::value, J = decltype(j_tag)::value;
            half8 wv[NT][2];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                wv[j][0] = wl8[((j * KB + kb) * 2 + 0) * 64];
                if (WLO) wv[j][1] = wl8[((j * KB + kb) * 2 + 1) * 64];
            }
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {  // (w_hi, x_hi), (w_hi, x_lo), (w_lo, x_hi)
                if (pr == 2 && !WLO) continue;
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[j][pr == 2 ? 1 : 0], x[SLOT][J][i][pr == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        load_tap(I0{}, 0);
        load_tap(I1{}, 1);
        block(I0{}, I0{}, 0); block(I0{}, I1{}, 1); block(I0{}, I2{}, 2);
        load_tap(I0{}, 2);
        block(I1{}, I0{}, 3); block(I1{}, I1{}, 4); block(I1{}, I2{}, 5);
        block(I0{}, I0{}, 6); block(I0{}, I1{}, 7); block(I0{}, I2{}, 8);
        // ReLU + split: conv5 rows leave as [hi 96 | lo 96], site-major, 13 rows per site: row R of the output IS R
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int R = blk * (16 * MTW) + i * 16 + li;
            if (R < n_rows) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    half4 h, l;
                    split4(acc[i][j], h, l);
                    half_t* o = c5 + (size_t)R * C5_ROW + j * 16 + 4 * lk;
                    *reinterpret_cast<half4*>(o) = h;
                    *reinterpret_cast<half4*>(o + C4_CH) = l;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
namespace {

struct T6 {
    static constexpr int S = TAIL_SITES;
    static constexpr int RS96 = 104, RS64 = 72, HRS = 260;  // halves, halves, floats
    static constexpr int C5_SS = (L5 + 2) * RS96, C6_SS = (L6 + 2) * RS96, C7_SS = (L7 + 2) * RS64;
    static constexpr int FCB = 4, RING_SS = L8 * RS64, RING = FCB * S * RING_SS;
    // buffer 0, per plane: conv6's output in front, the staged conv5 rows behind it (so the next group can be staged as
    // soon as conv6 is done, in one piece); buffer 1: conv7's output planes, later fc1's fp32 output
    static constexpr int C6_P = S * C6_SS, C5_P = S * C5_SS, P0 = C6_P + C5_P;
    static constexpr int B1 = FCB * S * HRS * 2;  // halves: fc1 output as floats
    static_assert(2 * S * C7_SS <= B1, "conv7's planes fit buffer 1");
    static constexpr int LDS_HALVES = 2 * P0 + B1 + 2 * RING;
};

template <int LOUT, int ORS, int OSS>
struct EpiPl {  // ReLU + split -> planes with one padding row in front of every site
    half_t* hi;
    half_t* lo;
    const float* __restrict__ bias;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        const int site = m / LOUT, p = m - site * LOUT;
        half4 h, l;
        split4(acc, h, l);
        *reinterpret_cast<half4*>(hi + site * OSS + (p + 1) * ORS + col) = h;
        *reinterpret_cast<half4*>(lo + site * OSS + (p + 1) * ORS + col) = l;
    }
};

template <int ORS>
struct EpiRg {  // conv8's rows for the batched fc1: [site][2][ORS], no padding rows
    half_t* hi;
    half_t* lo;
    const float* __restrict__ bias;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        half4 h, l;
        split4(acc, h, l);
        *reinterpret_cast<half4*>(hi + m * ORS + col) = h;
        *reinterpret_cast<half4*>(lo + m * ORS + col) = l;
    }
};

template <int HRS>
struct EpiF1 {
    float* out;
    const float* __restrict__ bias;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        *reinterpret_cast<float4*>(out + m * HRS + col) =
            make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    }
};

template <int S, int LOUT, int C>
__device__ __forceinline__ void zero_pads(half_t* hi, half_t* lo, int rs, int ss) {
    for (int i = threadIdx.x; i < S * 2 * C; i += blockDim.x) {
        const int site = i / (2 * C), rem = i - site * 2 * C;
        const int which = rem / C, c = rem - which * C;
        const int o = site * ss + (which ? (LOUT + 1) : 0) * rs + c;
        hi[o] = (half_t)0.f;
        lo[o] = (half_t)0.f;
    }
}

}  // namespace

// W16: plain fp16 weights in conv6..conv8 (precision modes 2 and 3); fc1 keeps split weights
template <bool W16>
__global__ __launch_bounds__(512) void tail6_kernel(SiteRange sr, CtxWeights W, const half_t* __restrict__ c5,
                                                     const half_t* __restrict__ zeros, float* __restrict__ logits,
                                                     float* __restrict__ prob, uint8_t* __restrict__ ml) {
    using T = T6;
    constexpr int S = T::S, NW = 8, FCB = T::FCB;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    __shared__ __attribute__((aligned(16))) half_t smem[T::LDS_HALVES];
    half_t* h0 = smem;              // plane hi of buffer 0: [conv6 out | conv5 rows]
    half_t* l0 = smem + T::P0;
    half_t* b1 = smem + 2 * T::P0;  // conv7 out: hi at b1, lo at b1 + S * C7_SS; fc1 out (fp32) over both
    half_t* c7h = b1;
    half_t* c7l = b1 + S * T::C7_SS;
    float* hfc = reinterpret_cast<float*>(b1);
    half_t* r_hi = b1 + T::B1;
    half_t* r_lo = r_hi + T::RING;
    half_t* c5h = h0 + T::C6_P;
    half_t* c5l = l0 + T::C6_P;
    __shared__ float fc2w[2 * 256 + 2];
    for (int i = threadIdx.x; i < 2 * 256 + 2; i += NW * 64) fc2w[i] = i < 512 ? W.fc2_w[i] : W.fc2_b[i - 512];
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    // LDS-DMA staging of a group's conv5 rows: a plane of the image is S * 15 rows of 13 sixteen-byte units (12 data + 1
    // pad); piece p = units [64 p, 64 p + 64) = one wave instruction (LDS side: wave-uniform base + 16 * lane)
    constexpr int UNITS = S * (L5 + 2) * 13, PIECES = (UNITS + 63) / 64;
    static_assert(T::RS96 * 2 == 13 * 16, "a row is 13 units");
    auto stage_piece = [&](const int grp, const int p, const int pl) __attribute__((always_inline)) {
        const int u = 64 * p + lane;
        if (u < UNITS) {
            const int R = u / 13, ch = u - 13 * R;
            const int site = R / (L5 + 2), prow = R - (L5 + 2) * site;
            const int gs = grp * S + site, pos = prow - 1;
            const half_t* src = zeros;  // padding rows, row pads, sites past the end
            if (pos >= 0 && pos < L5 && ch < 12 && gs < n_sites) src = c5 + ((size_t)gs * L5 + pos) * C5_ROW + pl * C4_CH + ch * 8;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)(pl ? c5l : c5h) + 1024u * (uint32_t)p);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    auto stage_group = [&](int grp, int w0, int nw) __attribute__((always_inline)) {  // both planes, dealt to waves w0 .. w0+nw-1
        for (int q = wave - w0; q < 2 * PIECES; q += nw) stage_piece(grp, q >> 1, q & 1);
    };

    if ((int)blockIdx.x * S < n_sites) {
        stage_group(blockIdx.x, 0, NW);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA is invisible to the compiler's own wait counting
    }
    int slot = 0, g_first = blockIdx.x;  // groups g_first, g_first + gridDim.x, ... wait in ring slots 0 .. slot-1
    for (int g = blockIdx.x; g * S < n_sites; g += gridDim.x) {
        const bool more = (g + (int)gridDim.x) * S < n_sites;
        __syncthreads();  // conv5 rows of group g staged (by the previous iteration's spare waves)

        // conv6 / conv7 / conv8 as 1xN grids: a wave owns one 16-channel tile and every row, each weight byte enters the CU once
        ConvH<NW, 96, 3, 96, L6, T::RS96, 1, 6, 3, S, T::C5_SS, 0, !W16>::run(c5h, c5l, wf(5), EpiPl<L6, T::RS96, T::C6_SS>{h0, l0, W.bias[5]});
        zero_pads<S, L6, 96>(h0, l0, T::RS96, T::C6_SS);
        __syncthreads();

        if (wave >= 4) {  // no tile in conv7: the next group's conv5 rows, all of them (their part of buffer 0 is dead)
            if (more) stage_group(g + gridDim.x, 4, 4);
        } else {
            ConvH<NW, 96, 3, 64, L7, T::RS96, 1, 4, 3, S, T::C6_SS, 0, !W16>::run(h0, l0, wf(6), EpiPl<L7, T::RS64, T::C7_SS>{c7h, c7l, W.bias[6]});
        }
        zero_pads<S, L7, 64>(c7h, c7l, T::RS64, T::C7_SS);
        __syncthreads();

        if (wave >= 4) {
            // everything this wave staged for the next group is in LDS before it reaches the barrier in front of conv6
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            ConvH<NW, 64, 3, 64, L8, T::RS64, 1, 4, 3, S, T::C7_SS, 0, !W16>::run(
                c7h, c7l, wf(7), EpiRg<T::RS64>{r_hi + slot * S * T::RING_SS, r_lo + slot * S * T::RING_SS, W.bias[7]});
        }
        if (slot == 0) g_first = g;
        ++slot;
        if (slot < FCB && more) continue;  // the loop-top barrier orders this conv8 before the next conv6
        __syncthreads();

        // fc1 as a 2-tap "conv" over conv8's two positions (k order l*64 + c; see hm_weights.cpp), FCB * S sites at once;
        // slots this batch did not fill hold stale rows whose results are never written out
        ConvH<NW, 64, 2, 256, 1, T::RS64, 1, 8, 2, FCB * S, T::RING_SS, 0>::run(r_hi, r_lo, wf(8), EpiF1<T::HRS>{hfc, W.bias[8]});
        __syncthreads();

        // fc2 + softmax (mod_batch.cpp:46-64) in fp32: 16 lanes per site = 2 outputs x 8 partial sums
        {
            static_assert(FCB * S * 16 == NW * 64, "one 16-lane team per site");
            const int bsite = threadIdx.x >> 4, o = (threadIdx.x >> 3) & 1, part = threadIdx.x & 7;
            const float* h = hfc + bsite * T::HRS + part * 32;
            const float* w2 = fc2w + o * 256 + part * 32;
            float sum = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) sum = fmaf(h[k], w2[k], sum);
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 1, 64);
            sum += fc2w[512 + o];
            const float other = __shfl_xor(sum, 8, 64);
            const int sl = bsite / S, site = bsite - sl * S;
            const int gs0 = (g_first + sl * (int)gridDim.x) * S;
            if ((threadIdx.x & 15) == 0 && sl < slot && gs0 + site < n_sites) {
                const float v0 = sum, v1 = other;
                const float mx = fmaxf(v0, v1);
                const float e0 = expf(v0 - mx), e1 = expf(v1 - mx);
                const float p1 = e1 / (e0 + e1);
                int q = (int)(255 * p1);
                q = q > 255 ? 255 : q;
                const int dst = sites[gs0 + site].uidx;
                logits[2 * (size_t)dst] = v0;
                logits[2 * (size_t)dst + 1] = v1;
                prob[dst] = p1;
                ml[dst] = (uint8_t)q;
            }
        }
        slot = 0;  // hfc and the ring are next written behind the loop-top barrier / the conv7 barrier
    }
}

// ------------------------------------------------------------------------------------------------------------------------
void launch_conv5_ws(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, const uint16_t* edge4,
                     const int32_t* e4row, uint16_t* c5, int grid, bool w16) {
    if (sr.cap <= 0) return;
    const dim3 g(sr.totals ? grid : max(1, min((sr.cap * L5 + 255) / 256, grid))), b(512);
    const half_t* e4 = reinterpret_cast<const half_t*>(maps.e4);
    const half_t* ed = reinterpret_cast<const half_t*>(edge4);
    const half_t* z = reinterpret_cast<const half_t*>(maps.zeros);
    if (w16)
        hipLaunchKernelGGL(conv5_ws_kernel<false>, g, b, 0, st, sr, w, e4, ed, e4row, z, reinterpret_cast<half_t*>(c5));
    else
        hipLaunchKernelGGL(conv5_ws_kernel<true>, g, b, 0, st, sr, w, e4, ed, e4row, z, reinterpret_cast<half_t*>(c5));
}

void launch_tail6(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, const uint16_t* c5,
                  float* logits, float* p, uint8_t* ml, int grid, bool w16) {
    if (sr.cap <= 0) return;
    const dim3 g(sr.totals ? grid : max(1, min((sr.cap + TAIL_SITES - 1) / TAIL_SITES, grid))), b(512);
    const half_t* z = reinterpret_cast<const half_t*>(maps.zeros);
    if (w16)
        hipLaunchKernelGGL(tail6_kernel<true>, g, b, 0, st, sr, w, reinterpret_cast<const half_t*>(c5), z, logits, p, ml);
    else
        hipLaunchKernelGGL(tail6_kernel<false>, g, b, 0, st, sr, w, reinterpret_cast<const half_t*>(c5), z, logits, p, ml);
}

}  // namespace hm
