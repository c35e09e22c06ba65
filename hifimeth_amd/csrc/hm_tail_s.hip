// hm_tail_s.hip -- the SPLIT tail (round 4): conv5 .. softmax of the dense-trunk path as TWO kernels,
//
//   tail_main_kernel : conv5 + conv6 for 8 sites per pass, weights resident (288 registers per wave); conv6's rows leave the chip
//                      (2.7 KB per site, laid out as the planes the head's flat LDS-DMA expects);
//   tail_head_kernel : conv7 + conv8 + fc1 + fc2 + softmax for 16 sites per pass, ALL weights resident (368 registers per wave).
//
// Why.  tail_kernel_r (hm_tail_r.hip) runs eight streaming-conv calls and seven barriers per pass of 8 sites, and
// tools/micro/tconv_ablate.hip (profiles/r04_tconv_ablations.txt) priced what surrounds the MFMAs of a call: ~530 ticks of prologue
// and drain and ~180 ticks per accumulator tile of the LAST group's epilogue, which nothing hides -- a third of a call of 135
// MFMAs, most of a call of 18 (conv8) or 54 (conv7); fc1 + fc2 ran at 18 % of the matrix pipe.  Its LDS is full at 8 sites, so
// the calls cannot get longer inside one kernel.  Split:
//   * main: conv6's output no longer overlays the input planes, so the WHOLE next group is gathered during conv6 and conv5 is ONE
//     call of ~290 MFMAs per wave instead of two (no late rows, no mid-layer drain); a pass is 2 calls and 2 barriers;
//   * head: conv7 / conv8 / fc1 see 16 sites per pass (M = 64 / 32 / 16 instead of 32 / 16 / 8-per-pass), every wave runs PAIRS
//     of n-tiles (half the LDS operand reads of one n-tile per wave), nothing is fetched per pass but the 7.5 KB of activations
//     per site-pair, double-buffered by LDS-DMA behind conv7 and conv8.
// The hand-off costs 2 x 3.7 KB of HBM traffic per site (conv6's 7 rows x 96 channels, hi and lo, in padded planes).
// Arithmetic per accumulator is unchanged (bias, then per k-block w_hi x_hi, w_hi x_lo, w_lo x_hi; fc2 and softmax as in
// tail_kernel_h): the calls are byte-identical to tail_kernel_r's and tail_kernel_h's (tests/test_gpu_parity.py).
//
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98); softmax -> ML byte:
// mod_batch.cpp:46-64.
#include "hm_convt.h"
#ifdef HM_TRUNK_STAMP
#include "hm_stamp.h"
namespace hm { __device__ unsigned long long g_tails_stamp[2][4][16]; }   // [main | head][wave][phase]
extern "C" int hm_debug_tails_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(hm::g_tails_stamp), sizeof(hm::g_tails_stamp)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[2][4][16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(hm::g_tails_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define TSS_DECL unsigned long long tts[12], tacc[12] = {}; unsigned long long tn = 0; const bool tst = blockIdx.x == 0; const unsigned long long tk0 = hm_stamp(), tr0 = __builtin_amdgcn_s_memrealtime()
#define TSS(i) do { if (tst) tts[i] = hm_stamp(); } while (0)
#define TSS_ACC(n) do { if (tst) { for (int i_ = 0; i_ < (n); ++i_) tacc[i_] += tts[i_ + 1] - tts[i_]; ++tn; } } while (0)
#define TSS_OUT(k, n) do { if (tst && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < (n); ++i_) atomicAdd(&g_tails_stamp[k][threadIdx.x >> 6][i_], tacc[i_]); atomicAdd(&g_tails_stamp[k][threadIdx.x >> 6][15], tn); atomicAdd(&g_tails_stamp[k][threadIdx.x >> 6][13], hm_stamp() - tk0); atomicAdd(&g_tails_stamp[k][threadIdx.x >> 6][14], __builtin_amdgcn_s_memrealtime() - tr0); } } while (0)
#else
#define TSS_DECL
#define TSS(i)
#define TSS_ACC(n)
#define TSS_OUT(k, n)
#endif

namespace hm {

namespace {

// ---- geometry shared by the two kernels: conv6's rows in HBM ------------------------------------------------------------------
constexpr int XS_RS96 = 104, XS_RS64 = 72;                 // halves per row: 96 / 64 channels + 16 bytes of pad (conflict-free reads)
constexpr int X6_SS = (7 + 2) * XS_RS96;                    // halves per site and plane: padding row, 7 rows, padding row

struct MGeo {
    static constexpr int S = TAIL_SITES, NW = 4;
    static constexpr int L4 = C4_LEN, L5 = 13, L6 = 7;
    static constexpr int RS96 = XS_RS96;
    static constexpr int IN_ROWS = L4 + 2;
    static constexpr int IN_SS = IN_ROWS * RS96, C5_SS = (L5 + 2) * RS96, C6_SS = X6_SS;
    static constexpr int P0 = S * IN_SS;  // plane of buffer 0: the input rows
    static constexpr int P1 = S * C5_SS;  // plane of buffer 1: conv5's output
    static constexpr int LDS_HALVES = 2 * P0 + 2 * P1;
    // gather: one LDS-DMA = four input rows of one plane
    static constexpr int PLANE_ROWS = S * IN_ROWS, QROWS = 4, NQ = PLANE_ROWS / QROWS, QBYTES = QROWS * RS96 * 2;
    static constexpr int NHOOK = (NQ + NW - 1) / NW;  // quads a wave requests for the first group
    // The gather of a group (90 KB through a path that gathers ~15 B/clk: 6 k cycles) is spread over a WHOLE pass: every wave runs
    // conv5's m-tiles in ascending order, so the rows of m-tiles 0 .. 3 (the EARLY rows: everything below the first row of m-tile 4)
    // are dead at a barrier in the middle of conv5 and the next group's can be fetched from there on (conv5's second half + conv6),
    // the LATE rows from conv6 on (conv6 + the next conv5's first half).  One quad is read by BOTH halves (m-tile 3's last row is
    // m-tile 4's first): it is the first of the late quads -- requested first in conv6 (wave 0's first DMA) -- and the only one of
    // them that the loop top waits for.
    static constexpr int MID_TILE = 4, MID_ROW = (MID_TILE * 16 / L5) * IN_ROWS + 2 * (MID_TILE * 16 % L5);  // first input row m-tile 4 reads
    static constexpr int EQ = MID_ROW / QROWS, LQ = NQ - EQ;                                                  // early / late quads
    static constexpr int NHE = (EQ + NW - 1) / NW, NHL = (LQ + NW - 1) / NW;                                  // hooks per wave
    static constexpr int LATE_VM = 2 * NHL + 2 * 6 ;  // vector-memory operations a wave issues in conv6: late-row DMAs (2 per quad) + its 6 tiles' stores
    static_assert(MID_ROW % QROWS == 0, "the early rows end on a quad boundary");
    static_assert(PLANE_ROWS % QROWS == 0 && RS96 * 2 == 13 * 16, "row-aligned pieces of 13 sixteen-byte units");
};

struct HGeo {
    static constexpr int S = 16, NW = 4;
    static constexpr int L7 = 4, L8 = 2;   // rows of conv7 / conv8 (conv6 hands over 7)
    static constexpr int RS96 = XS_RS96, RS64 = XS_RS64;
    // fc1's fp32 output for fc2: [site][8 parts][32 + 4 floats], and fc2's weights the same way.  A thread of fc2 sums 32 consecutive
    // k (tail_kernel_h's order); with the parts 128 bytes apart every ds_read_b128 of a lane group hit the same banks (8-way
    // conflicts: 2.4 k ticks for 16 sites).  Parts 144 bytes and sites / outputs 1152 bytes apart put a group's 16 addresses in 16
    // different 16-byte bank groups.
    static constexpr int PRS = 36, HRS = 8 * PRS;
    static constexpr int C6_SS = X6_SS, C7_SS = (L7 + 2) * RS64, C8_SS = L8 * RS64;
    static constexpr int INP = S * C6_SS, C7P = S * C7_SS, C8P = S * C8_SS;  // halves per plane
    static constexpr int LDS_HALVES = 4 * INP + 2 * C7P + 2 * C8P;           // two input buffers x (hi, lo), conv7's and conv8's outputs
    static constexpr int PIECES = (INP * 2 + 1023) / 1024;                   // 1 KB LDS-DMA pieces per input plane
    static constexpr int LAST_LANES = (INP * 2 - (PIECES - 1) * 1024) / 16;  // lanes of a plane's last piece
    static constexpr int NHOOK = 2 * PIECES / NW;                            // pieces a wave requests per pass: one per block of conv7 + conv8
    static_assert(2 * PIECES % NW == 0 && NHOOK == 9 + 6, "one piece per (conv7 | conv8) block and wave");
    static_assert(S * HRS * 4 <= INP * 2, "fc1's fp32 output overlays the hi plane of the pass's own input buffer");
    static_assert((INP * 2) % 16 == 0 && LAST_LANES > 0 && LAST_LANES <= 64, "whole 16-byte chunks");
};

// zero the two padding rows (physical rows 0 and LOUT + 1) of S stacked sites, C channels, 16 bytes per store
template <int S, int LOUT, int C, int RS, int SS, int NT>
__device__ __forceinline__ void zero_pad_rows(half_t* hi, half_t* lo, int t) {
    constexpr int PER = C / 8, N = S * 2 * 2 * PER;  // sites x {top, bottom} x {hi, lo} x 16-byte chunks
    for (int i = t; i < N; i += NT) {
        const int ch = i % PER, rest = i / PER, plane = rest & 1, which = (rest >> 1) & 1, site = rest >> 2;
        half_t* o = (plane ? lo : hi) + site * SS + (which ? (LOUT + 1) : 0) * RS + ch * 8;
        *reinterpret_cast<uint4*>(o) = make_uint4(0u, 0u, 0u, 0u);
    }
}

}  // namespace

// =====================================================================================================================================
// main: conv5 + conv6
// =====================================================================================================================================
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tail_main_kernel(SiteRange sr, CtxWeights W, const half_t* __restrict__ e4, const half_t* __restrict__ edge4,
                      const int32_t* __restrict__ e4row, const half_t* __restrict__ zeros, half_t* __restrict__ x6_hi,
                      half_t* __restrict__ x6_lo) {
    using T = MGeo;
    constexpr int S = T::S, NW = T::NW;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    // a workgroup takes a CONTIGUOUS range of 8-site groups: neighbouring sites read the same E4 rows
    const int n_groups = (n_sites + S - 1) / S, base_n = n_groups / (int)gridDim.x, rem_n = n_groups - base_n * (int)gridDim.x;
    const int g_begin = (int)blockIdx.x * base_n + min((int)blockIdx.x, rem_n), g_end = g_begin + base_n + ((int)blockIdx.x < rem_n);
    if (g_begin >= g_end) return;  // (whole workgroup: no barrier is left behind)

    __shared__ __attribute__((aligned(16))) half_t smem[T::LDS_HALVES];
    __shared__ __attribute__((aligned(16))) float bias_l[96 + 96];        // conv5, conv6 biases
    __shared__ unsigned long long rowsrc[T::PLANE_ROWS];                   // source address of every input row of the group being gathered
    half_t* h0 = smem;
    half_t* l0 = smem + T::P0;
    half_t* h1 = smem + 2 * T::P0;
    half_t* l1 = smem + 2 * T::P0 + T::P1;
    const float* b5 = bias_l;
    const float* b6 = bias_l + 96;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 192; i += NW * 64) bias_l[i] = i < 96 ? W.bias[4][i] : W.bias[5][i - 96];
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };

    // ---- resident weights: n-tiles (a, b) of conv5 and conv6 -----------------------------------------------------------------------
    // n-tile b is shared by an even and an odd wave, which run the pair (a, b) on complementary m-tiles and a alone on the others
    const int nta = wave == 0 ? 0 : wave == 1 ? 2 : wave == 2 ? 3 : 5, ntb = wave < 2 ? 1 : 4;
    const int nt56[2] = {nta, ntb}, col56[2] = {16 * nta, 16 * ntb};
    const bool odd = wave & 1;
    TW<9, 2> W5, W6;
    tw_load(wf(4), nt56, lane, W5);
    tw_load(wf(5), nt56, lane, W6);

    using C96 = TCfg<96, 3, T::RS96>;
    using R5 = TRows<T::L5, T::IN_SS, S * T::L5>;
    using R6 = TRows<T::L6, T::C5_SS, S * T::L6>;
    static_assert(S * T::L5 == 104 && S * T::L6 == 56, "m-tiles: 7 (6.5), 4 (3.5)");

    // ---- gather (as in tail_kernel_r): lane l < 52 of a piece = row l / 13 of the quad, 16-byte chunk l % 13 ---------------------------
    int tl = tid;
    const unsigned long long lanes52 = 0x000FFFFFFFFFFFFFull;
    const uint32_t lds_h0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)h0;
    const uint32_t lds_l0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)l0;
    auto dma_quad = [&](const int q) __attribute__((always_inline)) {  // q wave-uniform: rows 4q .. 4q + 3 of both input planes
        const int ln = tl & 63, q_row = min(ln / 13, T::QROWS - 1), q_chunk16 = (ln % 13) * 16;
        const unsigned long long src = rowsrc[T::QROWS * q + q_row] + (unsigned)q_chunk16;
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(lds_h0 + (uint32_t)(T::QBYTES * q));
        // (an LDS-DMA's immediate offset moves BOTH addresses, the global one and the LDS one: the lo plane's M0 takes it back)
        const uint32_t d1 = __builtin_amdgcn_readfirstlane(lds_l0 + (uint32_t)(T::QBYTES * q) - 192u);
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
            "global_load_lds_dwordx4 %5, off offset:192\n\t"
            "s_mov_b32 m0, %1\n\t"
            "s_mov_b64 exec, %0"
            : "=&s"(sv), "=&s"(km)
            : "s"(lanes52), "s"(d0), "s"(d1), "v"(src));
    };
    // E4 map row (view position off - 215) of this thread's site of group G: requested a pass before the table needs it
    auto load_e4r = [&](const int G) __attribute__((always_inline)) { return e4row[min(G * S + min(tl / T::IN_ROWS, S - 1), n_sites - 1)]; };
    // rows [r0, r1) of the table for group G
    auto build_table = [&](const int G, const int32_t e4r, const int r0, const int r1) __attribute__((always_inline)) {
        if (tl >= r0 && tl < r1) {
            const int t_site = tl / T::IN_ROWS, t_prow = tl - T::IN_ROWS * t_site;
            const int gs = G * S + t_site, pos = t_prow - 1;
            const half_t* a = zeros;  // padding rows, sites past the end
            if (gs < n_sites && pos >= 0 && pos < C4_LEN) {
                if (pos == 0) a = edge4 + (size_t)gs * (4 * C4_CH);
                else if (pos == C4_LEN - 1) a = edge4 + (size_t)gs * (4 * C4_CH) + 2 * C4_CH;
                else a = e4 + ((long long)e4r + 16 * pos) * (2 * C4_CH);
            }
            rowsrc[tl] = (unsigned long long)(uintptr_t)a;
        }
    };
    // conv5: in the middle (two blocks before the first group over m-tiles >= 4 starts reading) this group's late rows must be in
    // LDS and everybody must be through with the early rows; from then on the NEXT group's early rows are requested
    constexpr int C5_MID = 2 * 9 - 2;
    auto hook5 = [&](auto c_) __attribute__((always_inline)) {
        constexpr int C = decltype(c_)::value;
        if constexpr (C == C5_MID) {
            vm_drain();
            lds_barrier();
        }
        if constexpr (C > C5_MID && C - C5_MID - 1 < T::NHE) dma_quad(min(wave + NW * (C - C5_MID - 1), T::EQ - 1));
    };
    // conv6: the next group's late rows
    auto hook6 = [&](auto c_) __attribute__((always_inline)) {
        constexpr int C = decltype(c_)::value;
        if constexpr (C < T::NHL) dma_quad(T::EQ + min(wave + NW * C, T::LQ - 1));
    };

    // first group of this workgroup: everything at once, by everybody; conv5's padding rows are zeroed once (its epilogue never
    // writes them)
    zero_pad_rows<S, T::L5, 96, T::RS96, T::C5_SS, NW * 64>(h1, l1, tid);
    int32_t e4r_next = load_e4r(g_begin);
    build_table(g_begin, e4r_next, 0, T::PLANE_ROWS);
    e4r_next = load_e4r(g_begin + 1);
    lds_barrier();
#pragma unroll
    for (int k = 0; k < T::NHOOK; ++k) dma_quad(min(wave + NW * k, T::NQ - 1));
    vm_drain();
    lds_barrier();  // (the table is rewritten below)

    TSS_DECL;
    for (int g = g_begin; g < g_end; ++g) {
        tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        TSS(0);
        // the early rows of group g (and the quad both halves read: conv6's first DMA) have landed: everything this wave issued but the
        // rest of conv6's own operations -- the other late rows' DMAs, its stores -- is done
        e2_vmwait_n<T::LATE_VM - 2>();
        build_table(g + 1, e4r_next, 0, T::MID_ROW);  // the next group's early rows: requested from the middle of conv5 on
        lds_barrier();  // the early rows of group g are in LDS; conv6 of the previous pass is through with buffer 1
        TSS(1);
        // ---- conv5: ONE call (buffer 0 -> buffer 1), m-tiles in ascending order on every wave --------------------------------------------
        {
            const EpiStack<T::L5, T::RS96, T::C5_SS> e5{h1, l1};
            // both halves (m-tiles 0 .. 3 | 4 .. 6) are dealt evenly: the barrier in the middle must not wait for anybody --
            // even waves: pairs m0, m1; a on m2, m3 | pair m4 + a on m5; a on m6      (6 + 4 tiles)
            // odd waves : a on m0, m1; pairs m2, m3 | pair m5 + a on m4; pair m6      (6 + 5 tiles)
            if (odd) TConv<C96, R5, 8, 1, TG<0, 0, 0, 2>, TG<2, 2, 0, 0>, TG<5, 1, 4, 1>, TG<6, 1, 0, 0>>::run<1>(h0, l0, W5, b5, col56, e5, hook5);
            else TConv<C96, R5, 8, 1, TG<0, 2, 0, 0>, TG<0, 0, 2, 2>, TG<4, 1, 5, 1>, TG<0, 0, 6, 1>>::run<1>(h0, l0, W5, b5, col56, e5, hook5);
        }
        TSS(2);
        build_table(g + 1, e4r_next, T::MID_ROW, T::PLANE_ROWS);  // the next group's late rows (this group's were requested a pass ago)
        e4r_next = load_e4r(g + 2);
        lds_barrier();
        TSS(3);
        // ---- conv6 (buffer 1 -> HBM); the next group's late rows are requested meanwhile ---------------------------------------------------
        {
            const EpiGStack<T::L6, T::RS96, T::C6_SS> e6{x6_hi + (size_t)g * (S * T::C6_SS), x6_lo + (size_t)g * (S * T::C6_SS)};
            if (odd) TConv<C96, R6, 8, 1, TG<2, 2, 0, 0>, TG<0, 0, 0, 2>>::run<1>(h1, l1, W6, b6, col56, e6, hook6);   // pair m2, m3; a on m0, m1
            else TConv<C96, R6, 8, 1, TG<0, 2, 0, 0>, TG<0, 0, 2, 2>>::run<1>(h1, l1, W6, b6, col56, e6, hook6);      // pair m0, m1; a on m2, m3
        }
        TSS(4);
        TSS(5);
        TSS_ACC(5);
    }
    TSS_OUT(0, 5);
    vm_drain();  // nothing of this workgroup's gather is in flight when its LDS is handed on
}

// =====================================================================================================================================
// head: conv7 + conv8 + fc1 + fc2 + softmax
// =====================================================================================================================================
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tail_head_kernel(SiteRange sr, CtxWeights W, float* __restrict__ logits, float* __restrict__ prob, uint8_t* __restrict__ ml,
                      const half_t* __restrict__ x6_hi, const half_t* __restrict__ x6_lo) {
    using T = HGeo;
    constexpr int S = T::S, NW = T::NW;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    const int n_groups = (n_sites + S - 1) / S, base_n = n_groups / (int)gridDim.x, rem_n = n_groups - base_n * (int)gridDim.x;
    const int g_begin = (int)blockIdx.x * base_n + min((int)blockIdx.x, rem_n), g_end = g_begin + base_n + ((int)blockIdx.x < rem_n);
    if (g_begin >= g_end) return;

    __shared__ __attribute__((aligned(16))) half_t smem[T::LDS_HALVES];
    __shared__ __attribute__((aligned(16))) float fc2w[2 * T::HRS + 4];  // fc2 weights [output][part][36] + bias
    __shared__ __attribute__((aligned(16))) float bias_l[64 + 64];    // conv7, conv8 biases
    half_t* in = smem;                        // [buffer][plane][INP]
    half_t* c7h = smem + 4 * T::INP;
    half_t* c7l = c7h + T::C7P;
    half_t* c8h = c7l + T::C7P;
    half_t* c8l = c8h + T::C8P;
    const float* b7 = bias_l;
    const float* b8 = bias_l + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 256 + 2; i += NW * 64) {
        if (i < 512) fc2w[(i >> 8) * T::HRS + ((i & 255) >> 5) * T::PRS + (i & 31)] = W.fc2_w[i];
        else fc2w[2 * T::HRS + i - 512] = W.fc2_b[i - 512];
    }
    for (int i = tid; i < 128; i += NW * 64) bias_l[i] = i < 64 ? W.bias[6][i] : W.bias[7][i - 64];
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };

    // ---- resident weights: conv7 and conv8 as PAIRS of n-tiles (waves 0, 1: n-tiles 0, 1; waves 2, 3: n-tiles 2, 3 -- the two waves
    // of a pair take complementary m-tiles), fc1's n-tiles 4 wave .. 4 wave + 3 --------------------------------------------------------
    const int np = wave >> 1;
    const bool odd = wave & 1;
    const int nt78[2] = {2 * np, 2 * np + 1}, col78[2] = {32 * np, 32 * np + 16};
    const int ntf[4] = {4 * wave, 4 * wave + 1, 4 * wave + 2, 4 * wave + 3}, colf[4] = {64 * wave, 64 * wave + 16, 64 * wave + 32, 64 * wave + 48};
    const int colp[4] = {2 * wave * T::PRS, 2 * wave * T::PRS + 16, (2 * wave + 1) * T::PRS, (2 * wave + 1) * T::PRS + 16};  // the same in fc1's padded output rows
    TW<9, 2> W7;
    TW<6, 2> W8;
    TW<4, 4> WF;
    tw_load(wf(6), nt78, lane, W7);
    tw_load(wf(7), nt78, lane, W8);
    tw_load(wf(8), ntf, lane, WF);
    float4 bzf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bzf[j] = *reinterpret_cast<const float4*>(W.bias[8] + colf[j] + 4 * (lane >> 4));

    using C96 = TCfg<96, 3, T::RS96>;
    using C64 = TCfg<64, 3, T::RS64>;
    using CF = TCfg<64, 2, T::RS64>;
    using R7 = TRows<T::L7, T::C6_SS, S * T::L7>;
    using R8 = TRows<T::L8, T::C7_SS, S * T::L8>;
    using RF = TRows<1, T::C8_SS, S>;
    static_assert(S * T::L7 == 64 && S * T::L8 == 32, "m-tiles: 4, 2, 1");

    // ---- input: conv6's planes of 16 sites are contiguous in HBM in the LDS layout: flat 1 KB pieces --------------------------------
    int tl = tid;
    const uint32_t lds_in = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)in;
    auto dma_piece = [&](const int G, const int buf, const int k) __attribute__((always_inline)) {  // k wave-uniform: piece k % PIECES of plane k / PIECES
        const int plane = k >= T::PIECES, p = k - plane * T::PIECES;
        const half_t* src = (plane ? x6_lo : x6_hi) + (size_t)G * T::INP + p * 512 + (tl & 63) * 8;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_in + (uint32_t)(((buf * 2 + plane) * T::INP + p * 512) * 2));
        const unsigned long long mask = p == T::PIECES - 1 ? ((1ull << T::LAST_LANES) - 1ull) : ~0ull;
        unsigned long long sv;
        uint32_t km;
        asm volatile(
            "s_mov_b64 %0, exec\n\t"
            "s_mov_b32 %1, m0\n\t"
            "s_mov_b64 exec, %2\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %4, off\n\t"
            "s_mov_b32 m0, %1\n\t"
            "s_mov_b64 exec, %0"
            : "=&s"(sv), "=&s"(km)
            : "s"(mask), "s"(dst), "v"(src));
    };

    // conv7's padding rows are zeroed once (its epilogue never writes them; nothing overlays its planes)
    zero_pad_rows<S, T::L7, 64, T::RS64, T::C7_SS, NW * 64>(c7h, c7l, tid);
#pragma unroll
    for (int k = 0; k < T::NHOOK; ++k) dma_piece(g_begin, 0, wave + NW * k);
    vm_drain();

    int buf = 0;
    int dst_next;
    {
        const int idx = min(g_begin * S + (tid >> 4), n_sites - 1);
        dst_next = sites ? sites[idx].uidx : idx;
    }
    TSS_DECL;
    for (int g = g_begin; g < g_end; ++g, buf ^= 1) {
        tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const half_t* ih = in + (buf * 2) * T::INP;
        const half_t* il = ih + T::INP;
        float* hfc = reinterpret_cast<float*>(in + (buf * 2) * T::INP);  // fc1's output: over this pass's input once conv7 has read it
        // where this thread's site's results go (16 lanes per site in fc2): fetched a pass ahead -- behind the drain at the end of the
        // pass the compiler knows the load has completed; fetched in the pass itself, its wait in front of fc2 would be a vmcnt(0)
        // that also waits for every piece of the next group's input (the compiler does not see the inline-asm DMAs)
        const int bsite = tl >> 4, gsite = g * S + bsite;
        const int dst = dst_next;
        {
            const int idx = min(gsite + S, n_sites - 1);
            dst_next = sites ? sites[idx].uidx : idx;
        }
        TSS(0);
        lds_barrier();  // the rows of group g are in LDS (drained at the end of the previous pass); the previous pass's fc2 is through
        TSS(1);
        // the next group's planes go into the other buffer, one piece per block of conv7 and conv8
        auto hook7 = [&](auto c_) __attribute__((always_inline)) { dma_piece(g + 1, buf ^ 1, wave + NW * decltype(c_)::value); };
        auto hook8 = [&](auto c_) __attribute__((always_inline)) { dma_piece(g + 1, buf ^ 1, wave + NW * (9 + decltype(c_)::value)); };
        // ---- conv7 (input -> c7): pair on two m-tiles ---------------------------------------------------------------------------------
        {
            const EpiStack<T::L7, T::RS64, T::C7_SS> e7{c7h, c7l};
            if (odd) TConv<C96, R7, 8, 1, TG<2, 2, 0, 0>>::run(ih, il, W7, b7, col78, e7, hook7);
            else TConv<C96, R7, 8, 1, TG<0, 2, 0, 0>>::run(ih, il, W7, b7, col78, e7, hook7);
        }
        TSS(2);
        lds_barrier();
        TSS(3);
        // ---- conv8 (c7 -> c8): pair on one m-tile ----------------------------------------------------------------------------------------
        {
            const EpiRows<T::RS64> e8{c8h, c8l};
            if (odd) TConv<C64, R8, 8, 1, TG<1, 1, 0, 0>>::run(c7h, c7l, W8, b8, col78, e8, hook8);
            else TConv<C64, R8, 8, 1, TG<0, 1, 0, 0>>::run(c7h, c7l, W8, b8, col78, e8, hook8);
        }
        TSS(4);
        lds_barrier();
        TSS(5);
        // ---- fc1 = a 2-tap "conv" over conv8's two positions (k order l * 64 + c; hm_weights.cpp): four n-tiles on the one m-tile -------
        TConv<CF, RF, 8, 1, TG<0, 1, 0, 0, 4>>::run(c8h, c8l, WF, [&](int j) __attribute__((always_inline)) { return bzf[j]; }, colp, EpiFc1R<T::HRS>{hfc});
        TSS(6);
        lds_barrier();
        TSS(7);
        // ---- fc2 + softmax (mod_batch.cpp:46-64) in fp32: 16 lanes per site = 2 outputs x 8 partial sums ---------------------------------------
        {
            const int o = (tl >> 3) & 1, part = tl & 7;
            const float* h = hfc + bsite * T::HRS + part * T::PRS;
            const float* w2 = fc2w + o * T::HRS + part * T::PRS;
            float sum = 0.f;
            // (the same 32 sequential fmaf as tail_kernel_h's, operands fetched 16 bytes at a time)
            float4 hv[8], wv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                hv[k] = *reinterpret_cast<const float4*>(h + 4 * k);
                wv[k] = *reinterpret_cast<const float4*>(w2 + 4 * k);
            }
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
            sum += fc2w[2 * T::HRS + o];
            const float other = __shfl_xor(sum, 8, 64);
            if ((tl & 15) == 0 && gsite < n_sites) {
                const float v0 = sum, v1 = other;
                const float mx = fmaxf(v0, v1);
                const float e0 = expf(v0 - mx), e1 = expf(v1 - mx);
                const float p1 = e1 / (e0 + e1);
                int q = (int)(255 * p1);
                q = q > 255 ? 255 : q;
                logits[2 * (size_t)dst] = v0;
                logits[2 * (size_t)dst + 1] = v1;
                prob[dst] = p1;
                ml[dst] = (uint8_t)q;
            }
        }
        TSS(8);
        vm_drain();  // the next group's planes have arrived: nothing of the input stream is in flight at the loop top
        TSS(9);
        TSS_ACC(9);
    }
    TSS_OUT(1, 9);
}

// conv6's rows of one launch: two planes of (sites rounded up to whole passes of both kernels, + two head passes of slack for the
// head's look-ahead) x X6_SS halves.  The padding rows of a plane are never written: the buffer is zeroed once, and every launch
// must use the SAME plane stride (the engine passes the stride of its slice size, not of the launch's own site count).
size_t tail_split_x6_plane_halves(int64_t sites) { return (size_t)((sites + 15) / 16 * 16 + 2 * HGeo::S) * X6_SS; }
size_t tail_split_x6_bytes(int64_t sites) { return (size_t)2 * tail_split_x6_plane_halves(sites) * sizeof(uint16_t); }

void launch_tail_split(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, const uint16_t* edge4,
                       const int32_t* e4row, uint16_t* x6, size_t x6_plane_halves, float* logits, float* p, uint8_t* ml, int grid) {
    if (sr.cap <= 0) return;
    half_t* x6_hi = reinterpret_cast<half_t*>(x6);
    half_t* x6_lo = x6_hi + x6_plane_halves;
    const dim3 gm(sr.totals ? grid : max(1, min((sr.cap + MGeo::S - 1) / MGeo::S, grid)));
    hipLaunchKernelGGL(tail_main_kernel, gm, dim3(256), 0, st, sr, w, reinterpret_cast<const half_t*>(maps.e4),
                       reinterpret_cast<const half_t*>(edge4), e4row, reinterpret_cast<const half_t*>(maps.zeros), x6_hi, x6_lo);
    const dim3 gh(sr.totals ? grid : max(1, min((sr.cap + HGeo::S - 1) / HGeo::S, grid)));
    hipLaunchKernelGGL(tail_head_kernel, gh, dim3(256), 0, st, sr, w, logits, p, ml, x6_hi, x6_lo);
}

}  // namespace hm
