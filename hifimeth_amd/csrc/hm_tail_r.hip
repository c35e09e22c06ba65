// hm_tail_r.hip -- the RESIDENT tail kernel: conv5 .. conv8, fc1, fc2, softmax of the dense-trunk path with the weights of
// conv5, conv6 and conv7 held in registers for the whole launch (hm_convt.h: four waves, one per SIMD, 512 registers each).
//
// Same arithmetic, same LDS layout and the same results, bit for bit, as tail_kernel_h<0, GATHER> (hm_front_h.hip), which
// stays selectable (engine option "tail_impl" = 0) and is what the byte-identity test compares with.  What changed is the
// traffic: that kernel fetches 375 KB of weights from L2 per pass of 8 sites through the CU's 64 B/clk vector-memory path
// and starts every layer behind an L2 round trip; here a pass loads 49 KB (conv8, requested a layer ahead; fc1 once per
// four passes) and the vector-memory path carries the gather of the NEXT group's conv4 rows instead.
//
//   gather : a site's 25 conv4 rows = 23 rows of the dense E4 map (e4row + 16 s) + its two window-edge rows (edge kernel),
//            already split [hi 96 | lo 96].  Row-aligned LDS-DMA: one wave instruction (global_load_lds_dwordx4, lanes 0..51)
//            brings FOUR 208-byte LDS rows; a lane's (row in the quad, 16-byte chunk) never changes, the rows' source
//            addresses come from a 216-entry table in LDS that 216 threads rebuild once per pass -- a piece costs one
//            ds_read_b64 and one 64-bit add, for both planes (the lo plane is the same address + 192 bytes).
//   pass   : conv5 in two parts -- the tiles over the rows that arrived while conv6 / conv7 ran first, then (behind a wait
//            and a barrier) the tiles over the rows that could only be fetched once conv7 had read conv6's output, which
//            lives in the same LDS (the first 72 rows of each input plane) -- conv6 (+ the gather of the next group's other
//            144 rows), conv7, conv8, and once per four passes fc1 + fc2 + softmax for 32 sites.
//   sync   : barriers fence LDS only (__builtin_amdgcn_fence(..., "workgroup", "local")): a __syncthreads() would wait for
//            every vector-memory operation in flight, i.e. for the gather it is supposed to overlap.
//
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98); softmax -> ML byte:
// mod_batch.cpp:46-64.
#include "hm_convt.h"
#ifdef HM_TRUNK_STAMP
#include "hm_stamp.h"
namespace hm { __device__ unsigned long long g_tailr_stamp[4][16]; }
extern "C" int hm_debug_tailr_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(hm::g_tailr_stamp), sizeof(hm::g_tailr_stamp)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[4][16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(hm::g_tailr_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

namespace hm {

namespace {

struct RGeo {
    static constexpr int S = TAIL_SITES, NW = 4;
    static constexpr int L4 = C4_LEN, L5 = 13, L6 = 7, L7 = 4, L8 = 2;
    static constexpr int RS96 = 104, RS64 = 72, HRS = 260;  // halves, halves, floats
    static constexpr int IN_ROWS = L4 + 2;
    static constexpr int IN_SS = IN_ROWS * RS96, C5_SS = (L5 + 2) * RS96, C6_SS = (L6 + 2) * RS96, C7_SS = (L7 + 2) * RS64;
    static constexpr int P0 = S * IN_SS;  // plane of buffer 0: the input rows; conv6's output in its first rows
    static constexpr int P1 = S * C5_SS;  // plane of buffer 1: conv5's output; conv7's output; fc1's output as floats
    static constexpr int FCB = 4, RING_SS = L8 * RS64, RING = FCB * S * RING_SS;
    static constexpr int LDS_HALVES = 2 * P0 + 2 * P1 + 2 * RING;
    static_assert(S * C7_SS <= P1 && FCB * S * HRS * 2 <= 2 * P1, "buffer 1 plan");
    // gather: one LDS-DMA = four input rows of one plane
    static constexpr int PLANE_ROWS = S * IN_ROWS, QROWS = 4, NQ = PLANE_ROWS / QROWS, QBYTES = QROWS * RS96 * 2;
    static constexpr int LOW_ROWS = S * C6_SS / RS96, LOWQ = LOW_ROWS / QROWS;  // rows / quads that conv6's output overlays
    static_assert(PLANE_ROWS % QROWS == 0 && LOW_ROWS % QROWS == 0 && RS96 * 2 == 13 * 16, "row-aligned pieces of 13 sixteen-byte units");
    // conv5 tiles 3.. read input rows >= 99 only: they may run while rows [0, LOW_ROWS) are still in flight
    static_assert((3 * 16 / L5) * IN_ROWS + 2 * (3 * 16 % L5) >= LOW_ROWS, "conv5 tiles 3..6 stay clear of the late rows");
};

// zero the two padding rows (physical rows 0 and LOUT + 1) of S stacked sites, C channels, 16 bytes per store
template <int LOUT, int C, int RS, int SS>
__device__ __forceinline__ void zero_pads(half_t* hi, half_t* lo, int t) {
    constexpr int PER = C / 8, N = RGeo::S * 2 * 2 * PER;  // sites x {top, bottom} x {hi, lo} x 16-byte chunks
    for (int i = t; i < N; i += RGeo::NW * 64) {
        const int ch = i % PER, rest = i / PER, plane = rest & 1, which = (rest >> 1) & 1, site = rest >> 2;
        half_t* o = (plane ? lo : hi) + site * SS + (which ? (LOUT + 1) : 0) * RS + ch * 8;
        *reinterpret_cast<uint4*>(o) = make_uint4(0u, 0u, 0u, 0u);
    }
}

}  // namespace

// W16 (engine option precision = 2): conv8 and fc1 with plain fp16 weights (their w_lo x_hi product and lo-plane fetches dropped)
template <bool W16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tail_kernel_r(SiteRange sr, CtxWeights W, float* __restrict__ logits, float* __restrict__ prob, uint8_t* __restrict__ ml,
                   const half_t* __restrict__ e4, const half_t* __restrict__ edge4, const int32_t* __restrict__ e4row,
                   const half_t* __restrict__ zeros) {
    using T = RGeo;
    constexpr int S = T::S, NW = T::NW, FCB = T::FCB;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    // a workgroup takes a CONTIGUOUS range of 8-site groups: neighbouring sites read the same E4 rows
    const int n_groups = (n_sites + S - 1) / S, base_n = n_groups / (int)gridDim.x, rem_n = n_groups - base_n * (int)gridDim.x;
    const int g_begin = (int)blockIdx.x * base_n + min((int)blockIdx.x, rem_n), g_end = g_begin + base_n + ((int)blockIdx.x < rem_n);
    if (g_begin >= g_end) return;  // (whole workgroup: no barrier is left behind)

    __shared__ __attribute__((aligned(16))) half_t smem[T::LDS_HALVES];
    __shared__ __attribute__((aligned(16))) float fc2w[2 * 256 + 4];       // fc2 weights + bias
    __shared__ __attribute__((aligned(16))) float bias_l[96 + 96 + 64 + 64];  // conv5 .. conv8 biases
    __shared__ unsigned long long rowsrc[T::PLANE_ROWS];                   // source address of every input row of the group being gathered
    half_t* h0 = smem;
    half_t* l0 = smem + T::P0;
    half_t* h1 = smem + 2 * T::P0;
    half_t* l1 = smem + 2 * T::P0 + T::P1;
    float* hfc = reinterpret_cast<float*>(h1);
    half_t* r_hi = smem + 2 * T::P0 + 2 * T::P1;
    half_t* r_lo = r_hi + T::RING;
    const float* b5 = bias_l;
    const float* b6 = bias_l + 96;
    const float* b7 = bias_l + 192;
    const float* b8 = bias_l + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 256 + 2; i += NW * 64) fc2w[i] = i < 512 ? W.fc2_w[i] : W.fc2_b[i - 512];
    for (int i = tid; i < 320; i += NW * 64) bias_l[i] = i < 96 ? W.bias[4][i] : i < 192 ? W.bias[5][i - 96] : i < 256 ? W.bias[6][i - 192] : W.bias[7][i - 256];
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };

    // ---- resident weights: n-tiles (a, b) of conv5 and conv6, n-tile `wave` of conv7; conv8's arrive per pass -------------
    // waves 0 / 2 (even): pair (a, b) on the LOW m-tiles, a alone on the high ones; waves 1 / 3 (odd): the other way round
    const int nta = wave == 0 ? 0 : wave == 1 ? 2 : wave == 2 ? 3 : 5, ntb = wave < 2 ? 1 : 4;
    const int nt56[2] = {nta, ntb}, col56[2] = {16 * nta, 16 * ntb};
    const int nt78[1] = {wave}, col78[1] = {16 * wave};
    const bool odd = wave & 1;
    TW<9, 2> W5, W6;
    TW<9, 1> W7;
    TW<6, 1> W8;
    tw_load(wf(4), nt56, lane, W5);
    tw_load(wf(5), nt56, lane, W6);
    tw_load(wf(6), nt78, lane, W7);

    using C96 = TCfg<96, 3, T::RS96>;
    using C64 = TCfg<64, 3, T::RS64, !W16>;
    using R5 = TRows<T::L5, T::IN_SS, S * T::L5>;
    using R6 = TRows<T::L6, T::C5_SS, S * T::L6>;
    using R7 = TRows<T::L7, T::C6_SS, S * T::L7>;
    using R8 = TRows<T::L8, T::C7_SS, S * T::L8>;
    static_assert(S * T::L5 == 104 && S * T::L6 == 56 && S * T::L7 == 32 && S * T::L8 == 16, "m-tiles: 7 (6.5), 4 (3.5), 2, 1");

    // ---- gather ---------------------------------------------------------------------------------------------------------
    // lane l < 52 of a piece: row l / 13 of the quad, 16-byte chunk l % 13 (chunk 12 is the row's pad: it reads the 16 bytes
    // behind the 192 it needs -- the lo half, or the next row; never used)
    // (every per-lane constant below is re-derived from an opaque copy of the thread index inside the pass loop: hoisted out
    //  of it they would sit in registers beside 360 resident weight registers for the whole launch, and spill)
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
    auto build_table = [&](const int G, const int32_t e4r) __attribute__((always_inline)) {
        if (tl < T::PLANE_ROWS) {
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
    // conv5, first part: the late rows of THIS group (quads 0 .. LOWQ-1 dealt round-robin; a wave's spare slot repeats the last)
    auto hook_low = [&](auto c_) __attribute__((always_inline)) {
        constexpr int C = decltype(c_)::value;
        if constexpr (C < (T::LOWQ + NW - 1) / NW) dma_quad(min(wave + NW * C, T::LOWQ - 1));
    };
    // conv6: the other rows of the NEXT group
    auto hook_high = [&](auto c_) __attribute__((always_inline)) {
        constexpr int C = decltype(c_)::value;
        if constexpr (C < (T::NQ - T::LOWQ) / NW) dma_quad(T::LOWQ + wave + NW * C);
    };
    static_assert((T::NQ - T::LOWQ) % NW == 0, "the early quads deal out evenly");

    // first group of this workgroup: everything at once, by everybody
    int32_t e4r_next = load_e4r(g_begin);
    build_table(g_begin, e4r_next);
    e4r_next = load_e4r(g_begin + 1);
    lds_barrier();
#pragma unroll
    for (int k = 0; k < (T::NQ + NW - 1) / NW; ++k) dma_quad(min(wave + NW * k, T::NQ - 1));
    vm_drain();

    int slot = 0, g_first = g_begin;  // groups g_first, g_first + 1, ... wait in ring slots 0 .. slot-1
#ifdef HM_TRUNK_STAMP
    unsigned long long tts[12], tacc[12] = {};
    unsigned long long tn = 0;
    const bool tst = blockIdx.x == 0;
    const unsigned long long tk0 = hm_stamp(), tr0 = __builtin_amdgcn_s_memrealtime();
#define TTS(i) do { if (tst) tts[i] = hm_stamp(); } while (0)
#else
#define TTS(i)
#endif
    for (int g = g_begin; g < g_end; ++g) {
        const bool more = g + 1 < g_end;
        tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        lds_barrier();  // the early rows of group g are in LDS (drained before the previous conv8); conv8 / fc of the previous pass are done
        TTS(0);
        // ---- conv5, first part: m-tiles >= 3 (input rows >= 99); the late rows of this group are fetched meanwhile ----------------
        // (a group = up to two m-tiles, up to four accumulators: the activation ring and the accumulators stay at 64 registers)
        {
            const EpiStack<T::L5, T::RS96, T::C5_SS> e5{h1, l1};
            if (odd) TConv<C96, R5, 8, 1, TG<5, 2, 0, 0>, TG<0, 0, 3, 1>>::run<1>(h0, l0, W5, b5, col56, e5, hook_low);   // pair m5, m6; a on m3
            else TConv<C96, R5, 8, 1, TG<3, 1, 4, 1>, TG<0, 0, 5, 2>>::run<1>(h0, l0, W5, b5, col56, e5, hook_low);      // pair m3 + a on m4; a on m5, m6
        }
        TTS(1);
        vm_drain();     // this wave's late rows have landed ...
        lds_barrier();  // ... and everybody's
        TTS(2);
        build_table(g + 1, e4r_next);  // (the table of group g was last read by the hooks above)
        e4r_next = load_e4r(g + 2);
        // ---- conv5, second part ----------------------------------------------------------------------------------------------
        {
            const EpiStack<T::L5, T::RS96, T::C5_SS> e5{h1, l1};
            if (odd) TConv<C96, R5, 8, 1, TG<4, 1, 0, 1>, TG<0, 0, 1, 2>>::run(h0, l0, W5, b5, col56, e5);   // pair m4 + a on m0; a on m1, m2
            else TConv<C96, R5, 8, 1, TG<0, 2, 0, 0>, TG<2, 1, 0, 0>>::run(h0, l0, W5, b5, col56, e5);      // pair m0, m1; pair m2
        }
        zero_pads<T::L5, 96, T::RS96, T::C5_SS>(h1, l1, tl);
        TTS(3);
        lds_barrier();
        TTS(4);
        // ---- conv6 (buffer 1 -> the first rows of buffer 0); the next group's early rows are fetched meanwhile -------------------
        {
            const EpiStack<T::L6, T::RS96, T::C6_SS> e6{h0, l0};
            if (odd) TConv<C96, R6, 8, 1, TG<2, 2, 0, 0>, TG<0, 0, 0, 2>>::run<1>(h1, l1, W6, b6, col56, e6, hook_high);   // pair m2, m3; a on m0, m1
            else TConv<C96, R6, 8, 1, TG<0, 2, 0, 0>, TG<0, 0, 2, 2>>::run<1>(h1, l1, W6, b6, col56, e6, hook_high);      // pair m0, m1; a on m2, m3
        }
        zero_pads<T::L6, 96, T::RS96, T::C6_SS>(h0, l0, tl);
        TTS(5);
        lds_barrier();
        TTS(6);
        // ---- conv7 (buffer 0 -> buffer 1) ------------------------------------------------------------------------------------
        tw_load<!W16>(wf(7), nt78, lane, W8);  // conv8's weights: requested a layer ahead (earlier, conv6's working set spills)
        TConv<C96, R7, 8, 1, TG<0, 0, 0, 2>>::run(h0, l0, W7, b7, col78, EpiStack<T::L7, T::RS64, T::C7_SS>{h1, l1});
        zero_pads<T::L7, 64, T::RS64, T::C7_SS>(h1, l1, tl);
        TTS(7);
        vm_drain();  // the next group's early rows (and conv8's weights) have arrived: nothing of the gather is in flight at the loop top
        lds_barrier();
        TTS(8);
        // ---- conv8 (buffer 1 -> ring slot) --------------------------------------------------------------------------------------
        TConv<C64, R8, 8, 1, TG<0, 0, 0, 1>>::run(h1, l1, W8, b8, col78, EpiRows<T::RS64>{r_hi + slot * S * T::RING_SS, r_lo + slot * S * T::RING_SS});
        TTS(9);
#ifdef HM_TRUNK_STAMP
        if (tst) { for (int i = 0; i < 9; ++i) tacc[i] += tts[i + 1] - tts[i]; ++tn; }
#endif
        if (slot == 0) g_first = g;
        ++slot;
        if (slot < FCB && more) continue;  // the loop-top barrier orders this conv8 before the next conv5
        // ---- once per FCB passes: fc1 + fc2 + softmax for FCB * S sites -------------------------------------------------------
        // fc1 = a 2-tap "conv" over conv8's two positions (k order l*64 + c; hm_weights.cpp), FCB * S sites at once (slots this
        // batch did not fill hold stale rows whose results are never written out), in two halves of 128 outputs: a wave takes two
        // n-tiles of a half, whose weights (64 registers) it requests in one go -- the first half's before the barrier, where
        // their latency hides behind the other waves' conv8; so are the biases and the places this thread's two sites' results
        // go to.  (ConvH's k-block-ahead streaming was L2-latency-bound here: 12 MFMAs per k-block.)
        using CF = TCfg<64, 2, T::RS64, !W16>;
        using RF = TRows<1, T::RING_SS, FCB * S>;
        using FC1 = TConv<CF, RF, 8, 1, TG<0, 2, 0, 0>>;
        TW<4, 2> WF;
        const int ntf0[2] = {2 * wave, 2 * wave + 1}, ntf1[2] = {8 + 2 * wave, 9 + 2 * wave};
        const int colf0[2] = {32 * wave, 32 * wave + 16}, colf1[2] = {128 + 32 * wave, 144 + 32 * wave};
        tw_load<!W16>(wf(8), ntf0, tl & 63, WF);
        float4 bzf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bzf[j] = *reinterpret_cast<const float4*>(W.bias[8] + (j < 2 ? colf0[j] : colf1[j - 2]) + 4 * ((tl & 63) >> 4));
        int dst_pre[FCB * S * 16 / (NW * 64)];
#pragma unroll
        for (int rnd = 0; rnd < FCB * S * 16 / (NW * 64); ++rnd) {
            const int bsite = (tl >> 4) + rnd * (NW * 64 / 16), sl = bsite / S, idx = min((g_first + sl) * S + (bsite - sl * S), n_sites - 1);
            dst_pre[rnd] = sites ? sites[idx].uidx : idx;
        }
        lds_barrier();
        TTS(10);
        FC1::run(r_hi, r_lo, WF, [&](int j) __attribute__((always_inline)) { return bzf[j]; }, colf0, EpiFc1R<T::HRS>{hfc});
        tw_load<!W16>(wf(8), ntf1, tl & 63, WF);
        FC1::run(r_hi, r_lo, WF, [&](int j) __attribute__((always_inline)) { return bzf[2 + j]; }, colf1, EpiFc1R<T::HRS>{hfc});
        lds_barrier();
        // fc2 + softmax (mod_batch.cpp:46-64) in fp32: 16 lanes per site = 2 outputs x 8 partial sums (two rounds of 16 sites)
#pragma unroll
        for (int rnd = 0; rnd < FCB * S * 16 / (NW * 64); ++rnd) {
            const int bsite = (tl >> 4) + rnd * (NW * 64 / 16), o = (tl >> 3) & 1, part = tl & 7;
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
            const int gs0 = (g_first + sl) * S;
            if ((tl & 15) == 0 && sl < slot && gs0 + site < n_sites) {
                const float v0 = sum, v1 = other;
                const float mx = fmaxf(v0, v1);
                const float e0 = expf(v0 - mx), e1 = expf(v1 - mx);
                const float p1 = e1 / (e0 + e1);
                int q = (int)(255 * p1);
                q = q > 255 ? 255 : q;
                const int dst = dst_pre[rnd];
                logits[2 * (size_t)dst] = v0;
                logits[2 * (size_t)dst + 1] = v1;
                prob[dst] = p1;
                ml[dst] = (uint8_t)q;
            }
        }
        slot = 0;
        TTS(11);
#ifdef HM_TRUNK_STAMP
        if (tst) { tacc[9] += tts[11] - tts[10]; tacc[10] += 1; }
#endif
    }
#ifdef HM_TRUNK_STAMP
    if (tst && lane == 0) {
        for (int i = 0; i < 11; ++i) atomicAdd(&g_tailr_stamp[wave][i], tacc[i]);
        atomicAdd(&g_tailr_stamp[wave][11], tn);
        atomicAdd(&g_tailr_stamp[wave][13], hm_stamp() - tk0);
        atomicAdd(&g_tailr_stamp[wave][14], __builtin_amdgcn_s_memrealtime() - tr0);
    }
#endif
#undef TTS
    vm_drain();  // nothing of this workgroup's gather is in flight when its LDS is handed on
}

void launch_tail_gather_r(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, const uint16_t* edge4,
                          const int32_t* e4row, float* logits, float* p, uint8_t* ml, int grid, bool w16) {
    if (sr.cap <= 0) return;
    const dim3 g(sr.totals ? grid : max(1, min((sr.cap + TAIL_SITES - 1) / TAIL_SITES, grid)));
    if (w16) hipLaunchKernelGGL(tail_kernel_r<true>, g, dim3(256), 0, st, sr, w, logits, p, ml, reinterpret_cast<const half_t*>(maps.e4),
                                reinterpret_cast<const half_t*>(edge4), e4row, reinterpret_cast<const half_t*>(maps.zeros));
    else hipLaunchKernelGGL(tail_kernel_r<false>, g, dim3(256), 0, st, sr, w, logits, p, ml, reinterpret_cast<const half_t*>(maps.e4),
                            reinterpret_cast<const half_t*>(edge4), e4row, reinterpret_cast<const half_t*>(maps.zeros));
}

}  // namespace hm
