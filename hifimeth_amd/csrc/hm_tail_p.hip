// hm_tail_p.hip -- the STRIP tail kernel: conv5 .. conv8, fc1, fc2, softmax of the dense-trunk path for a context whose sites are
// DENSE (CHH: one view position in eight is a site), 16 sites per workgroup pass instead of tail_kernel_r's 8.
//
// What a site reads.  Its 25 conv4 rows are two rows of its own (the window-edge rows, edge kernel) and 23 rows of the dense E4
// map, 16 map rows apart: e4row + 16 s, s = 1 .. 23 (hm_device.h).  Two sites of one strand view whose first map rows agree
// mod 16 read rows of the same LATTICE: if they lie d lattice steps apart they share 23 - d rows.  tail_kernel_r takes its 8 sites
// in (read, qoff) order -- neighbours in qoff have different residues and share nothing -- so every site's 23 rows cross the
// memory system once per site (FETCH_SIZE: 8.5 KB per site from beyond L2, profiles/r04_pmc_derived.txt), and LDS, full with
// 8 x 25 rows, bounds the pass at 8 sites: half-empty last m-tiles in conv5 / conv6, and the fixed costs of a pass (eight
// streaming-conv calls with their prologues and exposed last epilogues, seven barriers) over 8 sites only.
//
// Here the sites of a launch are visited in (residue class, map row) order (class_sort_* below: a counting sort through a
// per-map-row mark array -- a map row is the first row of at most one site -- so no comparison sort is needed), and a pass takes
// up to 16 CONSECUTIVE sites of one class whose rows fit one STRIP of 144 consecutive lattice rows in LDS: at CHH's density
// (0.12 sites per view position) 14 sites on average, 60 KB of strip instead of 16 x 23 rows = 150 KB.  The sites of a pass lie
// along the ROWS of the MFMA tiles (hm_convp.h: m-tile p = position p of all 16 sites): full tiles in every layer, the taps on
// a site's zero padding skipped (exact zeros), one per-lane base address per buffer.  Same products in the same order per
// accumulator as tail_kernel_r: byte-identical calls (tests/test_gpu_parity.py).
//
// LDS (159.3 KB): A = [strip 144 rows | edge row 0 of 16 sites | edge row 24 of 16 sites] x (hi, lo) planes of 208-byte rows;
// B = conv5's output [13][16 sites] x (hi, lo).  conv6's output overlays the strip (rows 0 .. 111), conv7's, conv8's and fc1's
// outputs overlay B.  The next pass's gather (LDS-DMA, row-aligned quads as in hm_tail_r.hip; the strip's source rows are
// equidistant, no address table) runs in two parts: the edge rows and strip rows 112 .. 143 while conv6 / conv7 run, strip rows
// 0 .. 111 once conv7 has read conv6's output -- their lines are pulled into L2 by touch loads while conv5 runs, so that the
// late part is an L2 copy.
//
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98); softmax -> ML byte:
// mod_batch.cpp:46-64.
#include "hm_convp.h"
#ifdef HM_TRUNK_STAMP
#include "hm_stamp.h"
namespace hm { __device__ unsigned long long g_tailp_stamp[4][16]; }
extern "C" int hm_debug_tailp_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(hm::g_tailp_stamp), sizeof(hm::g_tailp_stamp)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[4][16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(hm::g_tailp_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

namespace hm {

namespace {

struct PGeo {
    static constexpr int S = 16, NW = 4;
    static constexpr int L4 = C4_LEN, L5 = 13, L6 = 7, L7 = 4, L8 = 2;
    static constexpr int RS = 104, RS64 = 72;               // halves
    // fc1's fp32 output for the VALU fc2: a site's 256 values in 8 parts of 32, parts 36 floats apart (fc2's 16 lanes of a site read 8 distinct
    // parts: 144-byte steps fall into different banks), sites 288 apart; fc2's weights likewise, the two outputs 324 apart
    static constexpr int HPS = 36, HRS = 8 * HPS, F2S = 324;
    static constexpr int R = TAILP_STRIP;                  // strip rows (lattice rows 16 map rows apart)
    static constexpr int SPAN = R - (L4 - 2);              // a pass's sites start at most this many lattice rows apart (0 .. SPAN)
    // plane A (halves from the plane pointer)
    static constexpr int STRIP = 0, EDGE0 = R * RS, EDGE24 = (R + S) * RS, PA = (R + 2 * S) * RS;
    // plane B
    static constexpr int P5 = L5 * S * RS;
    static constexpr int C7 = 0, C8 = L7 * S * RS64, HFC = C8 + L8 * S * RS64;   // conv7's / conv8's planes, fc1's fp32 output, inside B's hi | lo plane
    static constexpr int LDS_HALVES = 2 * PA + 2 * P5;
    static_assert(L6 * S * RS <= R * RS, "conv6's output overlays the strip");
    static_assert(HFC * 2 % 16 == 0 && HFC + S * HRS * 2 <= P5, "fc1's output fits behind conv8's, inside the hi plane");
    static constexpr int LATE_ROWS = L6 * S;               // strip rows conv6's output overlays: fetched once conv7 has read it
    static constexpr int QROWS = 4, QBYTES = QROWS * RS * 2, NQ = R / QROWS, LATEQ = LATE_ROWS / QROWS;
    static_assert(R % QROWS == 0 && LATE_ROWS % QROWS == 0 && RS * 2 == 13 * 16, "row-aligned pieces of 13 sixteen-byte units");
};

// conv5's operands: data row d of a site = its edge row 0 (d = 0), strip rows s + d - 1 (d = 1 .. 23), its edge row 24 (d = 24);
// output position p reads data rows 2p - 1 .. 2p + 1
template <class C>
struct PInStrip {
    using T = PGeo;
    int sb;  // s * RS + 8 * lk: this lane's site's first strip row
    int eb;  // li * RS + 8 * lk
    static constexpr int drow(int tile, int kb) { return 2 * tile - 1 + C::tap(kb); }
    static constexpr bool skip(int tile, int kb) { return drow(tile, kb) < 0 || drow(tile, kb) >= T::L4; }
    template <int TILE, int KB>
    __device__ __forceinline__ int off() const {
        constexpr int d = drow(TILE, KB), ch = C::ch0(KB);
        if constexpr (d == 0) return eb + (T::EDGE0 + ch);
        else if constexpr (d == T::L4 - 1) return eb + (T::EDGE24 + ch);
        else return sb + (T::STRIP + (d - 1) * T::RS + ch);
    }
};

}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tail_kernel_p(SiteRange sr, CtxWeights W, float* __restrict__ logits, float* __restrict__ prob, uint8_t* __restrict__ ml,
                   const half_t* __restrict__ e4, const half_t* __restrict__ edge4, const int32_t* __restrict__ order,
                   const int32_t* __restrict__ okey, int n_rows) {
    using T = PGeo;
    constexpr int NW = T::NW;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    // a workgroup takes a CONTIGUOUS range of the class-sorted list
    const int base_n = n_sites / (int)gridDim.x, rem_n = n_sites - base_n * (int)gridDim.x;
    const int i_begin = (int)blockIdx.x * base_n + min((int)blockIdx.x, rem_n), i_end = i_begin + base_n + ((int)blockIdx.x < rem_n);
    if (i_begin >= i_end) return;

    __shared__ __attribute__((aligned(16))) half_t smem[T::LDS_HALVES];
    __shared__ __attribute__((aligned(16))) float fc2w[T::F2S + 8 * T::HPS + 4];  // fc2 weights (padded like fc1's output) + bias
    __shared__ __attribute__((aligned(16))) float bias_l[96 + 96 + 64 + 64];  // conv5 .. conv8 biases
    __shared__ uint32_t touch_dump[64];                                       // where the touch loads' dwords go (never read)
    half_t* a_hi = smem;
    half_t* a_lo = smem + T::PA;
    half_t* b_hi = smem + 2 * T::PA;
    half_t* b_lo = smem + 2 * T::PA + T::P5;
    float* hfc = reinterpret_cast<float*>(b_hi + T::HFC);
    const float* b5 = bias_l;
    const float* b6 = bias_l + 96;
    const float* b7 = bias_l + 192;
    const float* b8 = bias_l + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 256 + 2; i += NW * 64) {
        if (i < 512) fc2w[(i >> 8) * T::F2S + ((i & 255) >> 5) * T::HPS + (i & 31)] = W.fc2_w[i];
        else fc2w[T::F2S + 8 * T::HPS + (i - 512)] = W.fc2_b[i - 512];
    }
    for (int i = tid; i < 320; i += NW * 64) bias_l[i] = i < 96 ? W.bias[4][i] : i < 192 ? W.bias[5][i - 96] : i < 256 ? W.bias[6][i - 192] : W.bias[7][i - 256];
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };

    // ---- resident weights: n-tiles (a, b) of conv5 and conv6 (as in tail_kernel_r); conv7's, conv8's and fc1's arrive per pass -------
    const int nta = wave == 0 ? 0 : wave == 1 ? 2 : wave == 2 ? 3 : 5, ntb = wave < 2 ? 1 : 4;
    const int nt56[2] = {nta, ntb}, col56[2] = {16 * nta, 16 * ntb};
    const int nt78[1] = {wave}, col78[1] = {16 * wave};
    const bool odd = wave & 1;
    TW<9, 2> W5, W6;
    TW<9, 1> W7;
    TW<6, 1> W8;
    tw_load(wf(4), nt56, lane, W5);
    tw_load(wf(5), nt56, lane, W6);

    using C96 = PCfg<96, 3>;
    using C64 = PCfg<64, 3>;
    using CF = PCfg<64, 2>;
    using I5 = PInStrip<C96>;
    using I6 = PInRows<C96, T::RS, T::L5, 0>;
    using I7 = PInRows<C96, T::RS, T::L6, 0>;
    using I8 = PInRows<C64, T::RS64, T::L7, T::C7>;
    using IF = PInRows<CF, T::RS64, T::L8, T::C8, 0, 0>;

    // ---- the plan of a pass: which sites, where their rows lie in the strip ------------------------------------------------------------
    // Lane li of every wave holds candidate li of the pass that starts at list position `pos`: its list entry (site index in the launch's
    // site list) and its first map row r1 = e4row + 16.  The pass takes the leading candidates of the first one's class (r1 mod 16) that
    // start within SPAN lattice rows of it; a lane beyond them stands in for the last site taken (same addresses, results not stored).
    int tl = tid;
    struct Plan {
        int take;      // sites of the pass (wave-uniform)
        int r1min;     // first map row of the strip (wave-uniform)
        int nrows;     // strip rows some site of the pass reads (wave-uniform)
        int s;         // this lane's site (li): its first strip row
        int oi;        // ... its index in the launch's site list
    };
    auto load_cand = [&](const int pos, int& c_oi, int& c_r1) __attribute__((always_inline)) {
        const int i = min(pos + (tl & 15), i_end - 1);
        // (clamped: a list entry is a site of this launch and its key a map row -- whatever the sort left in the arrays, no address below leaves the buffers)
        c_oi = min(max(order[i], 0), n_sites - 1);
        c_r1 = min(max(okey[i], 0), n_rows - 1);
    };
    auto make_plan = [&](const int pos, const int c_oi, const int c_r1) __attribute__((always_inline)) {
        Plan p;
        const int first = __shfl(c_r1, 0, 64);
        const bool ok = pos + (tl & 15) < i_end && ((c_r1 ^ first) & 15) == 0 && c_r1 >= first && c_r1 - first <= 16 * T::SPAN;
        const unsigned long long m = __ballot(ok) & 0xFFFFull;
        p.take = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(~m));   // leading candidates that fit (>= 1 while pos < i_end)
        const int e = min(tl & 15, max(p.take, 1) - 1);
        const int r1 = __shfl(c_r1, e, 64);
        p.oi = __shfl(c_oi, e, 64);
        p.s = (r1 - first) >> 4;
        p.r1min = __builtin_amdgcn_readfirstlane(first);
        p.nrows = __builtin_amdgcn_readfirstlane(__shfl(r1, 15, 64) - first) / 16 + (T::L4 - 2);
        return p;
    };

    // ---- gather ------------------------------------------------------------------------------------------------------------------------
    // strip quad q = strip rows 4q .. 4q + 3 of both planes: lane l < 52 brings chunk l % 13 of row l / 13 (chunk 12 is the row's pad: it reads
    // the 16 bytes behind the 192 it needs).  A quad none of whose rows is read by the pass is not fetched (EXEC = 0).
    const unsigned long long lanes52 = 0x000FFFFFFFFFFFFFull;
    const uint32_t lds_a_hi = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)a_hi;
    const uint32_t lds_a_lo = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)a_lo;
    const unsigned long long e4b = (unsigned long long)(uintptr_t)e4, edb = (unsigned long long)(uintptr_t)edge4;
    auto dma_quad = [&](const int q, const Plan& p) __attribute__((always_inline)) {  // q wave-uniform
        const int ln = tl & 63, q_row = min(ln / 13, T::QROWS - 1), q_chunk16 = (ln % 13) * 16;
        const unsigned long long src = e4b + (unsigned long long)(long long)(p.r1min + 16 * (T::QROWS * q + q_row)) * (2 * C4_CH * 2) + (unsigned)q_chunk16;
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(lds_a_hi + (uint32_t)(T::QBYTES * q));
        const uint32_t d1 = __builtin_amdgcn_readfirstlane(lds_a_lo + (uint32_t)(T::QBYTES * q) - 192u);
        const unsigned long long ex = T::QROWS * q < p.nrows ? lanes52 : 0ull;
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
            : "s"(ex), "s"(d0), "s"(d1), "v"(src));
    };
    // the edge rows of sites 4 wave .. 4 wave + 3: [row 0: hi | lo][row 24: hi | lo] = 768 contiguous bytes per site -> four LDS rows
    auto dma_edges = [&](const Plan& p) __attribute__((always_inline)) {
        const int ln = tl & 63, e_site = min(ln / 13, 3), e_chunk16 = (ln % 13) * 16;
        const int oi = __shfl(p.oi, 4 * wave + e_site, 64);
        const unsigned long long src = edb + (unsigned long long)oi * (4 * C4_CH * 2) + (unsigned)e_chunk16;
        const uint32_t rowq = (uint32_t)(4 * wave * T::RS * 2);
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(lds_a_hi + (uint32_t)(T::EDGE0 * 2) + rowq);
        const uint32_t d1 = __builtin_amdgcn_readfirstlane(lds_a_lo + (uint32_t)(T::EDGE0 * 2) + rowq - 192u);
        const uint32_t d2 = __builtin_amdgcn_readfirstlane(lds_a_hi + (uint32_t)(T::EDGE24 * 2) + rowq - 384u);
        const uint32_t d3 = __builtin_amdgcn_readfirstlane(lds_a_lo + (uint32_t)(T::EDGE24 * 2) + rowq - 576u);
        unsigned long long sv;
        uint32_t km;
        asm volatile(
            "s_mov_b64 %0, exec\n\t"
            "s_mov_b32 %1, m0\n\t"
            "s_mov_b64 exec, %2\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %7, off\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %7, off offset:192\n\t"
            "s_mov_b32 m0, %5\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %7, off offset:384\n\t"
            "s_mov_b32 m0, %6\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %7, off offset:576\n\t"
            "s_mov_b32 m0, %1\n\t"
            "s_mov_b64 exec, %0"
            : "=&s"(sv), "=&s"(km)
            : "s"(lanes52), "s"(d0), "s"(d1), "s"(d2), "s"(d3), "v"(src));
    };
    // touch loads: one dword per 128-byte line of the LATE strip rows (0 .. LATE_ROWS - 1: 3 lines per row) of the next pass, so that the copy
    // behind conv7 finds them in L2.  An LDS-DMA into a dump area nobody reads: a load into a register would have to be waited for by the
    // compiler's bookkeeping (or, hidden from it in inline asm, would write the register whenever it returns, long after the compiler has
    // given that register to something else).
    const uint32_t lds_dump = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)touch_dump;
    auto touch_late = [&](const Plan& p) __attribute__((always_inline)) {
        constexpr int LINES = T::LATE_ROWS * 3, PER_WAVE = (LINES + NW * 64 - 1) / (NW * 64);
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int line = min((tl & 255) + NW * 64 * k, LINES - 1), row = line / 3, part = line - 3 * row;
            const unsigned long long src = e4b + (unsigned long long)(long long)(p.r1min + 16 * min(row, max(p.nrows, 1) - 1)) * (2 * C4_CH * 2) + (unsigned)(part * 128);
            uint32_t km;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dword %2, off\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(km)
                : "s"(lds_dump), "v"(src));
        }
    };

    // ---- first pass of this workgroup: its plan, all of its rows ---------------------------------------------------------------------------
    int pos = i_begin;
    int c_oi, c_r1;
    load_cand(pos, c_oi, c_r1);
    Plan cur = make_plan(pos, c_oi, c_r1);
    int n_oi, n_r1;   // candidates of the pass after this one
    load_cand(pos + cur.take, n_oi, n_r1);
    lds_barrier();    // (fc2w / bias_l written)
#pragma unroll
    for (int k = 0; k < (T::NQ + NW - 1) / NW; ++k) dma_quad(min(wave + NW * k, T::NQ - 1), cur);
    dma_edges(cur);
    vm_drain();

#ifdef HM_TRUNK_STAMP
    unsigned long long tts[12], tacc[12] = {};
    unsigned long long tn = 0, tsites = 0;
    const bool tst = blockIdx.x == 0;
    const unsigned long long tk0 = hm_stamp(), tr0 = __builtin_amdgcn_s_memrealtime();
#define TTS(i) do { if (tst) tts[i] = hm_stamp(); } while (0)
#else
#define TTS(i)
#endif
    while (pos < i_end) {
        tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int li = tl & 15, lk = (tl & 63) >> 4;
        lds_barrier();  // this pass's rows are in LDS (every wave drained its own before); the previous pass's fc2 has read fc1's output
        TTS(0);
        // the next pass: its plan now (its candidates were requested a pass ago), its candidates' successors requested now
        const int pos_n = pos + cur.take;
        const bool more = pos_n < i_end;
        const Plan nxt = make_plan(pos_n, n_oi, n_r1);
        load_cand(pos_n + nxt.take, n_oi, n_r1);
        if (more) touch_late(nxt);
        // ---- conv5: strip + edge rows (A) -> B ----------------------------------------------------------------------------------------
        {
            const I5 ia{cur.s * T::RS + 8 * lk, li * T::RS + 8 * lk};
            const EpiP<T::RS> e5{b_hi + li * T::RS + 4 * lk, b_lo + li * T::RS + 4 * lk};
            // 13 m-tiles x 6 n-tiles: a wave runs n-tile a on all 13 and n-tile b (shared with its neighbour) on 7 | 6 of them; a group =
            // one tile with the pair + one with a alone (3 accumulators, 2 tiles: the operand ring holds this block's and the next one's reads)
            if (odd) PConv<C96, I5, 8, 1, TG<7, 1, 0, 1>, TG<8, 1, 1, 1>, TG<9, 1, 2, 1>, TG<10, 1, 3, 1>, TG<11, 1, 4, 1>, TG<12, 1, 5, 1>, TG<0, 0, 6, 1>>::run(a_hi, a_lo, W5, b5, col56, ia, e5);
            else PConv<C96, I5, 8, 1, TG<0, 1, 7, 1>, TG<1, 1, 8, 1>, TG<2, 1, 9, 1>, TG<3, 1, 10, 1>, TG<4, 1, 11, 1>, TG<5, 1, 12, 1>, TG<6, 1, 0, 0>>::run(a_hi, a_lo, W5, b5, col56, ia, e5);
        }
        TTS(1);
        lds_barrier();  // B complete; nobody reads A any more
        TTS(2);
        // ---- conv6: B -> the strip's first rows; the next pass's edge rows and high strip rows are fetched meanwhile ---------------------------
        {
            const I6 ia{li * T::RS + 8 * lk};
            const EpiP<T::RS> e6{a_hi + li * T::RS + 4 * lk, a_lo + li * T::RS + 4 * lk};
            auto hook_early = [&](auto c_) __attribute__((always_inline)) {
                constexpr int C = decltype(c_)::value;
                if constexpr (C == 0) { if (more) dma_edges(nxt); }
                else if constexpr (C - 1 < (T::NQ - T::LATEQ + NW - 1) / NW) { if (more) dma_quad(min(T::LATEQ + wave + NW * (C - 1), T::NQ - 1), nxt); }
            };
            // 7 m-tiles x 6 n-tiles
            if (odd) PConv<C96, I6, 8, 1, TG<4, 1, 0, 1>, TG<5, 1, 1, 1>, TG<6, 1, 2, 1>, TG<0, 0, 3, 1>>::run(b_hi, b_lo, W6, b6, col56, ia, e6, hook_early);
            else PConv<C96, I6, 8, 1, TG<0, 1, 4, 1>, TG<1, 1, 5, 1>, TG<2, 1, 6, 1>, TG<3, 1, 0, 0>>::run(b_hi, b_lo, W6, b6, col56, ia, e6, hook_early);
        }
        // conv7's weights are NOT resident (the register file holds conv5's and conv6's, 288 registers per wave, and the working set; with
        // conv7's 72 on top the allocator spills): requested here, behind conv6's last MFMA, they arrive in k order while the last epilogue,
        // the barrier and conv7's first blocks run
        tw_load(wf(6), nt78, tl & 63, W7);
        TTS(3);
        lds_barrier();
        TTS(4);
        // ---- conv7: strip's first rows -> B (conv7 plane) ----------------------------------------------------------------------------------
        tw_load(wf(7), nt78, tl & 63, W8);  // conv8's weights: requested a layer ahead
        {
            const I7 ia{li * T::RS + 8 * lk};
            const EpiP<T::RS64> e7{b_hi + T::C7 + li * T::RS64 + 4 * lk, b_lo + T::C7 + li * T::RS64 + 4 * lk};
            PConv<C96, I7, 8, 1, TG<0, 0, 0, 2>, TG<0, 0, 2, 2>>::run(a_hi, a_lo, W7, b7, col78, ia, e7);
        }
        TTS(5);
        lds_barrier();  // conv6's output has been read: the strip's low rows are free
        TTS(6);
        // ---- the next pass's low strip rows (from L2: touched while conv5 ran); conv8 -----------------------------------------------------------
        if (more) {
#pragma unroll
            for (int k = 0; k < T::LATEQ / NW; ++k) dma_quad(wave + NW * k, nxt);
        }
        static_assert(T::LATEQ % NW == 0, "the late quads deal out evenly");
        {
            const I8 ia{li * T::RS64 + 8 * lk};
            const EpiP<T::RS64> e8{b_hi + T::C8 + li * T::RS64 + 4 * lk, b_lo + T::C8 + li * T::RS64 + 4 * lk};
            PConv<C64, I8, 8, 1, TG<0, 0, 0, 2>>::run(b_hi, b_lo, W8, b8, col78, ia, e8);
        }
        TTS(7);
        // ---- fc1 + fc2 + softmax for the pass's 16 sites --------------------------------------------------------------------------------------
        // fc1 = a 2-tap "conv" over conv8's two positions (k order l*64 + c; hm_weights.cpp) in two halves of 128 outputs: a wave takes
        // two n-tiles of a half, whose weights (64 registers) it requests in one go
        using FC1 = PConv<CF, IF, 8, 1, TG<0, 1, 0, 0>>;
        TW<4, 2> WF;
        const int ntf0[2] = {2 * wave, 2 * wave + 1}, ntf1[2] = {8 + 2 * wave, 9 + 2 * wave};
        const int colf0[2] = {32 * wave, 32 * wave + 16}, colf1[2] = {128 + 32 * wave, 144 + 32 * wave};
        tw_load(wf(8), ntf0, tl & 63, WF);
        float4 bzf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bzf[j] = *reinterpret_cast<const float4*>(W.bias[8] + (j < 2 ? colf0[j] : colf1[j - 2]) + 4 * ((tl & 63) >> 4));
        // where the results of this thread's site (fc2: site tl >> 4) go
        const int my_oi = __shfl(cur.oi, (tl >> 4) & 15, 64);   // (lane i < 16 of every wave holds site i's entry)
        const int dst = sites ? sites[my_oi].uidx : my_oi;
        lds_barrier();  // conv8's rows are complete
        TTS(8);
        {
            const IF ia{li * T::RS64 + 8 * lk};
            const EpiFc1P<T::HPS> ef{hfc + li * T::HRS + 4 * lk};
            FC1::run(b_hi, b_lo, WF, [&](int j) __attribute__((always_inline)) { return bzf[j]; }, colf0, ia, ef);
            tw_load(wf(8), ntf1, tl & 63, WF);
            FC1::run(b_hi, b_lo, WF, [&](int j) __attribute__((always_inline)) { return bzf[2 + j]; }, colf1, ia, ef);
        }
        lds_barrier();
        TTS(9);
        // fc2 + softmax (mod_batch.cpp:46-64) in fp32: 16 lanes per site = 2 outputs x 8 partial sums
        {
            const int bsite = tl >> 4, o = (tl >> 3) & 1, part = tl & 7;
            const float* h = hfc + bsite * T::HRS + part * T::HPS;
            const float* w2 = fc2w + o * T::F2S + part * T::HPS;
            float sum = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) sum = fmaf(h[k], w2[k], sum);
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 1, 64);
            sum += fc2w[T::F2S + 8 * T::HPS + o];
            const float other = __shfl_xor(sum, 8, 64);
            if ((tl & 15) == 0 && bsite < cur.take) {
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
        TTS(10);
#ifdef HM_TRUNK_STAMP
        if (tst) { for (int i = 0; i < 10; ++i) tacc[i] += tts[i + 1] - tts[i]; ++tn; tsites += cur.take; }
#endif
        vm_drain();  // this wave's share of the next pass's rows has landed (and the stores above have left)
        pos = max(pos_n, pos + 1);
        cur = nxt;
    }
#ifdef HM_TRUNK_STAMP
    if (tst && lane == 0) {
        for (int i = 0; i < 10; ++i) atomicAdd(&g_tailp_stamp[wave][i], tacc[i]);
        atomicAdd(&g_tailp_stamp[wave][11], tn);
        atomicAdd(&g_tailp_stamp[wave][12], tsites);
        atomicAdd(&g_tailp_stamp[wave][13], hm_stamp() - tk0);
        atomicAdd(&g_tailp_stamp[wave][14], __builtin_amdgcn_s_memrealtime() - tr0);
    }
#endif
#undef TTS
    vm_drain();
}

// ---- the class sort: sites of a launch in (first map row mod 16, first map row) order ----------------------------------------------------
// A map row is the FIRST row (e4row + 16) of at most one site of a launch (a site's first row is a function of its read's map region
// and its view position), so the order follows from a scan over the map rows: mark[row] = site index, then per residue class the marks
// in row order.  Blocks of CS_ROWS map rows; counts per (class, block) scanned class-major.
constexpr int CS_ROWS = 4096;   // map rows per block: 256 threads x 16 consecutive rows (one lattice step, all 16 classes)

__global__ __launch_bounds__(256) void class_mark_kernel(SiteRange sr, const int32_t* __restrict__ e4row, int32_t* __restrict__ mark) {
    const Site* sites;
    const int n = resolve_sites(sr, sites);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) mark[e4row[i] + 16] = i;
}

// bit c of the result: row 16 t + c of the block carries a mark
__device__ __forceinline__ uint32_t class_bits(const int32_t* __restrict__ mark, int64_t row0, int64_t n_rows, int32_t (&m)[16]) {
    uint32_t bits = 0;
    const int4* p = reinterpret_cast<const int4*>(mark + row0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int4 v = make_int4(-1, -1, -1, -1);
        if (row0 + 4 * q + 3 < n_rows) v = p[q];
        else {
            if (row0 + 4 * q < n_rows) v.x = mark[row0 + 4 * q];
            if (row0 + 4 * q + 1 < n_rows) v.y = mark[row0 + 4 * q + 1];
            if (row0 + 4 * q + 2 < n_rows) v.z = mark[row0 + 4 * q + 2];
        }
        m[4 * q] = v.x; m[4 * q + 1] = v.y; m[4 * q + 2] = v.z; m[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) bits |= (m[c] >= 0 ? 1u : 0u) << c;
    return bits;
}

__global__ __launch_bounds__(256) void class_count_kernel(const int32_t* __restrict__ mark, int64_t n_rows, int n_blocks, int32_t* __restrict__ cnt) {
    __shared__ int32_t wc[4][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int32_t m[16];
    const uint32_t bits = class_bits(mark, (int64_t)blockIdx.x * CS_ROWS + 16 * t, n_rows, m);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int k = __popcll(__ballot((bits >> c) & 1u));
        if (lane == 0) wc[wave][c] = k;
    }
    __syncthreads();
    if (t < 16) cnt[(size_t)t * n_blocks + blockIdx.x] = wc[0][t] + wc[1][t] + wc[2][t] + wc[3][t];
}

// exclusive scan of n counters in place, one workgroup
__global__ __launch_bounds__(1024) void class_scan_kernel(int32_t* __restrict__ cnt, int n) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x, per = (n + 1023) / 1024, lo = min(t * per, n), hi = min(lo + per, n);
    int32_t s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int32_t run = part[t] - s;
    for (int i = lo; i < hi; ++i) {
        const int32_t v = cnt[i];
        cnt[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(256) void class_write_kernel(const int32_t* __restrict__ mark, int64_t n_rows, int n_blocks, const int32_t* __restrict__ offs,
                                                          int32_t* __restrict__ order, int32_t* __restrict__ okey) {
    __shared__ int32_t wc[4][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int32_t m[16];
    const int64_t row0 = (int64_t)blockIdx.x * CS_ROWS + 16 * t;
    const uint32_t bits = class_bits(mark, row0, n_rows, m);
    int rank[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const unsigned long long b = __ballot((bits >> c) & 1u);
        rank[c] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wc[wave][c] = __popcll(b);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (!((bits >> c) & 1u)) continue;
        int o = offs[(size_t)c * n_blocks + blockIdx.x] + rank[c];
        for (int w = 0; w < wave; ++w) o += wc[w][c];
        order[o] = m[c];
        okey[o] = (int32_t)(row0 + c);
    }
}

size_t tail_strip_mark_bytes(int64_t map_rows) { return (size_t)(map_rows + CS_ROWS) * sizeof(int32_t); }
size_t tail_strip_count_bytes(int64_t map_rows) { return (size_t)16 * (size_t)((map_rows + CS_ROWS - 1) / CS_ROWS + 1) * sizeof(int32_t); }

void launch_tail_strip(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, int n_views, const uint16_t* edge4,
                       const int32_t* e4row, int32_t* mark, int32_t* cnt, int32_t* order, int32_t* okey, float* logits, float* p, uint8_t* ml,
                       int grid) {
    if (sr.cap <= 0) return;
    const int64_t n_rows = maps.view_rows * n_views;
    const int n_blocks = (int)((n_rows + CS_ROWS - 1) / CS_ROWS);
    (void)hipMemsetAsync(mark, 0xFF, (size_t)n_rows * sizeof(int32_t), st);
    hipLaunchKernelGGL(class_mark_kernel, dim3(max(1, min((sr.cap + 255) / 256, 4 * grid))), dim3(256), 0, st, sr, e4row, mark);
    hipLaunchKernelGGL(class_count_kernel, dim3(n_blocks), dim3(256), 0, st, mark, n_rows, n_blocks, cnt);
    hipLaunchKernelGGL(class_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, 16 * n_blocks);
    hipLaunchKernelGGL(class_write_kernel, dim3(n_blocks), dim3(256), 0, st, mark, n_rows, n_blocks, cnt, order, okey);
    const dim3 g(sr.totals ? grid : max(1, min((sr.cap + PGeo::S - 1) / PGeo::S, grid)));
    hipLaunchKernelGGL(tail_kernel_p, g, dim3(256), 0, st, sr, w, logits, p, ml, reinterpret_cast<const half_t*>(maps.e4),
                       reinterpret_cast<const half_t*>(edge4), order, okey, (int)std::min<int64_t>(n_rows, INT32_MAX));
}

}  // namespace hm
