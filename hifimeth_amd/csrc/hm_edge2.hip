// hm_edge2.hip -- the edge kernel, second form: conv4 rows 0 and 24 of every site (the two that are not samples of the dense
// map E4, hm_trunk.hip), S = 32 sites per pass, left and right chains stacked along M (64 pseudo-rows), four one-row layers.
//
// Same arithmetic as edge_kernel (hm_trunk.hip) -- per accumulator: bias, then the k-blocks of the LIVE taps in ascending
// order, products w_hi x_hi, w_hi x_lo, w_lo x_hi -- so the edge rows are bit-identical; what changed is everything around
// the MFMAs, which were a quarter of edge_kernel's pass (profiles/r03_pmc_derived.txt: MFMA busy 0.26 - 0.30):
//   * taps are read WHERE THEY LIE.  A layer's operand row is (zero padding | the previous layer's edge output | one or two
//     rows of the dense map E(l-1)); edge_kernel copied all three side by side into an operand buffer before every layer
//     (a barrier + an LDS-to-LDS / register-to-LDS copy + a barrier).  Here the previous output stays in its plane, the map
//     rows in theirs, and a k-block reads from the plane its tap lives in.
//   * taps on the zero padding are skipped (left chains: tap 0 of conv2..conv4; right chains: tap 2 where the window ends on
//     the padding) -- a sixth to a third of conv2..conv4's MFMAs, exact zeros in edge_kernel.
//   * map rows arrive by LDS-DMA one LAYER ahead, into the other of two map sets, instead of being requested in one burst at
//     the start of the pass and parked in registers: no burst to wait for (edge_kernel: 8 - 11 k of a pass's 42 k ticks).
//   * a layer's weights (96 registers per wave; 196 KB per layer and CU, 3 k cycles of the CU's 64 B/clk vector-memory path --
//     as long as the layer's MFMAs) stream in during the PREVIOUS layer, each k-block's fragments into the registers of a
//     k-block that has just been used for the last time (a few registers of slack keep the loads ahead of the frees).  A layer
//     therefore waits for nothing: its weights and map rows landed before it started (one vmcnt(0) + barrier per layer).
//     The right and the left chains' blocks of a k-block run back to back so that the weights die in k order.
//   * the previous layer's output is overwritten in place, so a layer's epilogue runs behind a barrier of its own.
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98).
#include "hm_edge.h"
#ifdef HM_TRUNK_STAMP
#include "hm_stamp.h"
namespace hm { __device__ unsigned long long g_edge2_stamp[8][16]; }
extern "C" int hm_debug_edge2_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(hm::g_edge2_stamp), sizeof(hm::g_edge2_stamp)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[8][16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(hm::g_edge2_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

namespace hm {

namespace {

constexpr int E2_NW = 8;
constexpr int E2_MROWS = 3 * EG_S;                 // rows of a map set: left chains | right chains, first map tap | second map tap
constexpr int E2_SLACK = 256;                      // halves behind a map set that the last DMA piece may overrun
constexpr int E2_MRS = 264;                        // halves per map-set row: [hi 128 | lo 128 | 16 B pad] as the row lies in HBM (conflict-free reads)
constexpr int E2_MSET = E2_MROWS * E2_MRS + E2_SLACK;
constexpr int E2_SPLANE = EG_M * TR_RS;
constexpr int E2_XRS = EG_XROWS * TR_WRS + 8;        // halves per pseudo-row of feature rows: 16 rows of 8 halves + 16 B pad (conflict-free reads)
constexpr int E2_XH = EG_M * E2_XRS;
constexpr int E2_AHEAD = 2;                        // k-blocks of register slack: the next layer's k-block kb is requested once kb - 2 is dead
static_assert(E2_MRS * 2 == 33 * 16, "a map-set row is 33 sixteen-byte chunks (32 + pad)");

template <int I, int N, class F>
__device__ __forceinline__ void e2_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        e2_for<I + 1, N>(f);
    }
}

// LDS-only barrier: orders this workgroup's LDS traffic, leaves vector-memory operations (the DMAs) in flight
__device__ __forceinline__ void e2_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// s_waitcnt vmcnt(N) as a builtin, so that the compiler's own bookkeeping sees it (hm_tail_r.hip)
template <int N>
__device__ __forceinline__ void e2_vmwait() {
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// geometry of layer LAYER (2..4): which taps of a side are live, where they lie, which map rows a pass needs
template <int K1, int LAYER>
struct E2L {
    using G = EdgeGeo<K1>;
    static constexpr bool PAD = LAYER == 2 ? G::PAD2 : LAYER == 3 ? G::PAD3 : G::PAD4;
    static constexpr int R = LAYER == 2 ? G::R1 : LAYER == 3 ? G::R2 : G::R3, STEP = LAYER == 2 ? 2 : LAYER == 3 ? 4 : 8;
    static constexpr int COUT = LAYER == 4 ? C4_CH : 128;
    static constexpr int ROWS = PAD ? 2 * EG_S : 3 * EG_S;         // map rows per pass
    static constexpr int NI = (ROWS * 33 + 63) / 64;                // DMA pieces (1 KB of LDS each) per map set
    // tap kinds as in tap_source (hm_edge.h): side 0 = (zero, previous output, map); side 1 = (map, previous output, zero) or
    // (map, map, previous output)
    static constexpr int kind(int side, int tap) {
        if (side == 0) return tap == 0 ? SRC_ZERO : tap == 1 ? SRC_SPEC : SRC_MAP;
        if (PAD) return tap == 0 ? SRC_MAP : tap == 1 ? SRC_SPEC : SRC_ZERO;
        return tap == 2 ? SRC_SPEC : SRC_MAP;
    }
    // the j-th map tap of pseudo-row m sits in map-set row m + 32 j
    static constexpr int map_j(int side, int tap) { return side == 1 && !PAD && tap == 1 ? 1 : 0; }
    // schedule: k-blocks in ascending order, per k-block the right chains' block then the left chains' (whichever are live), so that
    // a k-block's weights are dead once its blocks are done; block b -> (side, k-block)
    static constexpr bool live(int side, int kb) { return kind(side, kb / 4) != SRC_ZERO; }
    static constexpr int nblocks() { int n = 0; for (int kb = 0; kb < 12; ++kb) n += live(1, kb) + live(0, kb); return n; }
    static constexpr int NB = nblocks();
    static constexpr int find(int b, bool want_side) {
        int n = 0;
        for (int kb = 0; kb < 12; ++kb)
            for (int s = 1; s >= 0; --s)
                if (live(s, kb) && n++ == b) return want_side ? s : kb;
        return -1;
    }
    static constexpr int side_of(int b) { return find(b, true); }
    static constexpr int kb_of(int b) { return find(b, false); }
    // the k-block whose last use is block b (-1: none)
    static constexpr int dies_at(int b) { return b + 1 < NB && kb_of(b + 1) == kb_of(b) ? -1 : kb_of(b); }
};
static_assert(E2L<13, 2>::NB == 20 && E2L<13, 4>::NB == 16 && E2L<11, 4>::NB == 20 && E2L<13, 2>::kb_of(4) == 4 && E2L<13, 2>::side_of(5) == 0 &&
              E2L<13, 2>::dies_at(4) == -1 && E2L<13, 2>::dies_at(5) == 4 && E2L<13, 4>::kb_of(15) == 11 && E2L<11, 2>::NB == 16 && E2L<11, 2>::side_of(15) == 0 && E2L<11, 2>::kb_of(15) == 11,
              "edge layer schedules");

struct E2Site {
    int64_t vrow;  // map row of view position 0 (incl. the view's plane offset)
    int32_t off, L;
    int64_t bo;
    int32_t view, valid;
};

}  // namespace

template <int K1>
__global__ __launch_bounds__(512) void edge2_kernel(SiteRange sr, const RInfo* __restrict__ rinfo, const uint8_t* __restrict__ bases,
                                                     const uint32_t* __restrict__ kin, CtxWeights W, TrunkMaps mp,
                                                     uint16_t* __restrict__ edge4, int32_t* __restrict__ e4row) {
    using G = EdgeGeo<K1>;
    constexpr int NW = E2_NW;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    __shared__ __attribute__((aligned(16))) half_t smem[2 * E2_MSET + 2 * E2_SPLANE + E2_XH];
    __shared__ E2Site sinfo2[2][EG_S];
    __shared__ int32_t rowtab[2][3][E2_MROWS];       // [pass buffer][layer - 2][map-set row] -> row of the layer's map (what a DMA piece gathers)
    half_t* mset = smem;                             // [set][E2_MSET]
    half_t* sp_hi = smem + 2 * E2_MSET;
    half_t* sp_lo = sp_hi + E2_SPLANE;
    half_t* xb = sp_lo + E2_SPLANE;                  // feature rows of conv1's edge outputs: [pseudo-row][16 rows][8 halves] + pad
    if ((int)blockIdx.x * EG_S >= n_sites) return;
    // (every per-lane constant is re-derived from an opaque copy of the thread index at the top of each pass: hoisted out of the
    //  pass loop, the DMA source arithmetic and the weight pointers of all layers would sit in registers for the whole launch)
    int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int li = lane & 15, lk = lane >> 4;
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };
    const uint32_t lds_mset = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)mset;

    // ---- site descriptors of the NEXT pass: a chain of three dependent loads, one link per layer of the current pass --------
    struct Desc {
        Site st;
        RInfo ri;
        int bs, valid;
    } nd;
    auto desc_a = [&](const int s0) __attribute__((always_inline)) {
        const int sc = min(s0, max(n_sites - 1, 0));                      // past the end: the last pass's sites again, dropped below
        const int nvalid = s0 < n_sites ? min(EG_S, n_sites - s0) : 0;
        const int t = tid & (EG_S - 1);
        nd.valid = t < nvalid;
        nd.st = sites[min(sc + t, max(n_sites - 1, 0))];  // pad slots repeat the last site; their results are dropped
    };
    auto desc_b = [&]() __attribute__((always_inline)) { nd.ri = rinfo[nd.st.read_idx]; };
    auto desc_c = [&]() __attribute__((always_inline)) { nd.bs = bases[nd.ri.base_off + nd.st.qoff]; };
    auto desc_d = [&](const int s0, E2Site* si, int32_t (*rt)[E2_MROWS]) __attribute__((always_inline)) {
        if (tid < EG_S) {
            E2Site es;
            es.bo = nd.ri.base_off;
            es.L = nd.ri.len;
            es.view = nd.bs == 2;
            es.off = es.view ? nd.ri.len - 1 - nd.st.qoff : nd.st.qoff;
            es.vrow = (int64_t)es.view * mp.view_rows + nd.ri.map_off + TR_PAD;
            es.valid = nd.valid;
            si[tid] = es;
            if (es.valid) e4row[s0 + tid] = (int32_t)(es.vrow + es.off - 215);
        }
        // map rows of the pass's DMA pieces: thread 96 l + r holds the descriptor of site r & 31 (every thread loaded one)
        if (tid < 3 * E2_MROWS) {
            const int l = tid / E2_MROWS, j = (tid - l * E2_MROWS) >> 5;
            const int view = nd.bs == 2, off = view ? nd.ri.len - 1 - nd.st.qoff : nd.st.qoff;
            const int r0 = l == 0 ? G::R1 : l == 1 ? G::R2 : G::R3, step = 2 << l;
            const int delta = j == 0 ? G::LEFT : r0 + (j - 1) * step;
            (&rt[0][0])[tid] = (int32_t)((int64_t)view * mp.view_rows + nd.ri.map_off + TR_PAD + off + delta);
        }
    };
    // feature rows of conv1's first / last output: K1 rows per pseudo-row, the one on the zero padding all zeros.
    // Thread t of NT: rows [NR (t % PER), + NR) of pseudo-row t / PER (PER = 16 / NR) -- the site's descriptor is read once, all loads
    // are issued before the first is used (clamped addresses, no branches); `between()` runs after the loads are issued.
    auto build_rows = [&](const E2Site* si, const int t, auto nt_, auto between) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_)::value, NR = EG_M * EG_XROWS / NT, PER = EG_XROWS / NR;
        static_assert(NR >= 1 && NR <= EG_XROWS && EG_XROWS % NR == 0, "a thread's rows lie in one pseudo-row");
        const int row = t / PER, t0 = (t - row * PER) * NR;
        const int side = row >= EG_S, site = row - side * EG_S;
        const E2Site es = si[site];
        uint32_t b[NR], k[NR];
        bool live[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int tt = t0 + u;
            const int x = es.off + (side ? G::X_RIGHT : G::X_LEFT) + tt;
            const bool is_pad = side ? tt == K1 - 1 : tt == 0;
            live[u] = tt < K1 && !is_pad && x >= 0 && x < es.L;
            const int xc = min(max(x, 0), es.L - 1);
            const int64_t j = es.bo + (es.view ? es.L - 1 - xc : xc);
            b[u] = bases[j];
            k[u] = kin[j];
        }
        between();
#pragma unroll
        for (int u = 0; u < NR; ++u)
            *reinterpret_cast<uint4*>(xb + row * E2_XRS + (t0 + u) * TR_WRS) = feature_row(live[u] ? (int)b[u] : -1, k[u], es.view);
    };

    // ---- map rows of layer LAYER into map set `set`: piece w0 + nw t, t = 0 .. (LDS-DMA: 64 lanes x 16 B = 1 KB of the set = 1.94
    // rows of 33 chunks -- a row's 512 bytes [hi | lo] as they lie in HBM + its pad, which re-reads chunk 31).  `dma_src` reads the
    // site's descriptor, `dma_issue` (a block of the MFMA stream later) forms the address and issues the piece -----------------------
    struct DmaSrc {
        int32_t rn;    // row of the layer's map
        uint32_t ch;   // 16-byte chunk of that row
    };
    // chunk c of a map set = chunk c % 33 of set row c / 33 (chunk 32 = the pad: chunk 31 again); c < 3 200, so c / 33 = c * 1986 >> 16
    auto dma_src = [&](auto ltag, const int32_t* rt, const int i) __attribute__((always_inline)) {
        using L = E2L<K1, decltype(ltag)::value>;
        const uint32_t c = (uint32_t)min(i, L::NI - 1) * 64u + (uint32_t)lane;
        const uint32_t row = (c * 1986u) >> 16;
        const uint32_t ch = min(c - 33u * row, 31u);
        return DmaSrc{rt[min(row, (uint32_t)(L::ROWS - 1))], ch};
    };
    auto dma_issue = [&](auto ltag, const DmaSrc& d, const int set, const int i) __attribute__((always_inline)) {
        using L = E2L<K1, decltype(ltag)::value>;
        const char* __restrict__ map = reinterpret_cast<const char*>(mp.e[decltype(ltag)::value - 2]);
        if (i < L::NI) {  // wave-uniform
            const char* src = map + (uint64_t)((uint32_t)d.rn * 32u + d.ch) * 16u;  // (a group's maps have < 2^27 rows: hm_engine.cpp caps group_bases)
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_mset + (uint32_t)((set * E2_MSET) * 2 + i * 1024));
            uint32_t km;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %2, off\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(km)
                : "s"(dst), "v"(src));
        }
    };
    // all pieces of a wave at once (first pass)
    auto dma_maps = [&](auto ltag, const int32_t* rt, const int set, const int w0, auto nw_) __attribute__((always_inline)) {
        using L = E2L<K1, decltype(ltag)::value>;
        constexpr int nw = decltype(nw_)::value;
#pragma unroll
        for (int t = 0; t < (L::NI + nw - 1) / nw; ++t) dma_issue(ltag, dma_src(ltag, rt, w0 + nw * t), set, w0 + nw * t);
    };
    using L2t = std::integral_constant<int, 2>;
    using L3t = std::integral_constant<int, 3>;
    using L4t = std::integral_constant<int, 4>;
    using NW8 = std::integral_constant<int, 8>;

    // ---- conv1 of the edge rows: bn0 folded into the weights, exact fp16 operand, the weights' hi and lo halves stacked along K
    // (ConvH<..., KSTACK = K1>): 7 k-blocks x 4 m-tiles, one product each; all its weights are requested a layer ahead -----------------
    constexpr int KB1 = (2 * K1 + 3) / 4 * 4 * 8 / 32;
    struct Head1 {
        half8 w[KB1];
        float4 bz;
    };
    auto load_head1 = [&](Head1& h) __attribute__((always_inline)) {
        const char* wp = reinterpret_cast<const char*>(W.c1f) + (size_t)wave * (KB1 * 1024);
        const uint32_t lo = (uint32_t)lane * 16u;
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) h.w[kb] = *reinterpret_cast<const half8*>(wp + kb * 1024 + lo);
        const int col = wave * 16 + 4 * lk;
        h.bz = *reinterpret_cast<const float4*>(W.c1f_bias + col);
    };
    auto conv1 = [&](const Head1& h) __attribute__((always_inline)) {
        // the folded bn0 constant must not count for the tap on the zero padding (c1f_corr, hm_weights.cpp): first output row (left
        // chains), last output row (right chains); requested here, used in the epilogue
        const float4 c0 = *reinterpret_cast<const float4*>(W.c1f_corr + wave * 16 + 4 * lk);
        const float4 c1 = *reinterpret_cast<const float4*>(W.c1f_corr + 128 + wave * 16 + 4 * lk);
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{h.bz.x, h.bz.y, h.bz.z, h.bz.w};
        // window row of tap slot 4 kb + lk: slots >= K1 walk the same rows again with the lo halves
        auto stack_off = [&](int kb) __attribute__((always_inline)) {
            int t = 4 * kb + lk;
            t = t >= 2 * K1 ? 0 : t >= K1 ? t - K1 : t;
            return t * TR_WRS;
        };
        const int a0 = li * E2_XRS;
        half8 x[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[0][i] = *reinterpret_cast<const half8*>(xb + a0 + i * (16 * E2_XRS) + stack_off(0));
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) {
            if (kb + 1 < KB1) {
                const int bo = stack_off(kb + 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) x[(kb + 1) & 1][i] = *reinterpret_cast<const half8*>(xb + a0 + i * (16 * E2_XRS) + bo);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h.w[kb], x[kb & 1][i], acc[i], 0, 0, 0);
        }
        const EpiSpecC1 epi{sp_hi, sp_lo, nullptr, c0, c1};
#pragma unroll
        for (int i = 0; i < 4; ++i) epi(i * 16 + li, wave * 16 + 4 * lk, acc[i]);
    };

    // ---- conv2..conv4 ------------------------------------------------------------------------------------------------------------------
    // a wave's weights of one layer: fragments [n-tile][k-block][plane][lane] half8 of n-tile `wave` (conv4 has 6: waves 6, 7 hold
    // copies they never use), and its four biases
    struct LW {
        half8 w[12][2];
        float4 bz;
    };
    // (uniform base + this lane's 32-bit byte offset: the loads take the scalar-base form, no 64-bit address arithmetic per load)
    auto wptr = [&](auto ltag) __attribute__((always_inline)) {
        constexpr int LAYER = decltype(ltag)::value;
        const int nt = LAYER == 4 ? min(wave, C4_CH / 16 - 1) : wave;
        return reinterpret_cast<const char*>(wf(LAYER - 1)) + (size_t)nt * (12 * 128 * 16);
    };
    auto load_kb = [&](auto ltag, auto kb_, LW& lw) __attribute__((always_inline)) {
        constexpr int LAYER = decltype(ltag)::value, kb = decltype(kb_)::value;
        const char* wp = wptr(ltag) + kb * 2048;
        const uint32_t lo = (uint32_t)lane * 16u;
        lw.w[kb][0] = *reinterpret_cast<const half8*>(wp + lo);
        lw.w[kb][1] = *reinterpret_cast<const half8*>(wp + 1024 + lo);
        if constexpr (kb == 0) lw.bz = *reinterpret_cast<const float4*>(W.bias[LAYER - 1] + (LAYER == 4 ? min(wave, C4_CH / 16 - 1) : wave) * 16 + 4 * lk);
    };
    // one layer on this wave's n-tile.  `slide(kb)` is called when the registers of k-block kb - E2_AHEAD are free: the caller
    // requests the NEXT layer's k-block kb there.  `job(b)` is called once per block (the DMA pieces ride there).
    auto layer = [&](auto ltag, const LW& lw, const half_t* mx, f32x4 (&acc)[4], auto slide, auto job) __attribute__((always_inline)) {
        constexpr int LAYER = decltype(ltag)::value;
        using L = E2L<K1, LAYER>;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{lw.bz.x, lw.bz.y, lw.bz.z, lw.bz.w};
        const int a0 = li * TR_RS + 8 * lk;  // this lane's row of m-tile 0, its 8 K elements of a 32-channel block (previous layer's planes)
        const int am = li * E2_MRS + 8 * lk; // the same in a map set (rows of [hi | lo | pad])
        constexpr int LA = 2, NX = LA + 1;  // operand blocks requested ahead of the MFMAs / ring slots
        half8 x[NX][2][2];
        auto reads = [&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value, side = L::side_of(b), kb = L::kb_of(b), tap = kb / 4, q = kb % 4;
            constexpr int kind = L::kind(side, tap);
            static_assert(kind != SRC_ZERO, "live taps only");
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (kind == SRC_MAP) {
                    const int o = am + ((2 * side + i) * 16 + 32 * L::map_j(side, tap)) * E2_MRS + 32 * q;
                    x[b % NX][i][0] = *reinterpret_cast<const half8*>(mx + o);
                    x[b % NX][i][1] = *reinterpret_cast<const half8*>(mx + o + 128);
                } else {
                    const int o = a0 + ((2 * side + i) * 16) * TR_RS + 32 * q;
                    x[b % NX][i][0] = *reinterpret_cast<const half8*>(sp_hi + o);
                    x[b % NX][i][1] = *reinterpret_cast<const half8*>(sp_lo + o);
                }
            }
        };
        e2_for<0, E2_AHEAD>(slide);
        e2_for<0, LA>(reads);
        e2_for<0, L::NB>([&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value, side = L::side_of(b), kb = L::kb_of(b);
            if constexpr (b + LA < L::NB) reads(std::integral_constant<int, (b + LA < L::NB ? b + LA : 0)>{});
            job(b_);
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[2 * side + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lw.w[kb][pr == 2 ? 1 : 0], x[b % NX][i][pr == 1 ? 1 : 0],
                                                                               acc[2 * side + i], 0, 0, 0);
            constexpr int dead = L::dies_at(b);
            constexpr bool sl = dead >= 0 && dead + E2_AHEAD < 12;
            if constexpr (sl) slide(std::integral_constant<int, (sl ? dead + E2_AHEAD : 0)>{});
            {   // the LDS reads of the block two ahead (and the slid-in weight loads) ride between this block's MFMAs
                constexpr int NRD = b + LA < L::NB ? 4 : 0;
#pragma unroll
                for (int r = 0; r < NRD; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (sl) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

#ifdef HM_TRUNK_STAMP
    unsigned long long ets[16], eacc[16] = {};
    unsigned long long en_it = 0, e_prev = 0;
    const bool est_on = blockIdx.x == 0;
#define ETS(i) do { if (est_on) ets[i] = hm_stamp(); } while (0)
#else
#define ETS(i)
#endif

    // ---- first pass of this workgroup: prepared by everybody ------------------------------------------------------------------------
    Head1 h1;
    LW lw;
    {
        const int s0 = blockIdx.x * EG_S;
        desc_a(s0);
        desc_b();
        desc_c();
        desc_d(s0, sinfo2[0], rowtab[0]);
        load_head1(h1);
        __syncthreads();
        build_rows(sinfo2[0], tid, std::integral_constant<int, NW * 64>{}, [] {});
        e2_vmwait<0>();
        dma_maps(L2t{}, rowtab[0][0], 0, wave, NW8{});
        e2_vmwait<0>();
    }
    int cur = 0, sel = 0;
    for (int s0 = blockIdx.x * EG_S; s0 < n_sites; s0 += gridDim.x * EG_S) {
        const int nvalid = min(EG_S, n_sites - s0);
        const E2Site* si_next = sinfo2[cur ^ 1];
        const half_t* mA = mset + (size_t)sel * E2_MSET;          // conv2 and conv4 read this set, conv3 the other
        const half_t* mB = mset + (size_t)(sel ^ 1) * E2_MSET;
        tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
        e2_barrier();  // feature rows, descriptors and conv2's map rows of this pass are in LDS; the previous pass is done with the planes
        ETS(0);
#ifdef HM_TRUNK_STAMP
        if (est_on && e_prev) eacc[9] += ets[0] - e_prev;
#endif
        const int sn = s0 + gridDim.x * EG_S;
        desc_a(sn);
        // conv2's weights: requested here, behind nothing but conv1's (the vector-memory path idles through conv1 otherwise)
        e2_for<0, 12>([&](auto k_) __attribute__((always_inline)) { load_kb(L2t{}, k_, lw); });
        conv1(h1);
        ETS(1);
        e2_vmwait<0>();
        e2_barrier();
        ETS(2);
        f32x4 acc[4];
        // ---- conv2 (conv3's map rows and weights arrive meanwhile) ----
        desc_b();
        {
            LW ln;
            DmaSrc ds;
            layer(L2t{}, lw, mA, acc, [&](auto k_) __attribute__((always_inline)) { load_kb(L3t{}, k_, ln); },
                  [&](auto b_) __attribute__((always_inline)) {  // conv3's map rows: a piece per block, its descriptor read a block earlier
                      constexpr int b = decltype(b_)::value, NP = (E2L<K1, 3>::NI + NW - 1) / NW;
                      static_assert(NP < E2L<K1, 2>::NB, "the pieces fit the layer's blocks");
                      if constexpr (b >= 1 && b - 1 < NP) dma_issue(L3t{}, ds, sel ^ 1, wave + NW * (b - 1));
                      if constexpr (b < NP) ds = dma_src(L3t{}, rowtab[cur][1], wave + NW * b);
                  });
            ETS(3);
            e2_vmwait<0>();
            e2_barrier();
            ETS(4);
            const EpiSpec epi{sp_hi, sp_lo, nullptr};
#pragma unroll
            for (int i = 0; i < 4; ++i) epi(i * 16 + li, wave * 16 + 4 * lk, acc[i]);
            lw = ln;
        }
        e2_barrier();
        ETS(5);
        // ---- conv3 (conv4's map rows and weights) ----
        desc_c();
        {
            LW ln;
            DmaSrc ds;
            layer(L3t{}, lw, mB, acc, [&](auto k_) __attribute__((always_inline)) { load_kb(L4t{}, k_, ln); },
                  [&](auto b_) __attribute__((always_inline)) {  // conv4's map rows
                      constexpr int b = decltype(b_)::value, NP = (E2L<K1, 4>::NI + NW - 1) / NW;
                      static_assert(NP < E2L<K1, 3>::NB, "the pieces fit the layer's blocks");
                      if constexpr (b >= 1 && b - 1 < NP) dma_issue(L4t{}, ds, sel, wave + NW * (b - 1));
                      if constexpr (b < NP) ds = dma_src(L4t{}, rowtab[cur][2], wave + NW * b);
                  });
            ETS(6);
            e2_vmwait<0>();
            e2_barrier();
            ETS(7);
            const EpiSpec epi{sp_hi, sp_lo, nullptr};
#pragma unroll
            for (int i = 0; i < 4; ++i) epi(i * 16 + li, wave * 16 + 4 * lk, acc[i]);
            desc_d(sn, sinfo2[cur ^ 1], rowtab[cur ^ 1]);
            lw = ln;
        }
        e2_barrier();
        ETS(8);
        // ---- conv4 (96 channels = waves 0..5, which also request the next pass's conv2 map rows and conv1 weights); waves 6 and 7
        // build the next pass's feature rows ----
        if (wave >= C4_CH / 16) {
            build_rows(si_next, tid - 64 * (C4_CH / 16), std::integral_constant<int, 128>{}, [&]() __attribute__((always_inline)) { load_head1(h1); });
        } else {
            DmaSrc ds;
            constexpr int NWC = C4_CH / 16;
            layer(L4t{}, lw, mA, acc, [&](auto k_) __attribute__((always_inline)) {
                if constexpr (decltype(k_)::value == 11) load_head1(h1);  // (behind the layer's last free: conv1's weights are not live beside conv4's)
            }, [&](auto b_) __attribute__((always_inline)) {  // the next pass's conv2 map rows
                constexpr int b = decltype(b_)::value, NP = (E2L<K1, 2>::NI + NWC - 1) / NWC;
                static_assert(NP < E2L<K1, 4>::NB, "the pieces fit the layer's blocks");
                if constexpr (b >= 1 && b - 1 < NP) dma_issue(L2t{}, ds, sel ^ 1, wave + NWC * (b - 1));
                if constexpr (b < NP) ds = dma_src(L2t{}, rowtab[cur ^ 1][0], wave + NWC * b);
            });
            const EpiEdgeOut epi{reinterpret_cast<half_t*>(edge4) + (size_t)s0 * (4 * C4_CH), nullptr, nvalid};
#pragma unroll
            for (int i = 0; i < 4; ++i) epi(i * 16 + li, wave * 16 + 4 * lk, acc[i]);
        }
        ETS(9);
        e2_vmwait<0>();
        cur ^= 1;
        sel ^= 1;
#ifdef HM_TRUNK_STAMP
        if (est_on) {
            for (int i = 0; i < 9; ++i) eacc[i] += ets[i + 1] - ets[i];
            e_prev = ets[9];
            ++en_it;
        }
#endif
    }
#ifdef HM_TRUNK_STAMP
    if (est_on && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 10; ++i) atomicAdd(&g_edge2_stamp[threadIdx.x >> 6][i], eacc[i]);
        atomicAdd(&g_edge2_stamp[threadIdx.x >> 6][10], en_it);
    }
#endif
#undef ETS
}

void launch_edge2(hipStream_t st, int k1, const SiteRange& sr, const RInfo* rinfo, const uint8_t* bases, const uint32_t* kin,
                  const CtxWeights& w, const TrunkMaps& maps, uint16_t* edge4, int32_t* e4row, int grid) {
    if (sr.cap <= 0) return;
    const dim3 g(sr.totals ? grid : max(1, min((sr.cap + EG_S - 1) / EG_S, grid))), b(512);
    if (k1 == 11) hipLaunchKernelGGL(edge2_kernel<11>, g, b, 0, st, sr, rinfo, bases, kin, w, maps, edge4, e4row);
    else hipLaunchKernelGGL(edge2_kernel<13>, g, b, 0, st, sr, rinfo, bases, kin, w, maps, edge4, e4row);
}

}  // namespace hm
