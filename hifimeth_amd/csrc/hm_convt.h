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
// WLO = false: the layer runs with plain fp16 WEIGHTS -- its w_lo x_hi product is dropped and the lo plane's fragments are not loaded
// (engine option precision = 2, for the layers that hold BASELINE.json configs[4]'s bar: conv8 and fc1)
template <int CIN_, int KT_, int IRS_, bool WLO_ = true>
struct TCfg {
    static constexpr int CIN = CIN_, KT = KT_, IRS = IRS_;
    static constexpr bool WLO = WLO_;
    static constexpr int NPR = WLO_ ? 3 : 2;
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
template <bool LO = true, int KB, int NTR>
__device__ __forceinline__ void tw_load(const half_t* __restrict__ wfrag, const int (&nt)[NTR], int lane, TW<KB, NTR>& W) {
    const half8* wp = reinterpret_cast<const half8*>(wfrag) + lane;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int j = 0; j < NTR; ++j) {
            W.w[kb][j][0] = wp[(size_t)(nt[j] * KB + kb) * 128];
            if constexpr (LO) W.w[kb][j][1] = wp[(size_t)(nt[j] * KB + kb) * 128 + 64];
        }
}

// One group of a wave's schedule: GP m-tiles from TP on which the wave's first NJ resident n-tiles run (NJ = 2, a pair: they share
// the activation reads; NJ = 4 for fc1 in the head kernel) and GS m-tiles from TS on which only resident n-tile 0 runs.  A group's
// NJ GP + GS accumulators are visited round-robin, product by product.  (How many accumulators a group has does not set its rate:
// a chain of v_mfma_f32_16x16x32_f16 onto ONE accumulator issues every 16 cycles -- tools/micro/mfma_chain.hip.)
template <int TP_, int GP_, int TS_, int GS_, int NJ_ = 2>
struct TG {
    static constexpr int TP = TP_, GP = GP_, TS = TS_, GS = GS_, NJ = NJ_;
    static constexpr int NT = GP_ + GS_, NA = NJ_ * GP_ + GS_;
    static constexpr int tile(int i) { return i < GP_ ? TP_ + i : TS_ + (i - GP_); }   // i-th tile of the group
    static constexpr int acc_tile(int a) { return a < NJ_ * GP_ ? a / NJ_ : GP_ + (a - NJ_ * GP_); }   // tile index (in the group) of accumulator a
    static constexpr int acc_j(int a) { return a < NJ_ * GP_ ? a % NJ_ : 0; }                          // resident n-tile of accumulator a
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
// previous group's epilogue and `hook(block)` -- the caller's slot for work that should ride between the MFMAs (the LDS-DMA
// gather of the next group of sites).
// Activation operands come through a RING of NS half8 registers: a block's reads are issued as early as the ring has room
// for them, at most LA blocks ahead.
//
// Epilogue (round 4).  tools/micro/tconv_ablate.hip showed that round 3's epilogue -- 26 vector instructions + 2 LDS stores per
// accumulator tile, all of a tile's inside ONE k-block, 3-4 per MFMA -- was not hidden at all: it cost its full issue time, 17 % of
// a call (an MFMA 16x16x32 holds the issue port 8 of its 16 cycles: two 4-cycle instructions per MFMA are free, the third is
// not).  Now (a) the rows a tile's results go to are computed ONCE per tile in the prologue, together with the operand rows
// (one division per tile instead of one per tile and one per accumulator), (b) an accumulator's epilogue is two STAGES --
// compute (ReLU, split) and store -- that ride in different k-blocks, and (c) a block's vector instructions are dealt evenly
// over ALL its MFMAs.
// Epi: struct St; row(site, p, m) -> offset of the output row; s0(acc, St&) compute; s1(off, col, St&) stores; NV0 / NV1 = vector
// instructions the scheduler sees in the two stages, NW = stores of stage 1, WMASK = their scheduling class (0x200 LDS store,
// 0x040 vector-memory store).  ncol[j]: first channel of resident n-tile j.
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
    static constexpr int mfmas() { int n = 0; for (int g = 0; g < NG; ++g) n += nas[g] * C::NPR * KB; return n; }
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
        // per tile of the schedule: this lane's first row in the input planes and the row its results go to (Epi's units).  Only the
        // first group's are computed here; a later group's ride between the MFMAs of the group before it (they are needed when
        // that group's last blocks prefetch) -- less prologue in front of the first MFMA, fewer registers live across the call
        int aoff[NTILES];
        int ooff[NTILES];
        auto tile_addr = [&](auto g_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            using G = Grp<g>;
            tstatic_for<0, G::NT>([&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value, tile = G::tile(i);
                int m = tile * 16 + li;
                if constexpr ((tile + 1) * 16 > RM::M) m = m < RM::M ? m : RM::M - 1;  // ragged last tile: re-read the last valid row
                const int site = m / RM::LOUT, p = m - site * RM::LOUT;
                aoff[tile_base(g) + i] = site * RM::ISS + 2 * p * C::IRS;
                ooff[tile_base(g) + i] = Epi::row(site, p, m);
            });
        };
        tile_addr(std::integral_constant<int, 0>{});
        f32x4 acc[2][AMAX];
        half8 x[NS];
        typename Epi::St est[AMAX];

        auto reads = [&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value, g = b / KB, kb = b % KB, s0 = slot_base(b);
            const int off = C::lane_off(kb, lk);
#ifdef HM_ABL_NOREAD   // tools/micro ablation: operands stay what the first blocks read
            if constexpr (b >= issued(0)) return;
#endif
#pragma unroll
            for (int i = 0; i < nts[g]; ++i) {
                x[(s0 + 2 * i) % NS] = *reinterpret_cast<const half8*>(in_hi + aoff[tile_base(g) + i] + off);
                x[(s0 + 2 * i + 1) % NS] = *reinterpret_cast<const half8*>(in_lo + aoff[tile_base(g) + i] + off);
            }
        };
        // stage s of accumulator a of group g
        auto stage = [&](auto g_, auto a_, auto s_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value, a = decltype(a_)::value, st = decltype(s_)::value;
            using G = Grp<g>;
            constexpr int ti = G::acc_tile(a), tile = G::tile(ti), j = G::acc_j(a);
#ifdef HM_ABL_NOEPI    // tools/micro ablation: the accumulator is kept alive, nothing is computed or stored from it
            if constexpr (st == 0) { const f32x4 keep = acc[g & 1][a]; asm volatile("" ::"v"(keep)); }
#else
            if constexpr (st == 0) {
                epi.s0(acc[g & 1][a], est[a]);
            } else {
                int col = ncol[j] + 4 * lk;
                // ragged last tile: rows past M write their (meaningless) values into the 16-byte pad behind the channels
                // of the last valid row instead of being branched around -- a branch here would cut the block in two
                if constexpr ((tile + 1) * 16 > RM::M) col = tile * 16 + li < RM::M ? col : Epi::PADCOL;
                epi.s1(ooff[tile_base(g) + ti], col, est[a]);
            }
#endif
        };
        tstatic_for<0, issued(0)>(reads);

        tstatic_for<0, NB>([&](auto c_) __attribute__((always_inline)) {
            constexpr int c = decltype(c_)::value, g = c / KB, kb = c % KB, s0 = slot_base(c);
            using G = Grp<g>;
            if constexpr (kb == 0) {
                float4 bz[WT::NTR];
                constexpr int NBZ = G::GP > 0 ? G::NJ : 1;  // resident n-tiles this group uses
#pragma unroll
                for (int j = 0; j < NBZ; ++j) {
                    if constexpr (std::is_pointer_v<Bias>) bz[j] = *reinterpret_cast<const float4*>(bias + ncol[j] + 4 * lk);
                    else bz[j] = bias(j);
                }
                tstatic_for<0, G::NA>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value, j = G::acc_j(a);
                    acc[g & 1][a] = f32x4{bz[j].x, bz[j].y, bz[j].z, bz[j].w};
                });
            }
            if constexpr (kb == 0 && g + 1 < NG) tile_addr(std::integral_constant<int, (g + 1 < NG ? g + 1 : 0)>{});
            constexpr int P0 = issued(c > 0 ? c - 1 : 0), P1 = issued(c);
            if constexpr (c > 0) tstatic_for<P0, P1>(reads);
            // product-major over the group's accumulators: an accumulator is revisited NA MFMAs later
            tstatic_for<0, C::NPR>([&](auto pr_) __attribute__((always_inline)) {
                constexpr int pr = decltype(pr_)::value;
                tstatic_for<0, G::NA>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value, i = G::acc_tile(a), j = G::acc_j(a);
                    acc[g & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.w[kb][j][pr == 2 ? 1 : 0], x[(s0 + 2 * i + (pr == 1 ? 1 : 0)) % NS],
                                                                           acc[g & 1][a], 0, 0, 0);
                });
            });
            // the previous group's accumulators leave between this group's MFMAs: 2 stages each, dealt over its k-blocks
            constexpr int NAP = g > 0 ? nas[g > 0 ? g - 1 : 0] : 0, NST = 2 * NAP;
            constexpr int E0 = kb * NST / KB, E1 = (kb + 1) * NST / KB;
            if constexpr (g > 0)
                tstatic_for<E0, E1>([&](auto e_) __attribute__((always_inline)) {
                    constexpr int e = decltype(e_)::value;
                    stage(std::integral_constant<int, (g > 0 ? g - 1 : 0)>{}, std::integral_constant<int, e / 2>{}, std::integral_constant<int, e % 2>{});
                });
            hook(c_);
            {   // interleave plan: this step's LDS reads behind the first MFMAs, the stages' vector instructions dealt over all MFMAs,
                // their LDS stores behind the last ones
                constexpr int NRD = [&]() constexpr { int n = 0; for (int b = (c > 0 ? P0 : P1); b < P1; ++b) n += nreads(b); return n; }();
                constexpr int ND = NRD + (kb == 0 && std::is_pointer_v<Bias> ? (G::GP > 0 ? G::NJ : 1) : 0) + HDS;
                constexpr int NM = C::NPR * G::NA;
                constexpr int NVA = kb == 0 && g + 1 < NG ? 9 * nts[g + 1 < NG ? g + 1 : 0] : 0;  // the next group's row offsets
                constexpr int NV = NVA + [&]() constexpr { int n = 0; for (int e = E0; e < E1; ++e) n += e % 2 == 0 ? Epi::NV0 : Epi::NV1; return g > 0 ? n : 0; }();
                constexpr int NWR = [&]() constexpr { int n = 0; for (int e = E0; e < E1; ++e) n += e % 2 == 1 ? Epi::NW : 0; return g > 0 ? n : 0; }();
                tstatic_for<0, NM>([&](auto q_) __attribute__((always_inline)) {
                    constexpr int q = decltype(q_)::value;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (q < ND) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    constexpr int nv = (q + 1) * NV / NM - q * NV / NM;
                    if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
                    if constexpr (q >= NM - NWR) __builtin_amdgcn_sched_group_barrier(Epi::WMASK, 1, 0);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // the last group's accumulators: nothing to hide behind -- stage-major, so that the tiles' dependent chains (accumulator
        // read, ReLU, convert, residual, store) overlap each other
        tstatic_for<0, nas[NG - 1]>([&](auto a_) __attribute__((always_inline)) { stage(std::integral_constant<int, NG - 1>{}, a_, std::integral_constant<int, 0>{}); });
        tstatic_for<0, nas[NG - 1]>([&](auto a_) __attribute__((always_inline)) { stage(std::integral_constant<int, NG - 1>{}, a_, std::integral_constant<int, 1>{}); });
    }
};

// LDS-only barrier: orders this workgroup's LDS traffic, leaves vector-memory operations (the gather) in flight
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// s_waitcnt vmcnt(0) as a BUILTIN (simm16: vmcnt 0, expcnt 7, lgkmcnt 15 = no wait): the compiler's own wait-count bookkeeping
// sees it.  Written as inline asm it would not -- the weights requested before the pass loop would still count as pending in
// the compiler's model, and the waits it then places in front of their first uses INSIDE the loop would, in steady state,
// wait for whatever is in flight there: the gather.
__device__ __forceinline__ void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// s_waitcnt vmcnt(N): all but the wave's N youngest vector-memory operations (loads, stores, LDS-DMAs alike, in issue order) are done
template <int N>
__device__ __forceinline__ void e2_vmwait_n() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// Epilogue functors for TConv (hm_convt.h): an accumulator tile leaves in two stages -- s0 computes (ReLU, hi / lo split),
// s1 stores -- to a row whose offset row(site, p, m) TConv computes once per tile.  NV0 / NV1: vector instructions of the two
// stages that the scheduler's interleave plan counts (split_lo2 is inline asm: not counted), NW: LDS stores of s1.
// ReLU + split -> planes of S stacked sites, physical row p + 1 (row 0 of a site is zero padding)
template <int LOUT, int ORS, int OSS>
struct EpiStack {
    static constexpr int PADCOL = ORS - 8;  // the 16-byte pad behind a row's channels: where a ragged tile's surplus rows write
    static constexpr int NV0 = 6, NV1 = 1, NW = 2, WMASK = 0x200;
    struct St { half4 h, l; };
    half_t* hi;
    half_t* lo;
    static __device__ __forceinline__ int row(int site, int p, int) { return site * OSS + (p + 1) * ORS; }
    __device__ __forceinline__ void s0(const f32x4& acc, St& s) const { split4(acc, s.h, s.l); }
    __device__ __forceinline__ void s1(int off, int col, const St& s) const {
        *reinterpret_cast<half4*>(hi + off + col) = s.h;
        *reinterpret_cast<half4*>(lo + off + col) = s.l;
    }
};
// conv8's output for the batched fc1: rows 0..LOUT-1 of a site back to back
template <int ORS>
struct EpiRows {
    static constexpr int PADCOL = ORS - 8;
    static constexpr int NV0 = 6, NV1 = 1, NW = 2, WMASK = 0x200;
    struct St { half4 h, l; };
    half_t* hi;
    half_t* lo;
    static __device__ __forceinline__ int row(int, int, int m) { return m * ORS; }
    __device__ __forceinline__ void s0(const f32x4& acc, St& s) const { split4(acc, s.h, s.l); }
    __device__ __forceinline__ void s1(int off, int col, const St& s) const {
        *reinterpret_cast<half4*>(hi + off + col) = s.h;
        *reinterpret_cast<half4*>(lo + off + col) = s.l;
    }
};
template <int HRS>
struct EpiFc1R {  // ReLU, fp32 h[site][256] for the VALU fc2
    static constexpr int PADCOL = 0;
    static constexpr int NV0 = 4, NV1 = 1, NW = 1, WMASK = 0x200;
    struct St { float4 v; };
    float* out;
    static __device__ __forceinline__ int row(int, int, int m) { return m * HRS; }
    __device__ __forceinline__ void s0(const f32x4& acc, St& s) const { s.v = make_float4(relu1(acc[0]), relu1(acc[1]), relu1(acc[2]), relu1(acc[3])); }
    __device__ __forceinline__ void s1(int off, int col, const St& s) const { *reinterpret_cast<float4*>(out + off + col) = s.v; }
};

// the same ReLU + split, straight to global memory: conv6's rows in the split tail (hm_tail_s.hip) leave the chip as the planes
// the head kernel's flat LDS-DMA expects: [site][LOUT + 2 rows][ORS halves] per plane, padding rows and columns never written
template <int LOUT, int ORS, int OSS>
struct EpiGStack {
    static constexpr int PADCOL = ORS - 8;
    static constexpr int NV0 = 6, NV1 = 1, NW = 2, WMASK = 0x040;
    struct St { half4 h, l; };
    half_t* __restrict__ hi;   // this pass's first site in the hi plane (wave-uniform)
    half_t* __restrict__ lo;
    static __device__ __forceinline__ int row(int site, int p, int) { return site * OSS + (p + 1) * ORS; }
    __device__ __forceinline__ void s0(const f32x4& acc, St& s) const { split4(acc, s.h, s.l); }
    __device__ __forceinline__ void s1(int off, int col, const St& s) const {
        *reinterpret_cast<half4*>(hi + (unsigned)(off + col)) = s.h;
        *reinterpret_cast<half4*>(lo + (unsigned)(off + col)) = s.l;
    }
};

}  // namespace hm
