// hm_trunk3.hip -- the dense trunk as a SLIDING WINDOW (round 4; engine option trunk_impl = 3): conv1 .. conv4 over exactly 112 rows
// per tile and layer.
//
// trunk2_kernel (hm_trunk.hip) computes a tile of 112 view positions from scratch: 144 / 144 / 128 / 112 rows of conv1 .. conv4 -- the
// halo of the dilated taps (E(l)[x] reads E(l-1)[x, x + d, x + 2 d], d = 2^(l-1), so 16 + 8 + 4 rows to the RIGHT of the tile, + K1 - 1
// feature rows) is recomputed by the neighbour: 6 912 MFMAs per tile where 5 936 are new results, +16 %.  The kernel runs at the chip's
// power / clock limit: its time follows the number of MFMAs it issues, not the idle cycles between them, so the halo
// is the one lever that is worth its size.
//
// Here a workgroup walks a CONTIGUOUS run of tiles (consecutive tiles of a read are consecutive in the tile list) and keeps the
// right-hand rows of every layer for the next tile: the layers of a step are SKEWED, each lagging the one below it by its halo,
//     step at u:  E4 rows [u, u + 112)        from E3 rows [u,      u + 128) = 16 kept  + 112 new  (plane A rows 0 .. 127)
//                 E3 rows [u + 16, u + 128)   from E2 rows [u + 16, u + 136) =  8 kept  + 112 new  (plane B rows 0 .. 119)
//                 E2 rows [u + 24, u + 136)   from E1 rows [u + 24, u + 140) =  4 kept  + 112 new  (plane A rows 0 .. 115)
//                 E1 rows [u + 28, u + 140)   from feature rows [u + 28, u + 152)                  (rebuilt per step: 124 rows)
// so that every layer computes 112 rows = 7 m-tiles per step.  A plane row of a layer's input is a position, taps are rows m,
// m + d, m + 2 d exactly as before; what changes is where a layer's output goes (behind the kept rows) and four small LDS copies
// per step (plane A holds E1 and then E3, so E1's and E3's kept rows wait in side buffers H1 / H3 while the other uses the plane).
// Kept rows are garbage at the start of a run (a workgroup's first tile, a read's first tile): a WARM-UP step at u - 112 comes
// first, whose E4 rows go to a dump buffer and whose copy slots write only rows that are valid and untouched by the garbage (the
// edge chains do read E1 .. E3 at positions -199 .. -173 of a read: those rows are new rows of the read's warm-up step).
// Same products in the same order per accumulator as trunk2_kernel: byte-identical maps (tests/test_gpu_parity.py).
//
// CONSTANT steps.  E1[x] .. E4[x] see the feature rows x .. x + 12 / 16 / 24 / 40: where none of them lies inside the read (all
// feature rows zero) every layer's row is ONE row that depends on nothing but the weights.  That holds for all new rows of a step at
// u <= -152 (a read's warm-up step and its first tile) and at u >= len (the tiles behind the read's end): 3 - 4 of a read's ~ 142 steps.
// A workgroup therefore starts with a CALIBRATION step -- an ordinary step on all-zero feature rows whose stores go to a scratch
// buffer -- keeps the four constant rows it produces (its kept rows ARE E1's .. E3's, its last conv4 row E4's: same instructions,
// same bytes as a computed step's), and a constant step only stores: the E4 rows of the tile, the E1 .. E3 rows the lists flag, and
// the constants as the kept rows of the step behind it.  No warm-up step is computed at a read's start any more.
//
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98).
#include "hm_convh.h"
#ifdef HM_TRUNK_STAMP  // diagnostic build (make stamp): per-wave shader-clock phase sums of workgroup 0, read by tools/trunk3_stamps.py
#include "hm_stamp.h"
namespace hm { __device__ unsigned long long g_trunk3_stamp[8][24]; }
extern "C" int hm_debug_trunk3_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(hm::g_trunk3_stamp), sizeof(hm::g_trunk3_stamp)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[8][24];
        if (hipMemcpyToSymbol(HIP_SYMBOL(hm::g_trunk3_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#include "hm_convs.h"
#include "hm_edge.h"

namespace hm {

namespace {

constexpr int T3_M = TR_OWN;                        // rows every layer computes per step
constexpr int T3_H1 = 4, T3_H2 = 8, T3_H3 = 16;     // rows of E1 / E2 / E3 kept for the next step (the halo of conv2 / conv3 / conv4)
constexpr int T3_S3 = T3_H3, T3_S2 = T3_S3 + T3_H2, T3_S1 = T3_S2 + T3_H1;  // first NEW row of E3 / E2 / E1 relative to the step's u: 16, 24, 28
constexpr int T3_AROWS = T3_M + T3_H3, T3_BROWS = T3_M + T3_H2;             // plane A: E1 (116 rows) | E3 (128 rows); plane B: E2 (120 rows)
constexpr int T3_XROWS = 128;                       // feature rows of a step: 112 + K1 - 1 <= 124
constexpr int T3_LDS_HALVES = 2 * T3_AROWS * TR_RS + 2 * T3_BROWS * TR_RS + 2 * (T3_H1 + T3_H3) * TR_RS;
// a step's record (rowlist3_kernel): three lists of TR_OWN plane-row numbers (the E1 / E2 / E3 rows an edge chain reads), the list of
// the E4 rows some site's tail reads (T3_N4 entries, meaningful when there are at most that many) and their number
constexpr int T3_N4 = 64, T3_RL3 = 3 * TR_OWN, T3_RL = T3_RL3 + T3_N4 + 4;
constexpr int T3_RLW = 3 * 32 + T3_N4 / 4 + 1;      // words of a step's record in LDS: lists at 128-byte strides, the E4 list, its count
static_assert(T3_M % 16 == 0 && T3_RL % 4 == 0 && T3_RL / 4 <= 256 && T3_XROWS >= T3_M + 12, "tile plan");

// conv4 over the NEEDED rows only: row m of the list -> plane A row rows[m] (its taps 8 and 16 rows further on), map row rows[m].
// A site reads E4 at off - 215 + 16 s, s = 1 .. 23: at the site densities of CpG and CHG (~3 % of the positions) about half of a
// tile's 112 rows are never read, and four m-tiles of listed rows replace seven.
struct ListRows4 {
    static constexpr bool DENSE = false;
    static constexpr int M = T3_N4;
    const uint8_t* rows = nullptr;  // LDS
    template <class C>
    __device__ __forceinline__ int off(int m) const { return (int)rows[m] * C::IRS; }
};
struct EpiE4L {
    half_t* __restrict__ g;
    const uint8_t* rows;  // LDS
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        half4 h, l;
        split4(acc, h, l);
        const size_t o = (size_t)rows[m] * (2 * C4_CH) + col;
        *reinterpret_cast<half4*>(g + o) = h;
        *reinterpret_cast<half4*>(g + o + C4_CH) = l;
    }
};

struct EpiTrunk3 {  // ReLU + split -> LDS planes, row m of the NEW rows (the pointers are offset by the kept rows)
    half_t* hi;
    half_t* lo;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        half4 h, l;
        split4(acc, h, l);
        *reinterpret_cast<half4*>(hi + m * TR_RS + col) = h;
        *reinterpret_cast<half4*>(lo + m * TR_RS + col) = l;
    }
};
struct EpiE43 {  // conv4 rows leave the trunk already split, [hi 96 | lo 96]
    half_t* __restrict__ g;
    __device__ __forceinline__ void operator()(int m, int col, const f32x4& acc) const {
        half4 h, l;
        split4(acc, h, l);
        *reinterpret_cast<half4*>(g + (size_t)m * (2 * C4_CH) + col) = h;
        *reinterpret_cast<half4*>(g + (size_t)m * (2 * C4_CH) + C4_CH + col) = l;
    }
};
template <int NWV_>
struct CopyRows3 {
    static constexpr int NWV = NWV_, CS = TR_OWN / (2 * NWV_);  // NWV waves x 2 rows x CS slots = one list entry each
    const uint8_t* rows;  // LDS: this layer's list of TR_OWN plane-row numbers
    half_t* g;            // map row of plane row 0
};

// workgroup barrier that orders LDS traffic only: vector-memory operations (a step's map-row stores) stay in flight across it
__device__ __forceinline__ void t3_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// nrows whole rows (hi and lo plane) LDS -> LDS, 16 bytes per thread and round
template <int NROWS, int NT, int K = 0>
__device__ __forceinline__ void move_rows(half_t* dh, half_t* dl, const half_t* sh, const half_t* sl, int t) {
    constexpr int PER = NROWS * TR_RS / 8, N = 2 * PER;  // 16-byte chunks per plane, in all
    if constexpr (K * NT < N) {
        const int c = min(t + K * NT, N - 1), plane = c >= PER, o = (c - plane * PER) * 8;
        const uint4 v = *reinterpret_cast<const uint4*>((plane ? sl : sh) + o);
        move_rows<NROWS, NT, K + 1>(dh, dl, sh, sl, t);   // (the other rounds' reads are issued before this round's store)
        *reinterpret_cast<uint4*>((plane ? dl : dh) + o) = v;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// Which map rows do the edge chains read?  Per step and layer a list of TR_OWN plane-row numbers of the layer's NEW rows: the flagged
// ones first, the rest filled with the LAST new row (a row that is always valid and never touched by a warm-up step's garbage; storing
// it once more is harmless).  Block w < n_work: the step of tile w; block n_work + w: the warm-up step in front of a read's first
// tile (u - 112); block 2 n_work: the list of a warm-up step in the middle of a read (a workgroup's first tile): nothing but fill.
// One WAVE makes T3_RPB consecutive step records.  The kernel waits on three dependent loads per record -- tile, read, site flags -- so its
// rate is the number of records in flight: the wave issues each of the three for all its records before it uses the first (one-wave
// blocks: 32 per CU).  Lane r owns rows r and r + 64 of a record; a record is assembled in LDS and leaves as 101 coalesced words.
constexpr int T3_RPB = 4;
template <int K1>
__global__ __launch_bounds__(64) void rowlist3_kernel(const TrunkTile* __restrict__ tiles, int n_tiles, int n_work, int ctx,
                                                       const RInfo* __restrict__ rinfo, const uint8_t* __restrict__ sctx, int64_t n_bases,
                                                       uint8_t* __restrict__ rowlist) {
    using G = EdgeGeo<K1>;
    const int r = threadIdx.x;
    constexpr int first_new[3] = {T3_H1, T3_H2, T3_H3}, shift[3] = {T3_S1, T3_S2, T3_S3};
    constexpr int W0 = -176, WN = 528;   // the site flags of every position any lookup of a record can touch go through LDS once: [u - 176, u + 352)
    __shared__ __attribute__((aligned(4))) uint8_t rec[T3_RPB][T3_RL];
    __shared__ __attribute__((aligned(4))) uint8_t sf[T3_RPB][WN];
    static_assert(-G::R1 - 2 + T3_S1 >= W0 && -G::R2 - 4 + T3_S2 >= W0 && -G::R3 - 8 + T3_S3 >= W0 && T3_S1 + TR_OWN - 1 - G::LEFT < W0 + WN &&
                  215 - 16 * 23 >= W0 && TR_OWN - 1 + 215 - 16 < W0 + WN, "the window covers every lookup");
    // record blk < n_work: the step of tile blk; n_work + w: the warm-up step in front of a read's first tile; 2 n_work: nothing but fill
    int kind[T3_RPB];   // 0: nothing to do, 1: a list record, 2: the all-fill record
    TrunkTile tl[T3_RPB];
    RInfo ri[T3_RPB];
    bool warm[T3_RPB];
    int view[T3_RPB];
#pragma unroll
    for (int k = 0; k < T3_RPB; ++k) {
        const int blk = (int)blockIdx.x * T3_RPB + k;
        kind[k] = blk > 2 * n_work ? 0 : blk == 2 * n_work ? 2 : 1;
        warm[k] = blk >= n_work;
        const int w = min(warm[k] ? blk - n_work : blk, n_work - 1);
        view[k] = w >= n_tiles;
        tl[k] = tiles[view[k] ? w - n_tiles : w];
    }
    bool any = false;
#pragma unroll
    for (int k = 0; k < T3_RPB; ++k) {
        if (kind[k] == 1 && warm[k] && tl[k].u0 != -TR_PAD) kind[k] = 0;  // (only a read's first tile has a warm-up step that stores map rows)
        any |= kind[k] != 0;
    }
    if (!any) return;   // (most blocks of the warm-up range: one tile in ~140 is a read's first)
#pragma unroll
    for (int k = 0; k < T3_RPB; ++k) ri[k] = rinfo[tl[k].read_idx];
    // The site bytes of a record's window, four positions per lane and load: sctx[j] = context | strand << 2 of forward position j
    // (hm_kernels.h), so view position y is a site of this context and view iff its byte equals ctx | view << 2.  View 0 walks the
    // array forwards from a 4-aligned address (read offsets, tile starts and the window's are multiples of 4); view 1 walks it
    // backwards (y -> L - 1 - y: an unaligned word, bytes swapped).  Every record's loads are issued -- at addresses inside the array
    // whatever the record's kind -- before the first is used; bytes of positions outside the read are masked afterwards.
    constexpr int NQ = (WN / 4 + 63) / 64;
    static_assert(WN % 4 == 0 && (-W0) % 4 == 0 && TR_PAD % 4 == 0 && TR_OWN % 4 == 0, "windows start on multiples of 4");
    uint32_t sw[T3_RPB][NQ];
#pragma unroll
    for (int k = 0; k < T3_RPB; ++k) {
        const int L = ri[k].len, u = tl[k].u0 - (warm[k] ? TR_OWN : 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int y0 = u + W0 + 4 * (r + 64 * q);   // the word's first view position
            const int64_t a = view[k] ? ri[k].base_off + (L - 1 - y0 - 3) : ri[k].base_off + y0;
            if (a >= 0 && a + 4 <= n_bases) {
                uint32_t v;
                __builtin_memcpy(&v, sctx + a, 4);
                sw[k][q] = view[k] ? __builtin_bswap32(v) : v;
            } else {   // at the array's ends: byte by byte (positions outside the array are outside the read)
                uint32_t v = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int64_t j = view[k] ? ri[k].base_off + (L - 1 - y0 - t) : ri[k].base_off + y0 + t;
                    v |= (uint32_t)(j >= 0 && j < n_bases ? sctx[j] : (uint8_t)CTX_NONE) << (8 * t);
                }
                sw[k][q] = v;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < T3_RPB; ++k) {
        for (int i = r; i < T3_RL3; i += 64) rec[k][i] = (uint8_t)(first_new[i / TR_OWN] + TR_OWN - 1);
        const int L = ri[k].len, u = tl[k].u0 - (warm[k] ? TR_OWN : 0);
        const uint32_t key = (uint32_t)(ctx | (view[k] << 2)) * 0x01010101u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int y0 = u + W0 + 4 * (r + 64 * q);
            const uint32_t x = sw[k][q] ^ key;                                               // a zero byte where the position is a site
            uint32_t hit = (~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu)) >> 7;    // 1 in exactly those bytes
            const int lo = min(max(-y0, 0), 4), hi = min(max(L - y0, 0), 4);                 // bytes [lo, hi) lie inside the read
            const uint32_t m_hi = hi >= 4 ? 0xFFFFFFFFu : (1u << (8 * hi)) - 1u, m_lo = lo >= 4 ? 0xFFFFFFFFu : (1u << (8 * lo)) - 1u;
            hit &= m_hi & ~m_lo;
            if (r + 64 * q < WN / 4) reinterpret_cast<uint32_t*>(sf[k])[r + 64 * q] = hit;
        }
    }
    __syncthreads();
    const uint64_t below = (1ull << r) - 1;
#pragma unroll
    for (int k = 0; k < T3_RPB; ++k) {
        if (kind[k] == 0) continue;
        const int blk = (int)blockIdx.x * T3_RPB + k;
        int n4 = 255;
        if (kind[k] == 1) {
            const int u = tl[k].u0 - (warm[k] ? TR_OWN : 0);
            auto sat = [&](int y) __attribute__((always_inline)) { return (int)sf[k][y - u - W0]; };   // y - u in [W0, W0 + WN)
            // E4 rows some site of this context reads: position x is row s of the site at x + 215 - 16 s (s = 1 .. 23: rows 0 and 24 are the
            // edge kernel's).  Not for CHH (K1 = 13: at its density nearly every row is read) nor for warm-up steps (their E4 rows go to the
            // dump): a count of 255 keeps the step on all 112 rows.
            const bool want4 = K1 == 11 && !warm[k];
            int f[2] = {0, 0}, need4[2] = {0, 0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = r + 64 * h;
                if (row < TR_OWN) {
                    {
                        const int x = u + shift[0] + row;
                        f[h] |= (sat(x - G::LEFT) | sat(x - G::R1) | (G::PAD2 ? 0 : sat(x - G::R1 - 2))) << 0;
                    }
                    {
                        const int x = u + shift[1] + row;
                        f[h] |= (sat(x - G::LEFT) | sat(x - G::R2) | (G::PAD3 ? 0 : sat(x - G::R2 - 4))) << 1;
                    }
                    {
                        const int x = u + shift[2] + row;
                        f[h] |= (sat(x - G::LEFT) | sat(x - G::R3) | (G::PAD4 ? 0 : sat(x - G::R3 - 8))) << 2;
                    }
                    if (want4) {
#pragma unroll
                        for (int sidx = 1; sidx <= 23; ++sidx) need4[h] |= sat(u + row + 215 - 16 * sidx);
                    }
                }
            }
            // per layer: the flagged rows first, in ascending order (the rest of the list keeps the fill entry written above)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const uint64_t b0 = __ballot((f[0] >> l) & 1), b1 = __ballot((f[1] >> l) & 1);
                if ((f[0] >> l) & 1) rec[k][l * TR_OWN + __popcll(b0 & below)] = (uint8_t)(first_new[l] + r);
                if ((f[1] >> l) & 1) rec[k][l * TR_OWN + __popcll(b0) + __popcll(b1 & below)] = (uint8_t)(first_new[l] + r + 64);
            }
            // the E4 list: the needed rows in ascending order, filled up with the first of them (a row is stored twice: harmless)
            const uint64_t n0 = __ballot(need4[0]), n1 = __ballot(need4[1]);
            if (want4) n4 = __popcll(n0) + __popcll(n1);
            if (want4 && n4 <= T3_N4) {
                if (need4[0]) rec[k][T3_RL3 + __popcll(n0 & below)] = (uint8_t)r;
                if (need4[1]) rec[k][T3_RL3 + __popcll(n0) + __popcll(n1 & below)] = (uint8_t)(r + 64);
                __syncthreads();
                if (r >= n4) rec[k][T3_RL3 + r] = n4 > 0 ? rec[k][T3_RL3] : (uint8_t)0;   // (r < 64 = T3_N4)
            }
        }
        if (r == 0) *reinterpret_cast<uint32_t*>(&rec[k][T3_RL3 + T3_N4]) = (uint32_t)n4;
        __syncthreads();
        uint32_t* out = reinterpret_cast<uint32_t*>(rowlist + (size_t)blk * T3_RL);
        for (int i = r; i < T3_RL / 4; i += 64) out[i] = reinterpret_cast<const uint32_t*>(rec[k])[i];   // (a dense step's E4 list is never read)
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// W3LO = false (engine option precision = 2, BASELINE.json configs[4]'s best variant that holds its bar): conv3 runs with plain fp16 WEIGHTS --
// its w_lo x_hi product is dropped, 12 of a tile's 53 MFMAs per position; activations stay hi + lo everywhere
template <int K1, bool W3LO = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void trunk3_kernel(const TrunkTile* __restrict__ tiles, int n_tiles, int n_views, int ctx, const RInfo* __restrict__ rinfo,
                   const uint8_t* __restrict__ bases, const uint32_t* __restrict__ kin, CtxWeights W, TrunkMaps mp, half_t* __restrict__ dump,
                   int32_t* __restrict__ list_steps, const int32_t* __restrict__ tcost) {
    constexpr int NW = 4, NTW = 2;
    __shared__ __attribute__((aligned(16))) half_t smem[T3_LDS_HALVES + T3_XROWS * TR_WRS];
    static_assert(sizeof(smem) == 147840, "LDS plan");
    __shared__ uint32_t rlist[2][T3_RLW];  // the step's record ([3][128] bytes of row lists, the E4 list, its count); two buffers: conv4 still reads one while the next step's arrives
    __shared__ int64_t s_grow0;
    __shared__ int s_kind;   // bit 0: warm-up step (E4 rows to the dump), bit 1: constant step, bit 2: the calibration step
    __shared__ __attribute__((aligned(16))) half_t kc[3][2][128];   // E1's, E2's, E3's constant row: [layer][hi | lo][channel]
    __shared__ __attribute__((aligned(16))) half_t e4c[2 * C4_CH];  // E4's, as a map row [hi 96 | lo 96]
    half_t* a_hi = smem;
    half_t* a_lo = a_hi + T3_AROWS * TR_RS;
    half_t* b_hi = a_lo + T3_AROWS * TR_RS;
    half_t* b_lo = b_hi + T3_BROWS * TR_RS;
    half_t* h1_hi = b_lo + T3_BROWS * TR_RS;   // E1's kept rows while plane A holds E3
    half_t* h1_lo = h1_hi + T3_H1 * TR_RS;
    half_t* h3_hi = h1_lo + T3_H1 * TR_RS;     // E3's kept rows while plane A holds E1
    half_t* h3_lo = h3_hi + T3_H3 * TR_RS;
    half_t* xb = h3_lo + T3_H3 * TR_RS;        // feature rows of the step (their own buffer: built while conv4 runs)
    const int n_work = n_tiles * n_views;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A workgroup takes a contiguous run of tiles: runs of equal COST (tcost: running cost of the group's tiles, n_tiles + 1 entries -- a
    // constant step is an eighth of a computed one; with equal counts the workgroup with the fewest of them sets the launch's time).
    // Workgroup k starts at the first work item w whose running cost reaches k / G of the total: a 64-ary search, every lane a probe.
    int w0, w1;
    if (tcost) {
        const int64_t c_view = tcost[n_tiles], c_all = c_view * n_views;
        auto cost_at = [&](const int wq) __attribute__((always_inline)) {   // running cost in front of work item wq (0 .. n_work)
            const int v = wq >= n_tiles && wq < n_work ? 1 : wq >= n_work ? n_views : 0;
            return v * c_view + (wq >= n_work ? 0 : tcost[wq - (v ? n_tiles : 0)]);
        };
        auto first_at = [&](const int64_t target) __attribute__((always_inline)) {
            int lo = 0, hi = n_work;   // the answer lies in [lo, hi]; cost_at(hi) >= target
            while (hi > lo) {
                const int stride = (hi - lo + 63) / 64;
                const int probe = lo + lane * stride;
                const bool ge = probe >= hi || cost_at(probe) >= target;
                const uint64_t bal = __ballot(ge);
                const int first = bal ? __builtin_ctzll(bal) : 64;
                if (first == 0) {
                    hi = lo;
                } else {
                    const int nlo = lo + (first - 1) * stride + 1;
                    hi = first < 64 ? min(lo + first * stride, hi) : hi;
                    lo = nlo;
                }
            }
            return __builtin_amdgcn_readfirstlane(lo);
        };
        const int64_t k = blockIdx.x, G = gridDim.x;
        w0 = first_at((k * c_all + G - 1) / G);
        w1 = k + 1 == G ? n_work : first_at(((k + 1) * c_all + G - 1) / G);
    } else {
        const int base_n = n_work / (int)gridDim.x, rem_n = n_work - base_n * (int)gridDim.x;
        w0 = (int)blockIdx.x * base_n + min((int)blockIdx.x, rem_n);
        w1 = w0 + base_n + ((int)blockIdx.x < rem_n);
    }
    if (w0 >= w1) return;

    for (int i = threadIdx.x; i < (T3_LDS_HALVES + T3_XROWS * TR_WRS) / 2; i += NW * 64) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
    for (int i = threadIdx.x; i < 2 * T3_RLW; i += NW * 64) rlist[0][i] = 0;
    __syncthreads();

    // The next step's feature rows and row lists, one row per thread, in steps that sit between the layers so that no load's
    // latency is exposed (as in trunk2_kernel).  The step after (w, warm-up) is (w, tile); the step after (w, tile) is (w + 1, warm-up)
    // if tile w + 1 is a read's first, else (w + 1, tile).
    struct Build {
        TrunkTile tl, tl2;  // the step's tile; the tile behind it in the list (a constant step builds its successor from it: no load to wait for)
        RInfo ri;
        int w, view, warm, cst, calib, ueff, b;
        uint32_t k, rl;
    } bd;
    auto build_desc1 = [&](const int w) __attribute__((always_inline)) {
        bd.w = w;
        bd.view = w >= n_tiles;
        bd.tl = tiles[bd.view ? w - n_tiles : w];
        const int w2 = min(w + 1, n_work - 1);
        bd.tl2 = tiles[w2 >= n_tiles ? w2 - n_tiles : w2];
    };
    // prev_warm: the step being computed is a warm-up (then the next one is the tile itself)
    auto build_desc2 = [&](const int prev_warm, const bool first) __attribute__((always_inline)) {
        bd.tl.read_idx = __builtin_amdgcn_readfirstlane(bd.tl.read_idx);
        bd.tl.u0 = __builtin_amdgcn_readfirstlane(bd.tl.u0);
        bd.warm = first ? 1 : (prev_warm ? 0 : (bd.tl.u0 == -TR_PAD));
        bd.calib = 0;
        bd.ueff = bd.tl.u0 - (bd.warm ? TR_OWN : 0);
        bd.ri = rinfo[bd.tl.read_idx];
    };
    auto build_desc3 = [&]() __attribute__((always_inline)) {
        bd.ri.len = __builtin_amdgcn_readfirstlane(bd.ri.len);
        bd.ri.map_off = __builtin_amdgcn_readfirstlane(bd.ri.map_off);
        bd.ri.base_off = ((int64_t)__builtin_amdgcn_readfirstlane((int)(bd.ri.base_off >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)bd.ri.base_off);
        // nothing of the read in any new row's receptive field: a read's warm-up step and first tile, the tiles behind its end
        bd.cst = bd.tl.u0 == -TR_PAD || (!bd.warm && bd.tl.u0 >= bd.ri.len);
    };
    auto build_loads = [&]() __attribute__((always_inline)) {
        const int r = threadIdx.x, L = bd.ri.len, x = bd.ueff + T3_S1 + r;
        const int xc = min(max(x, 0), L - 1);
        const int64_t j = bd.ri.base_off + (bd.view ? L - 1 - xc : xc);
        bd.b = bases[j];
        bd.k = kin[j];
        // the step's lists: the tile's own; a read's warm-up step's; the all-fill list of a warm-up step in the middle of a read
        const size_t li = bd.calib ? (size_t)2 * n_work : bd.warm ? (bd.tl.u0 == -TR_PAD ? (size_t)n_work + bd.w : (size_t)2 * n_work) : (size_t)bd.w;
        bd.rl = reinterpret_cast<const uint32_t*>(mp.rowlist + li * T3_RL)[min(r, T3_RL / 4 - 1)];
    };
    auto build_store = [&](const int buf) __attribute__((always_inline)) {
        const int r = threadIdx.x, x = bd.ueff + T3_S1 + r;
        if (r == 0) {
            s_grow0 = (int64_t)bd.view * mp.view_rows + bd.ri.map_off + (bd.ueff + TR_PAD);
            s_kind = bd.warm | (bd.cst << 1) | (bd.calib << 2);
        }
        if (r < T3_XROWS) {
            const bool in = !bd.calib && x >= 0 && x < bd.ri.len;
            *reinterpret_cast<uint4*>(xb + r * TR_WRS) = feature_row(in ? bd.b : -1, bd.k, bd.view);
        }
        if (r < T3_RL3 / 4) rlist[buf][r / (TR_OWN / 4) * 32 + r % (TR_OWN / 4)] = bd.rl;
        else if (r < T3_RL / 4) rlist[buf][3 * 32 + r - T3_RL3 / 4] = bd.rl;   // the E4 list and its count
    };

    using C1 = SCfg<8, (2 * K1 + 3) / 4 * 4, TR_WRS, 1, true, false, K1, 2, NTW>;
    using C2 = SCfg<128, 3, TR_RS, 2, true, true, 0, 1, NTW>;
    using C3 = SCfg<128, 3, TR_RS, 4, W3LO, true, 0, 1, NTW>;
    using C4 = SCfg<128, 3, TR_RS, 8, true, true, 0, 1, NTW>;
    using C4s = SCfg<128, 3, TR_RS, 8, true, true, 0, 1, 1>;   // one of the wave's two resident n-tiles alone
    // conv4's 6 n-tiles x 7 position tiles on four waves as in trunk2_kernel: waves 0 / 2 hold (a, b) = (0, 1) / (3, 4), waves 1 / 3
    // (b, a) = (1, 2) / (4, 5); the pair on one range of position tiles, a alone on the other
    const int nt04 = wave == 0 ? 0 : wave == 1 ? 1 : wave == 2 ? 3 : 4;
    using L1 = SConv<C1, C2, 0, 2, 2, 3>;
    using L2 = SConv<C2, C3, 0, 3, 4>;  // the longest group last: it is the window in which the next layer's weights can be fetched
    using L3 = SConv<C3, C4, 0, 3, 4>;
    static_assert(T3_XROWS <= NW * 64, "one feature row per thread");
    auto wf = [&](int i) { return reinterpret_cast<const half_t*>(W.wfrag_h[i]); };
    const half_t* c1f = reinterpret_cast<const half_t*>(W.c1f);
    const int nt0 = NTW * wave;

    WRegs wr;
    sconv_load_w<C1, 0, C1::KB>(c1f, nt0, lane, wr);
    sconv_load_bias<C1>(W.c1f_bias, nt0, lane, wr);
    build_desc1(w0);
    build_desc2(0, true);
    build_desc3();
    bd.calib = 1;          // the first step: calibration (all feature rows zero, every store into the scratch buffer)
    bd.cst = 0;
    build_loads();
    build_store(0);
    int buf = 0;
    int w = w0, warm = 1;  // the step being computed
    int n_list = 0;        // steps of this workgroup whose conv4 ran over the listed rows only
    int n_const = 0;       // tiles that were constant steps
    bool prev_calib = false;
    // scratch (engine: trunk3_dump_bytes): [0] a warm-up step's E4 rows (garbage), [1] the copy slots of the calibration step (128 map
    // rows), then per workgroup the calibration step's E4 rows (the last one is E4's constant row)
    half_t* const dump_rows = dump + TR_OWN * 2 * C4_CH;
    half_t* const dump_cal = dump_rows + 128 * 256 + (size_t)blockIdx.x * (TR_OWN * 2 * C4_CH);
    constexpr int first_new[3] = {T3_H1, T3_H2, T3_H3};
#ifdef HM_TRUNK_STAMP
    // per layer: [0] barrier -> run returns, [3] wait at the next barrier; slot 22 / 23: s_memtime / s_memrealtime of the whole loop
    unsigned long long ts[10], acc_t[16] = {};
    unsigned long long n_it = 0, acc_c = 0, n_c = 0;   // (acc_c, n_c: ticks and number of the constant steps, slots 17 / 18)
    const bool st_on = blockIdx.x == 0;
    const unsigned long long tk0 = hm_stamp(), tr0 = __builtin_amdgcn_s_memrealtime();
#define TS(i) do { if (st_on) ts[i] = hm_stamp(); } while (0)
#else
#define TS(i)
#endif
    while (true) {
        t3_lds_barrier();  // the step's feature rows, lists and descriptors are in LDS; the previous step is through with the planes (its stores may still drain)
        TS(0);
        const int64_t grow0 = s_grow0;
        const int kind = __builtin_amdgcn_readfirstlane(s_kind);
        const int cur_warm = kind & 1, cur_const = (kind >> 1) & 1, cur_calib = kind >> 2;
        if (prev_calib) {
            // the calibration step is through: its kept rows are the constant rows of E1 .. E3, its last conv4 row (scratch) is E4's
            prev_calib = false;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int t = threadIdx.x;
            if (t < 96) {
                const int l = t >> 5, plane = (t >> 4) & 1, col = (t & 15) * 8;
                const half_t* src = l == 0 ? (plane ? h1_lo : h1_hi) : l == 1 ? (plane ? b_lo : b_hi) : (plane ? h3_lo : h3_hi);
                *reinterpret_cast<uint4*>(&kc[l][plane][col]) = *reinterpret_cast<const uint4*>(src + col);
            } else if (t < 96 + 24) {
                reinterpret_cast<uint4*>(e4c)[t - 96] = reinterpret_cast<const uint4*>(dump_cal + (TR_OWN - 1) * 2 * C4_CH)[t - 96];
            }
            __syncthreads();
        }
        // the next step
        const int wn = cur_warm ? w : w + 1;
        const bool last = wn >= w1;
        // (a read's warm-up step starts 112 rows in front of the read's region: these bases may lie in front of it -- only rows inside are stored)
        half_t* g1 = cur_calib ? dump_rows : reinterpret_cast<half_t*>(mp.e[0]) + (grow0 + T3_S2) * 256;   // plane A row 0 as E1 = position u + 24
        half_t* g2 = cur_calib ? dump_rows : reinterpret_cast<half_t*>(mp.e[1]) + (grow0 + T3_S3) * 256;   // plane B row 0 = position u + 16
        half_t* g3 = cur_calib ? dump_rows : reinterpret_cast<half_t*>(mp.e[2]) + grow0 * 256;             // plane A row 0 as E3 = position u
        const uint8_t* rl = reinterpret_cast<const uint8_t*>(rlist[buf]);
        if (cur_const) {
            // ---- a constant step: nothing to compute.  The next step's descriptor comes from what is at hand (bd still describes this
            // step): the same tile (this was its warm-up step) or the tile behind it (fetched a step ago), the read's record only if the
            // read changes.  Its loads go out in front of this step's stores; the stores drain while they arrive ----
            const int t = threadIdx.x;
            n_const += !cur_warm;  // (tiles: a read's warm-up step is no tile)
            const int wb = min(wn, w1 - 1);
            bool new_read = false;
            if (wb != bd.w) {
                const int prev_read = bd.tl.read_idx;
                bd.w = wb;
                bd.view = wb >= n_tiles;
                bd.tl.read_idx = __builtin_amdgcn_readfirstlane(bd.tl2.read_idx);
                bd.tl.u0 = __builtin_amdgcn_readfirstlane(bd.tl2.u0);
                const int w2 = min(wb + 1, n_work - 1);
                bd.tl2 = tiles[w2 >= n_tiles ? w2 - n_tiles : w2];
                new_read = bd.tl.read_idx != prev_read;
                if (new_read) bd.ri = rinfo[bd.tl.read_idx];
            }
            auto finish_desc = [&]() __attribute__((always_inline)) {
                bd.warm = cur_warm ? 0 : (bd.tl.u0 == -TR_PAD);
                bd.calib = 0;
                bd.ueff = bd.tl.u0 - (bd.warm ? TR_OWN : 0);
                bd.cst = bd.tl.u0 == -TR_PAD || (!bd.warm && bd.tl.u0 >= bd.ri.len);
                build_loads();
            };
            if (!new_read) finish_desc();
            if (!cur_warm) {  // the tile's E4 rows (a read's warm-up step lies in front of its region)
                uint4* g4 = reinterpret_cast<uint4*>(reinterpret_cast<half_t*>(mp.e4) + grow0 * (2 * C4_CH));
                const uint4 v = reinterpret_cast<const uint4*>(e4c)[t % 24];   // (256 = 16 mod 24: a thread's chunk column moves by 16 per round)
                const uint4 v2 = reinterpret_cast<const uint4*>(e4c)[(t + 16) % 24], v3 = reinterpret_cast<const uint4*>(e4c)[(t + 8) % 24];
#pragma unroll
                for (int k = 0; k < (TR_OWN * 24 + 255) / 256; ++k) {
                    const int c = t + 256 * k;
                    if (c < TR_OWN * 24) g4[c] = k % 3 == 0 ? v : k % 3 == 1 ? v2 : v3;
                }
            }
            // the E1 .. E3 rows the lists flag (the fill entries -- the last new row -- once)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const uint8_t* rows = rl + 128 * l;
                half_t* g = l == 0 ? g1 : l == 1 ? g2 : g3;
                const int fill = first_new[l] + TR_OWN - 1, ch = t & 31;
                const uint4 v = reinterpret_cast<const uint4*>(&kc[l][0][0])[ch];
                int pr[TR_OWN / 8];   // (the 14 entries of this thread's row group are read before the first is used)
#pragma unroll
                for (int i = 0; i < TR_OWN / 8; ++i) pr[i] = rows[(t >> 5) + 8 * i];
#pragma unroll
                for (int i = 0; i < TR_OWN / 8; ++i)
                    if (pr[i] != fill) reinterpret_cast<uint4*>(g + (size_t)pr[i] * 256)[ch] = v;
                if (t < 32) reinterpret_cast<uint4*>(g + (size_t)fill * 256)[ch] = v;  // (the last new row may be flagged itself)
            }
            // the kept rows of the step behind this one: 4 + 8 + 16 rows of constants
            for (int c = t; c < (T3_H1 + T3_H2 + T3_H3) * 32; c += NW * 64) {
                const int row = c >> 5, plane = (c >> 4) & 1, col = (c & 15) * 8;
                const int l = row < T3_H1 ? 0 : row < T3_H1 + T3_H2 ? 1 : 2, r = row - (l == 0 ? 0 : l == 1 ? T3_H1 : T3_H1 + T3_H2);
                half_t* dst = (l == 0 ? (plane ? h1_lo : h1_hi) : l == 1 ? (plane ? b_lo : b_hi) : (plane ? h3_lo : h3_hi)) + r * TR_RS + col;
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&kc[l][plane][col]);
            }
            if (new_read) {
                build_desc3();
                finish_desc();
            }
            t3_lds_barrier();  // every wave has read this step's record and kind
            build_store(buf ^ 1);
            buf ^= 1;
#ifdef HM_TRUNK_STAMP
            if (st_on) {
                acc_c += hm_stamp() - ts[0];
                ++n_c;
            }
#endif
            if (last) break;
            w = wn;
            warm = bd.warm;
            continue;
        }
        // E1's kept rows come back into plane A (conv4 of the previous step is through with its rows 0 .. 3)
        move_rows<T3_H1, NW * 64>(a_hi, a_lo, h1_hi, h1_lo, threadIdx.x);
        L1::run(xb, xb, wr, EpiTrunk3{a_hi + T3_H1 * TR_RS, a_lo + T3_H1 * TR_RS}, wf(1), W.bias[1], nt0, nt0);
        build_desc1(min(wn, w1 - 1));  // unconditional (as are the loads below): the compiler's wait counts stay exact
        TS(1);
        __syncthreads();
        TS(2);
        build_desc2(cur_warm, cur_calib != 0);
        L2::run(a_hi, a_lo, wr, EpiTrunk3{b_hi + T3_H2 * TR_RS, b_lo + T3_H2 * TR_RS}, wf(2), W.bias[2], nt0, nt0, CopyRows3<NW>{rl, g1});
        // E1's last rows wait in H1 for the next step (conv3 is about to overwrite them)
        move_rows<T3_H1, NW * 64>(h1_hi, h1_lo, a_hi + T3_M * TR_RS, a_lo + T3_M * TR_RS, threadIdx.x);
        TS(3);
        __syncthreads();
        TS(4);
        build_desc3();
        build_loads();
        // E3's kept rows come back into plane A rows 0 .. 15 (conv2 is through with them; conv3 writes rows 16 ..)
        move_rows<T3_H3, NW * 64>(a_hi, a_lo, h3_hi, h3_lo, threadIdx.x);
        L3::run(b_hi, b_lo, wr, EpiTrunk3{a_hi + T3_H3 * TR_RS, a_lo + T3_H3 * TR_RS}, wf(3), W.bias[3], nt0, nt04, CopyRows3<NW>{rl + 128, g2});
        TS(5);
        __syncthreads();
        TS(6);
        // E3's and E2's last rows are kept for the next step: E3's in H3 (plane A is E1's next), E2's at the top of plane B itself
        move_rows<T3_H3, NW * 64>(h3_hi, h3_lo, a_hi + T3_M * TR_RS, a_lo + T3_M * TR_RS, threadIdx.x);
        move_rows<T3_H2, NW * 64>(b_hi, b_lo, b_hi + T3_M * TR_RS, b_lo + T3_M * TR_RS, threadIdx.x);
        // a warm-up step's E4 rows are not results (its kept rows were not): they go to the dump
        const EpiE43 e4{cur_calib ? dump_cal : cur_warm ? dump : reinterpret_cast<half_t*>(mp.e4) + grow0 * (2 * C4_CH)};
        const int n4 = __builtin_amdgcn_readfirstlane((int)rlist[buf][T3_RLW - 1]);
        if (!cur_warm && n4 <= T3_N4) {
            ++n_list;
            // ---- conv4 over the listed rows only: four m-tiles of needed rows instead of seven of all (CpG / CHG at their usual densities) ----
            const uint8_t* rows4 = rl + 3 * 128;
            const ListRows4 rm{rows4};
            const EpiE4L e4l{reinterpret_cast<half_t*>(mp.e4) + grow0 * (2 * C4_CH), rows4};
            using NoCp = typename SConvR<C4, C1, ListRows4, 2, 2>::NoCopy;
            if (wave & 1) {  // a = resident tile 1 alone on list tiles 0, 1 (with the E3 copy slots), the pair on list tiles 2, 3
                SConvR<C4s, void, ListRows4, 0, 1, 1>::template run<1>(a_hi, a_lo, wr, e4l, c1f, W.c1f_bias, nt04, nt0, CopyRows3<NW>{rl + 256, g3}, rm);
                SConvR<C4, C1, ListRows4, 2, 2>::run(a_hi, a_lo, wr, e4l, c1f, W.c1f_bias, nt04, nt0, NoCp{}, rm);
            } else {         // the pair on list tiles 0, 1, a = resident tile 0 alone on list tiles 2, 3
                SConvR<C4, void, ListRows4, 0, 1, 1>::run(a_hi, a_lo, wr, e4l, c1f, W.c1f_bias, nt04, nt0, CopyRows3<NW>{rl + 256, g3}, rm);
                SConvR<C4s, C1, ListRows4, 2, 2>::template run<0>(a_hi, a_lo, wr, e4l, c1f, W.c1f_bias, nt04, nt0, NoCp{}, rm);
            }
        } else
        // the four-tile part first (it carries the E3 copy slots), the three-tile part last (behind it the next step's conv1
        // weights are fetched into the registers it frees)
        if (wave & 1) {  // a = resident tile 1 alone on position tiles 0 .. 3, the pair on tiles 4 .. 6
            SConv<C4s, void, 0, 2, 2>::template run<1>(a_hi, a_lo, wr, e4, c1f, W.c1f_bias, nt04, nt0, CopyRows3<NW>{rl + 256, g3});
            SConv<C4, C1, 4, 3>::run(a_hi, a_lo, wr, e4, c1f, W.c1f_bias, nt04, nt0);
        } else {         // the pair on position tiles 0 .. 3, a = resident tile 0 alone on tiles 4 .. 6
            SConv<C4, void, 0, 2, 2>::run(a_hi, a_lo, wr, e4, c1f, W.c1f_bias, nt04, nt0, CopyRows3<NW>{rl + 256, g3});
            SConv<C4s, C1, 4, 3>::template run<0>(a_hi, a_lo, wr, e4, c1f, W.c1f_bias, nt04, nt0);
        }
        build_store(buf ^ 1);
        buf ^= 1;
        if (cur_calib) {  // its conv4 rows are read back at the top of the next step
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            prev_calib = true;
        }
        TS(7);
#ifdef HM_TRUNK_STAMP
        if (st_on) {
            __syncthreads();
            ts[8] = hm_stamp();
            for (int l = 0; l < 4; ++l) {
                acc_t[4 * l + 0] += ts[2 * l + 1] - ts[2 * l];
                acc_t[4 * l + 3] += ts[2 * l + 2] - ts[2 * l + 1];
            }
            ++n_it;
        }
#endif
        if (last) break;
        w = wn;
        warm = bd.warm;
    }
    (void)warm;
#ifdef HM_TRUNK_STAMP
    if (st_on && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 16; ++i) atomicAdd(&g_trunk3_stamp[threadIdx.x >> 6][i], acc_t[i]);
        atomicAdd(&g_trunk3_stamp[threadIdx.x >> 6][16], n_it);
        atomicAdd(&g_trunk3_stamp[threadIdx.x >> 6][17], acc_c);
        atomicAdd(&g_trunk3_stamp[threadIdx.x >> 6][18], n_c);
        atomicAdd(&g_trunk3_stamp[threadIdx.x >> 6][22], hm_stamp() - tk0);
        atomicAdd(&g_trunk3_stamp[threadIdx.x >> 6][23], __builtin_amdgcn_s_memrealtime() - tr0);
    }
#endif
#undef TS
    if (list_steps && threadIdx.x == 0 && n_list) atomicAdd(list_steps + ctx, n_list);
    if (list_steps && threadIdx.x == 0 && n_const) atomicAdd(list_steps + 4 + ctx, n_const);
}

size_t trunk3_rowlist_bytes(int64_t n_work) { return (size_t)(2 * n_work + 1) * T3_RL + 64; }
size_t trunk3_dump_bytes(int grid) { return ((size_t)(1 + grid) * TR_OWN * 2 * C4_CH + 128 * 256) * sizeof(uint16_t); }

void launch_trunk3(hipStream_t st, int k1, const TrunkTile* tiles, int n_tiles, int n_views, int ctx, const RInfo* rinfo,
                   const uint8_t* bases, const uint32_t* kin, const uint8_t* sctx, int64_t n_bases, const CtxWeights& w, const TrunkMaps& maps,
                   uint16_t* dump, int32_t* list_steps, const int32_t* tcost, int grid, bool w3_single) {
    if (n_tiles <= 0) return;
    const int n_work = n_tiles * n_views;
    const dim3 g(min(n_work, grid));
    if (k1 == 11) {
        hipLaunchKernelGGL(rowlist3_kernel<11>, dim3((2 * n_work + T3_RPB) / T3_RPB), dim3(64), 0, st, tiles, n_tiles, n_work, ctx, rinfo, sctx, n_bases, maps.rowlist);
        if (w3_single) hipLaunchKernelGGL((trunk3_kernel<11, false>), g, dim3(256), 0, st, tiles, n_tiles, n_views, ctx, rinfo, bases, kin, w, maps, reinterpret_cast<half_t*>(dump), list_steps, tcost);
        else hipLaunchKernelGGL((trunk3_kernel<11, true>), g, dim3(256), 0, st, tiles, n_tiles, n_views, ctx, rinfo, bases, kin, w, maps, reinterpret_cast<half_t*>(dump), list_steps, tcost);
    } else {
        hipLaunchKernelGGL(rowlist3_kernel<13>, dim3((2 * n_work + T3_RPB) / T3_RPB), dim3(64), 0, st, tiles, n_tiles, n_work, ctx, rinfo, sctx, n_bases, maps.rowlist);
        if (w3_single) hipLaunchKernelGGL((trunk3_kernel<13, false>), g, dim3(256), 0, st, tiles, n_tiles, n_views, ctx, rinfo, bases, kin, w, maps, reinterpret_cast<half_t*>(dump), list_steps, tcost);
        else hipLaunchKernelGGL((trunk3_kernel<13, true>), g, dim3(256), 0, st, tiles, n_tiles, n_views, ctx, rinfo, bases, kin, w, maps, reinterpret_cast<half_t*>(dump), list_steps, tcost);
    }
}

}  // namespace hm
