// hm_tail_p.hip -- the STRIP tail kernel: conv5 .. conv8 of the dense-trunk path for a context whose sites are DENSE (CHH: one view
// position in eight is a site), 16 sites per workgroup pass instead of tail_kernel_r's 8.  fc1, fc2 and the softmax of those sites run
// behind it in tail_fc_kernel (hm_tail_fc.hip), which this file's launcher starts; conv8's rows travel through x8 (512 B per site).
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
// B = conv5's output [13][16 sites] x (hi, lo).  conv6's output overlays the strip (rows 0 .. 111), conv7's overlays B (and conv8's
// and fc1's in the build that keeps fc1 .. softmax in the pass: make fcin).  The next pass's gather (LDS-DMA, row-aligned quads as in hm_tail_r.hip; the strip's source rows are
// equidistant, no address table) runs in two parts: the edge rows and strip rows 112 .. 143 while conv6 / conv7 run, strip rows
// 0 .. 111 once conv7 has read conv6's output -- their lines are pulled into L2 by touch loads while conv5 runs, so that the
// late part is an L2 copy.
//
// Reference for what is computed: training/model_cnn.py:8-85 / models/*.onnx (mod_main.cpp:32-98); softmax -> ML byte:
// mod_batch.cpp:46-64.
#include "hm_tail_p_geo.h"
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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tail_kernel_p(HM_TAILP_PARAMS) {
#ifdef HM_TAILP_FC_IN_KERNEL
    constexpr bool FC_IN_KERNEL = true;    // A/B build (make HM_TAILP_FC_IN_KERNEL=1): fc1, fc2, softmax per pass, as before the split
#else
    constexpr bool FC_IN_KERNEL = false;   // the pass ends with conv8; fc1 .. softmax: tail_fc_kernel (hm_tail_fc.hip)
#endif
    constexpr bool W16 = false;   // (engine option precision = 2 is the run-time parameter w16: see the body's header)
#include "hm_tail_p_body.inc"
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

// exclusive scan of n counters in place, one workgroup of 16 waves: a wave owns a contiguous range, sums it (coalesced, 64 counters per
// load), then rewrites it 256 counters per round -- four loads in flight, a wave-level inclusive scan (shuffles) each, the running sum
// carried in a scalar.  (A thread per contiguous chunk -- the first form -- read with a 350-byte stride and one counter per round trip:
// 146 us per launch for 90 k counters, more than the other three sort kernels together.)
__global__ __launch_bounds__(1024) void class_scan_kernel(int32_t* __restrict__ cnt, int n) {
    __shared__ int32_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = ((n + 15) / 16 + 255) / 256 * 256;   // counters per wave: whole rounds of 256
    const int lo = min(wave * per, n), hi = min(lo + per, n);
    int32_t s = 0;
    for (int i = lo + lane; i < hi; i += 64) s += cnt[i];
#pragma unroll
    for (int d = 32; d; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    int32_t run = 0;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    for (int i0 = lo; i0 < hi; i0 += 256) {
        int32_t v[4], x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 64 * k + lane;
            v[k] = i < hi ? cnt[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = v[k];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int32_t y = __shfl_up(x[k], d, 64);
                if (lane >= d) x[k] += y;
            }
            const int i = i0 + 64 * k + lane;
            if (i < hi) cnt[i] = run + x[k] - v[k];
            run += __shfl(x[k], 63, 64);
        }
    }
}

// (odst[o]: the site's slot in the batch's result arrays, for tail_fc_kernel)
__global__ __launch_bounds__(256) void class_write_kernel(SiteRange sr, const int32_t* __restrict__ mark, int64_t n_rows, int n_blocks, const int32_t* __restrict__ offs,
                                                          int32_t* __restrict__ order, int32_t* __restrict__ okey, int32_t* __restrict__ odst) {
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
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
        odst[o] = sites ? sites[min(max(m[c], 0), max(n_sites, 1) - 1)].uidx : m[c];
    }
}

size_t tail_strip_handover_bytes(int64_t sites) { return (size_t)std::max<int64_t>(sites, 1) * TAIL_STRIP_HANDOVER_BYTES; }
size_t tail_strip_mark_bytes(int64_t map_rows) { return (size_t)(map_rows + CS_ROWS) * sizeof(int32_t); }
size_t tail_strip_count_bytes(int64_t map_rows) { return (size_t)16 * (size_t)((map_rows + CS_ROWS - 1) / CS_ROWS + 1) * sizeof(int32_t); }

void launch_tail_strip(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, int n_views, const uint16_t* edge4,
                       const int32_t* e4row, int32_t* mark, int32_t* cnt, int32_t* order, int32_t* okey, int32_t* odst, uint16_t* x8, float* logits,
                       float* p, uint8_t* ml, int32_t* pass_count, int grid, bool w16) {
    if (sr.cap <= 0) return;
    const int64_t n_rows = maps.view_rows * n_views;
    const int n_blocks = (int)((n_rows + CS_ROWS - 1) / CS_ROWS);
    (void)hipMemsetAsync(mark, 0xFF, (size_t)n_rows * sizeof(int32_t), st);
    hipLaunchKernelGGL(class_mark_kernel, dim3(max(1, min((sr.cap + 255) / 256, 4 * grid))), dim3(256), 0, st, sr, e4row, mark);
    hipLaunchKernelGGL(class_count_kernel, dim3(n_blocks), dim3(256), 0, st, mark, n_rows, n_blocks, cnt);
    hipLaunchKernelGGL(class_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, 16 * n_blocks);
    hipLaunchKernelGGL(class_write_kernel, dim3(n_blocks), dim3(256), 0, st, sr, mark, n_rows, n_blocks, cnt, order, okey, odst);
    const dim3 g(sr.totals ? grid : max(1, min((sr.cap + PGeo::S - 1) / PGeo::S, grid)));
    hipLaunchKernelGGL(tail_kernel_p, g, dim3(256), 0, st, sr, w, logits, p, ml, reinterpret_cast<const half_t*>(maps.e4),
                            reinterpret_cast<const half_t*>(edge4), order, okey, (int)std::min<int64_t>(n_rows, INT32_MAX), pass_count,
                            reinterpret_cast<half_t*>(x8), w16 ? 1 : 0);
#ifndef HM_TAILP_FC_IN_KERNEL
    launch_tail_fc(st, sr, w, x8, odst, logits, p, ml, grid, w16);
#else
    if (w16) {   // (the A/B build keeps fc1 in the pass with full weights: it has no precision 2)
        fprintf(stderr, "hifimeth_hip (fcin build): precision 2 is not available with fc1 inside the strip kernel\n");
        abort();
    }
#endif
}

}  // namespace hm
