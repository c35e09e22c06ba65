// hm_tail_fc.hip -- fc1, fc2 and the softmax of the dense-trunk path as a kernel of their own (round 5), behind the strip tail kernel
// (hm_tail_p.hip), which stops at conv8.
//
// Inside the strip kernel the three cost 5.9 k of a pass's 33.1 k cycles for 2.7 k cycles' worth of MFMAs
// (profiles/r05_tailp_phase_stamps.txt): fc1's 131 KB of weights (hi + lo) cross the CU's vector-memory path once per pass of 16
// sites -- there are no registers left to keep them -- and fc2's VALU sum waits behind two barriers with nothing to overlap.  Here
// a wave's share of fc1's weights (4 n-tiles x 4 k-blocks x (hi, lo) = 128 registers) is RESIDENT for the whole launch, two
// workgroups share a CU so that one's fc2 rides under the other's MFMAs, and the sites arrive as full tiles of 16 in the strip
// kernel's list order, whatever pass they were taken in.
//
// Hand-over: conv8's output, 2 positions x 64 channels, split: [hi: l * 64 + c | lo: l * 64 + c] halves = 512 B per site, at the
// site's position in the class-sorted list (x8[pos]); dst[pos] = the site's slot in the batch's result arrays (written by the class sort).
//
// The arithmetic is tail_kernel_r's (and the strip kernel's former fc phase): per fc1 accumulator bias, then the k-blocks in ascending
// order with w_hi x_hi, w_hi x_lo, w_lo x_hi; fc2's 32 products per lane added in ascending k, the lanes' sums in the same butterfly;
// the same softmax -- byte-identical calls (tests/test_gpu_parity.py).
//
// Reference for what is computed: training/model_cnn.py:70-85 (fc1, fc2), softmax -> ML byte: mod_batch.cpp:46-64.
#include "hm_tail_p_geo.h"

#ifndef HM_FC_GRID_PER_CU
#define HM_FC_GRID_PER_CU 16
#endif

namespace hm {

namespace {

struct FcGeo {
    static constexpr int S = 16, NW = 4;
    static constexpr int RS64 = PGeo::RS64, HPS = PGeo::HPS, F2S = PGeo::F2S;
    // fc1's fp32 output, a site's 256 values in 8 parts of 32, parts HPS = 36 floats apart (fc2's 16 lanes of a site read 8 distinct parts from
    // different banks); sites 8 HPS + 4 = 292 floats apart: 292 = 36 mod 64, so the 16 sites of fc1's float4 stores (lane = site) fall into
    // 16 different bank quads.  (In the kernels that hold fc1 beside the convolutions the sites are 288 = 32 mod 64 apart -- every
    // store instruction an 8-way conflict, 1 k cycles per tile and workgroup: here the LDS is the busiest unit, there it was not.)
    static constexpr int HRS = 8 * HPS + 4;
    static constexpr int PLANE = 2 * S * RS64;   // halves: [2 positions][16 sites][RS64]
    static constexpr int X8 = TAIL_X8_HALVES;    // halves per site in the hand-over buffer
};

}  // namespace

// W16: engine option precision = 2 -- fc1 with plain fp16 weights (w_lo x_hi dropped, the lo plane not fetched), as tail_kernel_r<true>'s
template <bool W16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void tail_fc_kernel(SiteRange sr, CtxWeights W, float* __restrict__ logits, float* __restrict__ prob, uint8_t* __restrict__ ml,
                    const half_t* __restrict__ x8, const int32_t* __restrict__ dst) {
    using T = FcGeo;
    const Site* sites;   // (not read here: the sites' batch slots come through dst)
    const int n_sites = resolve_sites(sr, sites);
    const int n_tiles = (n_sites + T::S - 1) / T::S;
    if ((int)blockIdx.x >= n_tiles) return;

    struct Lds {
        half_t x_hi[T::PLANE], x_lo[T::PLANE];
        float hfc[T::S * T::HRS];
        float fc2w[T::F2S + 8 * T::HPS + 4];
    };
    __shared__ __attribute__((aligned(16))) Lds lds;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 256 + 2; i += T::NW * 64) {
        if (i < 512) lds.fc2w[(i >> 8) * T::F2S + ((i & 255) >> 5) * T::HPS + (i & 31)] = W.fc2_w[i];
        else lds.fc2w[T::F2S + 8 * T::HPS + (i - 512)] = W.fc2_b[i - 512];
    }
    // pad columns 64 .. 71 of the operand rows are never read (a k-block ends at channel 63)

    using CF = PCfg<64, 2, !W16>;
    using IF = PInRows<CF, T::RS64, 2, 0, 0, 0>;
    // one stream per tile: the wave's four n-tiles (64 of fc1's 256 outputs) on the tile's 16 sites -- 4 accumulators, 12 MFMAs per
    // k-block, the operand reads three k-blocks ahead (the strip kernel's fc1 ran two n-tiles at a time, reads one block ahead: its
    // registers held two more layers); an accumulator still sees bias, then its k-blocks in ascending order
    using FC1 = PConv<CF, IF, 8, 3, TG<0, 1, 0, 0, 4>>;
    const half_t* wf8 = reinterpret_cast<const half_t*>(W.wfrag_h[8]);
    TW<4, 4> WF;
    const int ntf[4] = {4 * wave, 4 * wave + 1, 4 * wave + 2, 4 * wave + 3}, colf[4] = {64 * wave, 64 * wave + 16, 64 * wave + 32, 64 * wave + 48};
    tw_load<!W16>(wf8, ntf, tid & 63, WF);
    float4 bzf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bzf[j] = *reinterpret_cast<const float4*>(W.bias[8] + colf[j] + 4 * ((tid & 63) >> 4));

    // a tile's 8 KB: thread t brings chunk t % 16 (16 bytes) of the hi and of the lo half of site t / 16
    const int ld_site = tid >> 4, ld_chunk = tid & 15;
    const int ld_dst = ((ld_chunk >> 3) * T::S + ld_site) * T::RS64 + (ld_chunk & 7) * 8;
    auto load_tile = [&](const int tile, uint4& vh, uint4& vl) __attribute__((always_inline)) {
        const int pos = min(tile * T::S + ld_site, n_sites - 1);   // (a ragged last tile re-reads the last site; its extra results are not stored)
        const uint4* src = reinterpret_cast<const uint4*>(x8 + (size_t)pos * T::X8) + ld_chunk;
        vh = src[0];
        vl = src[16];
    };
    // Everything a tile needs from global memory -- its operands and the batch slots of its results (dst[pos]: written beside the sorted
    // list by the class sort, so that no load here depends on another) -- is requested two tiles ahead, into the registers the tile
    // two turns back has just given up (the loop is unrolled by two: no register is copied while a load into it is in flight).  The
    // loop body is straight-line code, the result stores included, so that the compiler's wait counts are exact: with a load or a
    // store behind a branch (and it keeps a branch around any masked block that holds a store) it falls back to vmcnt(0), i.e. to
    // one full memory round trip per tile -- 0.93 ms per launch of 2.6 M sites in that form.
    const int G = (int)gridDim.x, last = n_tiles - 1;
    auto load_dst = [&](const int tile) __attribute__((always_inline)) { return dst[min(tile * T::S + (tid >> 4), n_sites - 1)]; };
    auto turn = [&](const int tile, uint4& vh, uint4& vl, int& dreg) __attribute__((always_inline)) {
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int li = tl & 15, lk = (tl & 63) >> 4;
        *reinterpret_cast<uint4*>(lds.x_hi + ld_dst) = vh;
        *reinterpret_cast<uint4*>(lds.x_lo + ld_dst) = vl;
        const int d = dreg;
        lds_barrier();     // the tile's operands (and, the first time round, fc2's weights) are in LDS; the previous tile's fc2 has read fc1's output
                           // (LDS-only barriers: the loads below and the result stores stay in flight across them)
        load_tile(min(tile + 2 * G, last), vh, vl);
        dreg = load_dst(min(tile + 2 * G, last));
        {
            const IF ia{li * T::RS64 + 8 * lk};
            const EpiFc1P<T::HPS> ef{lds.hfc + li * T::HRS + 4 * lk};
            FC1::run(lds.x_hi, lds.x_lo, WF, [&](int j) __attribute__((always_inline)) { return bzf[j]; }, colf, ia, ef);
        }
        lds_barrier();
        // fc2 + softmax (mod_batch.cpp:46-64) in fp32: 16 lanes per site = 2 outputs x 8 partial sums; a lane's 32 products are added in
        // ascending k as in tail_kernel_r's loop, with all of its LDS reads issued before the first is used
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
        // Every lane of a site's 16 computes the site's results and stores them -- the same values to the same addresses, one write per
        // site and array in the memory system -- so that the stores need no branch (see above).  A slot beyond the list's end (ragged last
        // tile) holds the last site's operands and slot: the same values once more.
        const float v0 = o ? other : sum, v1 = o ? sum : other;
        const float mx = fmaxf(v0, v1);
        const float e0 = expf(v0 - mx), e1 = expf(v1 - mx);
        const float p1 = e1 / (e0 + e1);
        int q = (int)(255 * p1);
        q = q > 255 ? 255 : q;
        *reinterpret_cast<float2*>(logits + 2 * (size_t)d) = make_float2(v0, v1);
        prob[d] = p1;
        ml[d] = (uint8_t)q;
    };
    int tile = blockIdx.x;
    uint4 vh0, vl0, vh1, vl1;   // operands of the tile of an even / odd turn
    int d0, d1;                 // batch slot of this thread's fc2 site of that tile
    load_tile(tile, vh0, vl0);
    d0 = load_dst(tile);
    load_tile(min(tile + G, last), vh1, vl1);
    d1 = load_dst(min(tile + G, last));
    while (true) {
        turn(tile, vh0, vl0, d0);
        if (tile + G >= n_tiles) break;
        turn(tile + G, vh1, vl1, d1);
        if (tile + 2 * G >= n_tiles) break;
        tile += 2 * G;
    }
}

size_t tail_fc_x8_bytes(int64_t sites) { return (size_t)std::max<int64_t>(sites, 1) * TAIL_X8_HALVES * sizeof(uint16_t); }

void launch_tail_fc(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const uint16_t* x8, const int32_t* dst, float* logits, float* p,
                    uint8_t* ml, int grid, bool w16) {
    if (sr.cap <= 0) return;
    // two workgroups fit a CU; eight times as many are launched (each loads its 128 KB of weights once for ~ 40 tiles): the dispatcher
    // evens out what a fixed two-per-CU placement would not; with a host-known count no more workgroups than there are tiles
    const int want = HM_FC_GRID_PER_CU * grid;
    const dim3 g(sr.totals ? want : max(1, min((sr.cap + FcGeo::S - 1) / FcGeo::S, want)));
    if (w16) hipLaunchKernelGGL(tail_fc_kernel<true>, g, dim3(256), 0, st, sr, w, logits, p, ml, reinterpret_cast<const half_t*>(x8), dst);
    else hipLaunchKernelGGL(tail_fc_kernel<false>, g, dim3(256), 0, st, sr, w, logits, p, ml, reinterpret_cast<const half_t*>(x8), dst);
}

}  // namespace hm
