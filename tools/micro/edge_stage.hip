// edge_stage.hip -- VERDICT r04 #3, measured: ONE layer of the window-edge chains (conv2's shape: 128 -> 128 channels, three taps) as a
// persistent kernel of its own with its weights RESIDENT in registers, sites streamed along M -- the "layer split" of edge2_kernel that
// DESIGN.md 3.4 had only argued about.  What it times is the stage a split would be made of:
//   * 4 waves, each holds two n-tiles of the layer for the whole launch (12 k-blocks x 2 x (hi, lo) = 192 registers, hm_convt.h's TW);
//   * a pass = 16 sites = 32 pseudo-rows: m-tile 0 the 16 left chains (taps: zero padding -- skipped --, the previous layer's edge row, one
//     map row), m-tile 1 the 16 right chains (K1 = 13 geometry: two map rows, the previous layer's row); sites along the MFMA rows
//     (hm_convp.h), 120 MFMAs per wave and pass;
//   * inputs double-buffered in LDS (80 rows of 272 B per plane and buffer = 87 KB; 32 sites per pass would need 174 KB): the NEXT pass's rows
//     arrive by LDS-DMA while this pass computes -- 48 map rows of 512 B gathered at random from a 1 GB map (HBM, as in the product), 32 rows of
//     the previous layer's output read contiguously; the 32 output rows leave as global stores from the epilogue.
// Prints ticks per site and CU, to be compared with conv2's share of an edge2_kernel pass: 9.8 k ticks per 32 sites = 306 per site
// (profiles/r04_edge2_phase_stamps.txt).  hipcc --offload-arch=gfx950 -O3 -std=c++20 -I hifimeth_amd/csrc tools/micro/edge_stage.hip
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hm_convp.h"

namespace hm {

namespace {

constexpr int ES = 16, RS = 136, NROW = 5 * ES;   // sites per pass; halves per LDS row (128 channels + 16 B pad); rows per buffer and plane
constexpr int SPEC_L = 0, SPEC_R = ES * RS, MAPL = 2 * ES * RS, MAPR0 = 3 * ES * RS, MAPR1 = 4 * ES * RS, PLANE = NROW * RS;
constexpr int QR = 3, CH = 17;                      // LDS rows per DMA instruction, 16-byte chunks per row (16 data + the pad)
constexpr int NDMA = (NROW + QR - 1) / QR;          // DMA instructions per plane and pass

using C = PCfg<128, 3>;
struct InEdge {
    int lb;  // li * RS + 8 * lk
    static constexpr bool skip(int tile, int kb) { return tile == 0 && C::tap(kb) == 0; }   // a left chain's first tap lies on the zero padding
    template <int TILE, int KB>
    __device__ __forceinline__ int off() const {
        constexpr int tap = C::tap(KB), ch = C::ch0(KB);
        constexpr int base = TILE == 0 ? (tap == 1 ? SPEC_L : MAPL) : (tap == 0 ? MAPR0 : tap == 1 ? MAPR1 : SPEC_R);
        return lb + (base + ch);
    }
};
// ReLU + split -> the layer's output rows in HBM: [site][side][hi 128 | lo 128]; tile t = side t
struct EpiOut {
    static constexpr int NV0 = 6, NV1 = 1, NW = 2, WMASK = 0x040;
    struct St { half4 h, l; };
    half_t* __restrict__ g;   // this lane's site: out + site * 512 + 4 * lk
    __device__ __forceinline__ void s0(const f32x4& acc, St& s) const { split4(acc, s.h, s.l); }
    template <int TILE>
    __device__ __forceinline__ void s1(int col, const St& s) const {
        *reinterpret_cast<half4*>(g + TILE * 256 + col) = s.h;
        *reinterpret_cast<half4*>(g + TILE * 256 + 128 + col) = s.l;
    }
};

}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void edge_stage_kernel(const half_t* __restrict__ wfrag, const float* __restrict__ bias, const half_t* __restrict__ spec,
                       const half_t* __restrict__ map, const int32_t* __restrict__ rows, half_t* __restrict__ out, int n_sites,
                       unsigned long long* __restrict__ ticks) {
    struct Lds {
        half_t buf[2][2][PLANE];           // [buffer][hi | lo][row][RS]
        float bias[128];
        unsigned long long src[2][NROW];   // source address of every LDS row of the pass being gathered
    };
    __shared__ __attribute__((aligned(16))) Lds lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 128) lds.bias[tid] = bias[tid];
    const int nt[2] = {2 * wave, 2 * wave + 1}, ncol[2] = {32 * wave, 32 * wave + 16};
    TW<12, 2> W;
    tw_load(wfrag, nt, lane, W);
    const int n_pass = (n_sites + ES - 1) / ES, per = (n_pass + (int)gridDim.x - 1) / (int)gridDim.x;
    const int p0 = (int)blockIdx.x * per, p1 = min(p0 + per, n_pass);
    if (p0 >= p1) return;

    auto build_table = [&](int pass, int b) {
        if (tid < NROW) {
            const int kind = tid / ES, s = min(pass * ES + tid % ES, n_sites - 1);
            const half_t* a = kind == 0 ? spec + (size_t)s * 512 : kind == 1 ? spec + (size_t)s * 512 + 256 : map + (size_t)rows[3 * s + (kind - 2)] * 256;
            lds.src[b][tid] = (unsigned long long)(uintptr_t)a;
        }
    };
    const unsigned long long lanes51 = 0x0007FFFFFFFFFFFFull;
    auto dma = [&](int q, int b) __attribute__((always_inline)) {   // LDS rows 3q .. 3q + 2 of both planes of buffer b
        const int ln = lane, r = min(QR * q + min(ln / CH, QR - 1), NROW - 1), chunk16 = (ln % CH) * 16;
        const unsigned long long src = lds.src[b][r] + (unsigned)chunk16;
        const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) half_t*)&lds.buf[b][0][0];
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(base + (uint32_t)(QR * q * RS * 2));
        const uint32_t d1 = __builtin_amdgcn_readfirstlane(base + (uint32_t)(PLANE * 2 + QR * q * RS * 2) - 256u);
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
            "global_load_lds_dwordx4 %5, off offset:256\n\t"
            "s_mov_b32 m0, %1\n\t"
            "s_mov_b64 exec, %0"
            : "=&s"(sv), "=&s"(km)
            : "s"(lanes51), "s"(d0), "s"(d1), "v"(src));
    };
    build_table(p0, 0);
    lds_barrier();
    for (int q = wave; q < NDMA; q += 4) dma(q, 0);
    build_table(min(p0 + 1, p1 - 1), 1);
    vm_drain();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int b = 0;
    for (int pass = p0; pass < p1; ++pass) {
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int li = tl & 15, lk = (tl & 63) >> 4;
        lds_barrier();   // this pass's rows and the next pass's address table are in LDS
        const bool more = pass + 1 < p1;
        auto hook = [&](auto c_) __attribute__((always_inline)) {
            constexpr int CB = decltype(c_)::value;
            if constexpr (CB < (NDMA + 3) / 4) { if (more && wave + 4 * CB < NDMA) dma(wave + 4 * CB, b ^ 1); }
        };
        const InEdge ia{li * RS + 8 * lk};
        const EpiOut eo{out + (size_t)min(pass * ES + li, n_sites - 1) * 512 + 4 * lk};
        PConv<C, InEdge, 8, 1, TG<0, 2, 0, 0>>::run<1>(&lds.buf[b][0][0], &lds.buf[b][1][0], W, (const float*)lds.bias, ncol, ia, eo, hook);
        vm_drain();      // the next pass's rows have landed (and this pass's stores have left)
        lds_barrier();   // everybody is through with this pass's address table
        build_table(min(pass + 2, p1 - 1), b);   // (the table of the pass after next goes where this pass's was)
        b ^= 1;
    }
    if (tid == 0 && ticks) atomicAdd(ticks, __builtin_amdgcn_s_memtime() - t0);
}

}  // namespace hm

int main(int argc, char** argv) {
    using namespace hm;
    const int n_sites = argc > 1 ? atoi(argv[1]) : 4 << 20;
    const size_t map_rows = size_t(1) << 21;   // 2 Mi rows x 512 B = 1 GB: gathers miss L2 and the Infinity Cache as the product's do
    std::vector<uint16_t> hw(8 * 12 * 2 * 64 * 8), hs((size_t)4096 * 512);
    srand(5);
    auto rnd16 = [](float scale) { _Float16 v = (_Float16)(((rand() & 0xffff) / 65536.0f - 0.5f) * scale); uint16_t u; memcpy(&u, &v, 2); return u; };
    for (auto& x : hw) x = rnd16(0.2f);
    std::vector<float> hb(128, 0.01f);
    std::vector<int32_t> hr((size_t)3 * n_sites);
    for (auto& x : hr) x = (int32_t)(((size_t)rand() * 32768 + rand()) % map_rows);
    half_t *d_w, *d_spec, *d_map, *d_out;
    float* d_b;
    int32_t* d_rows;
    unsigned long long* d_t;
    if (hipMalloc(&d_w, hw.size() * 2) != hipSuccess || hipMalloc(&d_spec, (size_t)n_sites * 512 * 2 + 1024) != hipSuccess ||
        hipMalloc(&d_map, map_rows * 512 + 1024) != hipSuccess || hipMalloc(&d_out, (size_t)n_sites * 512 * 2 + 1024) != hipSuccess ||
        hipMalloc(&d_b, 512) != hipSuccess || hipMalloc(&d_rows, hr.size() * 4) != hipSuccess || hipMalloc(&d_t, 8) != hipSuccess) return 1;
    (void)hipMemcpy(d_w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_b, hb.data(), 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_rows, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    // small random halves everywhere (a repeating 4 MB pattern: the values' bits are what the multipliers see, not their positions)
    for (auto& x : hs) x = rnd16(1.0f);
    for (size_t o = 0; o < (size_t)n_sites * 512 * 2; o += hs.size() * 2) (void)hipMemcpy((char*)d_spec + o, hs.data(), std::min(hs.size() * 2, (size_t)n_sites * 512 * 2 - o), hipMemcpyHostToDevice);
    for (size_t o = 0; o < map_rows * 512; o += hs.size() * 2) (void)hipMemcpy((char*)d_map + o, hs.data(), std::min(hs.size() * 2, map_rows * 512 - o), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(d_t, 0, 8);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(edge_stage_kernel, dim3(256), dim3(256), 0, 0, d_w, d_b, d_spec, d_map, d_rows, d_out, n_sites, d_t);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t = 0;
        (void)hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
        const double passes = (n_sites + 15) / 16;
        printf("edge stage (conv2 shape, resident weights, 16 sites per pass): %d sites in %.3f ms = %.3f ns per site; %.0f ticks per pass and workgroup = "
               "%.1f ticks per site (edge2_kernel's conv2 share: 306); %.1f ticks per MFMA (120 per wave and pass)\n",
               n_sites, ms, ms * 1e6 / n_sites, (double)t / passes, (double)t / passes / 16, (double)t / passes / 120);
    }
    return 0;
}
