// Does a second workgroup per CU hide what one wave per SIMD cannot?  hm_convt.h's TConv (conv5's geometry: 96 -> 96 channels,
// 3 taps, sites stacked along M) with a wave's two n-tiles resident in 144 registers, run ITERS times with an LDS barrier
// after every call (as the layers of the tail are), by
//   (a) 256 workgroups = one per CU  (one wave per SIMD: the shape of tail_kernel_r), and
//   (b) 512 workgroups = two per CU  (two waves per SIMD, <= 256 registers, <= 80 KB of LDS each),
// same code object.  Prints time per call and MFMA-pipe ticks per MFMA and SIMD.  Random weights and activations (the clock the
// chip holds depends on the data: MI355X_MICROARCH.md, DVFS give-back).
// hipcc --offload-arch=gfx950 -O3 -std=c++20 -I hifimeth_amd/csrc tools/micro/tconv_occ.hip -o tools/micro/_tconv_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "hm_convt.h"
using namespace hm;
constexpr int RS96 = 104, IN_SS = 27 * RS96, C5_SS = 15 * RS96;
template <int S>
struct EpiT {
    static constexpr int PADCOL = RS96 - 8;
    half_t* hi; half_t* lo;
    static constexpr int NV0 = 6, NV1 = 1, NW = 2, WMASK = 0x200;
    struct St { half4 h, l; };
    static __device__ __forceinline__ int row(int site, int p, int) { return site * C5_SS + (p + 1) * RS96; }
    __device__ __forceinline__ void s0(const f32x4& acc, St& s) const { split4(acc, s.h, s.l); }
    __device__ __forceinline__ void s1(int off, int col, const St& s) const {
        *reinterpret_cast<half4*>(hi + off + col) = s.h;
        *reinterpret_cast<half4*>(lo + off + col) = s.l;
    }
};
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <class TC, int S, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k(const half_t* w, unsigned long long* t, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) half_t smem[2 * S * IN_SS + 2 * S * C5_SS];
    __shared__ float bias[96];
    half_t* h0 = smem; half_t* l0 = smem + S * IN_SS; half_t* h1 = smem + 2 * S * IN_SS; half_t* l1 = h1 + S * C5_SS;
    for (int i = threadIdx.x; i < 2 * S * IN_SS; i += 256) {
        const unsigned h = hash(i * 977 + blockIdx.x);
        smem[i] = i < S * IN_SS ? (half_t)((h & 0xffff) * (1.0f / 65536.0f)) : (half_t)(((h >> 16) & 0xffff) * (1.0f / 65536.0f / 2048.0f));
    }
    if (threadIdx.x < 96) bias[threadIdx.x] = 0.1f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt[2] = {wave, (wave + 1) % 6}, col[2] = {16 * nt[0], 16 * nt[1]};
    TW<9, 2> W;
    tw_load(w, nt, lane, W);
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        TC::run(h0, l0, W, (const float*)bias, col, EpiT<S>{h1, l1});
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __syncthreads();
    if (lane == 0) t[blockIdx.x * 4 + wave] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = (float)h1[threadIdx.x + 200];
}
using C96 = TCfg<96, 3, RS96>;
template <class TC, int S, int WPE>
void run(const char* name, const half_t* w, unsigned long long* t, float* out, int grid) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f; unsigned long long ticks = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<TC, S, WPE>), dim3(grid), dim3(256), 0, 0, w, t, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[4]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        if (ms < best) best = ms, ticks = h[0];
    }
    const double per_cu = grid / 256.0, calls = (double)iters * per_cu;   // calls per CU
    printf("%-34s grid %4d (%.0f WG/CU): %7.2f ms  %6.0f ns per call and CU  %5.1f ticks per MFMA and SIMD (a WG's call: %6.0f ticks, clock %.2f GHz)\n",
           name, grid, per_cu, best, best * 1e6 / calls, (double)ticks / iters / per_cu / TC::mfmas(), (double)ticks / iters, (double)ticks / (best * 1e6));
}
int main() {
    half_t* w; unsigned long long* t; float* out;
    const size_t wn = 6 * 9 * 128 * 8;
    std::vector<_Float16> hw(wn);
    for (size_t i = 0; i < wn; ++i) hw[i] = (_Float16)(((int)(rand() & 0xffff) - 32768) * (1.0f / 32768.0f / 16.0f));
    (void)hipMalloc(&w, wn * 2); (void)hipMemcpy(w, hw.data(), wn * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&t, 512 * 4 * 8); (void)hipMalloc(&out, 512 * 256 * 4);
    using R4 = TRows<13, IN_SS, 4 * 13>;
    using T4 = TConv<C96, R4, 8, 1, TG<0, 2, 0, 0>, TG<2, 1, 3, 1>>;   // 52 rows: pairs on tiles 0..2, a alone on tile 3
    run<T4, 4, 1>("4 sites, 189 MFMAs, 1 wave/SIMD build", w, t, out, 256);
    run<T4, 4, 2>("4 sites, 189 MFMAs, 2 waves/SIMD build", w, t, out, 256);
    run<T4, 4, 2>("4 sites, 189 MFMAs, 2 waves/SIMD build", w, t, out, 512);
    return 0;
}
