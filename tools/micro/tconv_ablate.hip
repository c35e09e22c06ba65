// Where do the ticks of a TConv stream go?  conv5's geometry (8 sites, 104 rows), a wave's two n-tiles resident, one 4-wave
// workgroup per CU, random data; built four times: as is, -DHM_ABL_NOREAD (no LDS operand reads after the first block),
// -DHM_ABL_NOEPI (accumulators dropped instead of ReLU + split + LDS stores), and both.  Ticks per MFMA of wave 0.
// for f in "" -DHM_ABL_NOREAD -DHM_ABL_NOEPI "-DHM_ABL_NOREAD -DHM_ABL_NOEPI"; do hipcc --offload-arch=gfx950 -O3 -std=c++20 $f -I hifimeth_amd/csrc tools/micro/tconv_ablate.hip -o ...; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "hm_convt.h"
using namespace hm;
constexpr int RS96 = 104, IN_SS = 27 * RS96, C5_SS = 15 * RS96, S = 8;
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
template <class TC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const half_t* w, unsigned long long* t, float* out, int iters, int active_waves) {
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
    unsigned long long t0 = 0, t1 = 0;
    if (wave < active_waves) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; ++it) {
            TC::run(h0, l0, W, (const float*)bias, col, EpiT{h1, l1});
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
    __syncthreads();
    if (lane == 0) t[blockIdx.x * 4 + wave] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = (float)h1[threadIdx.x + 200];
}
using C96 = TCfg<96, 3, RS96>;
using R5 = TRows<13, IN_SS, S * 13>;
template <class TC>
void run(const char* name, const half_t* w, unsigned long long* t, float* out, int active) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<TC>, dim3(256), dim3(256), 0, 0, w, t, out, iters, active);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<TC>, dim3(256), dim3(256), 0, 0, w, t, out, iters, active);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s waves %d  %4d MFMAs per call: %6.1f ticks per MFMA (%.0f per call)  clock %.2f GHz\n", name, active, TC::mfmas(), (double)h[0] / iters / TC::mfmas(), (double)h[0] / iters, (double)h[0] / (ms * 1e6));
}
int main() {
#if defined(HM_ABL_NOREAD) && defined(HM_ABL_NOEPI)
    printf("== no LDS reads, no epilogue\n");
#elif defined(HM_ABL_NOREAD)
    printf("== no LDS reads\n");
#elif defined(HM_ABL_NOEPI)
    printf("== no epilogue\n");
#else
    printf("== as is\n");
#endif
    half_t* w; unsigned long long* t; float* out;
    const size_t wn = 6 * 9 * 128 * 8;
    std::vector<_Float16> hw(wn);
    for (size_t i = 0; i < wn; ++i) hw[i] = (_Float16)(((int)(rand() & 0xffff) - 32768) * (1.0f / 32768.0f / 16.0f));
    (void)hipMalloc(&w, wn * 2); (void)hipMemcpy(w, hw.data(), wn * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&t, 256 * 4 * 8); (void)hipMalloc(&out, 256 * 256 * 4);
    using P22 = TConv<C96, R5, 8, 1, TG<0, 2, 0, 0>, TG<2, 2, 0, 0>>;
    using P222 = TConv<C96, R5, 8, 1, TG<0, 2, 0, 0>, TG<2, 2, 0, 0>, TG<4, 2, 0, 0>>;
    using S22 = TConv<C96, R5, 8, 1, TG<0, 0, 0, 2>, TG<0, 0, 2, 2>>;
    using P22d = TConv<C96, R5, 16, 3, TG<0, 2, 0, 0>, TG<2, 2, 0, 0>>;
    run<P22>("pair x2, pair x2  NS 8 LA 1", w, t, out, 4);
    run<P22>("pair x2, pair x2  NS 8 LA 1", w, t, out, 1);
    run<P222>("pair x2 x3 groups NS 8 LA 1", w, t, out, 4);
    run<P22d>("pair x2, pair x2  NS 16 LA 3", w, t, out, 4);
    run<S22>("single x2, single x2 NS 8 LA 1", w, t, out, 4);
    run<S22>("single x2, single x2 NS 8 LA 1", w, t, out, 1);
    return 0;
}
