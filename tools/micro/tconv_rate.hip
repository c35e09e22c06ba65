// Calibration of hm_convt.h's TConv in isolation: one 4-wave workgroup per CU, planes of stacked sites in LDS, a layer run
// ITERS times back to back; prints shader-clock ticks per MFMA for several group shapes and look-aheads.
// hipcc --offload-arch=gfx950 -O3 -std=c++20 -I hifimeth_amd/csrc tools/micro/tconv_rate.hip -o tools/micro/_tconv_rate
#include <hip/hip_runtime.h>
#include <cstdio>
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
template <class TC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const half_t* w, unsigned long long* t, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) half_t smem[2 * S * IN_SS + 2 * S * C5_SS];
    __shared__ float bias[96];
    half_t* h0 = smem; half_t* l0 = smem + S * IN_SS; half_t* h1 = smem + 2 * S * IN_SS; half_t* l1 = h1 + S * C5_SS;
    for (int i = threadIdx.x; i < 2 * S * IN_SS; i += 256) smem[i] = (half_t)((i % 97) * 0.01f);
    if (threadIdx.x < 96) bias[threadIdx.x] = 0.1f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt[2] = {wave, (wave + 1) % 6}, col[2] = {16 * nt[0], 16 * nt[1]};
    TW<9, 2> W;
    tw_load(w, nt, lane, W);
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        TC::run(h0, l0, W, (const float*)bias, col, EpiT{h1, l1});
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __syncthreads();
    if (lane == 0) t[blockIdx.x * 4 + wave] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = (float)h1[threadIdx.x + 200];
}
using C96 = TCfg<96, 3, RS96>;
using R5 = TRows<13, IN_SS, S * 13>;
template <class TC>
void run(const char* name, const half_t* w, unsigned long long* t, float* out) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<TC>, dim3(256), dim3(256), 0, 0, w, t, out, iters);
    hipLaunchKernelGGL(k<TC>, dim3(256), dim3(256), 0, 0, w, t, out, iters);
    hipDeviceSynchronize();
    unsigned long long h[4]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %4d MFMAs per call: %6.1f ticks per MFMA (%.0f per call)\n", name, TC::mfmas(), (double)h[0] / iters / TC::mfmas(), (double)h[0] / iters);
}
int main() {
    half_t* w; unsigned long long* t; float* out;
    (void)hipMalloc(&w, 6 * 9 * 128 * 16); (void)hipMemset(w, 0, 6 * 9 * 128 * 16); (void)hipMalloc(&t, 256 * 4 * 8); (void)hipMalloc(&out, 256 * 256 * 4);
    run<TConv<C96, R5, 8, 1, TG<0, 0, 0, 1>>>("single x1  (1 acc)  NS 8 LA 1", w, t, out);
    run<TConv<C96, R5, 8, 3, TG<0, 0, 0, 1>>>("single x1  (1 acc)  NS 8 LA 3", w, t, out);
    run<TConv<C96, R5, 8, 1, TG<0, 0, 0, 2>>>("single x2  (2 accs) NS 8 LA 1", w, t, out);
    run<TConv<C96, R5, 12, 2, TG<0, 0, 0, 2>>>("single x2  (2 accs) NS 12 LA 2", w, t, out);
    run<TConv<C96, R5, 12, 1, TG<0, 0, 0, 3>>>("single x3  (3 accs) NS 12 LA 1", w, t, out);
    run<TConv<C96, R5, 8, 1, TG<0, 1, 0, 0>>>("pair x1    (2 accs) NS 8 LA 1", w, t, out);
    run<TConv<C96, R5, 8, 1, TG<0, 1, 1, 1>>>("pair+single(3 accs) NS 8 LA 1", w, t, out);
    run<TConv<C96, R5, 8, 1, TG<0, 2, 0, 0>>>("pair x2    (4 accs) NS 8 LA 1", w, t, out);
    run<TConv<C96, R5, 12, 2, TG<0, 2, 0, 0>>>("pair x2    (4 accs) NS 12 LA 2", w, t, out);
    run<TConv<C96, R5, 8, 1, TG<0, 2, 0, 0>, TG<2, 2, 0, 0>>>("pair x2, pair x2 (two groups) NS 8 LA 1", w, t, out);
    run<TConv<C96, R5, 8, 1, TG<0, 1, 4, 1>, TG<1, 1, 5, 1>, TG<2, 1, 6, 1>>>("3 x pair+single NS 8 LA 1", w, t, out);
    return 0;
}
