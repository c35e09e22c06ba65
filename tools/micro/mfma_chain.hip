// Calibration: v_mfma_f32_16x16x32_f16 issued by one wave per SIMD, round-robin over NACC independent accumulators -- i.e.
// an MFMA accumulates onto the result of the MFMA NACC issues before it.  Prints ns and shader-clock ticks per MFMA for
// NACC = 1 .. 8: the rate at which a chain of dependent accumulations can issue.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x4 acc[8] = {};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
            for (int u = 0; u < NACC; ++u) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int u = 0; u < 8; ++u) s += acc[u][0] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(float* out, unsigned long long* t) {
    const int iters = 4000, per = 24 / NACC * NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, t, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, t, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("accumulators %d: %.2f ns per MFMA and wave, %.1f ticks, %.1f TFLOP/s chip-wide\n", NACC, ms * 1e6 / (iters * (double)per), (double)h[0] / (iters * (double)per),
           256.0 * 4 * iters * per * 16384.0 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&t, 256 * 8);
    run<1>(out, t); run<2>(out, t); run<3>(out, t); run<4>(out, t); run<6>(out, t); run<8>(out, t);
    return 0;
}
