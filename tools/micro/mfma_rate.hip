// Calibration: shader-clock (s_memtime) cycles per v_mfma_f32_16x16x32_f16 issued back to back from one wave per SIMD,
// and the s_memtime rate against wall time.  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* t, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x4 acc[8] = {};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {  // asm: the builtin form made hipcc rotate the accumulators through AGPR copies
        asm volatile(
            "v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n\tv_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n\t"
            "v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n\tv_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n\t"
            "v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n\tv_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n\t"
            "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n\tv_mfma_f32_16x16x32_f16 %7, %8, %9, %7"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
            : "v"(a), "v"(b));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int u = 0; u < 8; ++u) s += acc[u][0] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 256 * 8);
    const int iters = 20000;
    for (int rep = 0; rep < 6; ++rep) {
        const int threads = rep < 3 ? 256 : 512;  // one, then two waves per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, t, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        double cyc = (double)h[0];
        printf("%d waves/SIMD: %.3f ms, stamp ticks %.0f -> %.2f ticks per MFMA and wave, tick rate %.3f GHz, %.1f TFLOP/s chip-wide\n", threads / 256, ms, cyc,
               cyc / (iters * 8.0), cyc / (ms * 1e6), 256.0 * (threads / 64) * iters * 8 * 16384.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
