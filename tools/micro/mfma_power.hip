// Calibration: what v_mfma_f32_16x16x32_f16 rate does this chip SUSTAIN, chip-wide, as a function of the operand data?
// Pure MFMA loop (operands in registers, 8 independent accumulators, every CU busy), one and two waves per SIMD, with
// (a) trivial operands (a few distinct small integers: what tools/micro/mfma_rate.hip uses) and (b) random fp16 operands
// (uniform in [-1, 1), a fresh pair of fragments per MFMA from a register pool of 16) -- the data an actual network sees.
// The chip lowers its clock under load (MI355X_MICROARCH.md, "DVFS give-back"): the random-data rate is the ceiling a
// real kernel can be priced against.   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o tools/micro/_mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <bool RANDOM>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* t, int iters) {
    half8 a[8], b[8];
    for (int r = 0; r < 8; ++r)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = hash((blockIdx.x * 512 + threadIdx.x) * 64 + r * 8 + i);
            a[r][i] = RANDOM ? (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f)) : (_Float16)(float)((i + r) & 3);
            b[r][i] = RANDOM ? (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f)) : (_Float16)(float)((i * 3 + r) & 3);
        }
    f32x4 acc[8] = {};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r)   // compile-time rotation: every MFMA sees a different (a, b) pair of the pool
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a[u]), "v"(b[(u + r) & 7]));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int u = 0; u < 8; ++u) s += acc[u][0] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <bool RANDOM>
void run(float* out, unsigned long long* t, int threads) {
    const int iters = 1600000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {   // ~0.1 s each: long enough for the clock to settle
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<RANDOM>, dim3(256), dim3(threads), 0, 0, out, t, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[1]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s operands, %d wave(s) per SIMD: %7.2f ms  %.2f ticks per MFMA and wave  tick rate %.3f GHz  %7.1f TFLOP/s chip-wide\n",
               RANDOM ? "random " : "trivial", threads / 256, ms, (double)h[0] / (iters * 8.0), (double)h[0] / (ms * 1e6),
               256.0 * (threads / 64) * iters * 8 * 16384.0 / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 256 * 8);
    run<false>(out, t, 256); run<true>(out, t, 256); run<false>(out, t, 512); run<true>(out, t, 512);
    return 0;
}
