// Which fp16 MFMA shape does this chip sustain more FLOP/s on, chip-wide, on RANDOM operands: v_mfma_f32_16x16x32_f16 or
// v_mfma_f32_32x32x16_f16?  Both do 1024 FLOP per cycle and SIMD on paper (16 / 32 cycles per instruction); the 32x32 form reads
// half the operand registers and issues half the instructions per FLOP, the chip holds its power, not its clock.
// Pure MFMA loops, operands in registers (a pool of 8 A and 8 B fragments, a different pair per instruction), every CU busy, one and
// two waves per SIMD, same accumulator bytes per wave (8 x f32x4 for 16x16, 2 x f32x16 for 32x32 -- and 8 x f32x16 as well).
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shapes.hip -o tools/micro/_mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int SHAPE, int NACC>   // SHAPE 16: 16x16x32, 32: 32x32x16
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* t, int iters) {
    half8 a[8], b[8];
    for (int r = 0; r < 8; ++r)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = hash((blockIdx.x * 512 + threadIdx.x) * 64 + r * 8 + i);
            a[r][i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
            b[r][i] = (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
        }
    unsigned long long t0, t1;
    float s = 0;
    if constexpr (SHAPE == 16) {
        f32x4 acc[NACC] = {};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; it += 8) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a[u]), "v"(b[(u + r) & 7]));
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][3];
    } else {
        f32x16 acc[NACC] = {};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; it += 8) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a[u]), "v"(b[(u + r) & 7]));
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int SHAPE, int NACC>
void run(float* out, unsigned long long* t, int threads) {
    const int iters = SHAPE == 16 ? 1600000 : 800000;
    const double flop = SHAPE == 16 ? 16384.0 : 32768.0;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {   // ~0.1 s each: long enough for the clock to settle
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, t, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[1]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s, %d accumulators, %d wave(s) per SIMD: %7.2f ms  %.2f ticks per MFMA and wave  tick rate %.3f GHz  %7.1f TFLOP/s chip-wide\n",
               SHAPE == 16 ? "16x16x32" : "32x32x16", NACC, threads / 256, ms, (double)h[0] / (iters * 8.0), (double)h[0] / (ms * 1e6),
               256.0 * (threads / 64) * iters * 8 * flop / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 256 * 8);
    run<16, 8>(out, t, 256); run<32, 2>(out, t, 256); run<32, 8>(out, t, 256);
    run<16, 8>(out, t, 512); run<32, 2>(out, t, 512);
    run<16, 8>(out, t, 256);   // again: order effects (temperature)
    return 0;
}
