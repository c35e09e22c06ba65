// WRITE_SIZE calibration (VERDICT r04 #6c): MI355X_MICROARCH.md calibrates the counter for 16-byte-per-lane streaming stores only; the
// trunk's epilogue stores 8 bytes per lane (half4: four channels of one map row).  Three kernels write the SAME 1 GiB -- 16 B / lane, 8 B / lane
// contiguous, and 8 B / lane in the trunk's pattern (a wave's 64 lanes = 16 rows x 4 lanes of 8 B at a 384-byte row stride, the other lanes'
// columns written by later instructions) -- under `rocprofv3 --pmc WRITE_SIZE`; tools/write_calib.sh prints counter / bytes per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void store16(uint4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4); }
__global__ void store8(uint2* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint2(1, 2); }
// rows of 384 B (48 uint2); a wave covers 16 rows x (4 lanes x 8 B) per instruction and walks the row's 12 column groups
__global__ void store8_rows(uint2* p, size_t rows) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    for (size_t r0 = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16; r0 < rows; r0 += (size_t)gridDim.x * (blockDim.x >> 6) * 16)
        for (int c = 0; c < 12; ++c) p[(r0 + li) * 48 + c * 4 + lk] = make_uint2(1, 2);
}
int main() {
    const size_t bytes = size_t(1) << 30;
    void* d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(store16, dim3(4096), dim3(256), 0, 0, (uint4*)d, bytes / 16);
        hipLaunchKernelGGL(store8, dim3(4096), dim3(256), 0, 0, (uint2*)d, bytes / 8);
        hipLaunchKernelGGL(store8_rows, dim3(4096), dim3(256), 0, 0, (uint2*)d, bytes / 384 / 16 * 16);
    }
    hipDeviceSynchronize();
    printf("bytes per launch: %zu\n", bytes);
    return 0;
}
