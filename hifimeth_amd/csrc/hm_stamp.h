// hm_stamp.h -- in-kernel shader-clock stamps for diagnostic builds (never compiled into the production kernels'
// hot loops: every use is behind a compile-time STAMP flag).
#pragma once
#include <hip/hip_runtime.h>

namespace hm {

// diagnostic builds only: shader-clock stamp (cdna_hip_programming.md section 7, "In-kernel stamps")
__device__ __forceinline__ unsigned long long hm_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
constexpr int N_STAMP = 20;

}  // namespace hm
