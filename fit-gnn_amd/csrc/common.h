// common.h -- small device helpers shared by the fitgnn HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fitgnn {

// Counter-based dropout decision: splitmix64 of (seed, element index) -> 24 uniform bits.
// Forward (spmm.hip) and backward (epilogue_bwd.hip) regenerate the same decision; no mask is stored.
__host__ __device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, float p) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);  // [0,1)
    return u >= p;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

}  // namespace fitgnn

#define FITGNN_RETURN_IF_HIP(expr)            \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)
