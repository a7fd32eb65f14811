// common.h -- small device helpers shared by the fitgnn HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fitgnn {

// Counter-based dropout: one splitmix64 per group of 4 consecutive elements (group g = (row*H + col) >> 2)
// yields 4 x 16 uniform bits; element (g, sub) is kept iff its 16 bits >= floor(p * 65536).
// Forward (spmm.hip) and backward (gcn_ops.hip) regenerate the same decisions; no mask is stored.
// FITGNN_EPI_SEED_DEVICE (= 8): `seed` carries a device pointer to the seed
__device__ __forceinline__ uint64_t resolve_seed(uint64_t seed, uint32_t epi) {
    return ((epi & 8u) && (epi & 4u)) ? *reinterpret_cast<const uint64_t *>(seed) : seed;
}

__host__ __device__ __forceinline__ uint64_t dropout_bits(uint64_t seed, uint64_t group) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (group + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) { return (uint32_t)(p * 65536.0f); }
__host__ __device__ __forceinline__ bool dropout_keep(uint64_t bits, int sub, uint32_t thresh) {
    return (uint32_t)((bits >> (16 * sub)) & 0xFFFFull) >= thresh;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

}  // namespace fitgnn

#define FITGNN_RETURN_IF_HIP(expr)            \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)
