// common.h -- small device helpers shared by the fitgnn HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fitgnn {

// Counter-based dropout: one 64-bit hash per group of 4 consecutive elements (group g = (row*H + col) >> 2)
// yields 4 x 16 uniform bits; element (g, sub) is kept iff its 16 bits >= floor(p * 65536).
// Forward (spmm.hip) and backward (gcn_ops.hip) regenerate the same decisions; no mask is stored.
// FITGNN_EPI_SEED_DEVICE (= 8): `seed` carries a device pointer to the seed
__device__ __forceinline__ uint64_t resolve_seed(uint64_t seed, uint32_t epi) {
    return ((epi & 8u) && (epi & 4u)) ? *reinterpret_cast<const uint64_t *>(seed) : seed;
}

// Two 32-bit murmur3 finalisers over (group, seed) instead of one splitmix64: 64-bit multiplies cost the vector ALU four
// 32-bit ones each, and the hash sits in the epilogue of HBM-bound kernels (the forward SpMM, the epilogue-backward kernels, the
// dH @ W GEMM) where it was most of a row's arithmetic.
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
}
__host__ __device__ __forceinline__ uint64_t dropout_bits(uint64_t seed, uint64_t group) {
    const uint32_t g = (uint32_t)group * 0x9E3779B1u + (uint32_t)(group >> 32) * 0x85EBCA77u;
    const uint32_t lo = fmix32(g ^ (uint32_t)seed);
    const uint32_t hi = fmix32((g + 0x7F4A7C15u) ^ (uint32_t)(seed >> 32) ^ 0x68E31DA4u);
    return ((uint64_t)hi << 32) | lo;
}
__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) { return (uint32_t)(p * 65536.0f); }
__host__ __device__ __forceinline__ bool dropout_keep(uint64_t bits, int sub, uint32_t thresh) {
    return (uint32_t)((bits >> (16 * sub)) & 0xFFFFull) >= thresh;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

}  // namespace fitgnn

// Raise a kernel's dynamic-LDS limit ONCE per process and device (done: one bit per device ordinal).  hipFuncSetAttribute is a
// host call into the runtime; issued before every launch it is a per-launch cost and a point where launches on different
// streams meet.
#include <atomic>
inline int fitgnn_lds_limit_once(const void *kernel, int bytes, std::atomic<uint64_t> &done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

#define FITGNN_RETURN_IF_HIP(expr)            \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)
