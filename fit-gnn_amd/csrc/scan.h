// scan.h -- single-workgroup exclusive prefix sum over int32 (in place allowed), no temporary storage.
// Used for one-off offset computations (family offsets, survivor ranks, CSR row pointers): N/1024
// iterations of a 1024-thread block; the data is tiny next to the kernels that consume it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fitgnn {

// out[i] = sum_{j<i} in[j] for i in [0, n]; out has n+1 entries (out[n] = total). in may alias out
// (in[n] is never read).
static __global__ __launch_bounds__(1024) void exclusive_scan_i32_kernel(const int32_t *in, int32_t *out, int32_t n) {
    __shared__ int32_t wave_tot[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int32_t v = i < n ? in[i] : 0;
        int32_t x = v;  // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wave_tot[wave] = x;
        __syncthreads();
        int32_t wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
        const int32_t carry = carry_s;
        __syncthreads();
        if (i < n) out[i] = carry + wbase + x - v;
        if (tid == 1023) carry_s = carry + wbase + x;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry_s;
}

inline void exclusive_scan_i32(const int32_t *in, int32_t *out, int32_t n, hipStream_t s) {
    hipLaunchKernelGGL(exclusive_scan_i32_kernel, dim3(1), dim3(1024), 0, s, in, out, n);
}

}  // namespace fitgnn
