// abi.hip -- version and error-string entry points of libfitgnn_hip.so, and the stream-copy probe of the bench line.
#include <hip/hip_runtime.h>

#include "fitgnn_hip.h"

extern "C" int fitgnn_abi_version(void) { return FITGNN_ABI_VERSION; }

extern "C" const char *fitgnn_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case FITGNN_E_BADARG: return "fitgnn: bad argument (null pointer, negative size or unsupported shape)";
        case FITGNN_E_WORKSPACE: return "fitgnn: workspace too small (see *_workspace_bytes)";
        case FITGNN_E_ALIGN: return "fitgnn: pointer or leading dimension misaligned";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "fitgnn: unknown error code";
}

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kChunk4 = 1024;   // 16 KiB per workgroup

// wave w of the workgroup takes the 1-KiB pieces w, w + 4, ... of the chunk, eight in flight per lane
__global__ __launch_bounds__(256) void stream_copy_kernel(const f4 *__restrict__ a, f4 *__restrict__ b, long n4) {
    const long base = (long)blockIdx.x * kChunk4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int U = 8;
    for (int p = wave * 64; p < kChunk4; p += 4 * 64 * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + p + u * 256 + lane;
            v[u] = (i < n4 && p + u * 256 < kChunk4) ? __builtin_nontemporal_load(a + i) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + p + u * 256 + lane;
            if (i < n4 && p + u * 256 < kChunk4) __builtin_nontemporal_store(v[u], b + i);
        }
    }
}
}  // namespace

extern "C" int fitgnn_stream_copy_f32(const float *src, float *dst, int64_t n, void *stream) {
    if (n < 0 || (n % 4) != 0) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!src || !dst) return FITGNN_E_BADARG;
    if ((((uintptr_t)src | (uintptr_t)dst) % 16) != 0) return FITGNN_E_ALIGN;
    const long n4 = (long)(n / 4);
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)((n4 + kChunk4 - 1) / kChunk4)), dim3(256), 0, (hipStream_t)stream,
                       (const f4 *)src, (f4 *)dst, n4);
    return (int)hipGetLastError();
}
