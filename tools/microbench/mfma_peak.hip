// mfma_peak.hip -- what v_mfma_f32_32x32x16_bf16 sustains on this box: W waves per CU, each issuing independent MFMAs back to back
// (8 accumulators, as gemm_nt's 64 x 128 wave tile), optionally with ds_read_b128 traffic beside them (LDS bytes per MFMA as in gemm_nt).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS_READS>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[2], b[4];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(float)(threadIdx.x + e);
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)(float)(threadIdx.x * 3 + e);
    const unsigned char *p = lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if (LDS_READS) {
#pragma unroll
                for (int q = 0; q < LDS_READS; ++q) {
                    bf16x8 t = *reinterpret_cast<const bf16x8 *>(p + ((g * LDS_READS + q) & 3) * 1024);
                    asm volatile("" :: "v"(t));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int L>
void run(const char *name, int threads) {
    float *out; hipMalloc(&out, 256 * 8 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * (threads / 64) * iters * 24 * 32768.0;
        printf("%s, %d waves/CU: %.2f ms, %.0f TFLOP/s\n", name, threads / 64, ms, flops / ms / 1e9);
    }
}
int main() {
    run<0>("MFMA only", 256);
    run<0>("MFMA only", 512);
    run<4>("MFMA + 4 ds_read_b128 per 8 MFMA (gemm_nt's ratio)", 512);
    run<2>("MFMA + 2 ds_read_b128 per 8 MFMA", 512);
    return 0;
}
