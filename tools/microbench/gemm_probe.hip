// gemm_probe.hip -- GPU-box microbenchmark of gemm_nt_kernel<4, false, true> (the layer-1 forward product) at the bench's size,
// with parts of the kernel compiled out (-DPROBE_NO_MFMA / -DPROBE_NO_ALOAD / -DPROBE_NO_STORE) to see which of its three
// streams (MFMA, the a side from HBM, the c store) bounds it and how well they overlap.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I fit-gnn_amd/csrc [-DPROBE_...] tools/microbench/gemm_probe.hip -o gemm_probe
#include "../../fit-gnn_amd/csrc/gemm_nt.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>

extern "C" int fitgnn_colsum_partials_f32(const float *, int32_t, int32_t, float *, void *) { return 0; }

int main(int argc, char **argv) {
    const long R = argc > 1 ? atol(argv[1]) : 8246057;
    const int N = 512, K = 512;
    float *a, *w, *c;
    void *img;
    hipMalloc(&a, (size_t)R * K * 4);
    hipMalloc(&c, (size_t)R * N * 4);
    hipMalloc(&w, (size_t)N * K * 4);
    hipMalloc(&img, fitgnn_gemm_nt_presplit_bytes(N, K));
    std::vector<float> h((size_t)N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(a, 0, (size_t)R * K * 4);
    for (long off = 0; off < R * K; off += (long)h.size()) hipMemcpy(a + off, h.data(), std::min<size_t>(h.size(), (size_t)(R * K - off)) * 4, hipMemcpyHostToDevice);
    fitgnn_gemm_nt_presplit_f32(w, K, 1, N, K, K, img, nullptr);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        const int n = 5;
        for (int i = 0; i < n; ++i) {
            int rc = fitgnn_gemm_nt_pre_f32(a, K, img, R, N, K, c, N, nullptr);
            if (rc) { printf("rc %d\n", rc); return 1; }
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= n;
        printf("R=%ld: %.3f ms  (%.2f TB/s on a + c, %.0f TFLOP/s bf16 x3)\n", R, ms, (double)R * (K + N) * 4 / ms / 1e9,
               2.0 * R * N * K * 3 / ms / 1e9);
    }
#if !defined(PROBE_NO_MFMA) && !defined(PROBE_NO_ALOAD) && !defined(PROBE_NO_STORE)
    // spot check against fp64 on a few rows (a is h repeated)
    double worst = 0;
    for (long r : {0L, 1L, 255L, 256L, R / 2 + 3, R - 1}) {
        std::vector<float> ar(K), cr(N);
        hipMemcpy(ar.data(), a + r * K, K * 4, hipMemcpyDeviceToHost);
        hipMemcpy(cr.data(), c + r * N, N * 4, hipMemcpyDeviceToHost);
        for (int n = 0; n < N; ++n) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) { ref += (double)ar[k] * h[(size_t)n * K + k]; mag += fabs((double)ar[k] * h[(size_t)n * K + k]); }
            worst = std::max(worst, fabs(ref - cr[n]) / mag);
        }
    }
    printf("spot check: max |err| / sum|terms| = %.2e\n", worst);
#endif
    return 0;
}
