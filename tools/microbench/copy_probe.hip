// copy_probe.hip -- what a read-once / write-once stream can reach on MI355X, by kernel shape (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o copy_probe copy_probe.hip ; run: ./copy_probe [MiB per buffer]
// The SpMM of a block-diagonal batch is such a stream with a little arithmetic: its ceiling is the best row of this table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ f4 ld(const f4 *p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <int NT>
__device__ __forceinline__ void st(f4 *p, f4 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// grid-stride, U loads in flight per lane, a wave instruction covers 1 KiB contiguous
template <int U, int NTL, int NTS>
__global__ __launch_bounds__(256) void copy_stride(const f4 *__restrict__ a, f4 *__restrict__ b, long n4) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NTL>(a + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(b + i + u * stride, v[u]);
    }
    for (; i < n4; i += stride) st<NTS>(b + i, ld<NTL>(a + i));
}

// each workgroup owns a contiguous chunk of CH KiB (like a row tile): wave w takes 1-KiB pieces w, w+4, ...; U in flight
template <int U, int NTL, int NTS>
__global__ __launch_bounds__(256) void copy_chunk(const f4 *__restrict__ a, f4 *__restrict__ b, long n4, int chunk4) {
    const long base = (long)blockIdx.x * chunk4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = wave * 64; p < chunk4; p += 4 * 64 * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = base + p + u * 256 + lane; v[u] = i < n4 && p + u * 256 < chunk4 ? ld<NTL>(a + i) : f4{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = base + p + u * 256 + lane; if (i < n4 && p + u * 256 < chunk4) st<NTS>(b + i, v[u]); }
    }
}

// through LDS, like the window kernel: stage CH KiB into LDS (register round trip), barrier, read back, store
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void copy_lds(const f4 *__restrict__ a, f4 *__restrict__ b, long n4, int chunk4) {
    extern __shared__ f4 lds[];
    const long base = (long)blockIdx.x * chunk4;
    for (int p = threadIdx.x; p < chunk4; p += 256) { const long i = base + p; lds[p] = i < n4 ? ld<NTL>(a + i) : f4{0, 0, 0, 0}; }
    __syncthreads();
    for (int p = threadIdx.x; p < chunk4; p += 256) { const long i = base + p; if (i < n4) st<NTS>(b + i, lds[p]); }
}

// the SpMM tile pattern: a workgroup owns ROWS rows x one SLAB4-float4 slab of a matrix whose rows are ROW4 float4 wide
// (slab = 64 float4 = 1 KiB: two workgroups per row tile at H = 512; slab = 128 float4: whole 2-KiB rows); staged through LDS
// with T threads, then stored -- what spmm_tile_kernel does around its row loop
template <int T, int NTL, int NTS>
__global__ __launch_bounds__(T) void copy_tile(const f4 *__restrict__ a, f4 *__restrict__ b, long n_rows, int rows, int row4, int slab4) {
    extern __shared__ f4 lds[];
    const int n_slabs = row4 / slab4;
    const long tile = blockIdx.x / n_slabs;
    const int slab = blockIdx.x % n_slabs;
    const long r0 = tile * rows;
    const int per = rows * slab4;
    for (int p = threadIdx.x; p < per; p += T) {
        const long r = r0 + p / slab4;
        lds[p] = r < n_rows ? ld<NTL>(a + r * row4 + slab * slab4 + p % slab4) : f4{0, 0, 0, 0};
    }
    __syncthreads();
    for (int p = threadIdx.x; p < per; p += T) {
        const long r = r0 + p / slab4;
        if (r < n_rows) st<NTS>(b + r * row4 + slab * slab4 + p % slab4, lds[p]);
    }
}

template <typename F>
float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    const long mib = argc > 1 ? atol(argv[1]) : 4096;
    const long bytes = mib << 20, n4 = bytes / 16;
    f4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    const int reps = mib >= 4096 ? 5 : 20;
    auto report = [&](const char *name, float ms) { printf("%-44s %9.1f us  %7.0f GB/s  (%.3f of 8 TB/s)\n", name, ms * 1e3, 2.0 * bytes / ms / 1e6, 2.0 * bytes / ms / 1e6 / 8000.0); fflush(stdout); };
    printf("buffer %ld MiB, read + write %.1f MB per launch\n", mib, 2.0 * bytes / 1e6);
    report("hipMemcpyDtoD", time_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, reps));
    for (int wgs_per_cu : {4, 8, 16, 32}) {
        const int grid = 256 * wgs_per_cu;
        char nm[96];
#define RUN_STRIDE(U, L, S) snprintf(nm, sizeof nm, "stride U=%d ntl=%d nts=%d grid=%d/CU", U, L, S, wgs_per_cu); \
        report(nm, time_ms([&] { hipLaunchKernelGGL((copy_stride<U, L, S>), dim3(grid), dim3(256), 0, 0, a, b, n4); }, reps));
        RUN_STRIDE(4, 0, 0) RUN_STRIDE(8, 0, 0) RUN_STRIDE(4, 1, 1) RUN_STRIDE(8, 1, 1) RUN_STRIDE(8, 0, 1) RUN_STRIDE(8, 1, 0)
    }
    for (int ch_kib : {16, 32, 64}) {
        const int chunk4 = ch_kib * 64;
        const int grid = (int)((n4 + chunk4 - 1) / chunk4);
        char nm[96];
#define RUN_CHUNK(U, L, S) snprintf(nm, sizeof nm, "chunk %d KiB U=%d ntl=%d nts=%d", ch_kib, U, L, S); \
        report(nm, time_ms([&] { hipLaunchKernelGGL((copy_chunk<U, L, S>), dim3(grid), dim3(256), 0, 0, a, b, n4, chunk4); }, reps));
        RUN_CHUNK(4, 0, 0) RUN_CHUNK(4, 1, 1) RUN_CHUNK(8, 0, 0) RUN_CHUNK(8, 1, 1)
#define RUN_LDS(L, S) snprintf(nm, sizeof nm, "lds-staged %d KiB ntl=%d nts=%d", ch_kib, L, S); \
        report(nm, time_ms([&] { hipLaunchKernelGGL((copy_lds<L, S>), dim3(grid), dim3(256), chunk4 * 16, 0, a, b, n4, chunk4); }, reps));
        if (ch_kib <= 32) { RUN_LDS(0, 0) RUN_LDS(1, 1) }
    }
    {
        const int row4 = 128;  // H = 512 floats
        const long n_rows = n4 / row4;
        char nm[96];
        struct Cfg { int rows, slab4, threads; } cfgs[] = {{16, 64, 256}, {8, 128, 256}, {16, 128, 256}, {16, 128, 512}, {12, 128, 256}, {32, 64, 256}, {8, 64, 256}, {4, 128, 256}};
        for (const Cfg &c : cfgs) {
            const long tiles = (n_rows + c.rows - 1) / c.rows;
            const unsigned grid = (unsigned)(tiles * (row4 / c.slab4));
            const size_t lds_b = (size_t)c.rows * c.slab4 * 16;
#define RUN_TILE(T, L, S) snprintf(nm, sizeof nm, "tile %2d rows x %4d B, %d thr, ntl=%d nts=%d", c.rows, c.slab4 * 16, T, L, S); \
            report(nm, time_ms([&] { hipLaunchKernelGGL((copy_tile<T, L, S>), dim3(grid), dim3(T), lds_b, 0, a, b, n_rows, c.rows, row4, c.slab4); }, reps));
            if (c.threads == 256) { RUN_TILE(256, 0, 0) RUN_TILE(256, 1, 1) RUN_TILE(256, 0, 1) }
            else { RUN_TILE(512, 0, 0) RUN_TILE(512, 1, 1) }
        }
    }
    return 0;
}
