// How fast can the y / z access patterns of the 256^3 passes stream when there is no arithmetic, no barrier, no LDS?
// A: contiguous grid-stride 4 reads + 1 write.  B: y tiles (32 columns x 256 rows, thread = (column, 8-cell segment)).
// C: z tiles (32 columns x 256 planes).   Build: hipcc -O3 --offload-arch=gfx950 -o scratch/tile_copy scratch/tile_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(1024) void k_stream(const double *a, const double *b, const double *c, double *y, long n)
{
    for (long i = blockIdx.x * 1024L + threadIdx.x; i < n; i += gridDim.x * 1024L) y[i] = y[i] + a[i] * b[i] + c[i];
}
template <int SEG, int TX, int T, int NT, int MAP>
__global__ __launch_bounds__(T) void k_tile2(const double *a, const double *b, const double *c, double *y, int nx, long sl, long outer_stride, int n)
{
    const int ixl = threadIdx.x % TX, seg = threadIdx.x / TX;
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (MAP == 1) {                                   // linear id -> (XCD = id % 8 keeps one row set): the 8 x tiles of a row set on one XCD, back to back
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x, xcd = id % 8, k = id / 8;    // k-th block of this XCD
        const unsigned gx = gridDim.x;                // tiles per row set
        const unsigned rs = (k / gx) * 8 + xcd;       // row set
        bx = k % gx; by = rs;
        if (by >= gridDim.y) return;
    }
    const long base = (long)by * outer_stride + bx * TX + ixl;
    double av[SEG], bv[SEG], cv[SEG], yv[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) { const long e = base + (long)(seg * SEG + i) * sl;
        if (NT & 1) { av[i] = __builtin_nontemporal_load(a + e); bv[i] = __builtin_nontemporal_load(b + e); cv[i] = __builtin_nontemporal_load(c + e); yv[i] = __builtin_nontemporal_load(y + e); }
        else { av[i] = a[e]; bv[i] = b[e]; cv[i] = c[e]; yv[i] = y[e]; } }
#pragma unroll
    for (int i = 0; i < SEG; ++i) { const long e = base + (long)(seg * SEG + i) * sl; const double v = yv[i] + av[i] * bv[i] + cv[i];
        if (NT & 2) __builtin_nontemporal_store(v, y + e); else y[e] = v; }
}
template <int SEG>
__global__ __launch_bounds__(1024) void k_tile(const double *a, const double *b, const double *c, double *y, int nx, long sl, long outer_stride, int n)
{
    const int TX = 32, ixl = threadIdx.x % TX, seg = threadIdx.x / TX;
    const long base = (long)blockIdx.y * outer_stride + blockIdx.x * TX + ixl;
    double av[SEG], bv[SEG], cv[SEG], yv[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) { const long e = base + (long)(seg * SEG + i) * sl; av[i] = a[e]; bv[i] = b[e]; cv[i] = c[e]; yv[i] = y[e]; }
#pragma unroll
    for (int i = 0; i < SEG; ++i) { const long e = base + (long)(seg * SEG + i) * sl; y[e] = yv[i] + av[i] * bv[i] + cv[i]; }
}
int main()
{
    const int n = 256; const long N = (long)n * n * n;
    double *a, *b, *c, *y;
    CK(hipMalloc(&a, N * 8)); CK(hipMalloc(&b, N * 8)); CK(hipMalloc(&c, N * 8)); CK(hipMalloc(&y, N * 8));
    CK(hipMemset(a, 0, N * 8)); CK(hipMemset(b, 0, N * 8)); CK(hipMemset(c, 0, N * 8)); CK(hipMemset(y, 0, N * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        hipEventRecord(e0); for (int r = 0; r < 20; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-40s %.4f ms  %.2f TB/s (40 B/cell)\n", name, ms, N * 40.0 / ms / 1e9);
    };
    timeit("A contiguous, grid 2048 x 1024", [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(1024), 0, 0, a, b, c, y, N); });
    timeit("A contiguous, grid 16384 x 1024", [&] { hipLaunchKernelGGL(k_stream, dim3(16384), dim3(1024), 0, 0, a, b, c, y, N); });
    timeit("B y tiles 32 x 256, SEG 8", [&] { hipLaunchKernelGGL(k_tile<8>, dim3(n / 32, n), dim3(1024), 0, 0, a, b, c, y, n, (long)n, (long)n * n, n); });
    timeit("C z tiles 32 x 256, SEG 8", [&] { hipLaunchKernelGGL(k_tile<8>, dim3(n / 32, n), dim3(1024), 0, 0, a, b, c, y, n, (long)n * n, (long)n, n); });
    for (int dir = 0; dir < 2; ++dir) {
        const long sl = dir == 0 ? n : (long)n * n, os = dir == 0 ? (long)n * n : n;
        printf("--- %s lines\n", dir == 0 ? "y" : "z");
        timeit("TX32 SEG8 T1024 nt-load", [&] { hipLaunchKernelGGL((k_tile2<8, 32, 1024, 1, 0>), dim3(n / 32, n), dim3(1024), 0, 0, a, b, c, y, n, sl, os, n); });
        timeit("TX32 SEG8 T1024 nt-store", [&] { hipLaunchKernelGGL((k_tile2<8, 32, 1024, 2, 0>), dim3(n / 32, n), dim3(1024), 0, 0, a, b, c, y, n, sl, os, n); });
        timeit("TX32 SEG8 T1024 nt both", [&] { hipLaunchKernelGGL((k_tile2<8, 32, 1024, 3, 0>), dim3(n / 32, n), dim3(1024), 0, 0, a, b, c, y, n, sl, os, n); });
        timeit("TX32 SEG8 T1024 xcd map", [&] { hipLaunchKernelGGL((k_tile2<8, 32, 1024, 0, 1>), dim3(n / 32, n), dim3(1024), 0, 0, a, b, c, y, n, sl, os, n); });
        timeit("TX64 SEG16 T1024", [&] { hipLaunchKernelGGL((k_tile2<16, 64, 1024, 0, 0>), dim3(n / 64, n), dim3(1024), 0, 0, a, b, c, y, n, sl, os, n); });
        timeit("TX64 SEG8 T1024 (half lines: 128 rows)", [&] { hipLaunchKernelGGL((k_tile2<8, 64, 1024, 0, 0>), dim3(n / 64, n), dim3(1024), 0, 0, a, b, c, y, n, sl, os, n); });
        timeit("TX16 SEG8 T512", [&] { hipLaunchKernelGGL((k_tile2<8, 16, 512, 0, 0>), dim3(n / 16, n), dim3(512), 0, 0, a, b, c, y, n, sl, os, n); });
        timeit("TX32 SEG16 T512", [&] { hipLaunchKernelGGL((k_tile2<16, 32, 512, 0, 0>), dim3(n / 32, n), dim3(512), 0, 0, a, b, c, y, n, sl, os, n); });
        timeit("TX128 SEG32 T1024", [&] { hipLaunchKernelGGL((k_tile2<32, 128, 1024, 0, 0>), dim3(n / 128, n), dim3(1024), 0, 0, a, b, c, y, n, sl, os, n); });
    }
    CK(hipDeviceSynchronize());
    return 0;
}
