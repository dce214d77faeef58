// micro-benchmark: how long does a freshly launched block wait for its first loads, as a function of how the arrays were allocated?
// kernel A writes NA arrays of n doubles; kernel B (same stream) reads them: per block, cycles from entry to "all loads returned".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { double *a[16]; };
__global__ void kA(Ptrs P, int na, long n, double v)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) for (int k = 0; k < na; ++k) P.a[k][i] = v + k;
}
template <int STRIDED>
__global__ void kB(Ptrs P, int na, long n, long stride, double *out, long long *cyc)
{
    const long long t0 = __builtin_readcyclecounter();
    double s = 0.0;
    const long i0 = blockIdx.x * 256L + threadIdx.x;
    // every thread: `na` arrays x 8 loads, all independent
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k < na) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const long i = i0 + (long)j * 32768L; s += P.a[k][i]; }
        }
    }
    out[i0] = s;
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    const long n = 262144; const int grid = 128, NA = 8;
    long long *d_cyc; CHK(hipMalloc((void **)&d_cyc, grid * sizeof(long long)));
    double *d_out; CHK(hipMalloc((void **)&d_out, n * sizeof(double)));
    hipStream_t st; CHK(hipStreamCreate(&st));
    for (int mode = 0; mode < 3; ++mode) {                       // 0: separate hipMallocs, 1: one arena (contiguous), 2: one arena, 2 MB aligned pieces
        Ptrs P; double *arena = nullptr;
        const size_t piece = mode == 2 ? (2u << 20) : ((n * sizeof(double) + 255) / 256 * 256);
        if (mode == 0) for (int k = 0; k < NA; ++k) CHK(hipMalloc((void **)&P.a[k], n * sizeof(double)));
        else { CHK(hipMalloc((void **)&arena, piece * NA)); for (int k = 0; k < NA; ++k) P.a[k] = (double *)((char *)arena + piece * k); }
        for (int na : {1, 4, 8}) {
            std::vector<long long> all;
            for (int rep = 0; rep < 50; ++rep) {
                hipLaunchKernelGGL(kA, dim3(grid), dim3(256), 0, st, P, NA, n, (double)rep);
                hipLaunchKernelGGL(kB<0>, dim3(grid), dim3(256), 0, st, P, na, n, 38L * 38L, d_out, d_cyc);
                std::vector<long long> h(grid); CHK(hipMemcpyAsync(h.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost, st)); CHK(hipStreamSynchronize(st));
                if (rep >= 10) all.insert(all.end(), h.begin(), h.end());
            }
            std::sort(all.begin(), all.end());
            printf("alloc mode %d, %d arrays x 8 loads per thread: cycles to all loads returned: median %lld  p10 %lld  p90 %lld\n", mode, na, all[all.size() / 2], all[all.size() / 10], all[all.size() * 9 / 10]);
        }
        // same, but the reader kernel follows ANOTHER reader (data clean in L2 / MALL, no producer in between)
        {
            std::vector<long long> all;
            for (int rep = 0; rep < 50; ++rep) {
                hipLaunchKernelGGL(kB<0>, dim3(grid), dim3(256), 0, st, P, 8, n, 38L * 38L, d_out, d_cyc);
                std::vector<long long> h(grid); CHK(hipMemcpyAsync(h.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost, st)); CHK(hipStreamSynchronize(st));
                if (rep >= 10) all.insert(all.end(), h.begin(), h.end());
            }
            std::sort(all.begin(), all.end());
            printf("alloc mode %d, 8 arrays, reader after reader: median %lld  p10 %lld  p90 %lld\n", mode, all[all.size() / 2], all[all.size() / 10], all[all.size() * 9 / 10]);
        }
        if (mode == 0) for (int k = 0; k < NA; ++k) (void)hipFree(P.a[k]); else (void)hipFree(arena);
    }
    return 0;
}
