// Access-pattern microbenchmark behind the long-line y / z passes: no arithmetic to speak of, no
// scans -- how fast do the row pieces of a 512^3 pass stream, as a function of (a) the width of a row piece (TX columns x 8 B),
// (b) the stride between the rows of a line (2 MB planes, padded planes), (c) the order a block walks its line in (whole line at
// once = one chunk; two-sweep order over NC chunks: a, b, c of chunks 0..NC-1, then y of chunks NC-1..0).
// 4 read streams + 1 write stream = 40 B/cell, like a y / z pass (x, L, 1/d, y in; y out).
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/tools/tile_copy512 profiles/tools/tile_copy512.hip ; run: tile_copy512 [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <bool NT> __device__ __forceinline__ double ldv(const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// block = TX columns x NS segment threads; a thread owns SEG consecutive cells of a chunk of NS * SEG cells; NC chunks per line.
// ORDER 0: chunk by chunk (all five streams).  ORDER 1: two-sweep order (forward a, b, c over all chunks with the running value
// parked in LDS per chunk, then y backwards).  XCD: 8-way contiguous block order.
template <int SEG, int TX, int NS, bool NT, int ORDER, bool XCD>
__global__ __launch_bounds__(TX * NS) void k_tile(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c, double *__restrict__ y,
                                                  long sl, long outer_stride, int n)
{
    extern __shared__ double park[];
    const int ixl = threadIdx.x % TX, seg = threadIdx.x / TX;
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (XCD) {
        const unsigned nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (nblk % 8 == 0) { const unsigned nl = (lin % 8) * (nblk / 8) + lin / 8; bx = nl % gridDim.x; by = nl / gridDim.x; }
    }
    const long base = (long)by * outer_stride + bx * TX + ixl;
    const int CH = NS * SEG, NC = (n + CH - 1) / CH;
    if (ORDER == 0) {
        for (int ch = 0; ch < NC; ++ch) {
            double av[SEG], bv[SEG], cv[SEG], yv[SEG];
#pragma unroll
            for (int i = 0; i < SEG; ++i) { const int cc = ch * CH + seg * SEG + i; const long e = base + (long)cc * sl; const bool ok = cc < n;
                av[i] = ok ? ldv<NT>(a + e) : 0.0; bv[i] = ok ? ldv<NT>(b + e) : 0.0; cv[i] = ok ? ldv<NT>(c + e) : 0.0; yv[i] = ok ? ldv<NT>(y + e) : 0.0; }
#pragma unroll
            for (int i = 0; i < SEG; ++i) { const int cc = ch * CH + seg * SEG + i; const long e = base + (long)cc * sl; if (cc < n) y[e] = yv[i] + av[i] * bv[i] + cv[i]; }
        }
    } else {
        double keep[SEG];
        for (int ch = 0; ch < NC; ++ch) {
            double av[SEG], bv[SEG], cv[SEG];
#pragma unroll
            for (int i = 0; i < SEG; ++i) { const int cc = ch * CH + seg * SEG + i; const long e = base + (long)cc * sl; const bool ok = cc < n;
                av[i] = ok ? ldv<NT>(a + e) : 0.0; bv[i] = ok ? ldv<NT>(b + e) : 0.0; cv[i] = ok ? ldv<NT>(c + e) : 0.0; }
#pragma unroll
            for (int i = 0; i < SEG; ++i) { keep[i] = av[i] * bv[i] + cv[i]; if (ch + 1 < NC) park[((ch * SEG + i) * NS + seg) * TX + ixl] = keep[i]; }
        }
        for (int ch = NC - 1; ch >= 0; --ch) {
            double yv[SEG];
#pragma unroll
            for (int i = 0; i < SEG; ++i) { const int cc = ch * CH + seg * SEG + i; const long e = base + (long)cc * sl; yv[i] = cc < n ? ldv<NT>(y + e) : 0.0; }
#pragma unroll
            for (int i = 0; i < SEG; ++i) { const int cc = ch * CH + seg * SEG + i; const long e = base + (long)cc * sl;
                const double v = ch + 1 < NC ? park[((ch * SEG + i) * NS + seg) * TX + ixl] : keep[i];
                if (cc < n) y[e] = yv[i] + v; }
        }
    }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512; const long N = (long)n * n * n;
    const long PADMAX = 4096;                       // doubles of padding per plane / per row at most
    const long NA = N + PADMAX * (long)n * n / 8 + (1 << 20);
    double *a, *b, *c, *y;
    CK(hipMalloc(&a, NA * 8)); CK(hipMalloc(&b, NA * 8)); CK(hipMalloc(&c, NA * 8)); CK(hipMalloc(&y, NA * 8));
    CK(hipMemset(a, 0, NA * 8)); CK(hipMemset(b, 0, NA * 8)); CK(hipMemset(c, 0, NA * 8)); CK(hipMemset(y, 0, NA * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        hipEventRecord(e0); for (int r = 0; r < 10; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        hipError_t e = hipGetLastError();
        printf("  %-58s %8.4f ms  %5.2f TB/s%s\n", name, ms, N * 40.0 / ms / 1e9, e == hipSuccess ? "" : "  LAUNCH ERROR");
    };
#define RUN(SEG, TX, NS, NT, ORDER, XCD, label) do { \
        const int CH = NS * SEG, NC = (n + CH - 1) / CH; const size_t lds = ORDER ? (size_t)(NC - 1) * CH * TX * 8 : 0; \
        if (lds > 160 * 1024) { printf("  %-58s skipped (LDS %zu KB)\n", label, lds / 1024); break; } \
        (void)hipFuncSetAttribute((const void *)k_tile<SEG, TX, NS, NT, ORDER, XCD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        timeit(label, [&] { hipLaunchKernelGGL((k_tile<SEG, TX, NS, NT, ORDER, XCD>), dim3(n / TX, n), dim3(TX * NS), lds, 0, a, b, c, y, sl, os, n); }); } while (0)
    for (int dir = 0; dir < 2; ++dir)
        for (int padi = 0; padi < 3; ++padi) {
            // y lines: rows of nx (+pad) doubles, planes of ny rows; z lines: planes of nx*ny (+pad) doubles
            const long pad = padi == 0 ? 0 : padi == 1 ? 32 : 544;          // 256 B ; 4 KB + 256 B
            long sl, os;
            if (dir == 0) { sl = n + (padi ? pad / 8 * 0 + (padi == 1 ? 16 : 48) : 0); os = sl * n; }
            else { sl = (long)n * n + pad; os = n; }
            printf("--- %s lines, n = %d, line stride %ld B (pad %ld B)\n", dir == 0 ? "y" : "z", n, sl * 8, (sl - (dir == 0 ? n : (long)n * n)) * 8);
            RUN(8, 16, 64, true, 0, false, "TX16 x 512 cells, one chunk (round-2 shape), nt");
            RUN(8, 16, 64, false, 0, false, "TX16 x 512 cells, one chunk, plain loads");
            RUN(8, 16, 64, true, 0, true, "TX16 x 512 cells, one chunk, nt, xcd order");
            RUN(8, 32, 32, true, 0, false, "TX32 x 256 cells x 2 chunks, chunk by chunk, nt");
            RUN(8, 32, 32, true, 1, false, "TX32 x 256 cells x 2 chunks, two-sweep order, nt");
            RUN(8, 32, 32, true, 1, true, "TX32 x 256 cells x 2 chunks, two-sweep order, nt, xcd");
            RUN(8, 64, 16, true, 0, false, "TX64 x 128 cells x 4 chunks, chunk by chunk, nt");
            RUN(8, 64, 16, true, 1, false, "TX64 x 128 cells x 4 chunks, two-sweep order, nt");
            RUN(4, 64, 16, true, 0, false, "TX64 x 64 cells x 8 chunks (SEG 4), chunk by chunk, nt");
            RUN(8, 32, 16, true, 1, false, "TX32 x 128 cells x 4 chunks, 512 threads, two-sweep, nt");
            RUN(8, 64, 8, true, 1, false, "TX64 x 64 cells x 8 chunks, 512 threads, two-sweep, nt");
        }
    CK(hipDeviceSynchronize());
    return 0;
}
