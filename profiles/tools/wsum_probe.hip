// Probe for DESIGN.md section 10 (hiding the all-reduce of the single-reduction CG): how long do the alpha-free weighted sums take on one slab?
// c = W.p_new with p_new = r - alpha q + beta p is W.r - alpha W.q + beta W.p: three sums per interface that need no CG scalar and could run while
// the reduction is in flight.  This kernel reads r, q, p, W_lo, W_hi (40 B per cell) of an nx x ny x nz slab and writes the six column sums per z line;
// same thread map as k_endpoint_w (64 columns x 4 z ranges per block, one column per lane, planes one after the other).  Second kernel: the combine
// step (six planes + two scalars -> c_lo, c_hi).  Times per launch with HIP events, 200 launches back to back, buffers re-used (cache-resident).
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/tools/wsum_probe profiles/tools/wsum_probe.hip ; run: wsum_probe [nx ny nz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_wsums(const double *__restrict__ p, const double *__restrict__ r, const double *__restrict__ q,
                                               const double *__restrict__ Wlo, const double *__restrict__ Whi, double *__restrict__ out, int nx, int ny, int nz)
{
    __shared__ double s[6][4][64];
    const int ixl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int ix = blockIdx.x * 64 + ixl;
    const bool valid = ix < nx;
    const long nxy = (long)nx * ny, line = (long)blockIdx.y * nx + ix;
    const int per = (nz + 3) >> 2, k0 = seg * per, k1 = k0 + per < nz ? k0 + per : nz;
    double a[6] = { 0, 0, 0, 0, 0, 0 };
    if (valid) {
#pragma unroll 4
        for (int k = k0; k < k1; ++k) {
            const long e = (long)k * nxy + line;
            const double pv = p[e], rv = r[e], qv = q[e], wl = Wlo[e], wh = Whi[e];
            a[0] = fma(wl, rv, a[0]); a[1] = fma(wl, qv, a[1]); a[2] = fma(wl, pv, a[2]);
            a[3] = fma(wh, rv, a[3]); a[4] = fma(wh, qv, a[4]); a[5] = fma(wh, pv, a[5]);
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) s[j][seg][ixl] = a[j];
    __syncthreads();
    if (seg == 0 && valid) {
#pragma unroll
        for (int j = 0; j < 6; ++j) out[(long)j * nxy + line] = ((s[j][0][ixl] + s[j][1][ixl]) + s[j][2][ixl]) + s[j][3][ixl];
    }
}

__global__ __launch_bounds__(256) void k_combine(const double *__restrict__ sums, const double *__restrict__ scal, double *__restrict__ clo, double *__restrict__ chi, long nxy)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nxy) return;
    const double alpha = scal[0], beta = scal[1];
    clo[i] = fma(beta, sums[2 * nxy + i], fma(-alpha, sums[1 * nxy + i], sums[i]));
    chi[i] = fma(beta, sums[5 * nxy + i], fma(-alpha, sums[4 * nxy + i], sums[3 * nxy + i]));
}

int main(int argc, char **argv)
{
    const int nx = argc > 3 ? atoi(argv[1]) : 256, ny = argc > 3 ? atoi(argv[2]) : 256, nz = argc > 3 ? atoi(argv[3]) : 32;
    const long N = (long)nx * ny * nz, nxy = (long)nx * ny;
    double *v[5], *sums, *scal, *clo, *chi;
    for (int i = 0; i < 5; ++i) { CK(hipMalloc(&v[i], N * 8)); CK(hipMemset(v[i], 0, N * 8)); }
    CK(hipMalloc(&sums, 6 * nxy * 8)); CK(hipMalloc(&scal, 16)); CK(hipMemset(scal, 0, 16)); CK(hipMalloc(&clo, nxy * 8)); CK(hipMalloc(&chi, nxy * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 gr((nx + 63) / 64, ny);
    const int reps = 200;
    for (int round = 0; round < 3; ++round) {
        float ms = 0.f;
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_wsums, gr, dim3(256), 0, 0, v[0], v[1], v[2], v[3], v[4], sums, nx, ny, nz);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_wsums, gr, dim3(256), 0, 0, v[0], v[1], v[2], v[3], v[4], sums, nx, ny, nz);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("k_wsums   %d x %d x %d: %7.2f us per launch, %6.1f GB/s on 40 B per cell\n", nx, ny, nz, us, 40.0 * N / (us * 1e-6) / 1e9);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_combine, dim3((unsigned)((nxy + 255) / 256)), dim3(256), 0, 0, sums, scal, clo, chi, nxy);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_combine %ld lines: %7.2f us per launch\n", nxy, ms * 1e3 / reps);
    }
    return 0;
}
