// Measure first: what does a grid barrier cost when every participating workgroup sits on ONE XCD (shared L2)?
// A CG iteration of a mid-size mesh (IAEA-3D 38x38x19) is two dependent launches today (7.2 + 4.3 us of kernels, 3.7 us of gap); a
// persistent kernel confined to one XCD would replace the two launch boundaries by two of these barriers.
//   G workgroups are launched; those whose HW_REG_XCC_ID equals `xcc` register and take part, the others leave at once (placement
//   is only observed, never assumed: the number of participants P is whatever registered).  Per round every participant writes a
//   4 KB record (round number in every word), passes the barrier, and checks its neighbour's record word by word.
//   mode 0: stores drained (vmcnt(0)) + relaxed agent atomic arrive, sc1 poll, agent ACQUIRE fence (buffer_inv sc1) + s_dcache_inv; plain loads
//   mode 1: the same arrive, no acquire; the record is read with sc1 loads (L1 bypassed, L2-served)
//   mode 3: as mode 1, the record read with non-temporal loads (__builtin_nontemporal_load: the flavour the solver's tile functions already have)
//   mode 2: agent RELEASE fence (buffer_wbl2 sc1) before the arrive + acquire after (the placement-independent form), plain loads
// Every spin is bounded; a timeout sets state->timeout and everybody leaves.
// build: hipcc -O3 --offload-arch=gfx950 -o xcd_barrier xcd_barrier.hip ; run: ./xcd_barrier [G=256] [threads=512] [rounds=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct State { unsigned arrived, nreg, count, timeout; unsigned long long t0, t1; unsigned stale, pad; };
__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool spin_until(const unsigned *p, unsigned target, unsigned *timeout)
{
    for (int i = 0; i < 4000000; ++i) {
        if (ld_sc1(p) >= target) return true;
        if ((i & 1023) == 1023 && ld_sc1(timeout)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k_bar(State *st, double *data, int rounds, int xcc, int words)
{
    __shared__ int s_widx, s_P, s_ok;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    id &= 0xf;
    const bool part = (int)id == xcc;
    if (threadIdx.x == 0) {
        s_widx = part ? (int)__hip_atomic_fetch_add(&st->nreg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        __hip_atomic_fetch_add(&st->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!part) return;
    if (threadIdx.x == 0) {
        s_ok = spin_until(&st->arrived, gridDim.x, &st->timeout) ? 1 : 0;
        s_P = (int)ld_sc1(&st->nreg);
    }
    __syncthreads();
    if (!s_ok) return;
    const int widx = s_widx, P = s_P;
    double *mine = data + (long)widx * words;
    const double *theirs = data + (long)((widx + 1) % P) * words;
    unsigned stale = 0;
    unsigned long long t0 = 0;
    if (widx == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) mine[i] = (double)r;
        // ---- barrier
        if (MODE == 2) { __syncthreads(); if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&st->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ok = spin_until(&st->count, (unsigned)P * (unsigned)r, &st->timeout) ? 1 : 0;
            if (MODE != 1 && MODE != 3) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __builtin_amdgcn_s_dcache_inv(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
        }
        __syncthreads();
        if (!s_ok) break;
        // ---- check the neighbour's record
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            double v;
            if (MODE == 1) v = __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else if (MODE == 3) v = __builtin_nontemporal_load(theirs + i); else v = theirs[i];
            if (v != (double)r) ++stale;
        }
        // a second barrier so that nobody overwrites a record that is still being checked (the real kernel has two phases as well)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&st->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ok = spin_until(&st->arrived, gridDim.x + (unsigned)P * (unsigned)r, &st->timeout) ? 1 : 0;
        }
        __syncthreads();
        if (!s_ok) break;
    }
    if (stale) atomicAdd(&st->stale, stale);
    if (widx == 0 && threadIdx.x == 0) { st->t0 = t0; st->t1 = __builtin_amdgcn_s_memrealtime(); }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 256, B = argc > 2 ? atoi(argv[2]) : 512, rounds = argc > 3 ? atoi(argv[3]) : 2000, words = 512;
    State *st; double *data;
    CK(hipMalloc((void **)&st, sizeof(State))); CK(hipMalloc((void **)&data, sizeof(double) * words * (size_t)G));
    printf("G = %d workgroups of %d threads, %d rounds of (4 KB record per workgroup, barrier, check the neighbour's record, plain barrier)\n", G, B, rounds);
    for (int xcc = 0; xcc < 8; xcc += 5)
        for (int mode = 0; mode < 4; ++mode)
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(st, 0, sizeof(State))); CK(hipMemset(data, 0, sizeof(double) * words * (size_t)G));
                if (mode == 0) hipLaunchKernelGGL(k_bar<0>, dim3(G), dim3(B), 0, 0, st, data, rounds, xcc, words);
                else if (mode == 1) hipLaunchKernelGGL(k_bar<1>, dim3(G), dim3(B), 0, 0, st, data, rounds, xcc, words);
                else if (mode == 3) hipLaunchKernelGGL(k_bar<3>, dim3(G), dim3(B), 0, 0, st, data, rounds, xcc, words);
                else hipLaunchKernelGGL(k_bar<2>, dim3(G), dim3(B), 0, 0, st, data, rounds, xcc, words);
                CK(hipDeviceSynchronize());
                State h; CK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
                const double us = (double)(h.t1 - h.t0) / 100.0 / rounds;      // s_memrealtime: 100 MHz
                printf("xcc %d mode %d: participants %u of %d, %.2f us per round (two barriers + record), stale words %u, timeout %u\n", xcc, mode, h.nreg, G, us, h.stale, h.timeout);
                fflush(stdout);
                if (h.timeout) { printf("timeout: stopping\n"); return 2; }
            }
    return 0;
}
