// nf_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the NeutFEM hot path.
//
// Everything here is fp64 and HBM-bound: stencil differences, batched tridiagonal line
// solves and streaming vector updates.  Design rules:
//   * A^-1 along a grid line is a first-order linear recurrence in each sweep direction, so it
//     is evaluated as a parallel scan of affine maps z -> a z + b instead of a serial Thomas
//     sweep: lanes of one wavefront scan an x-line with cross-lane shuffles (k_schur_x), and
//     y/z-lines are cut into register-resident segments whose summaries are combined through
//     LDS (k_schur_s).  No intermediate of the line solve ever goes to HBM.
//   * consecutive lanes always touch consecutive cells of the unit-stride x axis (coalesced
//     512 B - 1 KiB per wave instruction).
//   * reductions are two-stage with a fixed summation order (bitwise reproducible runs); the
//     scalars of CG (alpha, beta, stop test) never leave the device inside a solve.
#pragma once
#include <hip/hip_runtime.h>

// in-kernel cycle stamps: diagnostic builds only (-DNF_STAMPS, scratch library; never in the shipped one)
#ifdef NF_STAMPS
#define NF_STAMP(buf, i) do { if (buf) (buf)[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define NF_STAMP(buf, i) do { } while (0)
#endif

// waves per SIMD the single-reduction variants of the slab z passes are compiled for (A/B builds: profiles/tools/r04_d_call.sh)
#ifndef NF_SR_Z1_WAVES
#define NF_SR_Z1_WAVES 3
#endif
#ifndef NF_SR_Z2_WAVES
#define NF_SR_Z2_WAVES 4
#endif

namespace nf {

// device-resident state of one CG solve (SchurSolver::SolveSchurImplicit, src/solvers.cpp:577-636)
struct CgScalars {
    double rr, pAp, alpha, beta, rr_new, tol_sq, rhs_norm, tol;
    int done, its, maxit;
    int pend;      // fused CG only: the x += alpha p of the last FIN_RR has not been applied yet (k_schur_x / k_cg_flush do it)
    // Error slot.  Every reduction of the solve that crosses ranks carries one extra double next to its sum: a rank that hit a local
    // error adds a non-zero value there, and a non-finite sum counts as an error too.  The kernel that consumes the reduced value sets
    // `done` and `err` from it -- on every rank from the same bits, so all ranks leave the solve at the same iteration with an error
    // instead of one returning early and the others waiting in the next collective.  1 = non-finite sum, 2 = a rank raised its flag.
    int err;
    // lean CG (CgLean): |r|^2 and the iteration count after the iteration of parity q, so that a kernel whose blocks all read
    // the values of parity q^1 can have its block 0 store those of parity q without a race
    double rr2[2]; int its2[2];
};

// Low-latency readback of a few scalars: a one-thread kernel copies them into host memory that the device can write (mapped, coherent)
// and then stores a sequence number with system-scope release; the host polls the sequence number.  Replaces a 100-byte D2H copy
// through the copy engine + stream drain, which costs ~30 us of idle GPU per check (once per group solve and per outer iteration).
struct HostPub { CgScalars cg; double out[8]; unsigned long long seq; };
__global__ void k_publish(const CgScalars *__restrict__ cg, const double *__restrict__ out, int nout, HostPub *hp, unsigned long long seq)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (cg) hp->cg = *cg;
    for (int i = 0; i < nout && i < 8; ++i) hp->out[i] = out[i];
    __threadfence_system();
    __hip_atomic_store(&hp->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Sum over the 64 lanes of a wavefront, result in every lane.  Data-parallel-primitive moves (row_shr 1/2/4/8 inside each row of
// 16 lanes, then row_bcast 15 and 31 across rows) instead of ds_bpermute shuffles: six dependent steps of a few cycles each where
// the LDS crossbar costs ~100 cycles per step -- the reductions sit on the critical path of every latency-bound kernel here.
// Fixed order (deterministic); out-of-row sources read as zero (bound_ctrl).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
    return v + __hiloint2double(hi, lo);          // lanes outside ROW_MASK receive 0 (old = 0, bound_ctrl): v + 0
}
__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_add<0x111, 0xF>(v);                   // row_shr:1
    v = dpp_add<0x112, 0xF>(v);                   // row_shr:2
    v = dpp_add<0x114, 0xF>(v);                   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);                   // row_shr:8  -> lane 15 of every row holds the row's sum
    v = dpp_add<0x142, 0xA>(v);                   // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63 hold rows 0+1 / 2+3
    v = dpp_add<0x143, 0xC>(v);                   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// ---- cross-lane moves and scans of affine maps z -> A z + B with data-parallel primitives -------------------------------------
// lanes without a valid source (row / wavefront edge, or outside ROW_MASK) receive `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v, double old)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_above(double v) { return dpp_get<0x130, 0xF>(v, v); }   // wave_shl:1 = value of lane + 1
__device__ __forceinline__ double lane_below(double v) { return dpp_get<0x138, 0xF>(v, v); }   // wave_shr:1 = value of lane - 1
// Inclusive scan over groups of LPL consecutive lanes (LPL = 1, 2, 4 ... 64): afterwards lane li holds f_li o ... o f_0 of its group.
// row_shr 1/2/4/8 inside the rows of 16 lanes, then row_bcast 15 / 31 hand the totals of the lower rows upwards; no LDS crossbar.
#define NF_SCAN_STEP(CTRL, D) if (LPL > D) { double Ap = dpp_get<CTRL, 0xF>(A, 1.0), Bp = dpp_get<CTRL, 0xF>(B, 0.0); \
                                             if (pos < D) { Ap = 1.0; Bp = 0.0; } B = A * Bp + B; A = A * Ap; }
__device__ __forceinline__ void scan_maps_up(double &A, double &B, int LPL, int lane)
{
    const int pos = lane & ((LPL < 16 ? LPL : 16) - 1);
    NF_SCAN_STEP(0x111, 1) NF_SCAN_STEP(0x112, 2) NF_SCAN_STEP(0x114, 4) NF_SCAN_STEP(0x118, 8)
    if (LPL > 16) { const double Ap = dpp_get<0x142, 0xA>(A, 1.0), Bp = dpp_get<0x142, 0xA>(B, 0.0); B = A * Bp + B; A = A * Ap; }
    if (LPL > 32) { const double Ap = dpp_get<0x143, 0xC>(A, 1.0), Bp = dpp_get<0x143, 0xC>(B, 0.0); B = A * Bp + B; A = A * Ap; }
}
// The mirror image: lane li holds f_li o f_{li+1} o ... o f_{LPL-1}.  row_shl inside the rows; the totals of the higher rows come down
// through at most two crossbar reads (there is no downward row broadcast).
#define NF_SCAN_STEP_DN(CTRL, D) if (LPL > D) { double Ap = dpp_get<CTRL, 0xF>(A, 1.0), Bp = dpp_get<CTRL, 0xF>(B, 0.0); \
                                                if (pos + D >= W) { Ap = 1.0; Bp = 0.0; } B = A * Bp + B; A = A * Ap; }
__device__ __forceinline__ void scan_maps_down(double &A, double &B, int LPL, int lane)
{
    const int W = LPL < 16 ? LPL : 16, pos = lane & (W - 1);
    NF_SCAN_STEP_DN(0x101, 1) NF_SCAN_STEP_DN(0x102, 2) NF_SCAN_STEP_DN(0x104, 4) NF_SCAN_STEP_DN(0x108, 8)
    if (LPL > 16) {                                              // even rows take the suffix total of the row above (its first lane)
        const int src = (lane & ~31) + 16;
        double Ap = __shfl(A, src, 64), Bp = __shfl(B, src, 64);
        if (lane & 16) { Ap = 1.0; Bp = 0.0; }
        B = A * Bp + B; A = A * Ap;
    }
    if (LPL > 32) {                                              // the lower half takes the total of the upper half (lane 32)
        double Ap = __shfl(A, 32, 64), Bp = __shfl(B, 32, 64);
        if (lane >= 32) { Ap = 1.0; Bp = 0.0; }
        B = A * Bp + B; A = A * Ap;
    }
}

// thread t of the first 256 sums p[t], p[t + 256], ... in that order; the loads go out eight at a time (one memory round trip per
// eight terms instead of one per term: with 82 k partials -- split dot product at 512^3 -- the serial version took ~130 us)
__device__ __forceinline__ double strided_sum256(const double *__restrict__ p, int cnt)
{
    double s = 0.0;
    int i = threadIdx.x;
    if (threadIdx.x >= 256) return 0.0;
    for (; i + 7 * 256 < cnt; i += 8 * 256) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[i + j * 256];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; i < cnt; i += 256) s += p[i];
    return s;
}
// fixed-order block sum; result valid in thread 0.  sred: >= blockDim/64 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *sred)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) sred[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) s += sred[i];
    return s;
}

// three sums at once (the single-reduction CG's accumulation pass: p.q, q.q, r.q): one pair of barriers instead of three.  Same order of
// additions per sum as block_sum.  sred: >= 3 * blockDim / 64 doubles of LDS; results valid in thread 0.
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *sred)
{
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) { sred[w] = a; sred[nw + w] = b; sred[2 * nw + w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sb = 0.0, sc = 0.0;
        for (int i = 0; i < nw; ++i) { sa += sred[i]; sb += sred[nw + i]; sc += sred[2 * nw + i]; }
        a = sa; b = sb; c = sc;
    }
}

// ---------------------------------------------------------------------------------------------
// second reduction stage + CG scalar logic.  One block of 256 threads.
// partials: nq rows of `stride` doubles; each local slab owns a segment [off, off+cnt) of every row.
struct PartSegs { int n; int off[16]; int cnt[16]; };
enum FinOp { FIN_RHS = 0, FIN_PAP = 1, FIN_RR = 2, FIN_SUM = 3 };

__device__ __forceinline__ int reduce_err(double total, double flag)       // error slot of an all-reduced sum (CgScalars::err)
{
    return flag != 0.0 ? 2 : ((total - total) != 0.0 ? 1 : 0);            // x - x is 0 for every finite x, NaN for inf / NaN
}
__device__ __forceinline__ void cg_logic(int op, const double *tot, int nq, CgScalars *cg, double *out, double tol, int maxit, double flag = 0.0)
{
    if (op == FIN_RHS) { cg->err = 0; }
    if (op != FIN_SUM) { const int e = reduce_err(tot[0], flag); if (e) { cg->err = e; cg->done = 1; if (op == FIN_RHS) { cg->rr = tot[0]; cg->its = 0; cg->pend = 0; } return; } }
    if (op == FIN_RHS) {                       // src/solvers.cpp:587-592
        cg->rr = tot[0];
        cg->rhs_norm = sqrt(tot[0]);
        cg->tol = tol;
        cg->tol_sq = tol * tol * cg->rhs_norm * cg->rhs_norm;
        cg->done = 0; cg->its = 0; cg->maxit = maxit; cg->pend = 0;
        cg->rr2[0] = tot[0]; cg->its2[0] = 0;
        if (maxit <= 0) cg->done = 1;
    } else if (op == FIN_PAP) {                // src/solvers.cpp:602-606
        cg->pAp = tot[0];
        cg->pend = 0;                          // the x pass of this iteration has applied the deferred update
        if (fabs(tot[0]) < 1e-30) cg->done = 1;
        else cg->alpha = cg->rr / tot[0];
    } else if (op == FIN_RR) {                 // src/solvers.cpp:613-631
        cg->rr_new = tot[0];
        cg->its += 1;
        cg->pend = 1;
        if (tot[0] < cg->tol_sq) { cg->rr = tot[0]; cg->done = 1; }
        else {
            cg->beta = tot[0] / cg->rr;
            cg->rr = tot[0];
            if (cg->its >= cg->maxit) cg->done = 1;
        }
    } else {
        for (int q = 0; q < nq; ++q) out[q] = tot[q];
    }
}

// reduce_only = 1: write the process-local sums to red[] (an all-reduce over ranks follows, then k_cg_logic)
__global__ __launch_bounds__(256) void k_finalize(int op, const double *__restrict__ partials, PartSegs segs, long stride,
                                                  int nq, CgScalars *__restrict__ cg, double *__restrict__ out,
                                                  double tol, int maxit, int reduce_only, double *__restrict__ red, const double *__restrict__ errsrc = nullptr)
{
    __shared__ double sred[4];
    if ((op == FIN_PAP || op == FIN_RR) && !reduce_only) { if (cg->done) return; }
    double tot[4] = { 0, 0, 0, 0 };
    for (int q = 0; q < nq; ++q) {
        double acc = 0.0;
        for (int sgi = 0; sgi < segs.n; ++sgi) {
            double s = strided_sum256(partials + q * stride + segs.off[sgi], segs.cnt[sgi]);
            s = block_sum(s, sred);
            acc += s;
        }
        tot[q] = acc;
    }
    if (threadIdx.x != 0) return;
    // reduce_only: the sums go into an all-reduce over ranks, followed by this rank's error flag (errsrc, raised by the host)
    if (reduce_only) { for (int q = 0; q < nq; ++q) red[q] = tot[q]; if (errsrc) red[nq] = *errsrc; return; }
    cg_logic(op, tot, nq, cg, out, tol, maxit);
}
__global__ void k_cg_logic(int op, const double *__restrict__ red, int nq, CgScalars *__restrict__ cg, double *__restrict__ out,
                           double tol, int maxit, int with_flag)
{
    if (op == FIN_PAP || op == FIN_RR) { if (cg->done) return; }
    double tot[4] = { 0, 0, 0, 0 };
    for (int q = 0; q < nq; ++q) tot[q] = red[q];
    cg_logic(op, tot, nq, cg, out, tol, maxit, with_flag ? red[nq] : 0.0);   // red[nq]: the all-reduced error flags of the ranks
    if (with_flag == 2) out[nq] = red[nq];                       // FIN_SUM of the outer iteration: the host reads the flags next to the sums
}

// Lean CG (undivided mesh, fused): the two k_finalize launches of an iteration disappear.  The consumer of a reduction sums
// the producer's block partials itself -- every block redundantly, in k_finalize's order, so all blocks get the same bits --
// and its block 0 stores the scalars for the kernels (and the host) that follow:
//   k_cg_rupdate sums the p.q partials of the last direction pass  -> alpha            (FIN_PAP, src/solvers.cpp:602-606)
//   the next iteration's x pass sums the |r|^2 partials of k_cg_rupdate -> beta, stop tests  (FIN_RR, :613-631)
// No block reads a field that another block of the same kernel writes: |r|^2 and the iteration count are kept per parity of
// the iteration (CgScalars::rr2 / its2); `done` is sticky and every block derives the same stop decision on its own.
// k_cg_lean_rr applies the FIN_RR step alone (idempotent: same inputs, same parity slot) so that the host, which reads the
// scalars at the end of a batch, sees the outcome of the batch's last iteration.
__device__ __forceinline__ double strided_total(const double *p, int cnt, double *sred);
// count < 0: partials[0] already holds the total (slab teams: k_finalize with reduce_only + the all-reduce over ranks put it there)
// flag_at > 0 (count >= 0): partials[flag_at] is the error slot that travelled with the partials (multi-rank teams all-reduce the
// vector of block partials itself and let the consumers sum it: no k_finalize launch on the critical path)
struct CgLean { CgScalars *st; const double *partials; int count; int par; int first; int flag_at; };
__device__ __forceinline__ double lean_total(const CgLean &lean, double *sred)
{
    return lean.count < 0 ? lean.partials[0] : strided_total(lean.partials, lean.count, sred);
}
// count < 0: partials[1] is the error slot that travelled with the total (sum of the ranks' flags)
__device__ __forceinline__ int lean_err(const CgLean &lean, double total)
{
    return reduce_err(total, lean.count < 0 ? lean.partials[1] : (lean.flag_at > 0 ? lean.partials[lean.flag_at] : 0.0));
}
__device__ __forceinline__ double strided_total(const double *p, int cnt, double *sred)   // blocks of >= 256 threads; result in every thread
{
    // summed by the first 256 threads only, so that blocks of any size (and k_finalize) produce the same bits: the other
    // threads add exact zeros
    double s = strided_sum256(p, cnt);
    s = block_sum(s, sred);
    if (threadIdx.x == 0) sred[0] = s;
    __syncthreads();
    s = sred[0];
    __syncthreads();
    return s;
}
// FIN_RR for the iteration of parity lean.par; returns true when the solve stops here.  *beta_out is valid when it continues.
// The scalars of the previous iteration are loaded up front (LeanPre), so that a kernel can have them -- and its own tile loads -- in
// flight while the partial sums are reduced, instead of paying one memory round trip after the other.
struct LeanPre { double rr_old, tol_sq; int its_old, maxit; };
__device__ __forceinline__ LeanPre lean_preload(const CgLean &lean)
{
    const CgScalars *st = lean.st; const int qo = lean.par ^ 1;
    LeanPre p; p.rr_old = st->rr2[qo]; p.tol_sq = st->tol_sq; p.its_old = st->its2[qo]; p.maxit = st->maxit;
    return p;
}
__device__ __forceinline__ bool lean_rr_step(const CgLean &lean, const LeanPre &pre, bool writer, double *sred, double *beta_out)
{
    CgScalars *st = lean.st;
    const int q = lean.par;
    const double rr_new = lean_total(lean, sred);
    const int e = lean_err(lean, rr_new);
    if (e) {                                                     // every block (and every rank) takes this branch from the same bits
        if (writer) { st->rr_new = rr_new; st->rr = rr_new; st->err = e; st->done = 1; }
        *beta_out = 0.0;
        return true;
    }
    const double rr_old = pre.rr_old;
    const int its_new = pre.its_old + 1;
    const bool conv = rr_new < pre.tol_sq;
    const bool stop = conv || its_new >= pre.maxit;
    const double beta = rr_new / rr_old;
    if (writer) {
        st->rr2[q] = rr_new; st->its2[q] = its_new;
        st->rr_new = rr_new; st->rr = rr_new; st->its = its_new; st->pend = 1;
        if (!conv) st->beta = beta;
        if (stop) st->done = 1;
    }
    *beta_out = beta;
    return stop;
}
__device__ __forceinline__ bool lean_rr_step(const CgLean &lean, bool writer, double *sred, double *beta_out)
{
    return lean_rr_step(lean, lean_preload(lean), writer, sred, beta_out);
}
// what a direction pass does between issuing its loads and using them (nothing, by default)
struct NoMid { __device__ __forceinline__ bool operator()(double &, double &) const { return false; } };
// hp: publish the scalars to the host right away (end of a batch of iterations: the host is polling, see HostPub)
__global__ __launch_bounds__(256) void k_cg_lean_rr(CgLean lean, HostPub *hp, unsigned long long seq)
{
    __shared__ double sred[4];
    if (!lean.st->done) {
        double beta;
        (void)lean_rr_step(lean, threadIdx.x == 0, sred, &beta);
    }
    if (hp && threadIdx.x == 0) {                                // the same thread wrote the scalars above
        hp->cg = *lean.st;
        __threadfence_system();
        __hip_atomic_store(&hp->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------
// Single-reduction CG on slab teams (RT0-P0): ONE cross-rank reduction per iteration instead of two.
// The reference recurrence (src/solvers.cpp:577-636) needs p.Sp before it can form r_new and |r_new|^2 before it can form the next p:
// two dependent reductions.  Here the reduction of iteration j carries
//     red[0] = p_j.q_j   red[1] = q_j.q_j   red[2] = r_j.q_j   red[3] = |r_j|^2 (MEASURED, of the r the iteration started with)   red[4] = error flags
// and its consumer -- the first pass of iteration j+1 (the endpoint pass of the z lines, which reads p anyway) -- derives
//     alpha_j = |r_j|^2 / p.q          |r_{j+1}|^2 = |r_j|^2 - 2 alpha r.q + alpha^2 q.q   (the algebraic identity for |r - alpha q|^2)
//     beta_j  = |r_{j+1}|^2 / |r_j|^2
// and applies r -= alpha q, x += alpha p, p = r + beta p in one sweep.  The PREDICTED |r_{j+1}|^2 only feeds beta; the next reduction
// measures |r_{j+1}|^2 again, so nothing accumulates, and the stop test |r|^2 < tol^2 |b|^2 is taken on MEASURED values (one apply
// late: the solve that the reference ends after iteration j is ended here by the consumer of reduction j, with the same x_j).
// r.q is measured, not replaced by p.q: on IAEA-3D (blank assemblies with Sigma = 1e15, cond(S) ~ 1e17, |r|^2 swinging by ten orders
// of magnitude between consecutive iterations) the variants that lean on conjugacy or on r_new . r = 0 -- Chronopoulos-Gear, or
// |r_new|^2 = alpha^2 q.q - |r|^2 -- never converge (tests/test_single_reduction_cg.py shows it on the CPU oracle); this one needs
// the reference's iteration counts to within a few per cent there and identical counts on the well-conditioned benchmarks.
struct Cg1 { const double *red; CgScalars *st; int index; };   // index = iteration about to start (0: nothing to consume yet)
__device__ __forceinline__ bool cg1_step(const Cg1 &c, bool writer, double *alpha_out, double *beta_out)
{
    CgScalars *st = c.st;
    const double pq = c.red[0], qq = c.red[1], rq = c.red[2], rr = c.red[3], flag = c.red[4];
    const int j = c.index - 1;                                   // the iteration these sums belong to
    *alpha_out = 0.0; *beta_out = 0.0;
    const int e = flag != 0.0 ? 2 : ((((pq - pq) + (qq - qq)) + ((rq - rq) + (rr - rr))) != 0.0 ? 1 : 0);
    if (e) { if (writer) { st->err = e; st->done = 1; st->rr = rr; st->its = j; st->pend = 0; } return true; }
    if (j >= 1 && rr < st->tol_sq) { if (writer) { st->rr = rr; st->rr_new = rr; st->its = j; st->pend = 0; st->done = 1; } return true; }   // :624-628, on the measured |r_j|^2
    if (fabs(pq) < 1e-30) { if (writer) { st->pAp = pq; st->rr = rr; st->its = j; st->pend = 0; st->done = 1; } return true; }              // :604
    const double alpha = rr / pq;
    double rn = fma(alpha, fma(alpha, qq, -2.0 * rq), rr);       // |r - alpha q|^2
    if (!(rn > 0.0)) rn = 0.0;                                    // cancelled below rounding: restart from the residual (beta = 0)
    const double beta = rn / rr;
    const bool stop = j + 1 >= st->maxit;                        // the reference's loop bound: x still owes alpha_j p_j (k_cg_flush, pend)
    if (writer) {
        st->pAp = pq; st->alpha = alpha; st->beta = beta; st->its = j + 1; st->rr_new = rn;
        st->rr = stop ? rn : rr; st->pend = stop ? 1 : 0;
        if (stop) st->done = 1;
    }
    *alpha_out = alpha; *beta_out = beta;
    return stop;
}
// The four rows of block partials of one iteration -> red[0..3] (+ this rank's error flag -> red[4]): one wavefront per row, every lane
// sums the entries l, l + 64, ... of each slab's segment with eight loads in flight, one cross-lane sum at the end.  No barrier, no
// serial tail: ~3 us where k_finalize (one row and one segment at a time behind block-wide barriers) took 58 us for 4 x 8 x 256 partials.
__global__ __launch_bounds__(256) void k_reduce_rows(const double *__restrict__ partials, PartSegs segs, PartSegs segs3, long stride, double *__restrict__ red,
                                                     const double *__restrict__ errsrc)
{
    const int row = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double *p = partials + (long)row * stride;
    double s = 0.0;
    for (int sg = 0; sg < segs.n; ++sg) {
        const double *q = p + segs.off[sg]; const int cnt = row == 3 ? segs3.cnt[sg] : segs.cnt[sg];   // row 3 (|r|^2) comes from the endpoint pass, whose grid may differ
        int i = lane;
        for (; i + 7 * 64 < cnt; i += 8 * 64) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = q[i + j * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; i < cnt; i += 64) s += q[i];
    }
    s = wave_sum(s);
    if (lane == 0) red[row] = s;
    if (threadIdx.x == 0 && errsrc) red[4] = *errsrc;
}
// the consumer's step on its own (end of a batch of iterations: the host is about to read the scalars), idempotent
__global__ void k_cg1_logic(Cg1 c, HostPub *hp, unsigned long long seq)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!c.st->done && c.index > 0) { double a, b; (void)cg1_step(c, true, &a, &b); }
    if (hp) { hp->cg = *c.st; __threadfence_system(); __hip_atomic_store(&hp->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

// ---------------------------------------------------------------------------------------------
// BuildMatrices, per-DOF diagonals (src/NeutFEM.cpp:1163-1302): C, fission and scatter matrices are diagonal
// (Legendre orthogonality): value = xs * detJ * C-hat_pp, C-hat_pp = prod_axes 2/(2 i_t + 1).  Output is SoA [p][e].
// mode 0 (C): entries <= 1e-14 dropped.  mode 1 (fission / scatter): P0: kept iff |xs| > 1e-14 (value xs*V);
// P>=1: element skipped iff |xs| < 1e-14, entries <= 1e-14 dropped (:1204-1302).
struct ChatArgs { int nloc; double c[27]; };
__global__ void k_cell_coef(const double *__restrict__ xs, double *__restrict__ out, const double *__restrict__ hx,
                            const double *__restrict__ hy, const double *__restrict__ hz, int nx, int ny, long N,
                            int mode, int dim, ChatArgs ch)
{
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < N; e += (long)gridDim.x * blockDim.x) {
        const int ix = (int)(e % nx); const long r = e / nx; const int iy = (int)(r % ny); const int iz = (int)(r / ny);
        const double s = xs[e];
        if (ch.nloc == 1 && mode != 2) {
            const double V = hx[ix] * hy[iy] * hz[iz];
            double v = s * V;
            if (mode == 0) { if (!(fabs(v) > 1e-14)) v = 0.0; }
            else { if (!(fabs(s) > 1e-14)) v = 0.0; }
            out[e] = v;
        } else {
            double detJ = hx[ix] / 2.0;
            if (dim >= 2) detJ *= hy[iy] / 2.0;
            if (dim == 3) detJ *= hz[iz] / 2.0;
            const bool skip = mode >= 1 && fabs(s) < 1e-14;      // mode 2: AssembleWeightedMassMatrix (:1495-1529), any order
            for (int p = 0; p < ch.nloc; ++p) {
                double v = skip ? 0.0 : s * detJ * ch.c[p];
                if (!(fabs(v) > 1e-14)) v = 0.0;
                out[(long)p * N + e] = v;
            }
        }
    }
}
// host layout [e*nloc + p] <-> device layout [p*N + e]
__global__ void k_transpose_dofs(const double *__restrict__ in, double *__restrict__ out, long N, int nloc, int to_soa)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < N * nloc; i += gridDim.x * 256L) {
        const long e = to_soa ? i % N : i / nloc; const int p = to_soa ? (int)(i / N) : (int)(i % nloc);
        if (to_soa) out[i] = in[e * nloc + p]; else out[i] = in[(long)p * N + e];
    }
}

struct Geom {
    int dim, nx, ny, nz, k;
    const double *hx, *hy, *hz;
    // Line operators are stored for UNIT transverse scaling: for transverse Legendre mode a the whole chain matrix
    // (Dirichlet term included, src/NeutFEM.cpp:1458-1489) is T_a times the unit one, T_a = prod 2/(2 a_t + 1), so
    // the LDL^T multipliers are common to all modes and u = T^-1 t does not depend on T_a at all.
    // aLL/aLR: 1-D face block after the k bubbles of the cell have been condensed (RT0: [[2/3,1/3],[1/3,2/3]],
    // RT1: [[1/4,-1/12],..], RT2: [[2/15,1/30],..]); T0 = 2^(dim-1) is the mode-0 scaling (B-hat = -+T0 for RT0).
    double aLL, aLR, T0;
    int dir_lo[3], dir_hi[3];    // Dirichlet flags per direction (src/NeutFEM.cpp:2338-2347 attribute map)
};

// geometric factor of direction d for cell (ix,iy,iz), src/FEM.cpp:795-813 (2D quirk kept)
__device__ __forceinline__ double geom_factor(const Geom &G, int d, int ix, int iy, int iz)
{
    const double hx = G.hx[ix], hy = G.hy[iy], hz = G.hz[iz];
    if (G.dim == 1) return hx / 2.0;
    if (G.dim == 2) return d == 0 ? hy / hx : hx / hy;
    if (d == 0) return 2.0 * hx / (hy * hz);
    if (d == 1) return 2.0 * hy / (hx * hz);
    return 2.0 * hz / (hx * hy);
}
// Dirichlet diagonal term I_f(a) * 2 * D / T_a (src/NeutFEM.cpp:1350, 1458-1489): 2D, 4D/area, 8D/area
__device__ __forceinline__ double dirichlet_term(const Geom &G, int d, int ix, int iy, int iz, double D)
{
    double area = d == 0 ? G.hy[iy] * G.hz[iz] : d == 1 ? G.hx[ix] * G.hz[iz] : G.hx[ix] * G.hy[iy];
    double I = G.dim == 1 ? 1.0 : G.dim == 2 ? 2.0 / area : 4.0 / area;
    return I * 2.0 * D;
}
__device__ __forceinline__ void cell_a(const Geom &G, int d, int ix, int iy, int iz, double D, double &a2, double &a1)
{
    const double a = (1.0 / D) * geom_factor(G, d, ix, iy, iz);
    a2 = G.aLL * a; a1 = G.aLR * a;
    if (G.k == 0) {                                             // src/NeutFEM.cpp:1064 drop threshold on the real entries
        if (!(fabs(G.T0 * a2) > 1e-14)) a2 = 0.0;
        if (!(fabs(G.T0 * a1) > 1e-14)) a1 = 0.0;
    }
}

// One (direction, transverse mode) pass of the Schur apply for RT_k-P_m.  The phi moments (i along the line,
// transverse index a) of a cell couple to the line unknowns of mode a only:
//   faces    t_f = xR_{f-1} - xL_f,   xR = x_0 - sum_l eR_l G_l x_{l+1},  xL = x_0 + sum_l eL_l G_l x_{l+1}
//   solve    T_unit u = t             (bubbles condensed per cell: scalars because M^bb is diagonal for k <= 2)
//   bubbles  v_l = G_l x_{l+1} iM_l / c_e - (eL_l u_c + eR_l u_{c+1}),  c_e = factor_dir / D
//   output   y_0 += T_a (u_{c+1} - u_c),   y_{l+1} += T_a G_l v_l
// NB = number of bubble moments that exist in P_m (min(k, m)); NB = 0 is RT0-P0 (or RT_k-P0).
struct ModeArgs {
    double Ta;
    double eL[2], eR[2], Gc[2], iM[2];
    const double *x[3];          // moment arrays of the input vector: x[0] (along-index 0), x[1], x[2]
    double *y[3];
    const double *Cd[3];         // C diagonal of the same moments (first pass only)
    const double *D;             // diffusion coefficient per cell (for 1/c_e); used when NB > 0
    int dir;
};

// All transverse modes of one direction in ONE launch (they touch disjoint moments): the kernels get the ModeArgs of mode 0
// plus, per mode, T_a and the offset of its moments relative to mode 0's; the mode is the last grid dimension.
struct ModeTab { int n; double Ta[9]; long doff[9][3]; };
__device__ __forceinline__ ModeArgs select_mode(ModeArgs ma, const ModeTab &mt, int m, int nbp1)
{
    if (mt.n > 1) {
        ma.Ta = mt.Ta[m];
        for (int i = 0; i < nbp1; ++i) { const long o = mt.doff[m][i]; ma.x[i] += o; ma.y[i] += o; ma.Cd[i] += o; }
    }
    return ma;
}

// AssembleA + ApplyDirichletToA + SparseLU (src/NeutFEM.cpp:1036-1076,1328-1456; src/solvers.cpp:163)
// for one direction: one thread per grid line assembles the tridiagonal T of that line on the fly
// and stores its LDL^T factor cell-aligned: L[e] couples the lower to the upper face of cell e,
// DR[e] = 1/d'(upper face of e), D0[line] = 1/d'(first face).
// Slab decomposition (z lines only): if_lo / if_hi mark an interface with the neighbouring slab.  The
// interface face is a separator (partition method); the factored chain then covers the interior faces
// only: cells [fs, n-fe_off) with the edge cell's a2 added to the first/last chain diagonal.  Per line it
// also emits what the reduced (separator) system needs: alo/ahi = coupling a1 of the edge cells and
// hlo/hhi = this slab's half of the separator diagonal, a2_edge - a1_edge^2 (T_II^-1)[end,end].
// gfl = a1_lo a1_hi (T_II^-1)[first,last] measures the separator-to-separator coupling through the slab.
struct SlabOut { double *alo, *ahi, *hlo, *hhi, *gfl; };
__global__ void k_factor_lines(Geom G, int d, const double *__restrict__ D, double *__restrict__ L,
                               double *__restrict__ DR, double *__restrict__ D0, long nlines, int if_lo, int if_hi,
                               SlabOut so)
{
    const long line = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (line >= nlines) return;
    int n, ix = 0, iy = 0, iz = 0; long base, sl;
    const long nxy = (long)G.nx * G.ny;
    if (d == 0) { n = G.nx; iy = (int)(line % G.ny); iz = (int)(line / G.ny); base = line * G.nx; sl = 1; }
    else if (d == 1) { n = G.ny; ix = (int)(line % G.nx); iz = (int)(line / G.nx); base = iz * nxy + ix; sl = G.nx; }
    else { n = G.nz; ix = (int)(line % G.nx); iy = (int)(line / G.nx); base = line; sl = nxy; }
    int *ci = d == 0 ? &ix : d == 1 ? &iy : &iz;
    const int fs = if_lo ? 1 : 0, ce = if_hi ? n - 1 : n;       // chain cells [fs, ce)
    double a2, a1, ea2, ea1;
    // first chain face: lower contribution = Dirichlet term or the edge cell below
    *ci = fs;
    double Dc = D[base + (long)fs * sl];
    cell_a(G, d, ix, iy, iz, Dc, a2, a1);
    double dprev;
    double alo = 0.0, a2lo = 0.0;
    if (if_lo) { *ci = 0; cell_a(G, d, ix, iy, iz, D[base], ea2, ea1); alo = ea1; a2lo = ea2; dprev = ea2 + a2; }
    else { *ci = fs; dprev = a2 + (G.dir_lo[d] ? dirichlet_term(G, d, ix, iy, iz, Dc) : 0.0); }
    double ahi = 0.0, a2hi = 0.0;
    if (if_hi) { *ci = n - 1; cell_a(G, d, ix, iy, iz, D[base + (long)(n - 1) * sl], ea2, ea1); ahi = ea1; a2hi = ea2; }
    D0[line] = 1.0 / dprev;
    double prodl = 1.0, g00 = 1.0 / dprev;                      // (T^-1)[0,0] = sum_k (prod_{j<k} l_j)^2 / d'_k
    for (int c = fs; c < ce; ++c) {
        double diag_next, na2 = 0.0, na1 = 0.0;
        if (c + 1 < ce) {
            *ci = c + 1;
            const double Dn = D[base + (long)(c + 1) * sl];
            cell_a(G, d, ix, iy, iz, Dn, na2, na1);
            diag_next = a2 + na2;
        } else if (if_hi) {
            diag_next = a2 + a2hi;
        } else {
            *ci = c;
            diag_next = a2 + (G.dir_hi[d] ? dirichlet_term(G, d, ix, iy, iz, Dc) : 0.0);
        }
        const double l = a1 / dprev;
        const double dn = diag_next - l * a1;
        L[base + (long)c * sl] = l;
        DR[base + (long)c * sl] = 1.0 / dn;
        prodl *= -l; g00 += prodl * prodl / dn;
        dprev = dn;
        if (c + 1 < ce) { Dc = D[base + (long)(c + 1) * sl]; a2 = na2; a1 = na1; }
    }
    if (if_lo) { L[base] = 0.0; DR[base] = 0.0; }
    if (if_hi) { L[base + (long)(n - 1) * sl] = 0.0; DR[base + (long)(n - 1) * sl] = 0.0; }
    if (so.alo) {
        so.alo[line] = alo; so.ahi[line] = ahi;
        so.hlo[line] = a2lo - alo * alo * g00;
        so.hhi[line] = a2hi - ahi * ahi / dprev;                 // (T^-1)[last,last] = 1/d'_last
        so.gfl[line] = alo * ahi * prodl / dprev;
    }
}

// separator values of the partition method (slab interfaces).  The reduced system over the separators of a z line is
//   S_red(s) u_s - G_below u_{s-1} - G_above u_{s+1} = c_s,   c_s = c_hi(slab below) + c_lo(slab above),
// G_r = a_lo a_hi (T_II^-1)[first,last] of slab r (SlabOut::gfl) = coupling of a slab's two separators through it.  G decays
// like 0.27^planes: for thick slabs the system is diagonal to rounding (u = c / S_red, k_separators alone); for thin slabs
// a few Jacobi sweeps u <- (c + G_below u_{s-1} + G_above u_{s+1}) / S_red (k_sep_couple, one more neighbour exchange,
// k_sep_update) reach rounding level; the number of sweeps is fixed at build time from the measured coupling.
// Both sides of an interface evaluate the same expression in the same order, so their copies of u agree bitwise.
__global__ void k_separators(const double *__restrict__ clo, const double *__restrict__ chi, const double *__restrict__ rlo,
                             const double *__restrict__ rhi, const double *__restrict__ sinv_lo, const double *__restrict__ sinv_hi,
                             double *__restrict__ ulo, double *__restrict__ uhi, double *__restrict__ ctlo, double *__restrict__ cthi,
                             long nlines, long ntot, int if_lo, int if_hi, const CgScalars *__restrict__ cg)
{
    // entries: [transverse mode][line] (ntot = modes * nlines); S_red, G are per line (unit-scaled factors serve every mode)
    if (cg && cg->done) return;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= ntot) return;
    const long l = i % nlines;
    if (if_lo) { const double c = rlo[i] + clo[i]; ctlo[i] = c; ulo[i] = c * sinv_lo[l]; }   // below's c_hi + own c_lo (same order on both sides)
    if (if_hi) { const double c = chi[i] + rhi[i]; cthi[i] = c; uhi[i] = c * sinv_hi[l]; }   // own c_hi + above's c_lo
}
// what this slab contributes to its two separators through itself: to the upper one G u_lower, to the lower one G u_upper
__global__ void k_sep_couple(const double *__restrict__ gfl, const double *__restrict__ ulo, const double *__restrict__ uhi,
                             double *__restrict__ elo, double *__restrict__ ehi, long nlines, long ntot, int if_lo, int if_hi,
                             const CgScalars *__restrict__ cg)
{
    if (cg && cg->done) return;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= ntot) return;
    const bool both = if_lo && if_hi;
    const double g = gfl[i % nlines];
    if (if_hi) ehi[i] = both ? g * ulo[i] : 0.0;
    if (if_lo) elo[i] = both ? g * uhi[i] : 0.0;
}
// u = ((c + e_from_below) + e_from_above) / S_red ; relo = the slab below's e_hi, rehi = the slab above's e_lo
__global__ void k_sep_update(const double *__restrict__ ctlo, const double *__restrict__ cthi, const double *__restrict__ elo,
                             const double *__restrict__ ehi, const double *__restrict__ relo, const double *__restrict__ rehi,
                             const double *__restrict__ sinv_lo, const double *__restrict__ sinv_hi, double *__restrict__ ulo,
                             double *__restrict__ uhi, long nlines, long ntot, int if_lo, int if_hi, const CgScalars *__restrict__ cg)
{
    if (cg && cg->done) return;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= ntot) return;
    const long l = i % nlines;
    if (if_lo) ulo[i] = ((ctlo[i] + relo[i]) + elo[i]) * sinv_lo[l];
    if (if_hi) uhi[i] = ((cthi[i] + ehi[i]) + rehi[i]) * sinv_hi[l];
}
// S_red^-1 from the two halves (own + neighbour's)
__global__ void k_sred_inv(const double *__restrict__ own, const double *__restrict__ other, double *__restrict__ out, long n, int own_first)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = 1.0 / (own_first ? own[i] + other[i] : other[i] + own[i]);
}

// BuildDiagonalSchurCache (src/NeutFEM.cpp:483-597): S_inv(e) = 1/(C_ee + sum_faces B_ef^2 / A_ff)
// Slabs: the A_ff of an interface z face sums the edge cells of both slabs; nb_lo / nb_hi hold the neighbour's a2 per
// z line (k_edge_a2, exchanged once per group at cache-build time), nullptr on a domain boundary.
__global__ void k_edge_a2(Geom G, const double *__restrict__ D, double *__restrict__ lo, double *__restrict__ hi, long nlines)
{
    const long line = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (line >= nlines) return;
    const int ix = (int)(line % G.nx), iy = (int)(line / G.nx);
    double a2, a1;
    cell_a(G, 2, ix, iy, 0, D[line], a2, a1); lo[line] = a2;
    cell_a(G, 2, ix, iy, G.nz - 1, D[(long)(G.nz - 1) * G.nx * G.ny + line], a2, a1); hi[line] = a2;
}
__global__ void k_diag_cache(Geom G, const double *__restrict__ D, const double *__restrict__ Cd,
                             double *__restrict__ Sinv, long N, const double *__restrict__ nb_lo, const double *__restrict__ nb_hi)
{
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int ix = (int)(e % G.nx); const long r = e / G.nx; const int iy = (int)(r % G.ny); const int iz = (int)(r / G.ny);
    const long nxy = (long)G.nx * G.ny;
    double S = Cd[e];
    const double b2 = G.T0;                                      // B^2 / A_ff = T0^2 / (T0 A_unit)
    for (int d = 0; d < G.dim; ++d) {
        const int c = d == 0 ? ix : d == 1 ? iy : iz;
        const int n = d == 0 ? G.nx : d == 1 ? G.ny : G.nz;
        const long sl = d == 0 ? 1 : d == 1 ? G.nx : nxy;
        double a2, a1, b2n, b1n;
        cell_a(G, d, ix, iy, iz, D[e], a2, a1);
        // lower face
        double Alo = a2, Ahi = a2;
        if (c > 0) {
            int jx = ix - (d == 0), jy = iy - (d == 1), jz = iz - (d == 2);
            cell_a(G, d, jx, jy, jz, D[e - sl], b2n, b1n); Alo += b2n;
        } else if (d == 2 && nb_lo) Alo += nb_lo[e % nxy];
        else if (G.dir_lo[d]) Alo += dirichlet_term(G, d, ix, iy, iz, D[e]);
        if (c + 1 < n) {
            int jx = ix + (d == 0), jy = iy + (d == 1), jz = iz + (d == 2);
            cell_a(G, d, jx, jy, jz, D[e + sl], b2n, b1n); Ahi += b2n;
        } else if (d == 2 && nb_hi) Ahi += nb_hi[e % nxy];
        else if (G.dir_hi[d]) Ahi += dirichlet_term(G, d, ix, iy, iz, D[e]);
        if (fabs(G.T0 * Alo) > 1e-14) S += b2 / Alo;
        if (fabs(G.T0 * Ahi) > 1e-14) S += b2 / Ahi;
    }
    Sinv[e] = fabs(S) > 1e-14 ? 1.0 / S : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Schur apply, x direction (unit stride along the line).  One wavefront scans 64/LPL lines;
// a lane owns K consecutive cells per chunk of LPL*K cells, NCH chunks cover the line.
//   forward  z_{c+1} = t_{c+1} - L[c] z_c, w = z * dinv ; backward u_f = w_f - L[f] u_{f+1}   (ModeArgs: t, outputs)
// `first`: y_p = Cd_p x_p + ... (the x pass is the first to touch every moment), else accumulate.
typedef double nf_d2 __attribute__((ext_vector_type(2)));
template <bool NT = false>                                       // NT: streaming (non-temporal) loads, see ldg below
__device__ __forceinline__ void ld2(const double *p, long i, bool ok, bool vec, double &a, double &b, bool ok2)
{
    if (vec) {
        nf_d2 v = { 0.0, 0.0 };
        if (ok) v = NT ? __builtin_nontemporal_load(reinterpret_cast<const nf_d2 *>(p + i)) : *reinterpret_cast<const nf_d2 *>(p + i);
        a = v.x; b = v.y;
    } else if (NT) { a = ok ? __builtin_nontemporal_load(p + i) : 0.0; b = ok2 ? __builtin_nontemporal_load(p + i + 1) : 0.0; }
    else { a = ok ? p[i] : 0.0; b = ok2 ? p[i + 1] : 0.0; }
}
// Streaming loads for the big-mesh passes: every array of a pass is larger than the caches and read once per pass, and a load that
// carries the non-temporal hint does not push the lines the other streams still need out of L2 / MALL.  Tiled copy with the y / z
// access pattern of the 256^3 passes, no arithmetic (profiles/tools/r02/tile_copy.hip, one MI355X): 4.99 -> 5.92 TB/s (y), 4.87 -> 6.07 (z);
// non-temporal *stores* lose (5.00 / 4.88 alone, 5.40 / 5.54 combined).  Not for meshes that live in the caches between launches.
template <bool NT>
__device__ __forceinline__ double ldg(const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
// wave-uniform array base + 32-bit byte offset: ONE VGPR addresses the same cell in every array of a pass (global_load v, v_off, s[base])
// instead of a 64-bit address per array and cell -- what keeps the many-array passes (chunked long lines; RT1 / RT2 y and z passes) out of
// scratch.  The host takes such a kernel only while one group's array stays below 4 GiB.
template <bool NT> __device__ __forceinline__ double ldo(const double *p, unsigned byte_off)
{
    const double *q = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p) + byte_off);
    return NT ? __builtin_nontemporal_load(q) : *q;
}
__device__ __forceinline__ void sto(double *p, unsigned byte_off, double v) { *reinterpret_cast<double *>(reinterpret_cast<char *>(p) + byte_off) = v; }

// Fused CG (undivided mesh, any order: the x pass with all its transverse modes touches every moment exactly once): the
// vector updates that follow FIN_RR -- x_sol += alpha p and p = r + beta p
// (src/solvers.cpp:609,630) -- are deferred to the next iteration's x pass, which reads p anyway: p is read once instead
// of three times per iteration and one launch disappears.  Same operands, same expressions: bit-identical iterates.
// The last iteration's x_sol update is applied by k_cg_flush (CgScalars::pend).
// pout: where the new direction p = r + beta p goes (nullptr: in place).  The fused-direction launch (k_apply3) runs the x, y
// and z passes of one apply side by side, so the y / z blocks must still find the old p while the x blocks write the new
// one: they read p and r and form r + beta p themselves (mode "read-only fuse"), and the x blocks write into the other
// buffer of a pair.  Every site evaluates fma(beta, p, r) / fma(alpha, p, x_sol): same bits everywhere.
struct CgFuse { double *p; const double *r; double *xsol; double *pout; };

// One wave-task of the x pass: 64/LPL lines, lane = (line of the task, position in the line).  Returns the lane's share of x.y.
// DPPS: cross-lane traffic of the scans through data-parallel primitives (VALU) instead of ds_bpermute (LDS crossbar).  That halves
// the latency of a lone wave-task (resident kernel: 14.2 k -> 10.9 k cycles per x pass) but costs VALU issue slots: with many waves
// per SIMD the crossbar version is faster (fused-direction launch at 128^3: 76.8 vs 82.9 us per CG iteration on the same box).
// P2 (long lines inside CG, NCH >= 4): two load phases instead of one -- p, r, x_sol first (the deferred CG update consumes r and
// x_sol at once), then L, 1/d and the C diagonal.  Everything in flight at once costs 140 VGPRs at four chunks per lane (three waves
// per SIMD); in two phases the peak is the sweep's own working set.
template <int K, int NCH, bool VEC, int NB, class Mid = NoMid, bool DPPS = false, bool NT = false, bool P2 = false>
__device__ __forceinline__ double schur_x_task(const ModeArgs &ma, const Geom &G, const double *__restrict__ L, const double *__restrict__ DR,
                                               const double *__restrict__ D0, int nx, int ny, long nlines, int lpl_log2, int first,
                                               long task, int lane, bool active, bool fuse, double f_alpha, double f_beta, const CgFuse &fz, Mid mid = Mid())
{
    static_assert(K == 2, "two cells per lane and chunk");
    const int LPL = 1 << lpl_log2, LPW = 64 >> lpl_log2;
    const int li = lane & (LPL - 1), sub = lane >> lpl_log2;
    const long line = task * LPW + sub;
    const bool lv = active && line < nlines;
    const long base = lv ? line * nx : 0;
    const int iy = lv ? (int)(line % ny) : 0, iz = lv ? (int)(line / ny) : 0;
    double xm[NB + 1][NCH][K], yo[NB + 1][NCH][K], Ls[NCH][K], Rs[NCH][K], w[NCH][K], xL[NCH][K], xR[NCH][K], ic[NCH][K];
    double rq[NB + 1][NCH][K], sq[NB + 1][NCH][K], dq[NB > 0 ? NCH : 1][K];   // fused variants: r and x_sol of the same cells; D (1 / c_e)
    double *pw = fz.pout ? fz.pout : fz.p;
    // ---- loads only (every load of the pass is in flight before `mid` runs and before the first value is used)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = (ch * LPL + li) * K;
        const bool ok = lv && c0 < nx, ok2 = lv && c0 + 1 < nx;
        if (!P2) {
            ld2<NT>(L, base + c0, ok, VEC, Ls[ch][0], Ls[ch][1], ok2);
            ld2<NT>(DR, base + c0, ok, VEC, Rs[ch][0], Rs[ch][1], ok2);
        }
#pragma unroll
        for (int q = 0; q <= NB; ++q) {
            ld2<NT>(ma.x[q], base + c0, ok, VEC, xm[q][ch][0], xm[q][ch][1], ok2);
            if (!P2) ld2<NT>(first ? ma.Cd[q] : ma.y[q], base + c0, ok, VEC, yo[q][ch][0], yo[q][ch][1], ok2);
            if (fuse) {                                          // ma.x[q] points into p: the same offset addresses r and x_sol
                const long mo = (ma.x[q] - fz.p) + base + c0;
                ld2<NT>(fz.r, mo, ok, VEC, rq[q][ch][0], rq[q][ch][1], ok2);
                ld2<NT>(fz.xsol, mo, ok, VEC, sq[q][ch][0], sq[q][ch][1], ok2);
            }
        }
        if (NB > 0) ld2<NT>(ma.D, base + c0, ok, VEC, dq[NB > 0 ? ch : 0][0], dq[NB > 0 ? ch : 0][1], ok2);
    }
    if (mid(f_alpha, f_beta)) return 0.0;
    // ---- the deferred CG update of these cells, then the face values
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = (ch * LPL + li) * K;
        const bool ok = lv && c0 < nx, ok2 = lv && c0 + 1 < nx;
        if (fuse) {
#pragma unroll
            for (int q = 0; q <= NB; ++q) {
                const long mo = (ma.x[q] - fz.p) + base + c0;
                const double s0 = fma(f_alpha, xm[q][ch][0], sq[q][ch][0]), s1 = fma(f_alpha, xm[q][ch][1], sq[q][ch][1]);
                xm[q][ch][0] = fma(f_beta, xm[q][ch][0], rq[q][ch][0]); xm[q][ch][1] = fma(f_beta, xm[q][ch][1], rq[q][ch][1]);
                if (VEC) {
                    if (ok) { *reinterpret_cast<double2 *>(fz.xsol + mo) = make_double2(s0, s1);
                              *reinterpret_cast<double2 *>(pw + mo) = make_double2(xm[q][ch][0], xm[q][ch][1]); }
                } else {
                    if (ok) { fz.xsol[mo] = s0; pw[mo] = xm[q][ch][0]; }
                    if (ok2) { fz.xsol[mo + 1] = s1; pw[mo + 1] = xm[q][ch][1]; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            double pl = 0.0, pr = 0.0;
            ic[ch][j] = 0.0;
            if (NB > 0) {
                const double dd = dq[NB > 0 ? ch : 0][j];
                const bool okj = j == 0 ? ok : ok2;
                ic[ch][j] = okj ? dd / geom_factor(G, 0, c0 + j, iy, iz) : 0.0;      // 1 / c_e
#pragma unroll
                for (int l = 0; l < NB; ++l) { const double gx = ma.Gc[l] * xm[l + 1][ch][j]; pl += ma.eL[l] * gx; pr += ma.eR[l] * gx; }
            }
            xL[ch][j] = xm[0][ch][j] + pl; xR[ch][j] = xm[0][ch][j] - pr;
        }
    }
    if (P2) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {                       // second load phase: what the sweeps and the output need
            const int c0 = (ch * LPL + li) * K;
            const bool ok = lv && c0 < nx, ok2 = lv && c0 + 1 < nx;
            ld2<NT>(L, base + c0, ok, VEC, Ls[ch][0], Ls[ch][1], ok2);
            ld2<NT>(DR, base + c0, ok, VEC, Rs[ch][0], Rs[ch][1], ok2);
#pragma unroll
            for (int q = 0; q <= NB; ++q) ld2<NT>(first ? ma.Cd[q] : ma.y[q], base + c0, ok, VEC, yo[q][ch][0], yo[q][ch][1], ok2);
        }
    }
    const double d0 = lv ? D0[line] : 0.0;
    const double z0 = __shfl(-xL[0][0], 0, LPL);
    // ---- forward sweep over chunks
    double carry = z0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        double xn = DPPS ? lane_above(xL[ch][0]) : __shfl_down(xL[ch][0], 1, LPL);
        double xc = 0.0;
        if (ch + 1 < NCH) xc = __shfl(xL[ch + 1 < NCH ? ch + 1 : ch][0], 0, LPL);
        if (li == LPL - 1) xn = xc;
        double t[K];
#pragma unroll
        for (int j = 0; j < K; ++j) t[j] = xR[ch][j] - (j + 1 < K ? xL[ch][j + 1 < K ? j + 1 : j] : xn);
        double A = 1.0, B = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) { B = t[j] - Ls[ch][j] * B; A = -Ls[ch][j] * A; }
        if (DPPS) scan_maps_up(A, B, LPL, lane);
        else for (int d = 1; d < LPL; d <<= 1) {
            const double Ap = __shfl_up(A, d, LPL), Bp = __shfl_up(B, d, LPL);
            if (li >= d) { B = A * Bp + B; A = A * Ap; }
        }
        const double Ae = DPPS ? lane_below(A) : __shfl_up(A, 1, LPL), Be = DPPS ? lane_below(B) : __shfl_up(B, 1, LPL);
        double z = li == 0 ? carry : Ae * carry + Be;
#pragma unroll
        for (int j = 0; j < K; ++j) { z = t[j] - Ls[ch][j] * z; w[ch][j] = z * Rs[ch][j]; }
        if (ch + 1 < NCH) carry = __shfl(z, LPL - 1, LPL);
    }
    // ---- backward sweep over chunks
    double ucarry = 0.0, dot = 0.0;
#pragma unroll
    for (int ch = NCH - 1; ch >= 0; --ch) {
        double Ln = DPPS ? lane_above(Ls[ch][0]) : __shfl_down(Ls[ch][0], 1, LPL);
        double Lc = 0.0;
        if (ch + 1 < NCH) Lc = __shfl(Ls[ch + 1 < NCH ? ch + 1 : ch][0], 0, LPL);
        if (li == LPL - 1) Ln = Lc;
        double Lup[K];
#pragma unroll
        for (int j = 0; j < K; ++j) Lup[j] = j + 1 < K ? Ls[ch][j + 1 < K ? j + 1 : j] : Ln;
        double A = 1.0, B = 0.0;
#pragma unroll
        for (int j = K - 1; j >= 0; --j) { B = w[ch][j] - Lup[j] * B; A = -Lup[j] * A; }
        if (DPPS) scan_maps_down(A, B, LPL, lane);
        else for (int d = 1; d < LPL; d <<= 1) {
            const double Ap = __shfl_down(A, d, LPL), Bp = __shfl_down(B, d, LPL);
            if (li + d < LPL) { B = A * Bp + B; A = A * Ap; }
        }
        const double Ae = DPPS ? lane_above(A) : __shfl_down(A, 1, LPL), Be = DPPS ? lane_above(B) : __shfl_down(B, 1, LPL);
        double u = li == LPL - 1 ? ucarry : Ae * ucarry + Be;
        double uv[K];
#pragma unroll
        for (int j = K - 1; j >= 0; --j) { u = w[ch][j] - Lup[j] * u; uv[j] = u; }
        if (ch > 0) ucarry = __shfl(uv[0], 0, LPL);
        double ulo = DPPS ? lane_below(uv[K - 1]) : __shfl_up(uv[K - 1], 1, LPL);
        double wprev = 0.0;
        if (ch > 0) wprev = __shfl(w[ch > 0 ? ch - 1 : 0][K - 1], LPL - 1, LPL);
        if (li == 0) ulo = (ch == 0 ? z0 * d0 : wprev) - Ls[ch][0] * uv[0];
        double yv[NB + 1][K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double lo = j == 0 ? ulo : uv[j > 0 ? j - 1 : 0];
            yv[0][j] = (first ? yo[0][ch][j] * xm[0][ch][j] : yo[0][ch][j]) + ma.Ta * (uv[j] - lo);
            dot += xm[0][ch][j] * yv[0][j];           // padded cells have x = 0
#pragma unroll
            for (int l = 0; l < NB; ++l) {
                const double v = ma.Gc[l] * xm[l + 1][ch][j] * ma.iM[l] * ic[ch][j] - (ma.eL[l] * lo + ma.eR[l] * uv[j]);
                yv[l + 1][j] = (first ? yo[l + 1][ch][j] * xm[l + 1][ch][j] : yo[l + 1][ch][j]) + ma.Ta * ma.Gc[l] * v;
                dot += xm[l + 1][ch][j] * yv[l + 1][j];
            }
        }
        const int c0 = (ch * LPL + li) * K;
#pragma unroll
        for (int q = 0; q <= NB; ++q) {
            if (VEC) { if (lv && c0 < nx) *reinterpret_cast<double2 *>(ma.y[q] + base + c0) = make_double2(yv[q][0], yv[q][K - 1]); }
            else {
#pragma unroll
                for (int j = 0; j < K; ++j) if (lv && c0 + j < nx) ma.y[q][base + c0 + j] = yv[q][j];
            }
        }
    }
    return dot;
}

template <int K, int NCH, bool VEC, int NB, bool NT = false, bool P2 = false>
__global__ __launch_bounds__(256) void k_schur_x(ModeArgs ma0, ModeTab mt, Geom G, const double *__restrict__ L, const double *__restrict__ DR,
                                                 const double *__restrict__ D0, int nx, int ny, long nlines, int lpl_log2,
                                                 int first, int last, double *__restrict__ partials,
                                                 const CgScalars *__restrict__ cg, CgFuse fz, CgLean lean)
{
    __shared__ double sred[4];
    if (cg && cg->done) return;
    const ModeArgs ma = select_mode(ma0, mt, blockIdx.y, NB + 1);
    const bool lean_on = lean.st && !lean.first;                 // lean CG: this pass consumes the |r|^2 partials (FIN_RR)
    const bool fuse = fz.p != nullptr && (lean.st ? lean_on : cg->its > 0);
    const double alpha0 = fuse ? cg->alpha : 0.0, beta0 = (fuse && !lean.st) ? cg->beta : 0.0;
    LeanPre pre = { 0.0, 0.0, 0, 0 };
    if (lean_on) pre = lean_preload(lean);
    bool stopped = false;
    auto mid = [&](double &fa, double &fb) -> bool {             // runs with the pass's loads in flight
        (void)fa;
        if (lean_on) stopped = lean_rr_step(lean, pre, blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0, sred, &fb);
        return stopped;
    };
    // bit 1 of `first`: XCD-contiguous block order (consecutive workgroups go round-robin to the 8 XCDs; each then walks one eighth of the lines)
    unsigned bx = blockIdx.x;
    if ((first & 2) && gridDim.x % 8 == 0) bx = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    const double dot = schur_x_task<K, NCH, VEC, NB, decltype(mid), false, NT, P2>(ma, G, L, DR, D0, nx, ny, nlines, lpl_log2, first & 1, (long)bx * 4 + (threadIdx.x >> 6),
                                                                               threadIdx.x & 63, true, fuse, alpha0, beta0, fz, mid);
    if (stopped) return;                                         // decided inside, the same in every block: nothing to reduce
    if (last && partials) {
        const double s = block_sum(dot, sred);
        if (threadIdx.x == 0) partials[(long)blockIdx.y * gridDim.x + bx] = s;   // by position: the sum order does not depend on the block order
    }
}

// ---------------------------------------------------------------------------------------------
// Schur apply, y or z direction.  Thread = (ix lane, segment of SEG cells along the line) with
// the segment held in registers; segment summaries (affine maps) are exchanged through LDS.
// Always accumulates into y (the x pass ran first).  Grid: (ceil(nx/TX), n_outer).
// DIR (1 = y, 2 = z) only tags the instantiation so profilers list the two passes separately.
// SLAB = true (z lines of a slab with interfaces, RT0-P0): the kernel works on the interior chain of the
// slab-local line.  mode 1 = endpoint response (partition method, step 1): solve with the plain
// neighbour cells as boundary data, emit c_lo = -x_edge - a_lo u_first, c_hi = x_edge - a_hi u_last,
// touch no y.  mode 2 = final solve with the separator values u_lo/u_hi folded into the boundary data,
// accumulate y on the chain cells and on the edge cells.  mode 3 = the same solve, but instead of y it stores the face
// unknowns themselves, negated: the z currents of the slab (current reconstruction on decomposed meshes).
struct SlabArgs {
    int if_lo, if_hi, mode, xcd;
    // fold = 1 (mode 2, no separator sweeps): the separator values are formed here from the exchanged planes instead of by
    // k_separators -- u_lo = (r_lo + c_lo) S_red^-1, u_hi = (c_hi + r_hi) S_red^-1, the same expressions in the same order on
    // both sides of an interface, so the two copies agree bitwise
    int fold; const double *rlo, *rhi, *sinv_lo, *sinv_hi;
    int wsmin;                                // segments from which the wavefront scan of the summaries replaces the serial loops (0: default 64)
    // x || y on small slabs (team_schur_apply): the y pass runs beside the x pass on a second stream and stores its increment alone
    // (noacc) into a vector of its own; the accumulation pass of the z lines adds that vector (yadd) to y on the way
    int noacc; const double *yadd;
    const double *alo, *ahi, *ulo, *uhi;      // per line
    double *clo, *chi;                        // per line (mode 1 outputs)
    double *jz;                               // mode 3: J = -u on the slab's own z faces [((face) * nx * ny + line) * nfa + amode] (Sol_J_, src/solvers.cpp:228)
    double *jzb;                              // mode 3, RT1+: the z bubbles of the slab's cells [cell * ni + l + k * amode] (src/FEM.cpp:377-397)
    int nfa, ni, amode[9];                    // RT face / interior DOFs per face / cell and direction; RT transverse index of every P mode
    // single-reduction CG (template flag SR, see Cg1): mode 1 consumes the reduction of the previous iteration and carries r -= alpha q next
    // to the fused x / p update, leaving its blocks' shares of |r|^2 in sr_part; mode 2 leaves p.q, q.q and r.q in three rows of partials
    Cg1 sr; double *sr_r; const double *sr_q; double *sr_part; long sr_stride;
};
// One tile (TX columns x one line each, NSEG segments) of a y / z pass.  tid = thread index within the tile; threads with
// act == false only keep the barriers company.  acc: accumulate into y (the x pass ran before) or store the increment alone
// (fused-direction launch: every direction has an output vector of its own).  fro ("read-only fuse"): the input vector is
// r + beta x, formed on the fly (see CgFuse).  Returns the thread's share of x.y.
// SF (slab variants): the instantiation that can carry the fused CG update (r and x_sol of the cells in registers: 160 instead of
// 128 VGPRs); the accumulation / emit passes use the one without
// ZW (plain RT0-P0 lines): the pass's share of x.y is not summed as x_i y_i over the cells -- x.(S_d x) = T_a t^T A^-1 t with
// A = L D L^T is T_a sum_faces z_f^2 / d_f = T_a sum z_f w_f, and both factors are at hand in the forward sweep.  The x pass then
// contributes x.(C x + S_x x) and the consumer adds the three sets of partials: the backward half of a y / z pass needs no x at all
// (eight doubles fewer live through the scans, and the chunked long-line pass has nothing to re-read for its parked chunk).
// XC (k_cg_xcd: the producers of x and r are other workgroups of the SAME launch on the same XCD): every load of x and r bypasses the
// compute unit's L1 -- the overlap cell too
template <int SEG, int DIR, bool SLAB, int NB, bool SF = false, bool NTS = false, class Mid = NoMid, bool ZW = false, bool XC = false, bool SR = false>   // NTS: streaming loads in a slab variant
__device__ __forceinline__ double schur_s_tile(const ModeArgs &ma, const Geom &G, const double *__restrict__ L, const double *__restrict__ DR,
                                               const double *__restrict__ D0, int n, long sl, long outer_stride, int nx, int TX, int NSEG,
                                               unsigned bx, unsigned by, unsigned bz, unsigned gy, int tid, bool act, double *sm,
                                               const SlabArgs &sa, const CgFuse &fz, bool fuse, bool fro, double f_alpha, double f_beta, bool acc,
                                               long long *stamp = nullptr, Mid mid = Mid(), double *extra = nullptr)
{
    static_assert(!SR || (SLAB && NB == 0), "single-reduction CG: RT0-P0 slab passes");
    constexpr bool NT = SLAB ? NTS : SF;                         // big meshes: streaming loads (ldg); SF doubles as that flag on undivided meshes
    const double *x = ma.x[0];                                   // no __restrict__: the fused slab pass rewrites this vector (fz.p)
    double *__restrict__ y = ma.y[0];
    const int T = TX * NSEG;
    // Segment summaries: every thread composes the maps before (after) its own serially from LDS.  From 64 segments on (lines of
    // >= 505 cells) whole wavefronts scan the summaries of a column with cross-lane shuffles instead (lane = segment, log2 steps)
    // and hand every thread its incoming value through LDS: one more barrier per sweep, which only pays for the longest loops
    // (measured on one box, s_wsmin: 512-cell lines 77 -> 72 us, but 256-cell lines 134 -> 141 us, 384-cell 513 -> 540 us, 128-cell
    // 59 -> 66 us).  Needs full, aligned wavefronts.
    // (one wavefront spans a column's summaries: at most 64 segments -- a wider scan would need shuffles across wavefronts; longer
    // lines keep the serial loops)
    const bool wscan = NSEG >= (sa.wsmin > 0 ? sa.wsmin : 64) && NSEG <= 64 && (T & 63) == 0;
    const int NSP = wscan ? (NSEG | 1) : NSEG;                   // odd row length: conflict-free column-major rows
    const int TP = wscan ? TX * NSP : T;
    double *sA = sm, *sB = sm + TP, *sA2 = sm + 2 * TP, *sB2 = sm + 3 * TP, *sZ0 = sm + 4 * TP;
    act = act && tid < T;
    const int ixl = act ? tid % TX : 0, seg = act ? tid / TX : 0;
    const int si = wscan ? ixl * NSP + seg : seg * TX + ixl;     // this thread's slot in the summary arrays
    const int ix = bx * TX + ixl;
    const bool valid = act && ix < nx;
    long base = (long)by * outer_stride + ix;
    const long lineid = (long)by * nx + ix;
    const long lm = SLAB ? (long)bz * ((long)nx * gy) + lineid : 0;   // slab exchange planes: [mode][line]
    const long roff = fro ? (long)(fz.r - fz.p) : 0;                 // r relative to the input vector (same offset for every moment)
    // slab chain: cells [fs, fs+n) of the local line; x just outside the chain is real data (edge cells)
    double x_before = 0.0, x_after = 0.0, a_lo = 0.0, a_hi = 0.0, u_lo = 0.0, u_hi = 0.0, xe_lo = 0.0, xe_hi = 0.0;
    long edge_lo = 0, edge_hi = 0;
    if (SLAB) {
        const int fs = sa.if_lo ? 1 : 0;
        const int nloc = n;                                      // local cells on the line
        n = nloc - fs - (sa.if_hi ? 1 : 0);
        edge_lo = base; edge_hi = base + (long)(nloc - 1) * sl;
        base += (long)fs * sl;
        if (valid) {
            // the edge cells sit between a separator and the chain: towards the chain they contribute xR (lower edge) / xL (upper
            // edge), towards the separator xL / xR -- for RT0 all four are the cell value itself
            if (sa.if_lo) {
                a_lo = sa.alo[lineid]; x_before = x[edge_lo];
                if (fuse) { if (seg == 0) fz.xsol[edge_lo] = fma(f_alpha, x_before, fz.xsol[edge_lo]);
                            x_before = fma(f_beta, x_before, (SR && SF) ? fma(-f_alpha, sa.sr_q[edge_lo], fz.r[edge_lo]) : fz.r[edge_lo]); }
                xe_lo = x_before;
                if (NB > 0) {
                    const double g1 = ma.Gc[0] * ma.x[1][edge_lo], g2 = NB > 1 ? ma.Gc[1] * ma.x[2][edge_lo] : 0.0;
                    xe_lo = x_before + ma.eL[0] * g1 + (NB > 1 ? ma.eL[1] * g2 : 0.0);            // xL: towards the separator below
                    x_before = x_before - ma.eR[0] * g1 - (NB > 1 ? ma.eR[1] * g2 : 0.0);          // xR: towards the chain
                }
                if (sa.mode >= 2) { u_lo = sa.fold ? (sa.rlo[lm] + sa.clo[lm]) * sa.sinv_lo[lineid] : sa.ulo[lm]; x_before -= a_lo * u_lo; }
            }
            if (sa.if_hi) {
                a_hi = sa.ahi[lineid]; x_after = x[edge_hi];
                if (fuse) { if (seg == 0) fz.xsol[edge_hi] = fma(f_alpha, x_after, fz.xsol[edge_hi]);
                            x_after = fma(f_beta, x_after, (SR && SF) ? fma(-f_alpha, sa.sr_q[edge_hi], fz.r[edge_hi]) : fz.r[edge_hi]); }
                xe_hi = x_after;
                if (NB > 0) {
                    const double g1 = ma.Gc[0] * ma.x[1][edge_hi], g2 = NB > 1 ? ma.Gc[1] * ma.x[2][edge_hi] : 0.0;
                    xe_hi = x_after - ma.eR[0] * g1 - (NB > 1 ? ma.eR[1] * g2 : 0.0);             // xR: towards the separator above
                    x_after = x_after + ma.eL[0] * g1 + (NB > 1 ? ma.eL[1] * g2 : 0.0);           // xL: towards the chain
                }
                if (sa.mode >= 2) { u_hi = sa.fold ? (sa.chi[lm] + sa.rhi[lm]) * sa.sinv_hi[lineid] : sa.uhi[lm]; x_after += a_hi * u_hi; }
            }
        }
    }
    const bool wr = !SLAB || sa.mode == 2;
    const int c0 = seg * SEG;
    double xv[SEG + 1], Lv[SEG + 1], Rv[SEG], yo[SEG];          // xv: x_0 moment; for NB > 0 xL / xR are derived below
    double x1[NB > 0 ? SEG : 1], x2[NB > 1 ? SEG : 1], icv[NB > 0 ? SEG : 1];
    double xLn = 0.0;                                            // xL of the first cell of the next segment
    // ---- loads only.  Every load of this phase is issued before the first use of a loaded value: a use inside a predicated region
    // makes the compiler wait for ALL outstanding memory operations there (s_waitcnt vmcnt(0) per region), i.e. one memory round
    // trip per cell instead of one per phase (measured with in-kernel stamps: 9.6 k cycles for the 36 loads of a fused y / z block).
    if (SLAB && !SF) fuse = false;
    const bool fr = SLAB ? fuse : fro;                           // the input vector is r + beta x (wave-uniform)
    // higher-order passes of undivided meshes touch up to eleven arrays per cell: 32-bit byte offsets from the wave-uniform bases (ldo / sto)
    constexpr bool O32 = NB > 0 && !SLAB;
    const unsigned ob8 = (unsigned)(base * 8), sl8 = (unsigned)(sl * 8);
#define NF_A8(c) (ob8 + (unsigned)(c) * sl8)
    double rv[(SLAB && !SF) ? 1 : SEG + 1], sv[(SLAB && SF) ? SEG : 1];   // r of the same cells; slab fuse: x_sol of the owned cells
    double qv[(SR && SF) ? SEG + 1 : 1], rq_[(SR && !SF) ? SEG : 1];      // SR: q of the same cells (endpoint pass); r of the owned cells (accumulation pass)
    double srr = 0.0, sqq = 0.0, srq = 0.0;
    double r1[NB > 0 ? SEG + 1 : 1], r2[NB > 1 ? SEG + 1 : 1], v1a[NB > 0 ? SEG + 1 : 1], v2a[NB > 1 ? SEG + 1 : 1], Dv[NB > 0 ? SEG : 1];
#pragma unroll
    for (int i = 0; i <= SEG; ++i) {
        const int c = c0 + i; const bool ok = valid && c < n;
        const long a = base + (long)c * sl;
        // the overlap cell (i == SEG) is the next segment's first: keep that line for it (see k_schur_c on why this is not a ternary on i)
        if (O32) xv[i] = ok ? ldo<NT>(x, NF_A8(c)) : 0.0;
        else if (i < SEG || XC) xv[i] = ok ? ldg<NT>(x + a) : 0.0; else xv[i] = ok ? x[a] : 0.0;
        if (!SLAB || SF) rv[(SLAB && !SF) ? 0 : i] = (fr && ok) ? (SLAB ? fz.r[a] : (O32 ? ldo<NT>(x + roff, NF_A8(c)) : ldg<NT>(x + a + roff))) : 0.0;
        if (SLAB && SF && i < SEG) sv[(SLAB && SF) ? i : 0] = (fuse && ok) ? fz.xsol[a] : 0.0;
        if (SR && SF) qv[(SR && SF) ? i : 0] = (fuse && ok) ? sa.sr_q[a] : 0.0;
        if (O32) { Lv[i] = ok ? ldo<NT>(L, NF_A8(c)) : 0.0; if (i < SEG) Rv[i] = ok ? ldo<NT>(DR, NF_A8(c)) : 0.0; }
        else {
            if (i < SEG) Lv[i] = ok ? ldg<NT>(L + a) : 0.0; else Lv[i] = ok ? L[a] : 0.0;
            if (i < SEG) Rv[i] = ok ? ldg<NT>(DR + a) : 0.0;
        }
        if (NB > 0 && O32) {
            v1a[i] = ok ? ldo<XC>(ma.x[1], NF_A8(c)) : 0.0;
            if (NB > 1) v2a[NB > 1 ? i : 0] = ok ? ldo<XC>(ma.x[2], NF_A8(c)) : 0.0;
            r1[i] = (fro && ok) ? ldo<XC>(ma.x[1] + roff, NF_A8(c)) : 0.0;
            if (NB > 1) r2[NB > 1 ? i : 0] = (fro && ok) ? ldo<XC>(ma.x[2] + roff, NF_A8(c)) : 0.0;
            if (i < SEG) Dv[i] = ok ? ldo<false>(ma.D, NF_A8(c)) : 0.0;
        } else if (NB > 0) {
            v1a[i] = ok ? ldg<XC>(ma.x[1] + a) : 0.0;
            if (NB > 1) v2a[NB > 1 ? i : 0] = ok ? ldg<XC>(ma.x[2] + a) : 0.0;
            r1[i] = (!SLAB && fro && ok) ? ldg<XC>(ma.x[1] + a + roff) : 0.0;
            if (NB > 1) r2[NB > 1 ? i : 0] = (!SLAB && fro && ok) ? ldg<XC>(ma.x[2] + a + roff) : 0.0;
            if (i < SEG) Dv[i] = ok ? ma.D[a] : 0.0;
        }
    }
    double dinv_s = 0.0;
    if (valid && c0 < n) dinv_s = c0 == 0 ? D0[lineid] : DR[base + (long)(c0 - 1) * sl];
    NF_STAMP(stamp, 3);
    if (mid(f_alpha, f_beta)) return 0.0;                        // e.g. the reduction that yields beta: runs with the loads above in flight
    // ---- arithmetic on the loaded values (cells outside the line hold zeros throughout)
#pragma unroll
    for (int i = 0; i <= SEG; ++i) {
        const int c = c0 + i; const bool ok = valid && c < n;
        if ((!SLAB || SF) && fr) {
            if (SR && SF) rv[(SLAB && !SF) ? 0 : i] = fma(-f_alpha, qv[(SR && SF) ? i : 0], rv[(SLAB && !SF) ? 0 : i]);   // r -= alpha q (src/solvers.cpp:610)
            if (SLAB && SF && i < SEG) sv[(SLAB && SF) ? i : 0] = fma(f_alpha, xv[i], sv[(SLAB && SF) ? i : 0]);   // owned cell; the overlap cell (i == SEG) belongs to the next segment
            xv[i] = fma(f_beta, xv[i], rv[(SLAB && !SF) ? 0 : i]);
        }
        if (SR && SF && i < SEG) { const double rn = fr ? rv[(SLAB && !SF) ? 0 : i] : xv[i]; srr += ok ? rn * rn : 0.0; }   // |r|^2 of the owned cells (first iteration: p = r)
        if (SLAB && valid && c == n) xv[i] = x_after;
        if (NB > 0) {
            double v1 = v1a[i], v2 = NB > 1 ? v2a[NB > 1 ? i : 0] : 0.0;
            if (!SLAB && fro) { v1 = fma(f_beta, v1, r1[i]); if (NB > 1) v2 = fma(f_beta, v2, r2[NB > 1 ? i : 0]); }
            if (i < SEG) {
                x1[i] = v1; if (NB > 1) x2[NB > 1 ? i : 0] = v2;
                // cell coordinates for 1/c_e = D / factor_dir
                int cx = ix, cy = 0, cz = 0;
                if (DIR == 1) { cy = c; cz = by; } else { cy = by; cz = c + ((SLAB && sa.if_lo) ? 1 : 0); }   // chain cell c is local cell fs + c
                icv[i] = ok ? Dv[i] / geom_factor(G, DIR, cx, cy, cz) : 0.0;
            } else {
                xLn = xv[i] + ma.eL[0] * ma.Gc[0] * v1 + (NB > 1 ? ma.eL[1] * ma.Gc[1] * v2 : 0.0);
                if (SLAB && valid && c == n) xLn = x_after;          // the cell above the chain is the upper edge cell (xL, separator folded in)
            }
        }
    }
    if (SLAB && SF && fuse) {                                    // x_sol += alpha p of the owned cells
#pragma unroll
        for (int i = 0; i < SEG; ++i) if (valid && c0 + i < n) fz.xsol[base + (long)(c0 + i) * sl] = sv[(SLAB && SF) ? i : 0];
    }
    double t[NB > 0 ? SEG : 1];                                  // NB == 0: t_i = xv_i - xv_{i+1} is recomputed where needed (registers)
    double P = 1.0, lz = 0.0;
    double xL0 = xv[0];                                          // xL of this segment's first cell
    if (NB == 0) {
        (void)t;
    } else {
        double xLc[SEG + 1], xRc[SEG];
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
            const double g1 = ma.Gc[0] * x1[i], g2 = NB > 1 ? ma.Gc[1] * x2[i] : 0.0;
            xLc[i] = xv[i] + ma.eL[0] * g1 + (NB > 1 ? ma.eL[1] * g2 : 0.0);
            xRc[i] = xv[i] - ma.eR[0] * g1 - (NB > 1 ? ma.eR[1] * g2 : 0.0);
        }
        xLc[SEG] = xLn; xL0 = xLc[0];
#pragma unroll
        for (int i = 0; i < SEG; ++i) t[i] = xRc[i] - xLc[i + 1];
    }
#pragma unroll
    for (int i = 0; i < SEG; ++i) { const double ti = NB == 0 ? xv[i] - xv[i + 1] : t[NB > 0 ? i : 0]; lz = ti - Lv[i] * lz; P = -Lv[i] * P; }
    NF_STAMP(stamp, 4);
    if (act) { sA[si] = P; sB[si] = lz; }
    if (act && seg == 0) sZ0[ixl] = x_before - xL0;
    __syncthreads();
    NF_STAMP(stamp, 5);
    if (wscan) {
        // forward: incoming value of segment s = (f_{s-1} o ... o f_0)(z0); inclusive Hillis-Steele over the lanes of a column
        int W = 16; while (W < NSEG) W <<= 1;
        const int lane = tid & 63, cw = 64 / W, s_ = lane & (W - 1);
        if (tid < T)
            for (int c = (tid >> 6) * cw + lane / W; c < TX; c += (T >> 6) * cw) {
                double A = s_ < NSEG ? sA[c * NSP + s_] : 1.0, B = s_ < NSEG ? sB[c * NSP + s_] : 0.0;
                for (int d = 1; d < W; d <<= 1) {
                    const double Ap = __shfl_up(A, d, W), Bp = __shfl_up(B, d, W);
                    if (s_ >= d) { B = A * Bp + B; A = A * Ap; }
                }
                const double Ae = __shfl_up(A, 1, W), Be = __shfl_up(B, 1, W), z0 = sZ0[c];
                if (s_ < NSEG) sB[c * NSP + s_] = s_ == 0 ? z0 : Ae * z0 + Be;
            }
        __syncthreads();
    }
    if (SLAB && fuse && valid) {                                 // every read of the old p in this block is behind the barrier
#pragma unroll
        for (int i = 0; i < SEG; ++i) if (c0 + i < n) fz.p[base + (long)(c0 + i) * sl] = xv[i];
        if (seg == 0) { if (sa.if_lo) fz.p[edge_lo] = xe_lo; if (sa.if_hi) fz.p[edge_hi] = xe_hi; }
        if (SR && SF) {                                          // the updated residual: other threads of the block read the old one above (overlap and edge cells)
#pragma unroll
            for (int i = 0; i < SEG; ++i) if (c0 + i < n) sa.sr_r[base + (long)(c0 + i) * sl] = rv[(SLAB && !SF) ? 0 : i];
            // edge cells: every thread of the column formed r - alpha q for its p above; the owner (segment 0) forms it once more here,
            // behind the barrier, rather than keeping it alive across the scans -- the old r and q are still in place, it is their only writer
            if (seg == 0) {
                if (sa.if_lo) { const double re = fma(-f_alpha, sa.sr_q[edge_lo], fz.r[edge_lo]); sa.sr_r[edge_lo] = re; srr += re * re; }
                if (sa.if_hi) { const double re = fma(-f_alpha, sa.sr_q[edge_hi], fz.r[edge_hi]); sa.sr_r[edge_hi] = re; srr += re * re; }
            }
        }
    }
    if (SR && SF && !fuse && valid && seg == 0) {                // first iteration: r = p, untouched
        if (sa.if_lo) { const double re = x[edge_lo]; srr += re * re; }
        if (sa.if_hi) { const double re = x[edge_hi]; srr += re * re; }
    }
    double z;
    if (wscan) z = sB[si];
    else { z = sZ0[ixl]; for (int s = 0; s < seg; ++s) z = sA[s * TX + ixl] * z + sB[s * TX + ixl]; }
    const double zin = z;
    double w[SEG];
    double zw = (ZW && c0 == 0) ? zin * (zin * dinv_s) : 0.0;    // the line's first face (lanes outside the mesh hold zeros)
#pragma unroll
    for (int i = 0; i < SEG; ++i) { const double ti = NB == 0 ? xv[i] - xv[i + 1] : t[NB > 0 ? i : 0]; z = ti - Lv[i] * z; w[i] = z * Rv[i]; if (ZW) zw += z * w[i]; }
    if (ZW) asm volatile("" : "+v"(zw));                         // summed here, not sunk to the end of the kernel (which would keep every z alive)
    double Q = 1.0, lu = 0.0;
#pragma unroll
    for (int i = SEG - 1; i >= 0; --i) { lu = w[i] - Lv[i + 1] * lu; Q = -Lv[i + 1] * Q; }
    if (act) { sA2[si] = Q; sB2[si] = lu; }
    // y is only needed by the output stage: issue its loads here so they fly during the barrier + backward scan
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
        const int c = c0 + i; const bool ok = acc && valid && c < n && wr;
        yo[i] = ok ? (O32 ? ldo<NT>(y, NF_A8(c)) : ldg<NT>(y + base + (long)c * sl)) : 0.0;
        if (SLAB && NB == 0 && sa.yadd) yo[i] += ok ? sa.yadd[base + (long)c * sl] : 0.0;     // same order on every cell: (y_x + y_y) + z part
    }
    double y1o[NB > 0 ? SEG : 1], y2o[NB > 1 ? SEG : 1];
    if (NB > 0) {
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
            const bool okl = acc && valid && c0 + i < n && wr; const long a = base + (long)(c0 + i) * sl;
            y1o[i] = okl ? (O32 ? ldo<false>(ma.y[1], NF_A8(c0 + i)) : ma.y[1][a]) : 0.0;
            if (NB > 1) y2o[NB > 1 ? i : 0] = okl ? (O32 ? ldo<false>(ma.y[2], NF_A8(c0 + i)) : ma.y[2][a]) : 0.0;
        }
    }
    NF_STAMP(stamp, 6);
    __syncthreads();
    NF_STAMP(stamp, 7);
    if (SR && !SF) {                                             // r of the owned cells (for r.q): asked for behind the barrier, so that it is not alive across
#pragma unroll                                                   // the scans (the pass sits at 128 VGPRs = four blocks of 256 threads per CU); it flies during the backward replay
        for (int i = 0; i < SEG; ++i) rq_[(SR && !SF) ? i : 0] = (valid && c0 + i < n && wr) ? sa.sr_r[base + (long)(c0 + i) * sl] : 0.0;
    }
    double u = 0.0;
    if (wscan) {
        // backward: incoming value of segment s = (f_{s+1} o ... o f_{NSEG-1})(0) = the B part of the suffix composition at s + 1
        int W = 16; while (W < NSEG) W <<= 1;
        const int lane = tid & 63, cw = 64 / W, s_ = lane & (W - 1);
        if (tid < T)
            for (int c = (tid >> 6) * cw + lane / W; c < TX; c += (T >> 6) * cw) {
                double A = s_ < NSEG ? sA2[c * NSP + s_] : 1.0, B = s_ < NSEG ? sB2[c * NSP + s_] : 0.0;
                for (int d = 1; d < W; d <<= 1) {
                    const double Ap = __shfl_down(A, d, W), Bp = __shfl_down(B, d, W);
                    if (s_ + d < W) { B = A * Bp + B; A = A * Ap; }
                }
                const double Bn = __shfl_down(B, 1, W);
                if (s_ < NSEG) sA2[c * NSP + s_] = s_ == W - 1 ? 0.0 : Bn;
            }
        __syncthreads();
        if (act) u = sA2[si];
    } else if (act) for (int s = NSEG - 1; s > seg; --s) u = sA2[s * TX + ixl] * u + sB2[s * TX + ixl];
#pragma unroll
    for (int i = SEG - 1; i >= 0; --i) { u = w[i] - Lv[i + 1] * u; w[i] = u; }
    const double ulo = zin * dinv_s - Lv[0] * w[0];             // u at the lower face of this segment
    double dot = 0.0;
    if (wr) {
        // values first, branch-free (cells outside the line contribute exact zeros), then the stores: a store inside a predicated
        // region that also uses loaded data would wait for the previous cell's store to be acknowledged (vmcnt counts stores too)
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
            const double lo = i == 0 ? ulo : w[i > 0 ? i - 1 : 0];
            const bool in = valid && c0 + i < n;                 // a select, not a branch (slab chains: xv of the cell behind the chain end is the edge cell)
            yo[i] = yo[i] + ma.Ta * (w[i] - lo); if (!ZW) dot += in ? xv[i] * yo[i] : 0.0;
            if (SR && !SF) { sqq += in ? yo[i] * yo[i] : 0.0; srq += in ? rq_[(SR && !SF) ? i : 0] * yo[i] : 0.0; }
            if (NB > 0) {
                const double v = ma.Gc[0] * x1[i] * ma.iM[0] * icv[i] - (ma.eL[0] * lo + ma.eR[0] * w[i]);
                y1o[i] = y1o[i] + ma.Ta * ma.Gc[0] * v; dot += in ? x1[i] * y1o[i] : 0.0;
            }
            if (NB > 1) {
                const double v = ma.Gc[1] * x2[NB > 1 ? i : 0] * ma.iM[1] * icv[i] - (ma.eL[1] * lo + ma.eR[1] * w[i]);
                y2o[NB > 1 ? i : 0] = y2o[NB > 1 ? i : 0] + ma.Ta * ma.Gc[1] * v; dot += in ? x2[NB > 1 ? i : 0] * y2o[NB > 1 ? i : 0] : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
            const int c = c0 + i;
            if (valid && c < n) {
                const long a = base + (long)c * sl;
                if (O32) {
                    sto(y, NF_A8(c), yo[i]); sto(ma.y[1], NF_A8(c), y1o[i]);
                    if (NB > 1) sto(ma.y[2], NF_A8(c), y2o[NB > 1 ? i : 0]);
                } else {
                    y[a] = yo[i];
                    if (NB > 0) ma.y[1][a] = y1o[i];
                    if (NB > 1) ma.y[2][a] = y2o[NB > 1 ? i : 0];
                }
            }
        }
    }
    if (ZW) dot = ma.Ta * zw;
    if (SLAB && sa.mode == 3) {                                 // emit J_z = -u on every face of the local line (+ the z bubbles for RT1+)
        if (valid) {
            const long nxy = sl;                                 // z lines: stride between planes = nx * ny
            int am = 0;                                          // sa.amode[bz] without a dynamic index (that would put the whole argument struct in scratch)
#pragma unroll
            for (int m = 0; m < 9; ++m) if ((unsigned)m == bz) am = sa.amode[m];
            const int fs = sa.if_lo ? 1 : 0, nfa = sa.nfa, kb = G.k, ni_ = sa.ni;
            double *const jz_ = sa.jz, *const jzb_ = sa.jzb;     // locals: a lambda that captured `sa` itself would force the argument struct into scratch
            const double eL0 = ma.eL[0], eL1 = ma.eL[1], eR0 = ma.eR[0], eR1 = ma.eR[1], Gc0 = ma.Gc[0], Gc1 = ma.Gc[1], iM0 = ma.iM[0], iM1 = ma.iM[1];
            auto face = [=](long fpl) -> double & { return jz_[(fpl * nxy + lineid) * nfa + am]; };
            // bubbles of a cell between face values (lo, hi): v_l = G_l x_{l+1} iM_l / c_e - (eL_l lo + eR_l hi); inactive ones (l >= NB) have no source
            auto bubbles = [=](long cell, double lo, double hi, double xb1, double xb2, double ice) {
                for (int l = 0; l < kb; ++l) {
                    const double xb = l == 0 ? xb1 : xb2, Gl = l == 0 ? Gc0 : Gc1, iMl = l == 0 ? iM0 : iM1, eLl = l == 0 ? eL0 : eL1, eRl = l == 0 ? eR0 : eR1;
                    const double tb = l < NB ? Gl * xb * iMl * ice : 0.0;
                    jzb_[cell * ni_ + l + kb * am] = -(tb - (eLl * lo + eRl * hi));
                }
            };
#pragma unroll
            for (int i = 0; i < SEG; ++i) if (c0 + i < n) {
                face(fs + c0 + i + 1) = -w[i];
                if (kb > 0) bubbles((long)(fs + c0 + i) * nxy + lineid, i == 0 ? ulo : w[i > 0 ? i - 1 : 0], w[i], NB > 0 ? x1[NB > 0 ? i : 0] : 0.0, NB > 1 ? x2[NB > 1 ? i : 0] : 0.0,
                                    NB > 0 ? icv[NB > 0 ? i : 0] : 0.0);
            }
            if (seg == 0) {
                face(fs) = -ulo;                                 // lower face of the first chain cell
                if (sa.if_lo) {
                    const double us = sa.ulo[lm];
                    face(0) = -us;                               // the separator below
                    if (kb > 0) bubbles(lineid, us, ulo, NB > 0 ? ma.x[NB > 0 ? 1 : 0][edge_lo] : 0.0, NB > 1 ? ma.x[NB > 1 ? 2 : 0][edge_lo] : 0.0,
                                        NB > 0 ? ma.D[edge_lo] / geom_factor(G, DIR, ix, (int)by, 0) : 0.0);
                }
            }
            if (sa.if_hi && c0 <= n - 1 && n - 1 < c0 + SEG) {     // the thread that owns the last chain cell: separator above + upper edge cell
                double ulast = 0.0;
#pragma unroll
                for (int i = 0; i < SEG; ++i) if (c0 + i == n - 1) ulast = w[i];
                const double us = sa.uhi[lm];
                face(fs + n + 1) = -us;
                if (kb > 0) bubbles((long)(fs + n) * nxy + lineid, ulast, us, NB > 0 ? ma.x[NB > 0 ? 1 : 0][edge_hi] : 0.0, NB > 1 ? ma.x[NB > 1 ? 2 : 0][edge_hi] : 0.0,
                                    NB > 0 ? ma.D[edge_hi] / geom_factor(G, DIR, ix, (int)by, fs + n) : 0.0);
            }
        }
        return 0.0;
    }
    if (SLAB) {
        // chain end values: u_first by the segment-0 thread, u_last by the thread owning chain cell n-1
        // (edge values are re-read here instead of being kept live across the two scans: x holds the updated p by now, and the
        // per-line coefficients come from L2 -- the kernel sits at the 128-VGPR limit of a 1024-thread block)
        if (valid && seg == 0 && sa.if_lo) {
            const double a_lo = sa.alo[lineid];
            if (sa.mode == 1) sa.clo[lm] = -(NB == 0 ? x[edge_lo] : xe_lo) - a_lo * ulo;
            else {
                const double u_lo = sa.fold ? (sa.rlo[lm] + sa.clo[lm]) * sa.sinv_lo[lineid] : sa.ulo[lm];
                const double xe = x[edge_lo]; const double yv = (y[edge_lo] + ((NB == 0 && sa.yadd) ? sa.yadd[edge_lo] : 0.0)) + ma.Ta * (ulo - u_lo); y[edge_lo] = yv; dot += xe * yv;
                if (SR && !SF) { sqq += yv * yv; srq += sa.sr_r[edge_lo] * yv; }
                if (NB > 0) {                                    // bubbles of the edge cell: faces (separator, first chain face)
                    const double ice = ma.D[edge_lo] / geom_factor(G, DIR, ix, (int)by, 0);
#pragma unroll
                    for (int l = 0; l < NB; ++l) {
                        const double xb = ma.x[l + 1][edge_lo];
                        const double v = ma.Gc[l] * xb * ma.iM[l] * ice - (ma.eL[l] * u_lo + ma.eR[l] * ulo);
                        const double yb = ma.y[l + 1][edge_lo] + ma.Ta * ma.Gc[l] * v;
                        ma.y[l + 1][edge_lo] = yb; dot += xb * yb;
                    }
                }
            }
        }
        if (valid && sa.if_hi && c0 <= n - 1 && n - 1 < c0 + SEG) {
            double ulast = 0.0;
#pragma unroll
            for (int i = 0; i < SEG; ++i) if (c0 + i == n - 1) ulast = w[i];
            const double a_hi = sa.ahi[lineid];
            if (sa.mode == 1) sa.chi[lm] = (NB == 0 ? x[edge_hi] : xe_hi) - a_hi * ulast;
            else {
                const double u_hi = sa.fold ? (sa.chi[lm] + sa.rhi[lm]) * sa.sinv_hi[lineid] : sa.uhi[lm];
                const double xe = x[edge_hi]; const double yv = (y[edge_hi] + ((NB == 0 && sa.yadd) ? sa.yadd[edge_hi] : 0.0)) + ma.Ta * (u_hi - ulast); y[edge_hi] = yv; dot += xe * yv;
                if (SR && !SF) { sqq += yv * yv; srq += sa.sr_r[edge_hi] * yv; }
                if (NB > 0) {                                    // faces (last chain face, separator)
                    const int fsz = sa.if_lo ? 1 : 0;
                    const double ice = ma.D[edge_hi] / geom_factor(G, DIR, ix, (int)by, fsz + n);
#pragma unroll
                    for (int l = 0; l < NB; ++l) {
                        const double xb = ma.x[l + 1][edge_hi];
                        const double v = ma.Gc[l] * xb * ma.iM[l] * ice - (ma.eL[l] * ulast + ma.eR[l] * u_hi);
                        const double yb = ma.y[l + 1][edge_hi] + ma.Ta * ma.Gc[l] * v;
                        ma.y[l + 1][edge_hi] = yb; dot += xb * yb;
                    }
                }
            }
        }
    }
    if (SR && SF) return srr;                                    // endpoint pass: this thread's share of |r|^2
    if (SR && extra) { extra[0] = sqq; extra[1] = srq; }
    return dot;
#undef NF_A8
}

// Slab variants keep r and x_sol of their cells in registers next to x, L, 1/d (loads first, see schur_s_tile): blocks of at most 512
// threads, so that the register budget is 256 per thread (the host picks TX accordingly)
template <int SEG, int DIR, bool SLAB, int NB, bool SF = false, bool NTS = false, bool ZW = false, bool SR = false>
__global__ __launch_bounds__(SLAB ? 512 : 1024, SLAB ? (NB > 0 ? 2 : (SF ? (SR ? NF_SR_Z1_WAVES : 3) : (SR ? NF_SR_Z2_WAVES : 4))) : 1) void k_schur_s(ModeArgs ma0, ModeTab mt, Geom G, const double *__restrict__ L, const double *__restrict__ DR,
                          const double *__restrict__ D0, int n, long sl, long outer_stride, int nx, int TX, int NSEG,
                          int last, double *__restrict__ partials, const CgScalars *__restrict__ cg, SlabArgs sa, CgFuse fz, CgLean lean)
{
    extern __shared__ double sm[];
    if (cg && cg->done) return;
    const ModeArgs ma = select_mode(ma0, mt, blockIdx.z, NB + 1);
    // fused CG on slabs: the endpoint pass (mode 1) is the first to read p in an iteration, so it carries the deferred
    // x_sol += alpha p, p = r + beta p (see CgFuse); every cell of the local line is owned by exactly one thread.
    // With lean.st it is also the consumer of the all-reduced |r|^2 (FIN_RR: beta, stop tests), like the x pass of an undivided mesh.
    bool fuse = SLAB && SF && NB == 0 && sa.mode == 1 && fz.p != nullptr && (SR ? sa.sr.index > 0 : (lean.st ? !lean.first : cg->its > 0));
    double f_beta = fuse && !lean.st && !SR ? cg->beta : 0.0;
    if (SLAB && !SR && lean.st && !lean.first) {
        double *sred_l = sm + 4 * TX * (NSEG + 1) + TX;
        if (lean_rr_step(lean, blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0, sred_l, &f_beta)) return;
    }
    double f_alpha = fuse && !SR ? cg->alpha : 0.0;
    // single-reduction CG: every block derives alpha, beta and the stop decision from the same five all-reduced doubles (see Cg1)
    if (SR && SF && fuse) { if (cg1_step(sa.sr, blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0, &f_alpha, &f_beta)) return; }
    // XCD-aware tile order (experiment, sa.xcd): hardware deals consecutive workgroups round-robin to the 8 XCDs; remap so that
    // each XCD works on one contiguous range of tiles
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (sa.xcd) {
        const unsigned nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (nblk % 8 == 0) { const unsigned nl = (lin % 8) * (nblk / 8) + lin / 8; bx = nl % gridDim.x; by = nl / gridDim.x; }
    }
    static_assert(!ZW || (!SLAB && NB == 0), "the z.w form of the dot product is for plain RT0-P0 lines");
    double extra[2] = { 0.0, 0.0 };
    const double dot = schur_s_tile<SEG, DIR, SLAB, NB, SF, NTS, NoMid, ZW, false, SR>(ma, G, L, DR, D0, n, sl, outer_stride, nx, TX, NSEG, bx, by, blockIdx.z, gridDim.y,
                                                            (int)threadIdx.x, true, sm, sa, fz, fuse, false, f_alpha, f_beta, !(!SLAB && NB == 0 && sa.noacc),
                                                            nullptr, NoMid(), SR ? extra : nullptr);
    if (SLAB && sa.mode == 3) return;
    double *sred = sm + 4 * TX * (NSEG + 1) + TX;
    const long pidx = ((long)blockIdx.z * gridDim.y + by) * gridDim.x + bx;
    if (SR && SF) {                                              // endpoint pass: |r|^2 of the residual this iteration starts with
        const double s = block_sum(dot, sred);
        if (threadIdx.x == 0) sa.sr_part[pidx] = s;
        return;
    }
    if (last && partials) {
        if (SR) {                                                // accumulation pass: q.q and r.q next to p.q (24 doubles of reduction scratch: blocks of at most 512 threads)
            double s0 = dot, s1 = extra[0], s2 = extra[1];
            block_sum3(s0, s1, s2, sred);
            if (threadIdx.x == 0) { partials[pidx] = s0; partials[pidx + sa.sr_stride] = s1; partials[pidx + 2 * sa.sr_stride] = s2; }
        } else {
            const double s = block_sum(dot, sred);
            if (threadIdx.x == 0) partials[pidx] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The endpoint pass of the z lines WITHOUT a line solve (single-reduction CG on RT0-P0 slab teams).  What the partition method needs from
// the first chain solve of an apply are two numbers per z line: c_lo = -x_edge - a_lo u_first and c_hi = x_edge - a_hi u_last, with
// u = T^-1 t(x).  Both are LINEAR functionals of the line's cell values, c_lo = sum_k W_lo[k] x_k, c_hi = sum_k W_hi[k] x_k, and their weights
// depend on the factored operator only: the host measures them once per BuildMatrices by sending the unit vector of every plane through the
// endpoint pass itself (team_endpoint_weights: nz launches per group; |W_lo[k]| decays like 0.268^k away from its interface).  The pass then is
// a streaming kernel -- the deferred CG update of every cell (r -= alpha q, x += alpha p, p = r + beta p: the same fma expressions as the scan
// kernel's) and two multiply-adds per cell -- with no scan, no segment summaries and one barrier; on the 2 M-cell slabs of a strong-scaling run,
// where the scan kernel is latency-bound, that is 41 -> 30 us per launch for the same bytes (72 B per cell: W_lo, W_hi replace L, 1/d).
// Thread = (x column, one of four z ranges); the four partial sums of a line are added in a fixed order.
__global__ __launch_bounds__(256) void k_endpoint_w(double *__restrict__ p, double *__restrict__ r, const double *__restrict__ q, double *__restrict__ xsol,
                                                    const double *__restrict__ Wlo, const double *__restrict__ Whi, double *__restrict__ clo,
                                                    double *__restrict__ chi, int nx, int ny, int nz, int if_lo, int if_hi, int kcut, Cg1 sr,
                                                    const CgScalars *__restrict__ cg, double *__restrict__ rr_part)
{
    __shared__ double s_lo[4][64], s_hi[4][64], sred[4];
    if (cg && cg->done) return;
    double alpha = 0.0, beta = 0.0;
    const bool upd = sr.index > 0;
    if (upd && cg1_step(sr, blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0, &alpha, &beta)) return;   // every block: the same five doubles, the same decision
    const int ixl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int ix = blockIdx.x * 64 + ixl;
    const bool valid = ix < nx;
    const long nxy = (long)nx * ny, line = (long)blockIdx.y * nx + ix;
    const int per = (nz + 3) >> 2, k0 = seg * per, k1 = k0 + per < nz ? k0 + per : nz;
    double a_lo = 0.0, a_hi = 0.0, rr = 0.0;
    if (valid) {
#pragma unroll 4
        for (int k = k0; k < k1; ++k) {
            const long e = (long)k * nxy + line;
            double pv = p[e];
            if (upd) {
                const double rv = fma(-alpha, q[e], r[e]);           // src/solvers.cpp:610
                xsol[e] = fma(alpha, pv, xsol[e]);                   // :609
                pv = fma(beta, pv, rv);                              // :630
                r[e] = rv; p[e] = pv; rr += rv * rv;
            } else rr += pv * pv;                                    // first iteration: p = r
            // |W_lo[k]| <= 0.268^k |W_lo[0]| whatever the cross sections (T is a sum of cell blocks a_c [[2,1],[1,2]]: Jacobi-scaled condition <= 3), so
            // beyond kcut = 40 planes from its interface a weight is below 1e-22 of the sum: thick slabs do not read it (wave-uniform branch)
            if (if_lo && k < kcut) a_lo = fma(Wlo[e], pv, a_lo);
            if (if_hi && k >= nz - kcut) a_hi = fma(Whi[e], pv, a_hi);
        }
    }
    s_lo[seg][ixl] = a_lo; s_hi[seg][ixl] = a_hi;
    __syncthreads();
    if (seg == 0 && valid) {
        if (if_lo) clo[line] = ((s_lo[0][ixl] + s_lo[1][ixl]) + s_lo[2][ixl]) + s_lo[3][ixl];
        if (if_hi) chi[line] = ((s_hi[0][ixl] + s_hi[1][ixl]) + s_hi[2][ixl]) + s_hi[3][ixl];
    }
    const double s = block_sum(rr, sred);
    if (threadIdx.x == 0) rr_part[(long)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// Long y / z lines (RT0-P0, plain lines: no slab interfaces).  k_schur_s holds a whole line in the registers of one block, so at
// 512 cells per line a block is only 16 columns wide (128-byte row pieces) and at 1024 cells 8 (64 bytes).  Here the line is cut
// into TWO chunks of CH = 8 NS cells and the block is twice as wide for the same number of threads:
//   forward sweep of chunk 0 (its face values w and its factors L are parked in LDS: 2 x 64 KB for 1024 threads), forward and
//   backward sweep of chunk 1 out of registers (exactly k_schur_s's body), backward sweep of chunk 0 out of LDS.
// A chunk boundary is a segment boundary like any other -- the same affine-map summaries, composed serially per thread; the z
// (u) value that crosses it is handed over through sC.  Every global array is read once and written once per pass.  The pass's
// share of x.y is formed in the forward sweeps as T_a sum z_f w_f (see schur_s_tile, ZW), so the backward half needs no x.
// LDS (doubles): sA, sB [NS TX] (forward and backward summaries in turn), sC, sL1 [TX], 16 of reduction scratch, pW, pL [CH TX], pZ [NS TX].
// Per cell the same expressions as k_schur_s (the value crossing the chunk boundary is the swept one, not a composed summary:
// the two kernels agree to rounding, not bitwise).
// Addresses are 32-bit BYTE offsets from the (wave-uniform) array bases -- the host takes this kernel only while one group's array is
// below 4 GiB.  One VGPR per cell address, shared by x, L, 1/d and y, instead of a 64-bit pair per array: that is what keeps the
// two-chunk body inside the 128 registers of a 1024-thread block (with 64-bit addresses it spilled 10-12 registers to scratch,
// +31 % bytes written and +6 % fetched per pass by the PMC counters, profiles/r03_b_pmc_512_chunked.json).
template <int DIR, bool NT>
__global__ __launch_bounds__(1024, 1) void k_schur_c(const double *__restrict__ x, double *__restrict__ y, double Ta,
                                                     const double *__restrict__ L, const double *__restrict__ DR, const double *__restrict__ D0,
                                                     int n, long sl, long outer_stride, int nx, int TX, int NS, int last,
                                                     double *__restrict__ partials, const CgScalars *__restrict__ cg, int xcd)
{
    extern __shared__ double sm[];
    if (cg && cg->done) return;
    constexpr int SEG = 8;
    const int T = TX * NS, CH = NS * SEG;
    double *sA = sm, *sB = sm + T, *sC = sm + 2 * T, *sL1 = sC + TX, *sred = sL1 + TX, *pW = sred + 16, *pL = pW + (long)CH * TX, *pZ = pL + (long)CH * TX;
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (xcd) {                                                   // XCD-contiguous tile order, as in k_schur_s
        const unsigned nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (nblk % 8 == 0) { const unsigned nl = (lin % 8) * (nblk / 8) + lin / 8; bx = nl % gridDim.x; by = nl / gridDim.x; }
    }
    const int tid = (int)threadIdx.x;
    const bool act = tid < T;
    const int ixl = act ? tid % TX : 0, seg = act ? tid / TX : 0, si = seg * TX + ixl;
    const int ix = (int)bx * TX + ixl;
    const bool valid = act && ix < nx;
    const unsigned slb = (unsigned)(sl * 8);                     // bytes between the cells of a line
    const unsigned ob = (unsigned)(((long)by * outer_stride + ix) * 8);   // byte offset of the line's first cell
    const long lineid = (long)by * nx + ix;
    const bool need_dot = last && partials;
    double xv[SEG + 1], Lv[SEG + 1], Rv[SEG], w[SEG], yo[SEG];
    double zin = 0.0, dinv_s = 0.0, zc = 0.0, dot = 0.0;
    // ---- forward sweeps, chunk 0 then chunk 1
    for (int ch = 0; ch < 2; ++ch) {
        const int c0 = ch * CH + seg * SEG;
        const unsigned o0 = ob + (unsigned)c0 * slb;
#pragma unroll
        for (int i = 0; i <= SEG; ++i) {                         // loads only (see schur_s_tile)
            const bool ok = valid && c0 + i < n;
            const unsigned a = o0 + (unsigned)i * slb;
            // the overlap cell (i == SEG) is the next segment's first: a plain load keeps that line for it.  Two statements, not a
            // ternary on i: before the loop is unrolled a ternary is one load in each arm of a branch, which the optimiser merges
            // into a single load WITHOUT the hint (that is what round 2's kernels ran: every load of x was a plain one)
            if (i < SEG) { xv[i] = ok ? ldo<NT>(x, a) : 0.0; Lv[i] = ok ? ldo<NT>(L, a) : 0.0; Rv[i] = ok ? ldo<NT>(DR, a) : 0.0; }
            else { xv[i] = ok ? ldo<false>(x, a) : 0.0; Lv[i] = ok ? ldo<false>(L, a) : 0.0; }
        }
        double ds = 0.0;
        if (valid && c0 < n) ds = c0 == 0 ? D0[lineid] : ldo<false>(DR, o0 - slb);
        if (ch) __syncthreads();                                 // chunk 0's summaries and carry have been consumed
        double P = 1.0, lz = 0.0;
#pragma unroll
        for (int i = 0; i < SEG; ++i) { const double ti = xv[i] - xv[i + 1]; lz = ti - Lv[i] * lz; P = -Lv[i] * P; }
        if (act) { sA[si] = P; sB[si] = lz; }
        if (act && ch == 0 && seg == 0) sC[ixl] = -xv[0];        // z entering the line (Dirichlet / natural end: no cell below)
        if (act && ch == 1 && seg == NS - 1) sC[ixl] = zc;       // z leaving chunk 0
        if (act && ch == 1 && seg == 0) sL1[ixl] = Lv[0];        // the factor just above chunk 0 (its last segment's overlap cell)
        __syncthreads();
        double z = sC[ixl];
        for (int s = 0; s < seg; ++s) z = sA[s * TX + ixl] * z + sB[s * TX + ixl];
        zin = z; dinv_s = ds;
        if (c0 == 0) dot = zin * (zin * ds);                     // the line's first face
#pragma unroll
        for (int i = 0; i < SEG; ++i) { const double ti = xv[i] - xv[i + 1]; z = ti - Lv[i] * z; w[i] = z * Rv[i]; dot += z * w[i]; }
        asm volatile("" : "+v"(dot));                            // summed HERE: left alone, the optimiser sinks the products to the kernel's end and keeps every z and w of both chunks alive (in scratch) until then
        zc = z;
        if (ch == 0 && act) {
#pragma unroll
            for (int i = 0; i < SEG; ++i) { pW[(seg * SEG + i) * TX + ixl] = w[i]; pL[(seg * SEG + i) * TX + ixl] = Lv[i]; }
            pZ[si] = zin;
        }
    }
    // ---- backward sweeps and output, chunk 1 (still in registers) then chunk 0 (from LDS)
    double ucar = 0.0;
    for (int ch = 1; ch >= 0; --ch) {
        const int c0 = ch * CH + seg * SEG;
        const unsigned o0 = ob + (unsigned)c0 * slb;
        if (ch == 0) {
            // (the index is made opaque: otherwise the optimiser forwards the values this thread parked itself, i.e. keeps all of chunk 0
            // alive across chunk 1 -- in scratch memory, since the registers are full: 100 B per lane of spills instead of LDS reads)
            int pbase = seg * SEG * TX + ixl;
            asm volatile("" : "+v"(pbase));
#pragma unroll
            for (int i = 0; i < SEG; ++i) { w[i] = pW[pbase + i * TX]; Lv[i] = pL[pbase + i * TX]; }
            Lv[SEG] = seg < NS - 1 ? pL[(seg + 1) * SEG * TX + ixl] : sL1[ixl];   // written before the barriers in between
            zin = pZ[si];
            dinv_s = 0.0;
            if (valid && c0 < n) dinv_s = c0 == 0 ? D0[lineid] : ldo<false>(DR, o0 - slb);
        }
#pragma unroll
        for (int i = 0; i < SEG; ++i) yo[i] = (valid && c0 + i < n) ? ldo<NT>(y, o0 + (unsigned)i * slb) : 0.0;
        __syncthreads();                                         // the summaries of the previous sweep have been consumed
        double Q = 1.0, lu = 0.0;
#pragma unroll
        for (int i = SEG - 1; i >= 0; --i) { lu = w[i] - Lv[i + 1] * lu; Q = -Lv[i + 1] * Q; }
        if (act) { sA[si] = Q; sB[si] = lu; }
        if (act && ch == 0 && seg == 0) sC[ixl] = ucar;          // u entering chunk 0 from above: first face value of chunk 1
        __syncthreads();
        double u = ch == 1 ? 0.0 : sC[ixl];
        for (int s = NS - 1; s > seg; --s) u = sA[s * TX + ixl] * u + sB[s * TX + ixl];
#pragma unroll
        for (int i = SEG - 1; i >= 0; --i) { u = w[i] - Lv[i + 1] * u; w[i] = u; }
        ucar = w[0];
        const double ulo = zin * dinv_s - Lv[0] * w[0];         // u at the lower face of this segment
#pragma unroll
        for (int i = 0; i < SEG; ++i) {                          // values first, branch-free, then the stores (see schur_s_tile)
            const double lo = i == 0 ? ulo : w[i > 0 ? i - 1 : 0];
            yo[i] = yo[i] + Ta * (w[i] - lo);
        }
#pragma unroll
        for (int i = 0; i < SEG; ++i)
            if (valid && c0 + i < n) *reinterpret_cast<double *>(reinterpret_cast<char *>(y) + (o0 + (unsigned)i * slb)) = yo[i];
    }
    if (need_dot) {
        const double s = block_sum(Ta * dot, sred);
        if (threadIdx.x == 0) partials[(long)by * gridDim.x + bx] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused-direction Schur apply for small and medium meshes (undivided, lean CG).  Below a few million cells a CG iteration is
// a chain of short dependent kernels and their number, not the bytes, sets the pace.  The x, y and z passes of one apply
// depend on each other only through the read-modify-write of y, so here they run side by side in ONE launch, each direction
// into an output vector of its own (qx = C p + X p, qy = Y p, qz = Z p; k_cg_rupdate3 forms (qx + qy) + qz, the same bits as the
// three accumulating passes).  The deferred x_sol += alpha p, p = r + beta p rides in the x blocks, which write the new p into
// the other buffer of a pair; the y / z blocks form r + beta p from the old one on the fly (CgFuse).  A CG iteration is then two
// dependent launches (k_apply3, k_cg_rupdate3) instead of four; every block of a launch first sums the partials of the
// launch before it (lean CG), so the scalars alpha, beta and the stop test need no launch of their own.
// Blocks [0, nbx) do the x lines (one wave-task per wave), [nbx, nbx+nby) one y tile each, the rest one z tile each.
struct Apply3 {
    int nbx, nby, nbz;                  // blocks per role
    int ntask_x, lpl_log2;              // x: wave-tasks per transverse mode
    int n[2], TX[2], NSEG[2], gx[2], gy[2]; long sl[2], ostride[2];   // y (index 0) and z (index 1) tiles
    long long *stamps;                  // diagnostic builds (-DNF_STAMPS): in-kernel cycle stamps of three blocks, else unused
};
template <int NCH, bool VEC, int NB, int SEG>                    // (streaming loads lose here at every size up to 4 Mi cells: 128^3 79 -> 91 us per CG iteration)
__global__ __launch_bounds__(512) void k_apply3(ModeArgs max0, ModeArgs may0, ModeArgs maz0, ModeTab mtx, ModeTab mty, ModeTab mtz, Geom G,
                                                const double *__restrict__ Lx, const double *__restrict__ DRx, const double *__restrict__ D0x,
                                                const double *__restrict__ Ly, const double *__restrict__ DRy, const double *__restrict__ D0y,
                                                const double *__restrict__ Lz, const double *__restrict__ DRz, const double *__restrict__ D0z,
                                                int nx, int ny, long nlines_x, Apply3 A, double *__restrict__ partials,
                                                const CgScalars *__restrict__ cg, CgFuse fz, CgLean lean)
{
    extern __shared__ double sm[];
    double *sred = sm + 4 * (int)blockDim.x + 320;               // behind the largest tile's scan arrays (<= 4 TX (NSEG+1) + TX doubles, TX NSEG <= blockDim, TX <= 64)
    long long *stamp = nullptr;
#ifdef NF_STAMPS
    if (A.stamps && threadIdx.x == 0) {
        if (blockIdx.x == 0) stamp = A.stamps; else if (blockIdx.x == (unsigned)A.nbx) stamp = A.stamps + 16; else if (blockIdx.x == (unsigned)(A.nbx + A.nby)) stamp = A.stamps + 32;
    }
    long long st_r0 = 0, st_0 = 0, st_1 = 0;
    if (stamp) { st_r0 = (long long)__builtin_amdgcn_s_memrealtime(); st_0 = (long long)__builtin_readcyclecounter(); }
#endif
    // scalars of the previous iteration: requested here, consumed inside `mid` -- the pass's own loads go out in between, so the
    // block pays ONE memory round trip for flags, scalars, partial sums and tile data instead of four in a row
    const int done0 = cg->done;
    const bool fuse = !lean.first;
    const double alpha0 = fuse ? cg->alpha : 0.0;
    LeanPre pre = { 0.0, 0.0, 0, 0 };
    if (fuse) pre = lean_preload(lean);
    bool stopped = false;
    auto mid = [&](double &fa, double &fb) -> bool {
        (void)fa;
        if (done0) { stopped = true; return true; }
#ifdef NF_STAMPS
        if (stamp) st_1 = (long long)__builtin_readcyclecounter();
#endif
        if (fuse) stopped = lean_rr_step(lean, pre, blockIdx.x == 0 && threadIdx.x == 0, sred, &fb);   // FIN_RR of the previous iteration: beta, stop tests
        NF_STAMP(stamp, 2);
        return stopped;
    };
    const unsigned b = blockIdx.x;
    double dot = 0.0;
    if (b < (unsigned)A.nbx) {
        const int nw = blockDim.x >> 6;
        const long gt = (long)b * nw + (threadIdx.x >> 6);      // global wave-task: mode-major
        const int mode = (int)(gt / A.ntask_x);
        const bool active = mode < mtx.n;
        const ModeArgs ma = select_mode(max0, mtx, active ? mode : 0, NB + 1);
        dot = schur_x_task<2, NCH, VEC, NB>(ma, G, Lx, DRx, D0x, nx, ny, nlines_x, A.lpl_log2, 1, gt % A.ntask_x, threadIdx.x & 63, active,
                                            fuse, alpha0, 0.0, fz, mid);
    } else {
        const int r = b < (unsigned)(A.nbx + A.nby) ? 0 : 1;
        const unsigned t = b - A.nbx - (r ? A.nby : 0);
        const unsigned bx = t % A.gx[r], by = (t / A.gx[r]) % A.gy[r], bz = t / (A.gx[r] * A.gy[r]);
        SlabArgs sa; sa.if_lo = sa.if_hi = sa.mode = sa.xcd = sa.wsmin = sa.fold = sa.noacc = 0; sa.yadd = nullptr; sa.alo = sa.ahi = sa.ulo = sa.uhi = sa.rlo = sa.rhi = sa.sinv_lo = sa.sinv_hi = nullptr; sa.clo = sa.chi = sa.jz = sa.jzb = nullptr; sa.nfa = 1; sa.ni = 0;
        if (r == 0) {
            const ModeArgs ma = select_mode(may0, mty, bz, NB + 1);
            dot = schur_s_tile<SEG, 1, false, NB>(ma, G, Ly, DRy, D0y, A.n[0], A.sl[0], A.ostride[0], nx, A.TX[0], A.NSEG[0], bx, by, bz, A.gy[0],
                                                  (int)threadIdx.x, true, sm, sa, fz, false, fuse, alpha0, 0.0, false, stamp, mid);
        } else {
            const ModeArgs ma = select_mode(maz0, mtz, bz, NB + 1);
            dot = schur_s_tile<SEG, 2, false, NB>(ma, G, Lz, DRz, D0z, A.n[1], A.sl[1], A.ostride[1], nx, A.TX[1], A.NSEG[1], bx, by, bz, A.gy[1],
                                                  (int)threadIdx.x, true, sm, sa, fz, false, fuse, alpha0, 0.0, false, stamp, mid);
        }
    }
    if (stopped) return;
    NF_STAMP(stamp, 8);
    const double sdot = block_sum(dot, sred);
    if (threadIdx.x == 0) partials[b] = sdot;
    NF_STAMP(stamp, 9);
#ifdef NF_STAMPS
    if (stamp) { stamp[13] = (long long)__builtin_amdgcn_s_memrealtime(); stamp[12] = st_r0; stamp[0] = st_0; stamp[1] = st_1; }
#endif
}
// r -= alpha ((qx + qy) + qz), |r|^2 partials; consumes the x.y partials of k_apply3 (FIN_PAP).  qy / qz may be null (1D / 2D).
__global__ __launch_bounds__(256) void k_cg_rupdate3(double *__restrict__ r, const double *__restrict__ qx, const double *__restrict__ qy,
                                                     const double *__restrict__ qz, long n, const CgScalars *__restrict__ cg,
                                                     double *__restrict__ partials, CgLean lean)
{
    __shared__ double sred[4];
    // flags, |r|^2 of this iteration and the first trip's vector entries are requested before the reduction of the p.q partials
    // (one memory round trip instead of three in a row; on the latency-bound meshes this kernel is one trip long)
    const int done0 = cg->done;
    const double rr_cur = lean.st->rr2[lean.par];
    const long i0 = blockIdx.x * 256L + threadIdx.x;
    double r0 = 0.0, q0 = 0.0;
    if (i0 < n) { r0 = r[i0]; q0 = qx[i0]; if (qy) q0 += qy[i0]; if (qz) q0 += qz[i0]; }
    if (done0) return;
    const double pq = strided_total(lean.partials, lean.count, sred);
    const bool brk = fabs(pq) < 1e-30;
    const double alpha = brk ? 0.0 : rr_cur / pq;
    if (blockIdx.x == 0 && threadIdx.x == 0) { lean.st->pAp = pq; lean.st->pend = 0; if (brk) lean.st->done = 1; else lean.st->alpha = alpha; }
    if (brk) return;
    double s = 0.0;
    if (i0 < n) { const double rn = r0 - alpha * q0; r[i0] = rn; s += rn * rn; }
    for (long i = i0 + gridDim.x * 256L; i < n; i += gridDim.x * 256L) {
        double q = qx[i];
        if (qy) q += qy[i];
        if (qz) q += qz[i];
        const double rn = r[i] - alpha * q;
        r[i] = rn; s += rn * rn;
    }
    s = block_sum(s, sred);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// A whole CG solve (src/solvers.cpp:577-636) of a mid-size undivided mesh (any order) in ONE launch, on the workgroups of ONE XCD.
// Between the one-workgroup resident kernel (a few thousand unknowns) and the meshes that fill the chip, a CG iteration is two
// dependent launches (k_apply3, k_cg_rupdate3) whose cost is the launch boundary itself: 7.2 + 4.3 us of kernels that mostly wait for
// their first loads (L2 was written back at the boundary) plus 3.7 us of gap, on IAEA-3D 38x38x19.  The two reductions of an
// iteration need two grid-wide synchronisations; inside one launch those are barriers on a counter, and a barrier among the 32 compute
// units that share one L2 costs 0.93 us (profiles/tools/xcd_barrier.hip, profiles/r03_w_xcd_barrier.txt: two barriers + a 4 KB
// hand-off per workgroup 1.86 us with L1-bypassing loads, 3.7 us with an agent acquire per barrier, 8.9 us with release + acquire).
//   * 8 W workgroups are launched; a workgroup reads its XCD from HW_REG_XCC_ID, registers, and leaves unless it sits on XCD `xcc`.
//     Placement is observed, not assumed: the P workgroups that registered share the work, whatever P is (round-robin dispatch
//     gives W).  All participants share one L2, so a store that has been acknowledged (s_waitcnt vmcnt(0)) is visible to a load
//     that bypasses the reader's L1 (non-temporal / sc1 loads: the tile functions' streaming flavour) -- no L2 write-back
//     (buffer_wbl2), no L1 invalidate.  Uniform reads of exchanged words are explicit sc1 loads (a plain one may become a scalar
//     load through the scalar cache, which nothing here invalidates).
//   * every spin is bounded.  Workgroups that do not assemble (P = 0, P > 64, a start that times out: the XCD is busy with somebody
//     else's kernel) end the launch with CgScalars::err = 3 before a single vector has been touched and the host runs the solve
//     through the launch path; a barrier that times out in the middle of a solve (err = 4) is an error.
//   * every workgroup derives alpha, beta and the stop tests itself from the same partial sums in the same order: control flow is
//     uniform across workgroups without flags, and no scalar lives in memory during the solve.
// Phase A of an iteration = what k_apply3 does (q_x = C p + X p with the deferred x_sol += alpha p, p' = r + beta p written to the
// other buffer of the pair; q_y = Y p, q_z = Z p from p formed on the fly; p.q partials), phase B = k_cg_rupdate3.  Inside a
// 768-thread workgroup (XCD_THREADS) the three roles run side by side on different wavefronts (x: one wave-task per wave; y / z: sub-tiles of
// TX NSEG threads packed into the role's waves; the x waves keep the two barriers of a tile company).
// Partial sums: one per workgroup, added by wave_sum's fixed tree over the workgroups -- not the order of the launch path, so the iterates differ from it in
// the last bits (like the resident kernel's do); every run gives the same bits.
struct ResidentOut { double keff; int n_outer, status, cg_total, pad; };   // status 0 ok, 2 diverged (non-finite k or dphi); k_keff_xcd: 3 not assembled, 4 barrier timeout
struct XcdState { unsigned arrived, nreg, count, timeout, go; };   // go: 0 undecided, 1 start (every workgroup has registered), 2 do not start
struct XcdArgs {
    ModeArgs ma[3]; Geom G;
    const double *L[3], *DR[3], *D0[3];
    int nx, ny, dim; long nlines_x, N;
    int ntask_x, lpl_log2;
    int n[2], TX[2], NSEG[2], gx[2], gy[2]; long sl[2], ostride[2];   // y (0) and z (1) tiles
    ModeTab mt[3]; int nmodes;                                        // transverse modes of every direction (orders with bubble moments; one mode for P0)
    double *pA, *pB, *r, *xsol; const double *q[3];
    CgScalars *cg; double *part;                                      // part: 4 rotating slots of 64 partials (xcd_total), then p.q | |r|^2 partials of the CG, then diagnostics: 512 doubles
    XcdState *st; HostPub *hp; unsigned long long seq; int xcc;
};
__device__ __forceinline__ unsigned xld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// bounded by wall time (s_memrealtime: 100 MHz, independent of the shader clock), not by an iteration count: 0.5 s -- only a lost
// participant gets there; a workgroup that was merely preempted for a few milliseconds does not fail the solve
constexpr unsigned long long XCD_SPIN_TICKS = 50000000ull;
__device__ __forceinline__ bool xcd_spin(const unsigned *p, unsigned target, unsigned *timeout)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0;; ++i) {
        if (xld(p) >= target) return true;
        if ((i & 1023) == 1023) {
            if (xld(timeout)) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > XCD_SPIN_TICKS) break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}
// all threads; every wave's stores have been acknowledged by L2 before thread 0 arrives
__device__ __forceinline__ bool xcd_barrier(XcdState *st, unsigned target, int *s_ok)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&st->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_ok = xcd_spin(&st->count, target, &st->timeout) ? 1 : 0;
    }
    __syncthreads();
    return *s_ok != 0;
}
constexpr int XCD_THREADS = 768;                                 // 12 wavefronts = 3 per SIMD: 170 VGPRs (at 1024 threads the 128-VGPR budget spilled into the loop: a reload from scratch is a trip to L2 there)
// what a workgroup knows once the participants have assembled: its index, their number, and how its wavefronts split among the roles
struct XcdCtx {
    int widx, P; bool ok;
    int R, wx, dx, role;                                          // rounds of phase A; x role: wavefronts per round, wave-tasks of this workgroup
    int sT, sC, sPer, sNt, sTX, sNSEG, sGx, sGy, sN, sSlot, sLtid; long sSl, sOst; bool isz; double *sSm;   // tile waves: their direction's parameters
    unsigned nbar; int nred;                                      // barriers passed; reductions made (partial slots rotate)
};
// registration: returns false for a workgroup that leaves (not on the chosen XCD).  *none: this workgroup was the last of the grid to
// start and nobody registered -- it must tell the host.
__device__ __forceinline__ bool xcd_assemble(XcdState *st, int xcc, int *s_i, XcdCtx &C, bool *none)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    const bool part = (int)(id & 0xf) == xcc;
    *none = false;
    // The decision to start is ONE word that every participant reads (with a bounded spin per workgroup, one could time out
    // while another had just seen everybody arrive, and the two disagreed): the last workgroup of the grid to arrive -- every registration
    // is in by then -- writes go = 1 (or 2: nobody or too many registered); a participant whose wait expires tries to write 2; whoever
    // writes first decides for all.
    if (threadIdx.x == 0) {
        s_i[0] = part ? (int)__hip_atomic_fetch_add(&st->nreg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        const unsigned before = __hip_atomic_fetch_add(&st->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1 == gridDim.x) {
            const unsigned n = xld(&st->nreg);
            *none = n == 0;
            unsigned expect = 0u;
            (void)__hip_atomic_compare_exchange_strong(&st->go, &expect, (n > 0 && n <= 64) ? 1u : 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!part) return false;
    if (threadIdx.x == 0) {
        if (!xcd_spin(&st->go, 1u, &st->timeout)) { unsigned expect = 0u; (void)__hip_atomic_compare_exchange_strong(&st->go, &expect, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        s_i[2] = xld(&st->go) == 1u ? 1 : 0; s_i[1] = (int)xld(&st->nreg);
    }
    __syncthreads();
    C.widx = s_i[0]; C.P = s_i[1]; C.ok = s_i[2] != 0 && C.P <= 64;
    C.nbar = 0; C.nred = 0;
    return true;
}
__device__ __forceinline__ void xcd_plan(const XcdArgs &A, XcdCtx &C, int tid, int wave, double *sm)
{
    const int P = C.P, dim = A.dim;
    const int Ty = A.TX[0] * A.NSEG[0], Tz = A.TX[1] * A.NSEG[1];
    const int nty = dim >= 2 ? A.gx[0] * A.gy[0] * A.nmodes : 0, ntz = dim == 3 ? A.gx[1] * A.gy[1] * A.nmodes : 0;
    const int dx = (A.ntask_x * A.nmodes + P - 1) / P, ty = (nty + P - 1) / P, tz = (ntz + P - 1) / P;
    int R = 1, wx = 0, wy = 0, wz = 0, cy = 0, cz = 0;
    for (;; ++R) {
        wx = (dx + R - 1) / R; cy = (ty + R - 1) / R; cz = (tz + R - 1) / R;
        wy = (cy * Ty + 63) >> 6; wz = (cz * Tz + 63) >> 6;
        if (wx + wy + wz <= XCD_THREADS / 64 || R >= 4096) break;
    }
    if (wx + wy + wz > XCD_THREADS / 64) C.ok = false;
    C.R = R; C.wx = wx; C.dx = dx;
    C.role = wave < wx ? 0 : wave < wx + wy ? 1 : wave < wx + wy + wz ? 2 : 3;
    // y and z tiles run the same code (one unknown per cell: the direction is nothing but strides): a tile wave picks its direction's parameters once
    const bool isz = C.role == 2;
    C.isz = isz;
    C.sT = isz ? Tz : Ty; C.sC = isz ? cz : cy; C.sPer = isz ? tz : ty; C.sNt = isz ? ntz : nty;
    C.sTX = isz ? A.TX[1] : A.TX[0]; C.sNSEG = isz ? A.NSEG[1] : A.NSEG[0]; C.sGx = isz ? A.gx[1] : A.gx[0]; C.sGy = isz ? A.gy[1] : A.gy[0]; C.sN = isz ? A.n[1] : A.n[0];
    C.sSl = isz ? A.sl[1] : A.sl[0]; C.sOst = isz ? A.ostride[1] : A.ostride[0];
    const int sLt = tid - (isz ? wx + wy : wx) * 64;
    C.sSlot = sLt / C.sT; C.sLtid = sLt - C.sSlot * C.sT;
    C.sSm = sm + (isz ? cy * (4 * Ty + A.TX[0]) : 0) + ((C.sSlot >= 0 && C.sSlot < C.sC) ? C.sSlot : 0) * (4 * C.sT + C.sTX);
}
// grid-wide sum, the same bits in every thread of every participant: one partial per workgroup (block_sum), barrier, lane i fetches
// workgroup i's partial with one L1-bypassing load (P dependent loads would be P trips to L2), then wave_sum's fixed tree.  The partial
// slots rotate (four of them), so a slot is rewritten three barriers after its readers have moved on.  NV values share one barrier.
template <int NV>
__device__ __forceinline__ bool xcd_total(double (&v)[NV], double *part, XcdState *st, XcdCtx &C, double *sred, int *s_ok)
{
    const int tid = threadIdx.x, lane = tid & 63;
    double *slot[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        slot[k] = part + 64 * (C.nred++ & 3);
        const double sd = block_sum(v[k], sred);
        if (tid == 0) slot[k][C.widx] = sd;
        if (k + 1 < NV) __syncthreads();
    }
    if (!xcd_barrier(st, (unsigned)C.P * ++C.nbar, s_ok)) return false;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(lane < C.P ? __builtin_nontemporal_load(slot[k] + lane) : 0.0);
    return true;
}
// The CG solve of one group (src/solvers.cpp:577-636) by the assembled workgroups: x_sol = 0, r = p = rhs in pA and |rhs|^2 = rr are
// in place.  offN / offP / offL*: this group's offsets into the factor arrays (cells), the C diagonal (unknowns) and the first pivots (lines).
// err: 0, 1 (non-finite sum), 4 (a barrier timed out).  On return x_sol holds the solution (the deferred last update applied).
// NB > 0 (bubble moments): the wave-tasks and tiles of all transverse modes are numbered mode-major (ModeTab), y and z tiles run their own
// instantiations (the bubbles' geometry factor depends on the direction) with SEG = 4 cells per thread.
template <int NCH, bool VEC, int NB, int SEG>
__device__ __forceinline__ void xcd_cg(const XcdArgs &A, XcdCtx &C, long offN, long offP, long offL0, long offL1, long offL2, double *xsol,
                                       double &rr, double tol_sq, int maxit, int &its, int &err, double *sred, int *s_ok, long long *xs)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dim = A.dim; const long N = A.N; const int P = C.P, widx = C.widx;
    SlabArgs sa0; sa0.if_lo = sa0.if_hi = sa0.mode = sa0.xcd = sa0.wsmin = sa0.fold = sa0.noacc = 0; sa0.yadd = nullptr; sa0.alo = sa0.ahi = sa0.ulo = sa0.uhi = sa0.rlo = sa0.rhi = sa0.sinv_lo = sa0.sinv_hi = nullptr; sa0.clo = sa0.chi = sa0.jz = sa0.jzb = nullptr; sa0.nfa = 1; sa0.ni = 0;
    const double *const sL = (C.isz ? A.L[2] : A.L[1]) + offN, *const sDR = (C.isz ? A.DR[2] : A.DR[1]) + offN, *const sD0 = C.isz ? A.D0[2] + offL2 : A.D0[1] + offL1;
    ModeArgs ms = A.ma[0]; ms.Ta = C.isz ? A.ma[2].Ta : A.ma[1].Ta; ms.y[0] = C.isz ? A.ma[2].y[0] : A.ma[1].y[0];
    ModeArgs mx = A.ma[0], my = A.ma[1], mz = A.ma[2];
#pragma unroll
    for (int q = 0; q <= NB; ++q) mx.Cd[q] = A.ma[0].Cd[q] + offP;
    mx.D = A.ma[0].D + offN; my.D = A.ma[1].D + offN; mz.D = A.ma[2].D + offN;
    double alpha = 0.0, beta = 0.0, rr_new = rr;
    int pend = 0;
    double *const ppq = A.part + 256, *const prr = A.part + 320;
#ifdef NF_XSTAMPS
    long long xt = (long long)__builtin_amdgcn_s_memrealtime();
#define NF_XS(k) do { const long long t_ = (long long)__builtin_amdgcn_s_memrealtime(); xs[k] += t_ - xt; xt = t_; } while (0)
#else
    (void)xs;
#define NF_XS(k) do { } while (0)
#endif
    its = 0; err = 0;
    if (maxit > 0)
        for (;;) {
            const bool fuse = its > 0;
            double *const pin = (its == 0 || (its & 1)) ? A.pA : A.pB, *const pout = pin == A.pA ? A.pB : A.pA;
            const CgFuse fz = { pin, A.r, xsol, pout };
            // ---- phase A: q_d = S_d p for every direction, p.q
            double dot = 0.0;
            for (int ro = 0; ro < C.R; ++ro) {
                if (C.role == 1 || C.role == 2) {
                    const int k = ro * C.sC + C.sSlot, t = widx * C.sPer + k;
                    const bool act = C.sSlot < C.sC && k < C.sPer && t < C.sNt;
                    const unsigned tt = act ? t : 0;
                    if (NB == 0) {
                        ms.x[0] = pin;
                        dot += schur_s_tile<8, 1, false, 0, true, false, NoMid, false, true>(ms, A.G, sL, sDR, sD0, C.sN, C.sSl, C.sOst, A.nx, C.sTX, C.sNSEG, tt % C.sGx, tt / C.sGx, 0, C.sGy,
                                                                                              C.sLtid, act, C.sSm, sa0, fz, false, fuse, alpha, beta, false);
                    } else {
                        const unsigned per = (unsigned)(C.sGx * C.sGy), bz = tt / per, rem = tt - bz * per;
                        if (C.role == 1) {
                            ModeArgs m = my;
#pragma unroll
                            for (int q = 0; q <= NB; ++q) { m.x[q] = pin + (A.ma[1].x[q] - A.pA); m.Cd[q] = nullptr; }
                            m = select_mode(m, A.mt[1], bz, NB + 1);
                            dot += schur_s_tile<SEG, 1, false, NB, true, false, NoMid, false, true>(m, A.G, sL, sDR, sD0, C.sN, C.sSl, C.sOst, A.nx, C.sTX, C.sNSEG, rem % C.sGx, rem / C.sGx, bz, C.sGy,
                                                                                                    C.sLtid, act, C.sSm, sa0, fz, false, fuse, alpha, beta, false);
                        } else {
                            ModeArgs m = mz;
#pragma unroll
                            for (int q = 0; q <= NB; ++q) { m.x[q] = pin + (A.ma[2].x[q] - A.pA); m.Cd[q] = nullptr; }
                            m = select_mode(m, A.mt[2], bz, NB + 1);
                            dot += schur_s_tile<SEG, 2, false, NB, true, false, NoMid, false, true>(m, A.G, sL, sDR, sD0, C.sN, C.sSl, C.sOst, A.nx, C.sTX, C.sNSEG, rem % C.sGx, rem / C.sGx, bz, C.sGy,
                                                                                                    C.sLtid, act, C.sSm, sa0, fz, false, fuse, alpha, beta, false);
                        }
                    }
                } else {
                    const int k = ro * C.wx + wave; const long gt = (long)widx * C.dx + k;
                    const bool act = C.role == 0 && k < C.dx && gt < (long)A.ntask_x * A.nmodes;
                    const int mode = act ? (int)(gt / A.ntask_x) : 0;
                    ModeArgs m = mx;
#pragma unroll
                    for (int q = 0; q <= NB; ++q) m.x[q] = pin + (A.ma[0].x[q] - A.pA);
                    if (NB > 0) m = select_mode(m, A.mt[0], mode, NB + 1);
                    dot += schur_x_task<2, NCH, VEC, NB, NoMid, true, true>(m, A.G, A.L[0] + offN, A.DR[0] + offN, A.D0[0] + offL0, A.nx, A.ny, A.nlines_x, A.lpl_log2, 1,
                                                                           act ? gt - (long)mode * A.ntask_x : 0, lane, act, fuse, alpha, beta, fz);
                    __syncthreads(); __syncthreads();           // the two barriers inside a y / z tile
                }
                __syncthreads();                                 // the tiles' scan arrays are reused by the next round
            }
            NF_XS(0);
            { const double sd = block_sum(dot, sred); if (tid == 0) ppq[widx] = sd; }
            NF_XS(1);
            if (!xcd_barrier(A.st, (unsigned)P * ++C.nbar, s_ok)) { err = 4; break; }
            NF_XS(2);
            const double pq = wave_sum(lane < P ? __builtin_nontemporal_load(ppq + lane) : 0.0);     // src/solvers.cpp:602-606
            NF_XS(3);
            pend = 0;
            if ((pq - pq) != 0.0) { err = 1; rr = pq; break; }
            if (fabs(pq) < 1e-30) break;
            alpha = rr / pq;
            // ---- phase B: r -= alpha ((q_x + q_y) + q_z), |r|^2
            // (all loads of an element go out before the first use: a load inside a predicated region is waited for there, one trip
            // to L2 per operand instead of one per element; for dim < 3 the missing directions point at q_x and are not added)
            double s = 0.0;
            for (long i = (long)widx * XCD_THREADS + tid; i < N; i += (long)P * XCD_THREADS) {
                const double q0 = __builtin_nontemporal_load(A.q[0] + i), q1 = __builtin_nontemporal_load(A.q[1] + i), q2 = __builtin_nontemporal_load(A.q[2] + i);
                const double r0 = __builtin_nontemporal_load(A.r + i);
                double q = q0;
                if (dim >= 2) q += q1;
                if (dim == 3) q += q2;
                const double rn = r0 - alpha * q;
                A.r[i] = rn; s += rn * rn;
            }
            { const double sd = block_sum(s, sred); if (tid == 0) prr[widx] = sd; }
            NF_XS(4);
            if (!xcd_barrier(A.st, (unsigned)P * ++C.nbar, s_ok)) { err = 4; break; }
            NF_XS(5);
            rr_new = wave_sum(lane < P ? __builtin_nontemporal_load(prr + lane) : 0.0);               // :613-631
            if ((rr_new - rr_new) != 0.0) { err = 1; rr = rr_new; break; }
            ++its; pend = 1;
            NF_XS(6);
            if (rr_new < tol_sq) { rr = rr_new; break; }
            beta = rr_new / rr; rr = rr_new;
            if (its >= maxit) break;
        }
    // ---- the last iteration's x_sol += alpha p (:609); its direction: see the buffer pair in cg_solve
    if (pend && !err) {
        const double *plast = (its >= 2 && ((its - 1) & 1)) ? A.pB : A.pA;
        for (long i = (long)widx * XCD_THREADS + tid; i < N; i += (long)P * XCD_THREADS)
            xsol[i] = fma(alpha, __builtin_nontemporal_load(plast + i), __builtin_nontemporal_load(xsol + i));
    }
}
template <int NCH, bool VEC, int NB = 0, int SEG = 8>
__global__ __launch_bounds__(XCD_THREADS) void k_cg_xcd(XcdArgs A)
{
    extern __shared__ double sm[];
    __shared__ int s_i[4];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *const sred = sm + 5 * 1024 + 64;
    XcdCtx C; bool none;
    const bool part = xcd_assemble(A.st, A.xcc, s_i, C, &none);
    if (none) {                                                   // nobody sits on the chosen XCD: say so
        A.cg->err = 3; A.cg->done = 1;
        if (A.hp) { A.hp->cg = *A.cg; __threadfence_system(); __hip_atomic_store(&A.hp->seq, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
    if (!part) return;
    xcd_plan(A, C, tid, wave, sm);
    // ---- scalars as k_finalize(FIN_RHS) left them (a launch earlier)
    double rr = A.cg->rr;
    const double tol_sq = A.cg->tol_sq; const int maxit = A.cg->maxit;
    int its = 0, err = C.ok ? 0 : 3;
    long long xs[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (C.ok && !A.cg->done) xcd_cg<NCH, VEC, NB, SEG>(A, C, 0, 0, 0, 0, 0, A.xsol, rr, tol_sq, maxit, its, err, sred, &s_i[3], xs);
#ifdef NF_XSTAMPS
    if (C.widx == 0 && tid == 0) { for (int k = 0; k < 7; ++k) A.part[384 + k] += (double)xs[k]; A.part[391] += its; A.part[392] += 1; A.part[393] = C.P; A.part[394] = C.R; A.part[395] = C.wx; }
#endif
    if (C.widx == 0 && tid == 0) {
        CgScalars *cg = A.cg;
        if (!cg->done || err) { cg->rr = rr; cg->rr_new = rr; cg->its = its; cg->pend = 0; cg->done = 1; cg->err = err; }
        if (A.hp) { A.hp->cg = *cg; __threadfence_system(); __hip_atomic_store(&A.hp->seq, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

// The whole SolveKeff (src/NeutFEM.cpp:1627-1815: fission source, Gauss-Seidel group sweep with CG, k update, normalisation, Chebyshev,
// stop tests, history) of the same meshes in ONE launch on the workgroups of one XCD: what k_resident_keff does in one workgroup for
// meshes that fit its LDS, with grid barriers in the place of workgroup barriers.  Beside the CG iterations themselves this removes
// what surrounds them on the host-driven path: per group solve five stream operations and a host check, per outer iteration four
// launches and a host check (IAEA-3D 38x38x19: ~6 ms of 22).  Elementwise loops are grid-strided; every vector is read with
// L1-bypassing loads, and a barrier separates every phase that writes a vector from the phases that read it under another ownership.
// Every workgroup carries the scalars (k, Chebyshev state, stop tests) itself, from the same sums: uniform control flow, no flags.
struct XcdOuter {
    int ng; long NP, N;
    const double *Mf, *Chi; const double *const *Ms;
    double *phi, *raw, *p0, *p1, *tf;
    long nl[3];
    double keff0, tol_keff, tol_flux, cg_tol; int cg_max, max_outer;
    double ca1, a3[16], cb[16];
    double *hist; int *hist_cg; ResidentOut *out;                // status 3: the workgroups did not assemble (nothing touched), 4: a barrier timed out
};
template <int NCH, bool VEC, int NB = 0, int SEG = 8>
__global__ __launch_bounds__(XCD_THREADS) void k_keff_xcd(XcdArgs A, XcdOuter O)
{
    extern __shared__ double sm[];
    __shared__ int s_i[4];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *const sred = sm + 5 * 1024 + 64;
    XcdCtx C; bool none;
    const bool part = xcd_assemble(A.st, A.xcc, s_i, C, &none);
    if (none) { O.out->keff = O.keff0; O.out->n_outer = 0; O.out->status = 3; O.out->cg_total = 0; }
    if (!part) return;
    xcd_plan(A, C, tid, wave, sm);
    if (!C.ok) { if (C.widx == 0 && tid == 0) { O.out->keff = O.keff0; O.out->n_outer = 0; O.out->status = 3; O.out->cg_total = 0; } return; }
    const int ng = O.ng; const long NP = O.NP, N = O.N, NT = NP * ng;
    const long g0 = (long)C.widx * XCD_THREADS + tid, gs = (long)C.P * XCD_THREADS;
    double keff = O.keff0;
    int cheb_it = 0, n_outer = 0, status = 0, cg_total = 0;
    double *pa = O.p0, *pb = O.p1;
    long long xs[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int it = 0; it < O.max_outer && !status; ++it) {
        // total_fiss and prod_old (:1700-1707)
        double v1[1] = { 0.0 };
        for (long i = g0; i < NP; i += gs) {
            double v = 0.0;
            for (int g = 0; g < ng; ++g) v += O.Mf[g * NP + i] * __builtin_nontemporal_load(O.phi + g * NP + i);
            O.tf[i] = v; v1[0] += v;
        }
        if (!xcd_total<1>(v1, A.part, A.st, C, sred, &s_i[3])) { status = 4; break; }
        const double prod_old = v1[0];
        const double inv_k = 1.0 / keff;
        for (int g = 0; g < ng && !status; ++g) {
            double *const graw = O.raw + (long)g * NP;
            // rhs = chi_g tf / k + scatter (Gauss-Seidel) (:1716-1726); CG start x = 0, r = p = rhs (src/solvers.cpp:583-592)
            v1[0] = 0.0;
            for (long i = g0; i < NP; i += gs) {
                double v;
                if (NB == 0) v = inv_k * (O.Chi[g * N + i] * __builtin_nontemporal_load(O.tf + i));
                else { const double cv = O.Chi[g * N + i % N] * inv_k; v = fabs(cv) < 1e-14 ? 0.0 : cv * __builtin_nontemporal_load(O.tf + i); }   // BuildFissionRHS, P >= 1 (:1539-1561)
                for (int gp = 0; gp < ng; ++gp) {
                    const double *M = O.Ms[g * ng + gp];
                    if (gp == g || !M) continue;
                    v += M[i] * __builtin_nontemporal_load((gp < g ? O.raw : O.phi) + gp * NP + i);
                }
                graw[i] = 0.0; A.r[i] = v; A.pA[i] = v; v1[0] += v * v;
            }
            if (!xcd_total<1>(v1, A.part, A.st, C, sred, &s_i[3])) { status = 4; break; }
            double rr = v1[0];
            const double rhs_norm = sqrt(rr), tol_sq = O.cg_tol * O.cg_tol * rhs_norm * rhs_norm;
            int its = 0, err = 0;
            xcd_cg<NCH, VEC, NB, SEG>(A, C, g * N, g * NP, g * O.nl[0], g * O.nl[1], g * O.nl[2], graw, rr, tol_sq, O.cg_max, its, err, sred, &s_i[3], xs);
            if (err == 4) { status = 4; break; }
            if (err == 1) status = 2;                             // a non-finite sum: the outer iteration below reports the divergence
            if (tid == 0 && C.widx == 0) O.hist_cg[it * ng + g] = its;
            cg_total += its;
            if (!xcd_barrier(A.st, (unsigned)C.P * ++C.nbar, &s_i[3])) { status = 4; break; }   // this group's solution is complete before anybody reads it
        }
        if (status == 4) break;
        // prod_new, norms (:1766-1779)
        double v3[3] = { 0.0, 0.0, 0.0 };
        for (long i = g0; i < NT; i += gs) { const double v = __builtin_nontemporal_load(O.raw + i), d = v - __builtin_nontemporal_load(O.phi + i); v3[0] += O.Mf[i] * v; v3[1] += v * v; v3[2] += d * d; }
        if (!xcd_total<3>(v3, A.part, A.st, C, sred, &s_i[3])) { status = 4; break; }
        const double prod_new = v3[0], nsq = v3[1], dsq = v3[2];
        const double keff_new = keff * (prod_new / prod_old);
        const double dk = fabs(keff_new - keff);
        if (it >= 1) keff = keff_new;                               // :1774
        const double dphi = sqrt(dsq / nsq), norm = sqrt(nsq);
        if (tid == 0 && C.widx == 0) { O.hist[it] = keff; O.hist[O.max_outer + it] = dk; O.hist[2 * O.max_outer + it] = dphi; }
        n_outer = it + 1;
        if (!isfinite(keff_new) || !isfinite(dphi)) { status = 2; break; }
        status = 0;
        // normalise + Chebyshev (:1780-1788, src/solvers.cpp:720-756)
        int mode = 0; double ca = 0.0, cb = 0.0;
        if (it >= 2) {
            if (cheb_it == 15) cheb_it = 0;
            if (cheb_it == 0) mode = 1;
            else if (cheb_it == 1) { mode = 2; ca = O.ca1; }
            else { mode = 3; ca = O.a3[cheb_it]; cb = O.cb[cheb_it]; }
            ++cheb_it;
        }
        const bool do_norm = norm > 1e-14;
        for (long i = g0; i < NT; i += gs) {
            double v = __builtin_nontemporal_load(O.raw + i);
            if (do_norm) v /= norm;
            if (mode == 1) pa[i] = v;
            else if (mode == 2) { const double a = pa[i]; v = a + ca * (v - a); pb[i] = v; }
            else if (mode == 3) { const double a = pa[i], b = pb[i]; v = b + ca * (v - b) + cb * (b - a); pa[i] = v; }
            O.phi[i] = v;
        }
        if (mode == 3) { double *t = pa; pa = pb; pb = t; }
        if (!xcd_barrier(A.st, (unsigned)C.P * ++C.nbar, &s_i[3])) { status = 4; break; }   // the new phi before the next fission source
        if (dk < O.tol_keff && dphi < O.tol_flux) break;            // :1799-1802
    }
    if (tid == 0 && C.widx == 0) { O.out->keff = keff; O.out->n_outer = n_outer; O.out->status = status; O.out->cg_total = cg_total; }
}

// ---------------------------------------------------------------------------------------------
// CG vector kernels (src/solvers.cpp:577-631).  Fixed grids, grid-stride loops.
__global__ __launch_bounds__(256) void k_cg_init(const double *__restrict__ rhs, double *__restrict__ x,
                                                 double *__restrict__ r, double *__restrict__ p, long n,
                                                 double *__restrict__ partials)
{
    __shared__ double sred[4];
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const double b = rhs[i];
        x[i] = 0.0; r[i] = b; p[i] = b; s += b * b;
    }
    s = block_sum(s, sred);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cg_update(double *__restrict__ x, double *__restrict__ r,
                                                   const double *__restrict__ p, const double *__restrict__ q, long n,
                                                   const CgScalars *__restrict__ cg, double *__restrict__ partials)
{
    __shared__ double sred[4];
    if (cg->done) return;
    const double alpha = cg->alpha;
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        x[i] = fma(alpha, p[i], x[i]);
        const double rn = r[i] - alpha * q[i];
        r[i] = rn; s += rn * rn;
    }
    s = block_sum(s, sred);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// fused CG: r -= alpha q and |r|^2 only (x_sol and p are updated by the next x pass)
__global__ __launch_bounds__(256) void k_cg_rupdate(double *__restrict__ r, const double *__restrict__ q, long n,
                                                    const CgScalars *__restrict__ cg, double *__restrict__ partials, CgLean lean)
{
    __shared__ double sred[4];
    if (cg->done) return;
    double alpha;
    if (lean.st) {                                              // lean CG: this kernel consumes the p.q partials (FIN_PAP)
        const double pq = lean_total(lean, sred);
        const int e = lean_err(lean, pq);
        if (e) { if (blockIdx.x == 0 && threadIdx.x == 0) { lean.st->pAp = pq; lean.st->err = e; lean.st->done = 1; } return; }
        const bool brk = fabs(pq) < 1e-30;
        alpha = brk ? 0.0 : lean.st->rr2[lean.par] / pq;
        if (blockIdx.x == 0 && threadIdx.x == 0) { lean.st->pAp = pq; lean.st->pend = 0; if (brk) lean.st->done = 1; else lean.st->alpha = alpha; }
        if (brk) return;
    } else alpha = cg->alpha;
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const double rn = r[i] - alpha * q[i];                   // (streaming loads change nothing here: 102.3 vs 102.9 us of non-pass time per iteration)
        r[i] = rn; s += rn * rn;
    }
    s = block_sum(s, sred);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// fused CG: the x_sol update of the final iteration
__global__ void k_cg_flush(double *__restrict__ x, const double *__restrict__ p, long n, const CgScalars *__restrict__ cg)
{
    if (!cg->pend) return;
    const double alpha = cg->alpha;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) x[i] = fma(alpha, p[i], x[i]);
}
__global__ __launch_bounds__(256) void k_cg_pupdate(double *__restrict__ p, const double *__restrict__ r, long n,
                                                    const CgScalars *__restrict__ cg)
{
    if (cg->done) return;
    const double beta = cg->beta;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) p[i] = fma(beta, p[i], r[i]);
}

// ---------------------------------------------------------------------------------------------
// power iteration kernels (src/NeutFEM.cpp:1694-1788)
// total_fiss = sum_g M_fiss[g] phi_g ; partial sum of entries (prod_old)        (:1700-1707)
// With w != NULL (adjoint, :1930-1936) the partial sum is sum_{e < nw} w[e] * tf[e] (DOF 0 of every cell = first N entries).
__global__ __launch_bounds__(256) void k_fission(const double *__restrict__ Mf, const double *__restrict__ phi, int ng,
                                                 long n, double *__restrict__ tf, double *__restrict__ partials,
                                                 const double *__restrict__ w, long nw)
{
    __shared__ double sred[4];
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        double v = 0.0;
        for (int g = 0; g < ng; ++g) v += Mf[g * n + i] * phi[g * n + i];
        tf[i] = v;
        if (!w) s += v; else if (i < nw) s += w[i] * v;
    }
    s = block_sum(s, sred);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
struct ScatterArgs { const double *M[64]; int ng; };
// rhs = chi_g * total_fiss / k + sum_{gp != g} M_scatter[g<-gp] phi_gp  (Gauss-Seidel: gp < g from the new
// iterate, gp > g from the old one) (:1716-1726).  If sinv != NULL the diagonal solve phi = S_inv * rhs is
// fused (:607-613) and written to out, otherwise out = rhs.
// n = DOFs per group (SoA [p][e]), ncell = cells: chi is per cell; for P>=1 elements with |chi/k| < 1e-14 are skipped (:1551).
// With r0 != nullptr the CG start of src/solvers.cpp:583-592 rides along (x = 0, r = p = rhs, block partials of |rhs|^2: what k_cg_init
// does), one launch less per group solve.
__global__ __launch_bounds__(256) void k_group_rhs(ScatterArgs sa, int g, const double *__restrict__ chi,
                                                   const double *__restrict__ tf, double inv_k,
                                                   const double *__restrict__ phi_new, const double *__restrict__ phi_old,
                                                   const double *__restrict__ sinv, double *__restrict__ out, long n, long ncell,
                                                   double *__restrict__ x0 = nullptr, double *__restrict__ r0 = nullptr, double *__restrict__ p0 = nullptr,
                                                   double *__restrict__ partials = nullptr)
{
    __shared__ double sred[4];
    double s2 = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        double v;
        if (n == ncell) v = inv_k * (chi[i] * tf[i]);
        else { const double cv = chi[i % ncell] * inv_k; v = fabs(cv) < 1e-14 ? 0.0 : cv * tf[i]; }
        for (int gp = 0; gp < sa.ng; ++gp) {
            if (gp == g || !sa.M[gp]) continue;
            const double *ph = gp < g ? phi_new : phi_old;
            v += sa.M[gp][i] * ph[gp * n + i];
        }
        out[i] = sinv ? sinv[i] * v : v;
        if (r0) { x0[i] = 0.0; r0[i] = v; p0[i] = v; s2 += v * v; }
    }
    if (r0) {
        s2 = block_sum(s2, sred);
        if (threadIdx.x == 0) partials[blockIdx.x] = s2;
    }
}
// partial sums of prod_new, ||phi||^2, ||phi - phi_old||^2 over all groups       (:1766-1779)
__global__ __launch_bounds__(256) void k_outer_reduce(const double *__restrict__ Mf, const double *__restrict__ phi,
                                                      const double *__restrict__ old, long ntot,
                                                      double *__restrict__ partials, long stride)
{
    __shared__ double sred[4];
    double sp = 0.0, sn = 0.0, sd = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < ntot; i += gridDim.x * 256L) {
        const double v = phi[i], d = v - old[i];
        sp += Mf[i] * v; sn += v * v; sd += d * d;
    }
    sp = block_sum(sp, sred); sn = block_sum(sn, sred); sd = block_sum(sd, sred);
    if (threadIdx.x == 0) { partials[blockIdx.x] = sp; partials[stride + blockIdx.x] = sn; partials[2 * stride + blockIdx.x] = sd; }
}
// phi /= norm, then ChebyshevAccel::operator() (src/solvers.cpp:720-756).  mode 0: no acceleration,
// 1: store phi0, 2: phi1 = phi0 + a1 (phi - phi0), 3: three-term.  cur <- result.
__global__ __launch_bounds__(256) void k_normalize_cheb(const double *__restrict__ raw, double *__restrict__ cur,
                                                        double *__restrict__ p0, double *__restrict__ p1, long ntot,
                                                        double norm, int do_norm, int mode, double ca, double cb)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < ntot; i += gridDim.x * 256L) {
        double v = raw[i];
        if (do_norm) v /= norm;
        if (mode == 1) p0[i] = v;
        else if (mode == 2) { const double a = p0[i]; v = a + ca * (v - a); p1[i] = v; }
        else if (mode == 3) {
            const double a = p0[i], b = p1[i];
            v = b + ca * (v - b) + cb * (b - a);
            p0[i] = v;                       // caller swaps p0/p1: p0 <- old p1, p1 <- new
        }
        cur[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Diagonal-Schur power iteration with the whole outer loop on the device (undivided RT0-P0 mesh, no CMFD): an outer
// iteration is ng pointwise group solves, one scalar kernel and one normalise kernel, all a few microseconds on the
// benchmark meshes, so a host round trip per outer would dominate.  The scalar logic of src/NeutFEM.cpp:1766-1802 and of
// ChebyshevAccel (src/solvers.cpp:720-756) lives in k_outer_logic; the host launches batches of outers and reads the
// state once per batch.  Kernels of outers queued behind the last one exit on `done`.
struct OuterState {
    double keff, norm, a, b;
    double tol_keff, tol_flux;
    double ca1, a3[16], cb[16];     // Chebyshev coefficients: a_1, (4/sigma) a_n and b_n for n >= 2 (host-computed, same bits)
    int mode, do_norm, it, cheb_it, swap, swap_next, max_outer;
    int done;                       // 0 running, 1 finished (converged or max_outer), 2 diverged
    int done_next;                  // the current outer is the last one: its normalise kernel still runs (:1780-1802)
};
// group g: rhs = chi_g tf / k + scatter (Gauss-Seidel), raw_g = S_inv rhs (:607-613), and this group's share of
// prod_new = sum M_f raw, |raw|^2, |raw - phi_old|^2 (:1766-1779) accumulated block-wise over the groups (g = 0 stores)
__global__ __launch_bounds__(256) void k_diag_group(ScatterArgs sa, int g, const double *__restrict__ chi, const double *__restrict__ tf,
                                                    const double *__restrict__ raw_all, const double *__restrict__ phi_old,
                                                    const double *__restrict__ sinv, const double *__restrict__ Mf, double *__restrict__ out,
                                                    long n, const OuterState *__restrict__ st, double *__restrict__ partials, long stride)
{
    __shared__ double sred[4];
    if (st->done || st->done_next) return;
    const double inv_k = 1.0 / st->keff;
    double sp = 0.0, sn = 0.0, sd = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        double v = inv_k * (chi[i] * tf[i]);
        for (int gp = 0; gp < sa.ng; ++gp) {
            if (gp == g || !sa.M[gp]) continue;
            const double *ph = gp < g ? raw_all : phi_old;
            v += sa.M[gp][i] * ph[gp * n + i];
        }
        const double r = sinv[i] * v, d = r - phi_old[g * n + i];
        out[i] = r;
        sp += Mf[i] * r; sn += r * r; sd += d * d;
    }
    sp = block_sum(sp, sred); sn = block_sum(sn, sred); sd = block_sum(sd, sred);
    if (threadIdx.x == 0) {
        double *p1 = partials + stride + blockIdx.x, *p2 = partials + 2 * stride + blockIdx.x, *p3 = partials + 3 * stride + blockIdx.x;
        if (g == 0) { *p1 = sp; *p2 = sn; *p3 = sd; } else { *p1 += sp; *p2 += sn; *p3 += sd; }
    }
}
// one block: sums of the four partial rows (row 0: prod_old from k_fission / k_normalize_fission), then the scalar logic
__global__ __launch_bounds__(256) void k_outer_logic(const double *__restrict__ partials, int cnt0, int cnt, long stride,
                                                     OuterState *__restrict__ st, double *__restrict__ hist_k,
                                                     double *__restrict__ hist_dk, double *__restrict__ hist_dphi)
{
    __shared__ double sred[4];
    if (st->done) return;
    if (st->done_next) { if (threadIdx.x == 0) st->done = 1; return; }
    double tot[4];
    for (int q = 0; q < 4; ++q) {
        double s = 0.0;
        const int c = q == 0 ? cnt0 : cnt;
        for (int i = threadIdx.x; i < c; i += 256) s += partials[q * stride + i];
        tot[q] = block_sum(s, sred);
    }
    if (threadIdx.x != 0) return;
    const double prod_old = tot[0], prod_new = tot[1], nsq = tot[2], dsq = tot[3];
    const int it = st->it;
    double keff = st->keff;
    const double keff_new = keff * (prod_new / prod_old);
    const double dk = fabs(keff_new - keff);
    if (it >= 1) keff = keff_new;                               // :1774
    const double dphi = sqrt(dsq / nsq), norm = sqrt(nsq);
    st->keff = keff;
    hist_k[it] = keff; hist_dk[it] = dk; hist_dphi[it] = dphi;
    if (!isfinite(keff_new) || !isfinite(dphi)) { st->it = it + 1; st->done = 2; return; }
    int mode = 0; double a = 0.0, b = 0.0;
    int cheb_it = st->cheb_it;
    if (it >= 2) {                                              // src/solvers.cpp:720-756
        if (cheb_it == 15) cheb_it = 0;
        if (cheb_it == 0) mode = 1;
        else if (cheb_it == 1) { mode = 2; a = st->ca1; }
        else { mode = 3; a = st->a3[cheb_it]; b = st->cb[cheb_it]; }
        ++cheb_it;
    }
    st->cheb_it = cheb_it;
    st->swap = st->swap_next;                                   // orientation of (p0, p1) for this outer's normalise kernel
    st->swap_next = st->swap ^ (mode == 3 ? 1 : 0);
    st->mode = mode; st->a = a; st->b = b; st->norm = norm; st->do_norm = norm > 1e-14 ? 1 : 0;
    st->it = it + 1;
    if ((dk < st->tol_keff && dphi < st->tol_flux) || it + 1 >= st->max_outer) st->done_next = 1;   // :1799-1802
}
// phi_g = Chebyshev(raw_g / norm) for every group of a cell, and the next outer's total_fiss + prod_old partials (:1700-1707)
__global__ __launch_bounds__(256) void k_normalize_fission(const double *__restrict__ raw, double *__restrict__ cur,
                                                           double *__restrict__ p0, double *__restrict__ p1,
                                                           const double *__restrict__ Mf, int ng, long n,
                                                           const OuterState *__restrict__ st, double *__restrict__ tf,
                                                           double *__restrict__ partials)
{
    __shared__ double sred[4];
    if (st->done) return;
    const double norm = st->norm, ca = st->a, cb = st->b;
    const int do_norm = st->do_norm, mode = st->mode;
    double *pa = st->swap ? p1 : p0, *pb = st->swap ? p0 : p1;
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        double t = 0.0;
        for (int g = 0; g < ng; ++g) {
            const long j = g * n + i;
            double v = raw[j];
            if (do_norm) v /= norm;
            if (mode == 1) pa[j] = v;
            else if (mode == 2) { const double a = pa[j]; v = a + ca * (v - a); pb[j] = v; }
            else if (mode == 3) { const double a = pa[j], b = pb[j]; v = b + ca * (v - b) + cb * (b - a); pa[j] = v; }
            cur[j] = v;
            t += Mf[j] * v;
        }
        tf[i] = t; s += t;
    }
    s = block_sum(s, sred);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// Resident solve: the whole SolveKeff (full Schur path: power iteration, Gauss-Seidel group sweep, CG, Chebyshev;
// src/NeutFEM.cpp:1627-1815, src/solvers.cpp:577-636, 664-756) of a SMALL undivided mesh in ONE workgroup and ONE launch.
// The reference's own benchmark meshes (IAEA-2D 38x38, KOEBERG 34x34, every coarse-mesh initialisation) are a few thousand
// unknowns: as separate launches a CG iteration is 4 x ~5 us of launch and cross-chip latency for ~100 KB of data.  Here every
// phase is separated by a workgroup barrier only; the direction passes are the same device functions as the big-mesh kernels
// (schur_x_task, schur_s_tile: identical per-cell arithmetic), run over "virtual" tiles; reductions are fixed-order
// (thread-strided, then wave, then across waves), so runs are reproducible; every thread derives the same scalars from them.
struct ResidentArgs {
    Geom G; int ng, dim, nmodes; long N, nphi;
    ModeArgs ma[3]; ModeTab mt[3];          // per direction, mode 0, group 0, x -> p, y -> q; Cd and D advance with the group
    const double *L[3], *DR[3], *D0[3]; long nlines[3];
    const double *Mf, *Chi; const double *const *Ms;             // Ms: ng*ng table of scatter diagonals (null = empty block, :1722)
    double *phi, *raw, *p0, *p1, *tf, *r, *p, *q;
    int lpl_log2, ntask_x;
    int n[2], TX[2], NSEG[2], gx[2], gy[2]; long sl[2], ostride[2];
    double keff0, tol_keff, tol_flux, cg_tol; int cg_max, max_outer;
    double ca1, a3[16], cb[16];
    double *hist; int *hist_cg; ResidentOut *out;
    // LDS residency (host plan, greedy by priority): bit 0 p, 1 q, 2 r, 3 x_sol, 4+2d L[d], 5+2d DR[d], 10 C diagonal.  One CU
    // moves ~10 B/cycle to and from L2 but 128 B/cycle to and from LDS, and a barrier no longer waits for global store acks.
    const double *Cd0; int lds_mask;
    // Line-per-lane variants (k_resident_keff<.., NB, PITCH != 0>): first lane slot of every direction (each direction starts on a
    // wavefront boundary; slot0[3] = number of slots), and whether a direction's lines are swept by pairs of lanes
    int slot0[4], tw[3];                                         // tw: two lanes per line in that direction (lines of >= 4 cells)
    // the same for RT_k-P_m with bubble moments (k_resident_keff<.., NB > 0, -1>): cell pitch, the moment index of every (direction,
    // transverse mode, along-index), and per (moment, direction) the factor T_a G_l^2 / M^bb_l of the bubble's diagonal term (0 when
    // the moment has no bubble in that direction), which is folded into the C diagonal once per group
    int PC, mom[3][9][3]; double diagc[27][3];
};
__device__ __forceinline__ double block_total(double v, double *sred)     // fixed-order sum over the block, result in every thread
{
    v = block_sum(v, sred);
    if (threadIdx.x == 0) sred[0] = v;
    __syncthreads();
    v = sred[0];
    __syncthreads();
    return v;
}
// One barrier instead of four: every wavefront leaves its partial in the buffer of the given parity and every thread adds the (at most 8)
// partials in the same fixed order.  Two reductions of the same parity must be separated by a barrier (the CG loop alternates).
__device__ __forceinline__ double block_total_1b(double v, double *sred, int parity)
{
    v = wave_sum(v);
    double *b = sred + parity * 8;
    const int nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) b[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += b[w];
    return s;
}
// RT0-P0 on a mesh that lives in LDS: one lane per line, both sweeps serial along the line -- no scans, no barriers.  For lines of a few
// dozen cells the dependent chain of n multiply-adds per sweep is several times shorter than the instruction stream of the segmented
// scans (in-kernel stamps, IAEA-2D 38 x 38: 10.9 k + 6.2 k cycles for the x and y passes of the scan kernels).
//   forward  z_{c+1} = (x_c - x_{c+1}) - L_c z_c, z_0 = -x_0, w_f = z_f / d_f ; backward u_f = w_f - L_f u_{f+1} ; out_c = Ta (u_{c+1} - u_c)
// blk: the direction's block of three arrays at a compile-time pitch -- L, 1/d, out -- so that one address register serves all three
// (LDS instructions carry a 16-bit byte offset).  The n - 1 cells that have an upper neighbour go in unpredicated pieces of 8 / 4 / 2 / 1
// cells (a predicated or clamped load costs more instructions than the arithmetic), the last cell on its own.
typedef __attribute__((address_space(3))) double lds_f64;      // explicit LDS pointers: ds_read / ds_write with 32-bit addresses, not flat accesses
// Three per-direction values as named members.  A local array T v[3] that is indexed by a run-time direction anywhere (or through a chain
// d == 0 ? v[0] : ..., which the optimiser folds back into v[d]) is kept in scratch memory for the whole kernel, and every use inside the
// CG loop becomes a scratch load in front of the dependent LDS access.
template <class T> struct Tri {
    T a, b, c;
    __device__ __forceinline__ T operator[](int d) const { return d == 0 ? a : d == 1 ? b : c; }
    __device__ __forceinline__ void set(int d, T v) { if (d == 0) a = v; else if (d == 1) b = v; else c = v; }
};
template <int CH, int PITCH>
__device__ __forceinline__ void serial_fwd(const lds_f64 *&xp, lds_f64 *&bp, int sl, double &z, double &xc)
{
    double xv[CH + 1], Lv[CH], Rv[CH];
    xv[0] = xc;
#pragma unroll
    for (int i = 0; i < CH; ++i) { Lv[i] = bp[i * sl]; Rv[i] = bp[i * sl + PITCH]; xv[i + 1] = xp[(i + 1) * sl]; }
#pragma unroll
    for (int i = 0; i < CH; ++i) { z = (xv[i] - xv[i + 1]) - Lv[i] * z; bp[i * sl + 2 * PITCH] = z * Rv[i]; }
    xc = xv[CH]; xp += CH * sl; bp += CH * sl;
}
template <int CH, int PITCH>
__device__ __forceinline__ void serial_bwd(lds_f64 *&bp, int sl, double Ta, double &u2, double &Ln)
{
    double wv[CH], Lv[CH];
    bp -= CH * sl;
#pragma unroll
    for (int i = 0; i < CH; ++i) { wv[i] = bp[i * sl + 2 * PITCH]; Lv[i] = bp[i * sl]; }
#pragma unroll
    for (int i = CH - 1; i >= 0; --i) {
        const double u1 = wv[i] - Ln * u2;                       // u_{c+1}
        bp[(i + 1) * sl + 2 * PITCH] = Ta * (u2 - u1);           // cell c + 1
        u2 = u1; Ln = Lv[i];
    }
}
// Two lanes per line ("burn at both ends"): the even lane eliminates from face 0 up to the middle face k = n / 2, the odd lane -- on the
// mirrored line, with the factors of the mirrored elimination in its cells' slots (serial_twist_factors) -- from face n down to it; they
// meet in u_k = (b_k - l_{k-1} z_{k-1} - l~ z~) / d'_k and substitute outwards.  Mirroring turns b into -b and u into -u, the outputs
// T_a (u_{c+1} - u_c) come out the same, so both lanes run the same code with `cnt` own cells, a stride of opposite sign and their own
// first pivot.  Half the dependent chain, half the instructions per wavefront.
__device__ __forceinline__ double quad_swap(double v)            // value of the partner lane (lane ^ 1)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pair_even(double v)            // value of the even lane of the pair
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xA0, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xA0, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int PITCH>
__device__ __forceinline__ void serial_line_rt0(const lds_f64 *x, lds_f64 *blk, double d0, double dm, double Ta, int cnt, int base, int sl,
                                                bool two_sided, bool top)
{
    const lds_f64 *xp = x + base; lds_f64 *bp = blk + base;
    double xc = xp[0], z = -xc;
    const double w0 = z * d0;
    const int m = cnt - 1;
    for (int k = 0; k < (m >> 3); ++k) serial_fwd<8, PITCH>(xp, bp, sl, z, xc);
    if (m & 4) serial_fwd<4, PITCH>(xp, bp, sl, z, xc);
    if (m & 2) serial_fwd<2, PITCH>(xp, bp, sl, z, xc);
    if (m & 1) serial_fwd<1, PITCH>(xp, bp, sl, z, xc);
    double Ln = bp[0], u2;
    if (two_sided) {
        const double Pown = Ln * z, b = xc - xp[sl];             // xp[sl]: the partner's last cell
        const double Poth = quad_swap(Pown);
        const double um = ((b + Poth) - Pown) * dm;
        const double ut = pair_even(um);                         // one value for both halves
        u2 = top ? ut : -ut;
    } else {
        z = xc - Ln * z;                                         // the last cell: nothing above it
        u2 = z * bp[PITCH];                                      // u_n = w_n (L_n = 0)
    }
    if (m & 1) serial_bwd<1, PITCH>(bp, sl, Ta, u2, Ln);
    if (m & 2) serial_bwd<2, PITCH>(bp, sl, Ta, u2, Ln);
    if (m & 4) serial_bwd<4, PITCH>(bp, sl, Ta, u2, Ln);
    for (int k = 0; k < (m >> 3); ++k) serial_bwd<8, PITCH>(bp, sl, Ta, u2, Ln);
    const double u0 = w0 - Ln * u2;
    bp[2 * PITCH] = Ta * (u2 - u0);
}
// Factors of the mirrored elimination for the upper part of a line (cells k .. n-1), from the stored top-down factors: d_f = 1 / (1/d_f),
// e_c = l_c d_c, t_f = d_f + l_{f-1} e_{f-1}; then d^_n = t_n, l~_f = e_f / d^_{f+1}, d^_f = t_f - l~_f e_f downwards.  Cell c gets l~_c in
// its L slot and 1 / d^_c in its 1/d slot (what the mirrored sweep expects there); returns 1 / d^_n, and 1 / d'_k of the middle face
// (d'_k = d_k - l~_k e_k) through dm.  L: the line's first cell, oR: offset of the 1/d slots.
__device__ __forceinline__ double serial_twist_factors(lds_f64 *L, int oR, double d0inv, int n, int k, int sl, double &dm)
{
    auto dpiv = [&](int f) { return 1.0 / (f == 0 ? d0inv : (double)L[(f - 1) * sl + oR]); };   // standard pivot of face f
    double dlo = dpiv(n - 1), elo = L[(n - 1) * sl] * dlo;       // d_{n-1}, e_{n-1}
    double dhat = dpiv(n) + L[(n - 1) * sl] * elo;               // t_n
    const double first = 1.0 / dhat;
    for (int f = n - 1; f > k; --f) {                            // face f: cell f holds l~_f and 1 / d^_f
        const double dlo2 = dpiv(f - 1), elo2 = L[(f - 1) * sl] * dlo2;
        const double tf = dlo + L[(f - 1) * sl] * elo2;
        const double lt = elo / dhat;
        L[f * sl] = lt;
        dhat = tf - lt * elo;
        L[f * sl + oR] = 1.0 / dhat;
        dlo = dlo2; elo = elo2;
    }
    const double lk = elo / dhat;                                // dlo = d_k, elo = e_k here
    L[k * sl] = lk;
    dm = 1.0 / (dlo - lk * elo);
    return first;
}
// The same for RT_k-P_m with NB bubble moments (formulas above ModeArgs).  x / y: moment 0 of the line's transverse mode at the first cell,
// o1 / o2: offsets of the along-moments 1 and 2; L at the first cell, 1/d at offset oR.  The bubbles' own diagonal term
// T_a G_l^2 x_{l+1} / (M^bb_l c_e) is not formed here: it sits in the thread-private C diagonal of the cell pass.
struct HiConst { double eL[2], eR[2], Gc[2]; };
template <int NB>
__device__ __forceinline__ void hi_faces(double x0, double x1, double x2, const HiConst &c, double &xL, double &xR)
{
    double pl = 0.0, pr = 0.0;
    if (NB > 0) { const double g = c.Gc[0] * x1; pl += c.eL[0] * g; pr += c.eR[0] * g; }
    if (NB > 1) { const double g = c.Gc[1] * x2; pl += c.eL[1] * g; pr += c.eR[1] * g; }
    xL = x0 + pl; xR = x0 - pr;
}
template <int CH, int NB>
__device__ __forceinline__ void hi_fwd(const lds_f64 *&xp, int o1, int o2, const lds_f64 *&Lp, int oR, lds_f64 *&wp, int sl, double &z, double &xRc,
                                       const HiConst &c)
{
    double x0[CH], x1[CH], x2[CH], Lv[CH], Rv[CH];              // xp: the cell above the current one; Lp, wp: the current cell
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        x0[i] = xp[i * sl]; x1[i] = NB > 0 ? xp[i * sl + o1] : 0.0; x2[i] = NB > 1 ? xp[i * sl + o2] : 0.0;
        Lv[i] = Lp[i * sl]; Rv[i] = Lp[i * sl + oR];
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        double xLn, xRn;
        hi_faces<NB>(x0[i], x1[i], x2[i], c, xLn, xRn);
        z = (xRc - xLn) - Lv[i] * z; wp[i * sl] = z * Rv[i]; xRc = xRn;
    }
    xp += CH * sl; Lp += CH * sl; wp += CH * sl;
}
template <int NB>
__device__ __forceinline__ void hi_emit(lds_f64 *yp, int o1, int o2, double Ta, double lo, double hi, const HiConst &c)
{
    yp[0] = Ta * (hi - lo);
    if (NB > 0) yp[o1] = -(Ta * c.Gc[0]) * (c.eL[0] * lo + c.eR[0] * hi);
    if (NB > 1) yp[o2] = -(Ta * c.Gc[1]) * (c.eL[1] * lo + c.eR[1] * hi);
}
template <int CH, int NB>
__device__ __forceinline__ void hi_bwd(const lds_f64 *&Lp, lds_f64 *&yp, int o1, int o2, int sl, double Ta, double &u2, double &Ln, const HiConst &c)
{
    double wv[CH], Lv[CH];
    Lp -= CH * sl; yp -= CH * sl;
#pragma unroll
    for (int i = 0; i < CH; ++i) { wv[i] = yp[i * sl]; Lv[i] = Lp[i * sl]; }
#pragma unroll
    for (int i = CH - 1; i >= 0; --i) {
        const double u1 = wv[i] - Ln * u2;                       // u_{c+1}: lower face of cell c + 1, u2 its upper face
        hi_emit<NB>(yp + (i + 1) * sl, o1, o2, Ta, u1, u2, c);
        u2 = u1; Ln = Lv[i];
    }
}
// cnt own cells; two_sided: see serial_line_rt0 -- the mirrored half sees the odd along-moment with the opposite sign, which is the
// caller's business (HiConst::Gc[0] negated: it multiplies x_1 on the way in and y_1 on the way out)
template <int NB>
__device__ __forceinline__ void serial_line_hi(const lds_f64 *x, int o1, int o2, const lds_f64 *L, int oR, lds_f64 *y, double d0, double dm, double Ta,
                                               const HiConst &c, int cnt, int sl, bool two_sided, bool top)
{
    const lds_f64 *xp = x + sl, *Lp = L; lds_f64 *wp = y;
    double xL0, xRc;
    hi_faces<NB>(x[0], NB > 0 ? x[o1] : 0.0, NB > 1 ? x[o2] : 0.0, c, xL0, xRc);
    double z = -xL0;
    const double w0 = z * d0;
    const int m = cnt - 1;
    for (int k = 0; k < (m >> 2); ++k) hi_fwd<4, NB>(xp, o1, o2, Lp, oR, wp, sl, z, xRc, c);
    if (m & 2) hi_fwd<2, NB>(xp, o1, o2, Lp, oR, wp, sl, z, xRc, c);
    if (m & 1) hi_fwd<1, NB>(xp, o1, o2, Lp, oR, wp, sl, z, xRc, c);
    double Ln = Lp[0], u2;
    if (two_sided) {
        double xLn, xRn;
        hi_faces<NB>(xp[0], NB > 0 ? xp[o1] : 0.0, NB > 1 ? xp[o2] : 0.0, c, xLn, xRn);   // the partner's last cell, seen from this side
        const double Pown = Ln * z, b = xRc - xLn;
        const double Poth = quad_swap(Pown);
        const double um = ((b + Poth) - Pown) * dm;
        const double ut = pair_even(um);
        u2 = top ? ut : -ut;
    } else {
        z = xRc - Ln * z;                                        // the last cell: nothing above it
        u2 = z * Lp[oR];
    }
    lds_f64 *yp = wp;
    if (m & 1) hi_bwd<1, NB>(Lp, yp, o1, o2, sl, Ta, u2, Ln, c);
    if (m & 2) hi_bwd<2, NB>(Lp, yp, o1, o2, sl, Ta, u2, Ln, c);
    for (int k = 0; k < (m >> 2); ++k) hi_bwd<4, NB>(Lp, yp, o1, o2, sl, Ta, u2, Ln, c);
    const double u0 = w0 - Ln * u2;
    hi_emit<NB>(yp, o1, o2, Ta, u0, u2, c);
}
// all tiles of one y / z pass, nconc = blockDim / (TX NSEG) of them side by side; every thread runs every round (barriers inside)
template <int SEG, int DIR, int NB>
__device__ __forceinline__ double resident_s_pass(const ResidentArgs &A, int r, const ModeArgs &mad, const double *Ld, const double *DRd, int g, double *sm,
                                                  const SlabArgs &sa0, const CgFuse &fz)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int T = A.TX[r] * A.NSEG[r];
    const int nconc = nt / T, slot = tid / T, ltid = tid - slot * T;
    const int ntiles = A.gx[r] * A.gy[r] * A.nmodes;
    double *smt = sm + (slot < nconc ? slot : 0) * (4 * T + (A.NSEG[r] >= 64 ? 5 : 1) * A.TX[r]);   // padded rows when the wavefront scan is on
    double dot = 0.0;
    for (int t0 = 0; t0 < ntiles; t0 += nconc) {
        const int t = t0 + slot;
        const bool act = slot < nconc && t < ntiles;
        const unsigned tt = act ? t : 0;
        const unsigned bx = tt % A.gx[r], by = (tt / A.gx[r]) % A.gy[r], bz = tt / (A.gx[r] * A.gy[r]);
        const ModeArgs ma = select_mode(mad, A.mt[DIR], bz, NB + 1);
        dot += schur_s_tile<SEG, DIR, false, NB>(ma, A.G, Ld, DRd, A.D0[DIR] + g * A.nlines[DIR], A.n[r], A.sl[r], A.ostride[r],
                                                 A.G.nx, A.TX[r], A.NSEG[r], bx, by, bz, A.gy[r], ltid, act, smt, sa0, fz, false, false, 0.0, 0.0, true);
        __syncthreads();                                         // the tile's scan arrays are reused by the next round
    }
    return dot;
}
template <bool VEC, int NB, int PITCH = 0>                        // PITCH > 0: the line-per-lane variant (RT0-P0, everything in LDS)
__global__ __launch_bounds__(512) void k_resident_keff(ResidentArgs A)
{
    constexpr bool SERIAL = PITCH > 0;                             // RT0-P0 line-per-lane variant
    constexpr bool SERH = PITCH < 0;                               // line-per-lane variant with bubble moments
    static_assert(!SERIAL || NB == 0, "a compile-time pitch is the RT0-P0 variant");
    static_assert(!SERH || NB > 0, "the run-time pitch variant is for NB > 0");
    constexpr int SEG = NB > 0 ? 4 : 8;
    extern __shared__ double sm[];
    const int scr = (SERIAL || SERH) ? 0 : 5 * (int)blockDim.x;   // scratch of the tiled passes
    double *sred = sm + scr + 64;
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, nw = nt >> 6;
    const int ng = A.ng; const long N = A.N, NP = A.nphi, NT = NP * ng;
    SlabArgs sa0; sa0.if_lo = sa0.if_hi = sa0.mode = sa0.xcd = sa0.wsmin = sa0.fold = sa0.noacc = 0; sa0.yadd = nullptr; sa0.alo = sa0.ahi = sa0.ulo = sa0.uhi = sa0.rlo = sa0.rhi = sa0.sinv_lo = sa0.sinv_hi = nullptr; sa0.clo = sa0.chi = sa0.jz = sa0.jzb = nullptr; sa0.nfa = 1; sa0.ni = 0;
    // vectors and factors that fit are kept in LDS behind the scratch area (ResidentArgs::lds_mask); the rest stays in global memory
    double *lds = sm + scr + 64 + 16;
    long lo = 0;
    auto carve = [&](int bit, long n) -> double * { if (!((A.lds_mask >> bit) & 1)) return nullptr; double *q_ = lds + lo; lo += (n + 1) & ~1L; return q_; };
    double *vp = carve(0, NP), *vq = carve(1, NP), *vr = carve(2, NP), *vx = carve(3, NP);
    double *fL[3], *fR[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { fL[d] = carve(4 + 2 * d, N); fR[d] = carve(5 + 2 * d, N); }
    double *fC = carve(10, NP);
    // Line-per-lane variant.  LDS: p, then per direction a block {L, 1/d, contribution} at pitch PITCH and the first pivots of its lines;
    // rows padded to an odd length nxp (x lines: lane = line, so the lanes of a wavefront are nxp doubles apart -- odd means no bank
    // conflicts; an even nx = 38 gave 3-way conflicts on every access and an LDS-bound sweep).  Every thread owns the KC = PITCH / 512
    // padded cells tid + 512 k for the whole solve and keeps their r, x_sol and C diagonal in registers; cells of the padding and beyond
    // the mesh hold zeros everywhere, so nothing below is predicated.
    constexpr int KC = SERIAL ? PITCH / 512 : 1;
    lds_f64 *lp = nullptr; Tri<lds_f64 *> lb = { nullptr, nullptr, nullptr }; Tri<const lds_f64 *> lD = { nullptr, nullptr, nullptr };
    Tri<lds_f64 *> lDm = { nullptr, nullptr, nullptr }, lDr = { nullptr, nullptr, nullptr };   // 1 / d'_k and 1 / d^_n per line (two-sided sweeps)
    int gi[KC], nxp = 0;
    if (SERIAL) {
        nxp = A.G.nx | 1;
        lp = (lds_f64 *)lds; lo = PITCH;
#pragma unroll
        for (int d = 0; d < 3; ++d) if (d < A.dim) { lb.set(d, (lds_f64 *)(lds + lo)); lo += 3 * PITCH; }
#pragma unroll
        for (int d = 0; d < 3; ++d) if (d < A.dim) {
            const long nl2 = (A.nlines[d] + 1) & ~1L;
            lD.set(d, (const lds_f64 *)(lds + lo)); lDm.set(d, (lds_f64 *)(lds + lo + nl2)); lDr.set(d, (lds_f64 *)(lds + lo + 2 * nl2)); lo += 3 * nl2;
        }
        const int Np = nxp * A.G.ny * A.G.nz;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int ip = tid + k * 512, row = ip / nxp, ix = ip - row * nxp;
            gi[k] = (ip < Np && ix < A.G.nx) ? row * A.G.nx + ix : -1;
            lp[ip] = 0.0;
#pragma unroll
            for (int d = 0; d < 3; ++d) if (d < A.dim) lb[d][2 * PITCH + ip] = 0.0;
        }
    }
    // Higher orders: DOF j = moment * PC + padded cell; p and one contribution vector per direction (nloc PC each), {L, 1/d} per
    // direction (PC each), first pivots; small tables (moment indices, T_a, diagonal factors) in LDS because a dynamically indexed
    // kernel argument would move the whole argument struct to scratch.  Every thread owns the DOFs tid + 512 k, k < KD.
    constexpr int KD = SERH ? 10 : 1;
    lds_f64 *hp = nullptr; Tri<lds_f64 *> hc = { nullptr, nullptr, nullptr }, hL = { nullptr, nullptr, nullptr }, hD = { nullptr, nullptr, nullptr };
    Tri<lds_f64 *> hDm = { nullptr, nullptr, nullptr }, hDr = { nullptr, nullptr, nullptr };
    lds_f64 *tTa = nullptr, *tDg = nullptr; __attribute__((address_space(3))) int *tMom = nullptr;
    int gj[KD], hPC = 0, hNPp = 0, hNp = 0;
    if (SERH) {
        nxp = A.G.nx | 1; hPC = A.PC; hNp = nxp * A.G.ny * A.G.nz;
        const int nloc = (int)(NP / N); hNPp = nloc * hPC;
        tTa = (lds_f64 *)lds; tDg = tTa + 32; tMom = (__attribute__((address_space(3))) int *)(tDg + 96); lo = 32 + 96 + 48;
        hp = (lds_f64 *)(lds + lo); lo += hNPp;
#pragma unroll
        for (int d = 0; d < 3; ++d) if (d < A.dim) { hc.set(d, (lds_f64 *)(lds + lo)); lo += hNPp; }
#pragma unroll
        for (int d = 0; d < 3; ++d) if (d < A.dim) { hL.set(d, (lds_f64 *)(lds + lo)); lo += 2 * hPC; }
#pragma unroll
        for (int d = 0; d < 3; ++d) if (d < A.dim) {
            const long nl2 = (A.nlines[d] + 1) & ~1L;
            lds_f64 *const b_ = (lds_f64 *)(lds + lo);
            hD.set(d, b_); hDm.set(d, b_ + nl2); hDr.set(d, b_ + 2 * nl2); lo += 3 * nl2;
        }
        if (tid < 27) { const int d = tid / 9, m = tid % 9; tTa[tid] = A.mt[d].n > 1 ? A.mt[d].Ta[m] : A.ma[d].Ta; }
        if (tid < 81) { tDg[tid] = A.diagc[tid / 3][tid % 3]; tMom[tid] = A.mom[tid / 27][(tid / 3) % 9][tid % 3]; }
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int j = tid + k * 512, mo = j / hPC, ip = j - mo * hPC, row = ip / nxp, ix = ip - row * nxp;
            gj[k] = (j < hNPp && ip < hNp && ix < A.G.nx) ? mo * (int)N + row * A.G.nx + ix : -1;
            if (j < hNPp) { hp[j] = 0.0; hc[0][j] = 0.0; if (A.dim >= 2) hc[1][j] = 0.0; if (A.dim == 3) hc[2][j] = 0.0; }
        }
        __syncthreads();
    }
    double *const wp = vp ? vp : A.p, *const wq = vq ? vq : A.q, *const wr = vr ? vr : A.r;
    double keff = A.keff0;
    int cheb_it = 0, n_outer = 0, status = 0, cg_total = 0;
    double *pa = A.p0, *pb = A.p1;
    for (int it = 0; it < A.max_outer; ++it) {
        // total_fiss and prod_old (:1700-1707)
        double s = 0.0;
        for (long i = tid; i < NP; i += nt) {
            double v = 0.0;
            for (int g = 0; g < ng; ++g) v += A.Mf[g * NP + i] * A.phi[g * NP + i];
            A.tf[i] = v; s += v;
        }
        const double prod_old = block_total(s, sred);
        const double inv_k = 1.0 / keff;
        for (int g = 0; g < ng; ++g) {
            if constexpr (SERH) {
                double *const graw = A.raw + (long)g * NP;
                double rv[KD], xv[KD], cv[KD], pv[KD];
                const int nx = A.G.nx, ny = A.G.ny, n_ = (int)N;
                // this group's factors and first pivots
#pragma unroll
                for (int d = 0; d < 3; ++d) if (d < A.dim) {
                    for (int ip = tid; ip < hPC; ip += nt) {
                        const int row = ip / nxp, ix = ip - row * nxp;
                        const bool ok = ip < hNp && ix < nx;
                        const long e = g * N + row * nx + ix;
                        hL[d][ip] = ok ? A.L[d][e] : 0.0; hL[d][hPC + ip] = ok ? A.DR[d][e] : 0.0;
                    }
                    for (int i = tid; i < (int)A.nlines[d]; i += nt) hD[d][i] = A.D0[d][g * A.nlines[d] + i];
                }
                // two-sided directions: mirrored factors for the upper half of every line (one thread per line; the modes share them)
                __syncthreads();
#pragma unroll
                for (int d = 0; d < 3; ++d) if (d < A.dim && A.tw[d]) {
                    for (int l = tid; l < (int)A.nlines[d]; l += nt) {
                        const int base = d == 0 ? l * nxp : d == 1 ? (l / nx) * nxp * ny + l % nx : (l / nx) * nxp + l % nx;
                        const int sl = d == 0 ? 1 : d == 1 ? nxp : nxp * ny, n = d == 0 ? nx : d == 1 ? ny : A.G.nz;
                        double dm;
                        hDr[d][l] = serial_twist_factors(hL[d] + base, hPC, hD[d][l], n, n / 2, sl, dm);
                        hDm[d][l] = dm;
                    }
                }
                // rhs (:1716-1726), CG start (src/solvers.cpp:583-592); C diagonal + the bubbles' diagonal terms of every direction
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < KD; ++k) {
                    double v = 0.0; cv[k] = 0.0;
                    if (gj[k] >= 0) {
                        const int i = gj[k], mo = i / n_, e = i - mo * n_;
                        const double chv = A.Chi[g * N + e] * inv_k;
                        v = fabs(chv) < 1e-14 ? 0.0 : chv * A.tf[i];
                        for (int gp = 0; gp < ng; ++gp) {
                            const double *M = A.Ms[g * ng + gp];
                            if (gp == g || !M) continue;
                            v += M[i] * (gp < g ? A.raw : A.phi)[gp * NP + i];
                        }
                        const int ex = e % nx, ey = (e / nx) % ny, ez = e / (nx * ny);
                        const double Dc = A.ma[0].D[g * N + e];
                        double cd = A.Cd0[g * NP + i];
                        for (int d = 0; d < A.dim; ++d) { const double f = tDg[mo * 3 + d]; if (f != 0.0) cd += f * (Dc / geom_factor(A.G, d, ex, ey, ez)); }
                        cv[k] = cd;
                    }
                    rv[k] = v; pv[k] = v; xv[k] = 0.0; s += v * v;
                    if (tid + k * 512 < hNPp) hp[tid + k * 512] = v;
                }
                double rr = block_total(s, sred);
                const double rhs_norm = sqrt(rr), tol_sq = A.cg_tol * A.cg_tol * rhs_norm * rhs_norm;
                int its = 0;
                HiConst hcst;
                for (int l = 0; l < 2; ++l) { hcst.eL[l] = A.ma[0].eL[l]; hcst.eR[l] = A.ma[0].eR[l]; hcst.Gc[l] = A.ma[0].Gc[l]; }
                const int nm = A.nmodes;
                // which (direction, mode, line half) a lane slot sweeps; slot `tid` is decoded once per group
                struct LaneLine { bool ok, tw, top; int cnt, sl, o1, o2; double d0, dm, Ta; const lds_f64 *x, *L; lds_f64 *y; HiConst c; };
                auto lane_line = [&](int sl_) -> LaneLine {
                    LaneLine o; o.ok = false; o.tw = o.top = false; o.cnt = o.sl = o.o1 = o.o2 = 0; o.d0 = o.dm = o.Ta = 0.0; o.x = o.L = hp; o.y = hc[0]; o.c = hcst;
                    if (sl_ >= A.slot0[3]) return o;
                    const int d = sl_ >= A.slot0[2] ? 2 : sl_ >= A.slot0[1] ? 1 : 0;
                    const int nl = d == 0 ? (int)A.nlines[0] : d == 1 ? (int)A.nlines[1] : (int)A.nlines[2];
                    const int twd = d == 0 ? A.tw[0] : d == 1 ? A.tw[1] : A.tw[2];
                    const int rem = sl_ - (d == 0 ? A.slot0[0] : d == 1 ? A.slot0[1] : A.slot0[2]);
                    const int lane_l = twd ? rem >> 1 : rem;
                    const int mode = lane_l / nl, l = lane_l - mode * nl;
                    if (mode >= nm) return o;
                    o.ok = true; o.tw = twd != 0; o.top = !twd || !(rem & 1);
                    const int m0 = tMom[(d * 9 + mode) * 3];
                    o.o1 = (tMom[(d * 9 + mode) * 3 + 1] - m0) * hPC; o.o2 = NB > 1 ? (tMom[(d * 9 + mode) * 3 + 2] - m0) * hPC : 0;
                    const int base0 = d == 0 ? l * nxp : d == 1 ? (l / nx) * nxp * ny + l % nx : (l / nx) * nxp + l % nx;
                    const int sl0 = d == 0 ? 1 : d == 1 ? nxp : nxp * ny, n = d == 0 ? nx : d == 1 ? ny : A.G.nz;
                    o.cnt = !twd ? n : o.top ? n / 2 : n - n / 2; o.sl = o.top ? sl0 : -sl0;
                    const int base = o.top ? base0 : base0 + (n - 1) * sl0;
                    o.x = hp + m0 * hPC + base; o.L = (d == 0 ? hL[0] : d == 1 ? hL[1] : hL[2]) + base; o.y = (d == 0 ? hc[0] : d == 1 ? hc[1] : hc[2]) + m0 * hPC + base;
                    o.d0 = o.top ? (d == 0 ? hD[0] : d == 1 ? hD[1] : hD[2])[l] : (d == 0 ? hDr[0] : d == 1 ? hDr[1] : hDr[2])[l];
                    o.dm = twd ? (d == 0 ? hDm[0] : d == 1 ? hDm[1] : hDm[2])[l] : 0.0;
                    o.Ta = tTa[d * 9 + mode];
                    if (!o.top) o.c.Gc[0] = -o.c.Gc[0];          // mirrored line: the odd along-moment changes sign
                    return o;
                };
                const LaneLine my = lane_line(tid);
                while (its < A.cg_max) {
#ifdef NF_STAMPS
                    long long ts0 = (long long)__builtin_readcyclecounter();
#endif
                    // ---- every (direction, transverse mode, line) at once, one lane or a pair of lanes each
                    {
                        LaneLine o = my;
                        for (int sl_ = tid;;) {                  // one call site: slot tid from the registers, any further slot decoded on the way
                            if (o.ok) serial_line_hi<NB>(o.x, o.o1, o.o2, o.L, hPC, o.y, o.d0, o.dm, o.Ta, o.c, o.cnt, o.sl, o.tw, o.top);
                            sl_ += nt;
                            if (sl_ >= A.slot0[3]) break;
                            o = lane_line(sl_);
                        }
                    }
                    __syncthreads();
#ifdef NF_STAMPS
                    long long ts1 = (long long)__builtin_readcyclecounter();
#endif
                    double qv[KD], dot = 0.0;
#pragma unroll
                    for (int k = 0; k < KD; ++k) {
                        const int j = tid + k * 512; const bool in = j < hNPp;
                        const int jj = in ? j : 0;
                        double q_ = cv[k] * pv[k] + hc[0][jj];
                        if (A.dim >= 2) q_ += hc[1][jj];
                        if (A.dim == 3) q_ += hc[2][jj];
                        qv[k] = in ? q_ : 0.0; dot += pv[k] * qv[k];
                    }
#ifdef NF_STAMPS
                    long long ts2 = (long long)__builtin_readcyclecounter();
#endif
                    const double pq = block_total_1b(dot, sred, 0);   // src/solvers.cpp:602-606
#ifdef NF_STAMPS
                    long long ts3 = (long long)__builtin_readcyclecounter();
#endif
                    if (fabs(pq) < 1e-30) break;
                    const double alpha = rr / pq;
                    s = 0.0;
#pragma unroll
                    for (int k = 0; k < KD; ++k) { rv[k] = rv[k] - alpha * qv[k]; s += rv[k] * rv[k]; }
                    const double rr_new = block_total_1b(s, sred, 1);   // :613-631
                    ++its;
                    const bool conv = rr_new < tol_sq;
                    const double beta = rr_new / rr; rr = rr_new;
#pragma unroll
                    for (int k = 0; k < KD; ++k) {                // :609, :630
                        xv[k] = fma(alpha, pv[k], xv[k]);
                        pv[k] = fma(beta, pv[k], rv[k]);
                        if (tid + k * 512 < hNPp) hp[tid + k * 512] = pv[k];
                    }
                    __syncthreads();
#ifdef NF_STAMPS
                    if (tid == 0 && A.hist) { long long ts4 = (long long)__builtin_readcyclecounter(); long long *acc = (long long *)(A.hist + 3 * A.max_outer);
                                              acc[0] += ts1 - ts0; acc[1] += ts2 - ts1; acc[2] += ts3 - ts2; acc[3] += ts4 - ts3; acc[4] += 1; }
#endif
                    if (conv) break;
                }
#pragma unroll
                for (int k = 0; k < KD; ++k) if (gj[k] >= 0) graw[gj[k]] = xv[k];
                if (tid == 0) A.hist_cg[it * ng + g] = its;
                cg_total += its;
                __syncthreads();
                continue;
            }
            if constexpr (SERIAL) {
                double *const graw = A.raw + (long)g * N;
                double rv[KC], xv[KC], cv[KC], pv[KC];
                // this group's factors into the directions' blocks, first pivots, C diagonal into registers
#pragma unroll
                for (int d = 0; d < 3; ++d) if (d < A.dim) {
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const int ip = tid + k * 512;
                        lb[d][ip] = gi[k] >= 0 ? A.L[d][g * N + gi[k]] : 0.0;
                        lb[d][PITCH + ip] = gi[k] >= 0 ? A.DR[d][g * N + gi[k]] : 0.0;
                    }
                    for (int i = tid; i < (int)A.nlines[d]; i += nt) ((lds_f64 *)lD[d])[i] = A.D0[d][g * A.nlines[d] + i];
                }
                const int nx = A.G.nx, ny = A.G.ny;
                // two-sided directions: the odd lane of every pair rewrites the upper half of its line with the mirrored factors
                __syncthreads();
                for (int sl_ = tid; sl_ < A.slot0[3]; sl_ += nt) {
                    const int d = sl_ >= A.slot0[2] ? 2 : sl_ >= A.slot0[1] ? 1 : 0;
                    const int rem = sl_ - (d == 0 ? A.slot0[0] : d == 1 ? A.slot0[1] : A.slot0[2]);
                    const int twd = d == 0 ? A.tw[0] : d == 1 ? A.tw[1] : A.tw[2], nl = d == 0 ? (int)A.nlines[0] : d == 1 ? (int)A.nlines[1] : (int)A.nlines[2];
                    const int l = rem >> 1;
                    if (twd && (rem & 1) && l < nl) {
                        const int base = d == 0 ? l * nxp : d == 1 ? (l / nx) * nxp * ny + l % nx : (l / nx) * nxp + l % nx;
                        const int sl = d == 0 ? 1 : d == 1 ? nxp : nxp * ny, n = d == 0 ? nx : d == 1 ? ny : A.G.nz;
                        lds_f64 *const Lb = (d == 0 ? lb[0] : d == 1 ? lb[1] : lb[2]) + base;
                        const lds_f64 *const D0d = d == 0 ? lD[0] : d == 1 ? lD[1] : lD[2];
                        double dm;
                        const double first = serial_twist_factors(Lb, PITCH, D0d[l], n, n / 2, sl, dm);
                        (d == 0 ? lDm[0] : d == 1 ? lDm[1] : lDm[2])[l] = dm;
                        (d == 0 ? lDr[0] : d == 1 ? lDr[1] : lDr[2])[l] = first;
                    }
                }
                // rhs = chi_g tf / k + scatter (Gauss-Seidel) (:1716-1726); CG start x = 0, r = p = rhs (src/solvers.cpp:583-592)
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    double v = 0.0; cv[k] = 0.0;
                    if (gi[k] >= 0) {
                        const int i = gi[k];
                        v = inv_k * (A.Chi[g * N + i] * A.tf[i]);
                        for (int gp = 0; gp < ng; ++gp) {
                            const double *M = A.Ms[g * ng + gp];
                            if (gp == g || !M) continue;
                            v += M[i] * (gp < g ? A.raw : A.phi)[gp * N + i];
                        }
                        cv[k] = A.Cd0[g * N + i];
                    }
                    rv[k] = v; pv[k] = v; xv[k] = 0.0; lp[tid + k * 512] = v; s += v * v;
                }
                double rr = block_total(s, sred);
                const double rhs_norm = sqrt(rr), tol_sq = A.cg_tol * A.cg_tol * rhs_norm * rhs_norm;
                int its = 0;
                // which line (half) a lane slot sweeps; slot `tid` is decoded once per group, not once per iteration
                struct LaneLine { bool ok, tw, top; int cnt, base, sl; double d0, dm, Ta; lds_f64 *blk; };
                auto lane_line = [&](int sl_) -> LaneLine {
                    LaneLine o; o.ok = false; o.tw = o.top = false; o.cnt = o.base = o.sl = 0; o.d0 = o.dm = o.Ta = 0.0; o.blk = lb[0];
                    if (sl_ >= A.slot0[3]) return o;
                    const int d = sl_ >= A.slot0[2] ? 2 : sl_ >= A.slot0[1] ? 1 : 0;
                    const int rem = sl_ - (d == 0 ? A.slot0[0] : d == 1 ? A.slot0[1] : A.slot0[2]);
                    const int twd = d == 0 ? A.tw[0] : d == 1 ? A.tw[1] : A.tw[2], nl = d == 0 ? (int)A.nlines[0] : d == 1 ? (int)A.nlines[1] : (int)A.nlines[2];
                    const int l = twd ? rem >> 1 : rem;
                    if (l >= nl) return o;
                    o.ok = true; o.tw = twd != 0; o.top = !twd || !(rem & 1);
                    const int base0 = d == 0 ? l * nxp : d == 1 ? (l / nx) * nxp * ny + l % nx : (l / nx) * nxp + l % nx;
                    const int sl0 = d == 0 ? 1 : d == 1 ? nxp : nxp * ny, n = d == 0 ? nx : d == 1 ? ny : A.G.nz;
                    o.cnt = !twd ? n : o.top ? n / 2 : n - n / 2;
                    o.base = o.top ? base0 : base0 + (n - 1) * sl0; o.sl = o.top ? sl0 : -sl0;
                    o.Ta = d == 0 ? A.ma[0].Ta : d == 1 ? A.ma[1].Ta : A.ma[2].Ta;
                    o.blk = d == 0 ? lb[0] : d == 1 ? lb[1] : lb[2];
                    o.d0 = o.top ? (d == 0 ? lD[0] : d == 1 ? lD[1] : lD[2])[l] : (d == 0 ? lDr[0] : d == 1 ? lDr[1] : lDr[2])[l];
                    o.dm = twd ? (d == 0 ? lDm[0] : d == 1 ? lDm[1] : lDm[2])[l] : 0.0;
                    return o;
                };
                const LaneLine my = lane_line(tid);
                while (its < A.cg_max) {
#ifdef NF_STAMPS
                    long long ts0 = (long long)__builtin_readcyclecounter();
#endif
                    // ---- every line of every direction at once, one lane (or a pair of lanes) each: X p, Y p, Z p into the directions' blocks
                    {
                        LaneLine o = my;
                        for (int sl_ = tid;;) {                  // one call site: slot tid from the registers, any further slot decoded on the way
                            if (o.ok) serial_line_rt0<PITCH>(lp, o.blk, o.d0, o.dm, o.Ta, o.cnt, o.base, o.sl, o.tw, o.top);
                            sl_ += nt;
                            if (sl_ >= A.slot0[3]) break;
                            o = lane_line(sl_);
                        }
                    }
                    __syncthreads();
#ifdef NF_STAMPS
                    long long ts1 = (long long)__builtin_readcyclecounter();
#endif
                    // ---- own cells: q = C p + X p + Y p + Z p and p.q
                    double qv[KC], dot = 0.0;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const int ip = tid + k * 512;
                        double q_ = cv[k] * pv[k] + lb[0][2 * PITCH + ip];
                        if (A.dim >= 2) q_ += lb[1][2 * PITCH + ip];
                        if (A.dim == 3) q_ += lb[2][2 * PITCH + ip];
                        qv[k] = q_; dot += pv[k] * q_;
                    }
#ifdef NF_STAMPS
                    long long ts2 = (long long)__builtin_readcyclecounter();
#endif
                    const double pq = block_total_1b(dot, sred, 0);   // src/solvers.cpp:602-606
#ifdef NF_STAMPS
                    long long ts3 = (long long)__builtin_readcyclecounter();
#endif
                    if (fabs(pq) < 1e-30) break;
                    const double alpha = rr / pq;
                    s = 0.0;
#pragma unroll
                    for (int k = 0; k < KC; ++k) { rv[k] = rv[k] - alpha * qv[k]; s += rv[k] * rv[k]; }
                    const double rr_new = block_total_1b(s, sred, 1);   // :613-631
                    ++its;
                    const bool conv = rr_new < tol_sq;
                    const double beta = rr_new / rr; rr = rr_new;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {                // :609, :630
                        xv[k] = fma(alpha, pv[k], xv[k]);
                        pv[k] = fma(beta, pv[k], rv[k]);
                        lp[tid + k * 512] = pv[k];
                    }
                    __syncthreads();
#ifdef NF_STAMPS
                    if (tid == 0 && A.hist) { long long ts4 = (long long)__builtin_readcyclecounter(); long long *acc = (long long *)(A.hist + 3 * A.max_outer);
                                              acc[0] += ts1 - ts0; acc[1] += ts2 - ts1; acc[2] += ts3 - ts2; acc[3] += ts4 - ts3; acc[4] += 1; }
#endif
                    if (conv) break;
                }
#pragma unroll
                for (int k = 0; k < KC; ++k) if (gi[k] >= 0) graw[gi[k]] = xv[k];
                if (tid == 0) A.hist_cg[it * ng + g] = its;
                cg_total += its;
                __syncthreads();
                continue;
            }
            double *const graw = A.raw + (long)g * NP;
            double *const xsol = vx ? vx : graw;
            // this group's line factors and C diagonal into LDS (when planned)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (fL[d]) for (long i = tid; i < N; i += nt) fL[d][i] = A.L[d][g * N + i];
                if (fR[d]) for (long i = tid; i < N; i += nt) fR[d][i] = A.DR[d][g * N + i];
            }
            if (fC) for (long i = tid; i < NP; i += nt) fC[i] = A.Cd0[g * NP + i];
            // rhs = chi_g tf / k + scatter (Gauss-Seidel) (:1716-1726); CG start x = 0, r = p = rhs (src/solvers.cpp:583-592)
            s = 0.0;
            for (long i = tid; i < NP; i += nt) {
                double v;
                if (NP == N) v = inv_k * (A.Chi[g * N + i] * A.tf[i]);
                else { const double cv = A.Chi[g * N + i % N] * inv_k; v = fabs(cv) < 1e-14 ? 0.0 : cv * A.tf[i]; }
                for (int gp = 0; gp < ng; ++gp) {
                    const double *M = A.Ms[g * ng + gp];
                    if (gp == g || !M) continue;
                    v += M[i] * (gp < g ? A.raw : A.phi)[gp * NP + i];
                }
                xsol[i] = 0.0; wr[i] = v; wp[i] = v; s += v * v;
            }
            double rr = block_total(s, sred);
            const double rhs_norm = sqrt(rr), tol_sq = A.cg_tol * A.cg_tol * rhs_norm * rhs_norm;
            int its = 0, pend = 0; double alpha = 0.0, beta = 0.0;
            const CgFuse fz = { wp, wr, xsol, nullptr };
            ModeArgs mad[3];
            const double *dL[3], *dR[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                mad[d] = A.ma[d]; mad[d].D += g * N;
                for (int q = 0; q <= NB; ++q) {
                    const long off = A.ma[d].x[q] - A.p;         // offset of this moment inside a flux vector
                    mad[d].x[q] = wp + off; mad[d].y[q] = wq + off;
                    mad[d].Cd[q] = fC ? fC + off : A.Cd0 + g * NP + off;
                }
                dL[d] = fL[d] ? fL[d] : A.L[d] + g * N; dR[d] = fR[d] ? fR[d] : A.DR[d] + g * N;
            }
            while (its < A.cg_max) {
                const bool fuse = its > 0;
                double dot = 0.0;
#ifdef NF_STAMPS
                long long ts0 = (long long)__builtin_readcyclecounter();
#endif
                // ---- x lines: q = C p + X p, carrying the deferred x_sol += alpha p, p = r + beta p of the previous iteration
                const int ntask = A.ntask_x * A.nmodes;
                for (int gt0 = 0; gt0 < ntask; gt0 += nw) {
                    const int gt = gt0 + wave;
                    const bool active = gt < ntask;
                    const int mode = active ? gt / A.ntask_x : 0;
                    const ModeArgs ma = select_mode(mad[0], A.mt[0], mode, NB + 1);
                    const double dx = schur_x_task<2, 1, VEC, NB, NoMid, true>(ma, A.G, dL[0], dR[0], A.D0[0] + g * A.nlines[0], A.G.nx, A.G.ny,
                                                                  A.nlines[0], A.lpl_log2, 1, gt % A.ntask_x, tid & 63, active, fuse, alpha, beta, fz);
                    if (A.dim == 1) dot += dx;
                }
                __syncthreads();
#ifdef NF_STAMPS
                long long ts1 = (long long)__builtin_readcyclecounter();
#endif
                // ---- y, then z lines: accumulate into q; the last direction also gives p.q
                if (A.dim >= 2) { const double ds = resident_s_pass<SEG, 1, NB>(A, 0, mad[1], dL[1], dR[1], g, sm, sa0, fz); if (A.dim == 2) dot += ds; }
                if (A.dim == 3) dot += resident_s_pass<SEG, 2, NB>(A, 1, mad[2], dL[2], dR[2], g, sm, sa0, fz);
#ifdef NF_STAMPS
                long long ts2 = (long long)__builtin_readcyclecounter();
#endif
                const double pq = block_total(dot, sred);       // src/solvers.cpp:602-606
#ifdef NF_STAMPS
                long long ts3 = (long long)__builtin_readcyclecounter();
#endif
                pend = 0;
                if (fabs(pq) < 1e-30) break;
                alpha = rr / pq;
                s = 0.0;
                for (long i = tid; i < NP; i += nt) { const double rn = wr[i] - alpha * wq[i]; wr[i] = rn; s += rn * rn; }
                const double rr_new = block_total(s, sred);     // :613-631
#ifdef NF_STAMPS
                if (tid == 0 && A.hist) { long long ts4 = (long long)__builtin_readcyclecounter(); long long *acc = (long long *)(A.hist + 3 * A.max_outer);
                                          acc[0] += ts1 - ts0; acc[1] += ts2 - ts1; acc[2] += ts3 - ts2; acc[3] += ts4 - ts3; acc[4] += 1; }
#endif
                ++its; pend = 1;
                if (rr_new < tol_sq) { rr = rr_new; break; }
                beta = rr_new / rr; rr = rr_new;
            }
            // the last iteration's x_sol update; the group's solution goes (back) to global memory
            if (pend) for (long i = tid; i < NP; i += nt) graw[i] = fma(alpha, wp[i], xsol[i]);
            else if (vx) for (long i = tid; i < NP; i += nt) graw[i] = xsol[i];
            if (tid == 0) A.hist_cg[it * ng + g] = its;
            cg_total += its;
            __syncthreads();
        }
        // prod_new, norms (:1766-1779)
        double sp = 0.0, sn = 0.0, sd = 0.0;
        for (long i = tid; i < NT; i += nt) { const double v = A.raw[i], d = v - A.phi[i]; sp += A.Mf[i] * v; sn += v * v; sd += d * d; }
        const double prod_new = block_total(sp, sred), nsq = block_total(sn, sred), dsq = block_total(sd, sred);
        const double keff_new = keff * (prod_new / prod_old);
        const double dk = fabs(keff_new - keff);
        if (it >= 1) keff = keff_new;                               // :1774
        const double dphi = sqrt(dsq / nsq), norm = sqrt(nsq);
        if (tid == 0) { A.hist[it] = keff; A.hist[A.max_outer + it] = dk; A.hist[2 * A.max_outer + it] = dphi; }
        n_outer = it + 1;
        if (!isfinite(keff_new) || !isfinite(dphi)) { status = 2; break; }
        // normalise + Chebyshev (:1780-1788, src/solvers.cpp:720-756)
        int mode = 0; double ca = 0.0, cb = 0.0;
        if (it >= 2) {
            if (cheb_it == 15) cheb_it = 0;
            if (cheb_it == 0) mode = 1;
            else if (cheb_it == 1) { mode = 2; ca = A.ca1; }
            else { mode = 3; ca = A.a3[cheb_it]; cb = A.cb[cheb_it]; }
            ++cheb_it;
        }
        const bool do_norm = norm > 1e-14;
        for (long i = tid; i < NT; i += nt) {
            double v = A.raw[i];
            if (do_norm) v /= norm;
            if (mode == 1) pa[i] = v;
            else if (mode == 2) { const double a = pa[i]; v = a + ca * (v - a); pb[i] = v; }
            else if (mode == 3) { const double a = pa[i], b = pb[i]; v = b + ca * (v - b) + cb * (b - a); pa[i] = v; }
            A.phi[i] = v;
        }
        if (mode == 3) { double *t = pa; pa = pb; pb = t; }
        __syncthreads();
        if (dk < A.tol_keff && dphi < A.tol_flux) break;            // :1799-1802
    }
    if (tid == 0) { A.out->keff = keff; A.out->n_outer = n_outer; A.out->status = status; A.out->cg_total = cg_total; }
}

// ---------------------------------------------------------------------------------------------
// Explicit-S branch of the Schur solver (src/solvers.cpp:114-124, 259-509: direct solver types, a never-pushed solver type,
// n_phi < 200): S is formed column by column with the matrix-free apply (S e_j), inverted once per BuildMatrices, and a
// group solve is one dense matrix-vector product.  Column-major n x n.
__global__ void k_unit_vector(double *__restrict__ v, long n, long j)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) v[i] = i == j ? 1.0 : 0.0;
}
// one step of the in-place-style Gauss-Jordan inversion (S is SPD: no pivoting), written out of place so that every entry reads
// the step's pivot row and column as they were: Mo = step_k(M)
__global__ void k_gj_step(const double *__restrict__ M, double *__restrict__ Mo, int n, int k)
{
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)n * n) return;
    const int i = (int)(idx % n), j = (int)(idx / n);
    const double p = M[(long)k * n + k], rkj = M[(long)j * n + k], fik = M[(long)k * n + i];
    double out;
    if (i == k && j == k) out = 1.0 / p;
    else if (i == k) out = rkj / p;
    else if (j == k) out = -fik / p;
    else out = M[idx] - fik * (rkj / p);
    Mo[idx] = out;
}
// y = Sinv x: one thread per row, consecutive threads read consecutive rows of a column (coalesced)
__global__ void k_dense_matvec(const double *__restrict__ Sinv, const double *__restrict__ x, double *__restrict__ y, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += Sinv[(long)j * n + i] * x[j];
    y[i] = s;
}

// adjoint helpers: sum over groups, weighted dot product, scaling
__global__ void k_sum_groups(const double *__restrict__ a, double *__restrict__ out, long n, int ng)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) { double v = 0.0; for (int g = 0; g < ng; ++g) v += a[g * n + i]; out[i] = v; }
}
// partial sums of sum_g sum_i a[g][i] b[g][i] m[i]   (bi-orthonormalisation <phi, phi+>, :2020-2066)
__global__ __launch_bounds__(256) void k_dot3(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ m,
                                              long n, int ng, double *__restrict__ partials)
{
    __shared__ double sred[4];
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n * ng; i += gridDim.x * 256L) s += a[i] * b[i] * m[i % n];
    s = block_sum(s, sred);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ void k_scale(double *__restrict__ v, long n, double divisor)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) v[i] /= divisor;
}
__global__ void k_fill_const(double *__restrict__ v, long n, double c)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) v[i] = c;
}

// SolveCoarse: arithmetic volume-weighted block means (src/NeutFEM.cpp:2494-2556), one thread per coarse cell
__global__ void k_coarsen(const double *__restrict__ fine, double *__restrict__ coarse, const double *__restrict__ xb,
                          const double *__restrict__ yb, const double *__restrict__ zb, int dim, int nx, int ny, int nz,
                          int rx, int ry, int rz, int nfields)
{
    const int nxc = nx / rx, nyc = ny / ry, nzc = nz / rz;
    const long Nc = (long)nxc * nyc * nzc, Nf = (long)nx * ny * nz;
    const long ec = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (ec >= Nc) return;
    const int kx = (int)(ec % nxc), ky = (int)((ec / nxc) % nyc), kz = (int)(ec / ((long)nxc * nyc));
    for (int f = 0; f < nfields; ++f) {
        double vt = 0.0, s = 0.0;
        for (int sz = 0; sz < rz; ++sz) for (int sy = 0; sy < ry; ++sy) for (int sx = 0; sx < rx; ++sx) {
            const int ixf = kx * rx + sx, iyf = ky * ry + sy, izf = kz * rz + sz;
            double vol = xb[ixf + 1] - xb[ixf];
            if (dim >= 2) vol *= yb[iyf + 1] - yb[iyf];
            if (dim >= 3) vol *= zb[izf + 1] - zb[izf];
            vt += vol;
            s += vol * fine[f * Nf + ((long)izf * ny + iyf) * nx + ixf];
        }
        coarse[f * Nc + ec] = s / vt;
    }
}
// prolongation: piecewise-constant injection (src/NeutFEM.cpp:2585-2606)
// fine has gstride doubles per group (nloc*Nf, SoA): only the P0 moment is written, the rest is zeroed by the caller
__global__ void k_prolong(const double *__restrict__ coarse, double *__restrict__ fine, int nx, int ny, int nz, int rx,
                          int ry, int rz, int ng, long gstride)
{
    const long Nf = (long)nx * ny * nz;
    const int nxc = nx / rx, nyc = ny / ry; const long Nc = (long)nxc * nyc * (nz / rz);
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= Nf) return;
    const int ix = (int)(e % nx), iy = (int)((e / nx) % ny), iz = (int)(e / ((long)nx * ny));
    const long ec = ((long)(iz / rz) * nyc + iy / ry) * nxc + ix / rx;
    for (int g = 0; g < ng; ++g) fine[g * gstride + e] = coarse[g * Nc + ec];
}

// J reconstruction (src/solvers.cpp:227-228 full path: J = -A^-1 B^T phi ; src/NeutFEM.cpp:620-633 diagonal
// path: J_f = +(B^T phi)_f / A_ff).  One thread per (line, transverse mode), Thomas in place in the face array (run
// once, after convergence).  Face DOF (face, a) lives at face*nf + a; bubble (l, a) of cell e at e*ni + l + k*a behind the
// faces (src/FEM.cpp:264-334, 377-397).  jface / jbub point at this direction's face / bubble block; nfa = (k+1)^(dim-1).
__global__ void k_flux_to_J(Geom G, ModeArgs ma, int nb, int amode, int nfa, int ni, const double *__restrict__ D,
                            const double *__restrict__ L, const double *__restrict__ DR, const double *__restrict__ D0,
                            double *__restrict__ jface, double *__restrict__ jbub, long nlines, int diag)
{
    const long line = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (line >= nlines) return;
    const int d = ma.dir;
    int n, ix = 0, iy = 0, iz = 0; long base, sl, fbase, fsl;
    const long nxy = (long)G.nx * G.ny;
    if (d == 0) { n = G.nx; iy = (int)(line % G.ny); iz = (int)(line / G.ny); base = line * G.nx; sl = 1; fbase = line * (G.nx + 1); fsl = 1; }
    else if (d == 1) { n = G.ny; ix = (int)(line % G.nx); iz = (int)(line / G.nx); base = iz * nxy + ix; sl = G.nx;
                       fbase = (long)iz * (G.ny + 1) * G.nx + ix; fsl = G.nx; }
    else { n = G.nz; ix = (int)(line % G.nx); iy = (int)(line / G.nx); base = line; sl = nxy; fbase = line; fsl = nxy; }
    int *ci = d == 0 ? &ix : d == 1 ? &iy : &iz;
#define JF(f) jface[(fbase + (long)(f) * fsl) * nfa + amode]
    if (!diag) {
        // xL / xR of a cell (ModeArgs), forward sweep, scaling, backward sweep; faces hold z, then u, then -u
        auto xlr = [&](int c, double &xl, double &xr) {
            const long a = base + (long)c * sl;
            double pl = 0.0, pr = 0.0;
            for (int l = 0; l < nb; ++l) { const double gx = ma.Gc[l] * ma.x[l + 1][a]; pl += ma.eL[l] * gx; pr += ma.eR[l] * gx; }
            xl = ma.x[0][a] + pl; xr = ma.x[0][a] - pr;
        };
        double xl, xr; xlr(0, xl, xr);
        double z = -xl;
        JF(0) = z * D0[line];
        for (int c = 0; c < n; ++c) {
            double nxl = 0.0, nxr = 0.0;
            if (c + 1 < n) xlr(c + 1, nxl, nxr);
            z = (xr - nxl) - L[base + (long)c * sl] * z;
            JF(c + 1) = z * DR[base + (long)c * sl];
            xr = nxr;
        }
        double u = JF(n);
        for (int c = n - 1; c >= 0; --c) {
            const double uhi = u;
            u = JF(c) - L[base + (long)c * sl] * u;
            JF(c + 1) = -uhi;
            // bubbles of cell c: v_l = G_l x_{l+1} iM_l / c_e - (eL_l u_c + eR_l u_{c+1}); inactive bubbles (l >= nb) have t_b = 0
            if (ni > 0) {
                *ci = c;
                const long a = base + (long)c * sl;
                const double ic = D[a] / geom_factor(G, d, ix, iy, iz);
                for (int l = 0; l < G.k; ++l) {
                    const double tb = l < nb ? ma.Gc[l] * ma.x[l + 1][a] * ma.iM[l] * ic : 0.0;
                    const double v = tb - (ma.eL[l] * u + ma.eR[l] * uhi);
                    jbub[a * ni + l + G.k * amode] = -v;
                }
            }
        }
        JF(0) = -u;
    } else {
        double a2p = 0.0, a1, xp = 0.0, Dp = 0.0;
        for (int f = 0; f <= n; ++f) {
            double a2 = 0.0, xc = 0.0, Dc = 0.0;
            if (f < n) { *ci = f; Dc = D[base + (long)f * sl]; cell_a(G, d, ix, iy, iz, Dc, a2, a1); xc = ma.x[0][base + (long)f * sl]; }
            double Aff = a2p + a2;
            if (f == 0 && G.dir_lo[d]) { *ci = 0; Aff += dirichlet_term(G, d, ix, iy, iz, Dc); }
            if (f == n && G.dir_hi[d]) { *ci = n - 1; Aff += dirichlet_term(G, d, ix, iy, iz, Dp); }
            const double tt = xp - xc;                           // (B^T phi)_f / A_ff = T0 (..) / (T0 A_unit)
            JF(f) = fabs(G.T0 * Aff) < 1e-14 ? 0.0 : tt / Aff;
            a2p = a2; xp = xc; Dp = Dc;
        }
    }
#undef JF
}

// Diagonal-path currents of the z faces of a slab (src/NeutFEM.cpp:620-633: J_f = (B^T phi)_f / A_ff): the interface faces take the
// neighbour's edge-cell value and a2 (one plane each, exchanged by the caller); both slabs of an interface report the same number.
__global__ void k_diag_Jz_slab(Geom G, const double *__restrict__ D, const double *__restrict__ x, double *__restrict__ jz, long nlines,
                               int if_lo, int if_hi, const double *__restrict__ nb_x_lo, const double *__restrict__ nb_a2_lo,
                               const double *__restrict__ nb_x_hi, const double *__restrict__ nb_a2_hi)
{
    const long line = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (line >= nlines) return;
    const int ix = (int)(line % G.nx), iy = (int)(line / G.nx), n = G.nz;
    const long nxy = (long)G.nx * G.ny;
    double a2p = if_lo ? nb_a2_lo[line] : 0.0, xp = if_lo ? nb_x_lo[line] : 0.0, Dp = 0.0, a1;
    for (int f = 0; f <= n; ++f) {
        double a2 = 0.0, xc = 0.0, Dc = 0.0;
        if (f < n) { Dc = D[line + (long)f * nxy]; cell_a(G, 2, ix, iy, f, Dc, a2, a1); xc = x[line + (long)f * nxy]; }
        else if (if_hi) { a2 = nb_a2_hi[line]; xc = nb_x_hi[line]; }
        double Aff = a2p + a2;
        if (f == 0 && G.dir_lo[2]) Aff += dirichlet_term(G, 2, ix, iy, 0, Dc);
        if (f == n && G.dir_hi[2]) Aff += dirichlet_term(G, 2, ix, iy, n - 1, Dp);
        const double tt = xp - xc;
        jz[(long)f * nxy + line] = fabs(G.T0 * Aff) < 1e-14 ? 0.0 : tt / Aff;
        a2p = a2; xp = xc; Dp = Dc;
    }
}

// ---------------------------------------------------------------------------------------------
// CMFD acceleration (src/NeutFEM.cpp:662-1017).  A 7-point finite-volume operator per group on the cell grid,
//   M = diag(C_00 + sum_faces Deff A_face) - offdiag(Deff A_face),  Deff = D~ + D^,
// solved with Eigen's diagonally preconditioned CG (tolerance 1e-8 on |r|/|b|, at most 100 iterations, x0 = 0).  Face
// coefficients are stored per direction with the reference's face numbering (x: iz*ny*(nx+1) + iy*(nx+1) + ix, ...).
struct CmfdFaces { const double *Dt[3]; const double *Dh[3]; };
struct CmfdScalars { double absNew, alpha, beta, thr; int done, its; };

__device__ __forceinline__ double cmfd_face_area(const Geom &G, int d, int ix, int iy, int iz)
{
    return d == 0 ? G.hy[iy] * G.hz[iz] : d == 1 ? G.hx[ix] * G.hz[iz] : G.hx[ix] * G.hy[iy];
}
__device__ __forceinline__ long cmfd_face(const Geom &G, int d, int ix, int iy, int iz)
{
    if (d == 0) return ((long)iz * G.ny + iy) * (G.nx + 1) + ix;
    if (d == 1) return ((long)iz * (G.ny + 1) + iy) * G.nx + ix;
    return ((long)iz * G.ny + iy) * G.nx + ix;
}
// ComputeDtildeCoefficients (:723-821): harmonic mean inside, 2D/h on the boundary.  One thread per face of direction d.
__global__ void k_cmfd_dtilde(Geom G, int d, const double *__restrict__ D, double *__restrict__ Dt, long nfaces)
{
    const long f = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (f >= nfaces) return;
    const int n0 = d == 0 ? G.nx + 1 : G.nx, n1 = d == 1 ? G.ny + 1 : G.ny;
    int c[3] = { (int)(f % n0), (int)((f / n0) % n1), (int)(f / ((long)n0 * n1)) };
    const int nd = d == 0 ? G.nx : d == 1 ? G.ny : G.nz;
    const double *h = d == 0 ? G.hx : d == 1 ? G.hy : G.hz;
    const int cd = c[d];
    const long st = d == 0 ? 1 : d == 1 ? G.nx : (long)G.nx * G.ny;
    c[d] = cd == 0 ? 0 : cd - 1;                                  // lower (or only) neighbour
    const long eL = ((long)c[2] * G.ny + c[1]) * G.nx + c[0];
    double v;
    if (cd == 0 || cd == nd) v = 2.0 * D[eL] / h[c[d]];
    else { const double DL = D[eL], DR = D[eL + st], dL = h[cd - 1], dR = h[cd]; v = 2.0 * DL * DR / (DL * dR + DR * dL); }
    Dt[f] = v;
}
// UpdateDhatCoefficients (:823-869): x faces only; Jx holds the mode-0 face current at stride nfa.
__global__ void k_cmfd_dhat(Geom G, const double *__restrict__ phi, const double *__restrict__ Jx, int nfa,
                            const double *__restrict__ Dt, double *__restrict__ Dh, long nfaces)
{
    const long f = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (f >= nfaces) return;
    const int ix = (int)(f % (G.nx + 1)); const long line = f / (G.nx + 1);
    const long eR = line * G.nx + ix, eL = eR - 1;
    const double pd = ix == 0 ? -phi[eR] : ix == G.nx ? phi[eL] : phi[eL] - phi[eR];
    Dh[f] = fabs(pd) > 1e-14 ? Jx[f * nfa] / pd - Dt[f] : 0.0;
}
__device__ __forceinline__ void cmfd_cell(const Geom &G, long e, int &ix, int &iy, int &iz)
{
    ix = (int)(e % G.nx); iy = (int)((e / G.nx) % G.ny); iz = (int)(e / ((long)G.nx * G.ny));
}
// diag, rhs = chi tf / k, x = 0, r = rhs, p = z = r / diag ; partials: [0] |rhs|^2, [1] r.z
__global__ __launch_bounds__(256) void k_cmfd_setup(Geom G, CmfdFaces F, const double *__restrict__ C00, const double *__restrict__ chi,
                                                    const double *__restrict__ tf, double inv_k, double *__restrict__ diag,
                                                    double *__restrict__ x, double *__restrict__ r, double *__restrict__ pp,
                                                    long N, double *__restrict__ partials, long stride)
{
    __shared__ double sred[4];
    double s0 = 0.0, s1 = 0.0;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < N; e += gridDim.x * 256L) {
        int c[3]; cmfd_cell(G, e, c[0], c[1], c[2]);
        double dg = C00[e];
        for (int d = 0; d < G.dim; ++d) {
            const double A = cmfd_face_area(G, d, c[0], c[1], c[2]);
            const long fl = cmfd_face(G, d, c[0], c[1], c[2]);
            c[d] += 1; const long fu = cmfd_face(G, d, c[0], c[1], c[2]); c[d] -= 1;
            dg += ((F.Dt[d][fl] + F.Dh[d][fl]) + (F.Dt[d][fu] + F.Dh[d][fu])) * A;
        }
        const double b = chi[e] * tf[e] * inv_k;
        const double z = (dg != 0.0 ? 1.0 / dg : 1.0) * b;             // DiagonalPreconditioner stores 1/diag
        diag[e] = dg; x[e] = 0.0; r[e] = b; pp[e] = z;
        s0 += b * b; s1 += b * z;
    }
    s0 = block_sum(s0, sred); s1 = block_sum(s1, sred);
    if (threadIdx.x == 0) { partials[blockIdx.x] = s0; partials[stride + blockIdx.x] = s1; }
}
// q = M p ; partial p.q
// nb_lo / nb_hi (slabs): the neighbouring slab's edge plane of p across the interface below / above, else nullptr
__global__ __launch_bounds__(256) void k_cmfd_matvec(Geom G, CmfdFaces F, const double *__restrict__ diag, const double *__restrict__ pp,
                                                     double *__restrict__ q, long N, const CmfdScalars *__restrict__ sc,
                                                     double *__restrict__ partials, const double *__restrict__ nb_lo, const double *__restrict__ nb_hi)
{
    __shared__ double sred[4];
    if (sc->done) return;
    const long nxy = (long)G.nx * G.ny;
    double s0 = 0.0;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < N; e += gridDim.x * 256L) {
        int c[3]; cmfd_cell(G, e, c[0], c[1], c[2]);
        const double pe = pp[e];
        double s = diag[e] * pe;
        for (int d = 0; d < G.dim; ++d) {
            const int cd = c[d], nd = d == 0 ? G.nx : d == 1 ? G.ny : G.nz;
            const long st = d == 0 ? 1 : d == 1 ? G.nx : nxy;
            const double A = cmfd_face_area(G, d, c[0], c[1], c[2]);
            const long fl = cmfd_face(G, d, c[0], c[1], c[2]);
            c[d] += 1; const long fu = cmfd_face(G, d, c[0], c[1], c[2]); c[d] -= 1;
            if (cd > 0) s -= (F.Dt[d][fl] + F.Dh[d][fl]) * A * pp[e - st];
            else if (d == 2 && nb_lo) s -= (F.Dt[d][fl] + F.Dh[d][fl]) * A * nb_lo[e % nxy];
            if (cd < nd - 1) s -= (F.Dt[d][fu] + F.Dh[d][fu]) * A * pp[e + st];
            else if (d == 2 && nb_hi) s -= (F.Dt[d][fu] + F.Dh[d][fu]) * A * nb_hi[e % nxy];
        }
        q[e] = s; s0 += pe * s;
    }
    s0 = block_sum(s0, sred);
    if (threadIdx.x == 0) partials[blockIdx.x] = s0;
}
// x += alpha p ; r -= alpha q ; z = r / diag ; partials: [0] |r|^2, [1] r.z
__global__ __launch_bounds__(256) void k_cmfd_update(const double *__restrict__ diag, const double *__restrict__ pp, const double *__restrict__ q,
                                                     double *__restrict__ x, double *__restrict__ r, double *__restrict__ z, long N,
                                                     const CmfdScalars *__restrict__ sc, double *__restrict__ partials, long stride)
{
    __shared__ double sred[4];
    if (sc->done) return;
    const double alpha = sc->alpha;
    double s0 = 0.0, s1 = 0.0;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < N; e += gridDim.x * 256L) {
        x[e] += alpha * pp[e];
        const double re = r[e] - alpha * q[e];
        const double dg = diag[e], ze = (dg != 0.0 ? 1.0 / dg : 1.0) * re;
        r[e] = re; z[e] = ze;
        s0 += re * re; s1 += re * ze;
    }
    s0 = block_sum(s0, sred); s1 = block_sum(s1, sred);
    if (threadIdx.x == 0) { partials[blockIdx.x] = s0; partials[stride + blockIdx.x] = s1; }
}
__global__ void k_cmfd_pupdate(const double *__restrict__ z, double *__restrict__ pp, long N, const CmfdScalars *__restrict__ sc)
{
    if (sc->done) return;
    const double beta = sc->beta;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < N; e += gridDim.x * 256L) pp[e] = z[e] + beta * pp[e];
}
// Eigen's conjugate_gradient() scalar logic.  op 0: after setup (rhs2, r.z) ; op 1: after matvec (p.q) ; op 2: after update (|r|^2, r.z)
__device__ __forceinline__ void cmfd_logic(int op, const double *tot, CmfdScalars *sc)
{
    if (op == 0) {
        sc->its = 0; sc->alpha = sc->beta = 0.0;
        const double rhs2 = tot[0];
        sc->thr = fmax(1e-8 * 1e-8 * rhs2, 2.2250738585072014e-308);
        sc->absNew = tot[1];
        sc->done = (rhs2 == 0.0 || rhs2 < sc->thr) ? 1 : 0;
    } else if (op == 1) {
        sc->alpha = sc->absNew / tot[0];
    } else {
        sc->its += 1;
        if (tot[0] < sc->thr) { sc->done = 1; return; }
        sc->beta = tot[1] / sc->absNew;
        sc->absNew = tot[1];
        if (sc->its >= 100) sc->done = 1;
    }
}
// second reduction stage + the scalar logic.  One block of 256 threads.
__global__ __launch_bounds__(256) void k_cmfd_logic(int op, const double *__restrict__ partials, int cnt, long stride,
                                                    CmfdScalars *__restrict__ sc)
{
    __shared__ double sred[4];
    if (op != 0 && sc->done) return;
    double tot[2] = { 0.0, 0.0 };
    const int nq = op == 1 ? 1 : 2;
    for (int q = 0; q < nq; ++q) {
        double s = 0.0;
        for (int i = threadIdx.x; i < cnt; i += 256) s += partials[q * stride + i];
        tot[q] = block_sum(s, sred);
    }
    if (threadIdx.x != 0) return;
    cmfd_logic(op, tot, sc);
}
// slab teams: the totals come from k_finalize (reduce_only) + the all-reduce over ranks
__global__ void k_cmfd_logic_tot(int op, const double *__restrict__ red, CmfdScalars *__restrict__ sc)
{
    if (op != 0 && sc->done) return;
    double tot[2] = { red[0], op == 1 ? 0.0 : red[1] };
    cmfd_logic(op, tot, sc);
}
// D-tilde of the interface z faces of a slab (ComputeDtildeCoefficients interior formula, :760-790): the cell on the other side belongs
// to the neighbouring slab (its D and cell height arrive as planes).  Both slabs evaluate 2 DL DR / (DL dR + DR dL) with the same operands.
__global__ void k_cmfd_dtilde_iface(const double *__restrict__ D_own, const double *__restrict__ D_nb, const double *__restrict__ h_nb, double h_own,
                                    double *__restrict__ Dt_plane, long nlines, int own_is_upper)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= nlines) return;
    const double DL = own_is_upper ? D_nb[i] : D_own[i], DR = own_is_upper ? D_own[i] : D_nb[i];
    const double dL = own_is_upper ? h_nb[i] : h_own, dR = own_is_upper ? h_own : h_nb[i];
    Dt_plane[i] = 2.0 * DL * DR / (DL * dR + DR * dL);
}
// ratio clamp + relaxation (:994-1014); the correction multiplies every moment of the cell
__global__ void k_cmfd_correct(const double *__restrict__ x, double *__restrict__ phi, long N, int nloc, double omega)
{
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= N) return;
    const double pc = phi[e];
    double ratio = 1.0;
    if (fabs(pc) > 1e-14) { ratio = x[e] / pc; ratio = fmax(0.5, fmin(2.0, ratio)); }
    const double corr = omega * ratio + (1.0 - omega) * 1.0;
    for (int l = 0; l < nloc; ++l) phi[l * N + e] *= corr;
}

// streaming copy (HBM microbenchmark for the roofline's "of measured copy" figure): 16 bytes per lane and access, U independent loads in
// flight per lane before the first store, every block walks contiguous chunks of U x 4 KiB (a wave instruction covers 1 KiB); the grid is a
// multiple of the CU count.  NT = non-temporal loads and stores as whole 16-byte accesses (round 3's NT variant split them into two 8-byte
// halves and never got past 4.8 TB/s).
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_copy(const nf_d2 *__restrict__ src, nf_d2 *__restrict__ dst, long n2)
{
    const long chunk = 256L * U;
    for (long base = (long)blockIdx.x * chunk; base < n2; base += (long)gridDim.x * chunk) {
        nf_d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256L + threadIdx.x;
            if (i < n2) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256L + threadIdx.x;
            if (i < n2) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
        }
    }
}

// input validation on the device (nf_upload_xs): bit `bit` of *flags is raised if the array holds a non-finite value, or
// (need_nonzero) an exact zero -- 1/D of such a cell would put inf into every line system it touches
__global__ void k_check_xs(const double *__restrict__ v, long n, int need_nonzero, int bit, int *__restrict__ flags)
{
    bool bad = false;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const double x = v[i];
        bad |= !isfinite(x) || (need_nonzero && x == 0.0);
    }
    if (bad) atomicOr(flags, 1 << bit);
}

// fill with a deterministic pseudo-random pattern (profiling helper)
__global__ void k_fill_pattern(double *__restrict__ v, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        unsigned long long s = (unsigned long long)i * 6364136223846793005ULL + 1442695040888963407ULL;
        s ^= s >> 29;
        v[i] = (double)(s & 0xFFFFFF) / 16777216.0 - 0.5;
    }
}

}  // namespace nf
