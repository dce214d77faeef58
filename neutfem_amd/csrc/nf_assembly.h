// nf_assembly.h -- literal element-matrix assembly on the device: LocalMatrices::Compute (src/FEM.cpp:748-953), one element per
// workgroup, quadrature points and basis values staged in LDS.
//
// The hot path never forms local matrices: on a Cartesian mesh A_loc, B_loc, C_loc are closed forms folded into a handful of scalars
// per mode.  This kernel is the literal counterpart -- the reference's tensor quadrature, basis function by basis
// function -- kept (i) so that the closed forms can be checked against the quadrature ON THE DEVICE (tests/test_gpu_assembly.py, also
// against the oracle's literal nfo_local_matrices), and (ii) because it is the one place of this code base where the work is a dense
// contraction: A_d = Psi_d^T W Psi_d over up to 125 quadrature points x 36 functions per direction, B = Phi^T W divPsi, C = Phi^T W Phi.
// It exists in two variants, plain fp64 FMA and v_mfma_f64_16x16x4_f64, so that SURVEY section 7-6 ("measure whether MFMA beats plain
// FMA for RT1+/P1+, and say so") is answered with a number (profiles/r03_i_assembly_mfma.txt).
//
// LDS (doubles): 1-D tables (<= 6 points x {P_0..2, P'_0..2, (1-x)/2, (1+x)/2, bubbles, bubble derivatives, weights}); the basis
// values of ONE direction at all tensor points Psi[Q][a], their divergences Dv[Q][a] and the flux basis Phi[Q][p], rows padded to
// multiples of 16 functions and 4 points (zeros) so that the MFMA tiles need no edge handling: 128 x 48 x 2 + 128 x 32 = 16384 doubles.
#pragma once
#include <hip/hip_runtime.h>

namespace nf {

struct AsmArgs {
    int dim, k, m;                 // mesh dimension, RT order, P order (src/NeutFEM.cpp:149-169)
    int nq;                        // 1-D Gauss points: order 2 max(k, m) + 3, with the reference's fallback (7 -> 5 points, include/FEM.hpp:115-120)
    double qp[6], qw[6];
    int nx, ny, nz;
    const double *hx, *hy, *hz;    // cell widths (device)
    const double *D, *Sig;         // the group's diffusion coefficient and removal cross-section per cell (device)
    const int *elems; int n_elems; // elements to assemble
    double *A, *B, *C;             // outputs, element-major: nJ x nJ, nP x nJ, nP x nP (row-major, like LocalMatrices::GetA/B/C)
};

__device__ __forceinline__ double asm_legP(int n, double x) { return n == 0 ? 1.0 : (n == 1 ? x : 0.5 * (3.0 * x * x - 1.0)); }
__device__ __forceinline__ double asm_legdP(int n, double x) { return n == 0 ? 0.0 : (n == 1 ? 1.0 : 3.0 * x); }

typedef double asm_d4 __attribute__((ext_vector_type(4)));

// One 16 x 16 tile of  sum_Q  Lt[Q][i0 + i] * w[Q] * Rt[Q][j0 + j]  over Q in [0, nQ4) (nQ4 a multiple of 4, rows beyond the real
// points are zero), by one wavefront.  Lt / Rt: LDS tables with row pitches pl / pr.  Result: acc[r] = element (row (lane >> 4) + 4 r,
// column lane & 15) (the f64 MFMA's own C/D map, cdna_hip_programming.md "f64 MFMA does NOT use these maps").
__device__ __forceinline__ asm_d4 asm_tile_mfma(const double *Lt, int pl, int i0, const double *Rt, int pr, int j0, const double *wq, int nQ4, int lane)
{
    asm_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
    const int r = lane & 15, kk = lane >> 4;
    for (int Q0 = 0; Q0 < nQ4; Q0 += 4) {
        const int Q = Q0 + kk;
        const double a = Lt[Q * pl + i0 + r] * wq[Q];           // A operand: A[i = lane & 15][k = lane >> 4]
        const double b = Rt[Q * pr + j0 + r];                    // B operand: B[k = lane >> 4][j = lane & 15]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}

template <bool MFMA>
__global__ __launch_bounds__(256) void k_local_matrices(AsmArgs P)
{
    extern __shared__ double sm[];
    const int dim = P.dim, k = P.k, m = P.m, nq = P.nq;
    int nf = 1, ni = k; for (int t = 1; t < dim; ++t) { nf *= k + 1; ni *= k + 1; }
    const int nper = 2 * nf + ni, nJ = dim * nper, n1 = m + 1;
    int nP = 1; for (int t = 0; t < dim; ++t) nP *= n1;
    const int nyl = dim >= 2 ? nq : 1, nzl = dim == 3 ? nq : 1, nQ = nq * nyl * nzl, nQ4 = (nQ + 3) & ~3;
    const int PA = 48, PP = 32;                                  // row pitches of the staged tables (functions padded to multiples of 16)
    // ---- LDS carve-up
    double *t_P = sm;                  // [3][6]  P_n(x_q)
    double *t_fL = t_P + 18, *t_fR = t_fL + 6;                   // (1 -+ x) / 2
    double *t_b = t_fR + 6, *t_db = t_b + 12;                    // [2][6] bubbles (1 - x^2) P_l and their derivatives
    double *wq = t_db + 12;                                      // [128] tensor weights w_base
    double *Psi = wq + 128, *Dv = Psi + 128 * PA, *Phi = Dv + 128 * PA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = P.elems[blockIdx.x];
    const int nxy = P.nx * P.ny;
    const int iz = e / nxy, iy = (e % nxy) / P.nx, ix = e % P.nx;
    // geometric factors of LocalMatrices::Compute (src/FEM.cpp:795-813; the 2D pair is inverted w.r.t. 1D / 3D -- kept)
    const double hx = P.hx[ix], hy = dim >= 2 ? P.hy[iy] : 1.0, hz = dim == 3 ? P.hz[iz] : 1.0;
    double fac[3] = { 0.0, 0.0, 0.0 }, detJ;
    if (dim == 1) { detJ = hx / 2.0; fac[0] = hx / 2.0; }
    else if (dim == 2) { detJ = (hx / 2.0) * (hy / 2.0); fac[0] = hy / hx; fac[1] = hx / hy; }
    else { detJ = (hx / 2.0) * (hy / 2.0) * (hz / 2.0); fac[0] = 2.0 * hx / (hy * hz); fac[1] = 2.0 * hy / (hx * hz); fac[2] = 2.0 * hz / (hx * hy); }
    const double invD = 1.0 / P.D[e], Sigma = P.Sig[e];
    // ---- 1-D tables (include/FEM.hpp:139-201, src/FEM.cpp:403-453)
    if (tid < nq) {
        const double x = P.qp[tid];
        for (int n = 0; n < 3; ++n) t_P[n * 6 + tid] = asm_legP(n, x);
        t_fL[tid] = 0.5 * (1.0 - x); t_fR[tid] = 0.5 * (1.0 + x);
        for (int l = 0; l < 2; ++l) {
            t_b[l * 6 + tid] = (1.0 - x * x) * asm_legP(l, x);
            t_db[l * 6 + tid] = -2.0 * x * asm_legP(l, x) + (1.0 - x * x) * asm_legdP(l, x);
        }
    }
    for (int Q = tid; Q < 128; Q += 256) {
        double w = 0.0;
        if (Q < nQ) { const int qz = Q % nzl, qy = (Q / nzl) % nyl, qx = Q / (nzl * nyl); w = P.qw[qx] * (dim >= 2 ? P.qw[qy] : 1.0) * (dim == 3 ? P.qw[qz] : 1.0); }
        wq[Q] = w;
    }
    __syncthreads();
    // ---- the flux basis at every tensor point (src/FEM.cpp:626-671): Phi[Q][p] = P_i(xi) P_j(eta) P_k(zeta), p = i + n j + n^2 k
    for (int idx = tid; idx < 128 * PP; idx += 256) {
        const int Q = idx / PP, p = idx % PP;
        double v = 0.0;
        if (Q < nQ && p < nP) {
            const int qz = Q % nzl, qy = (Q / nzl) % nyl, qx = Q / (nzl * nyl);
            const int i = p % n1, j = (p / n1) % n1, kk = p / (n1 * n1);
            v = t_P[i * 6 + qx];
            if (dim >= 2) v *= t_P[j * 6 + qy];
            if (dim == 3) v *= t_P[kk * 6 + qz];
        }
        Phi[idx] = v;
    }
    double *Ae = P.A + (size_t)blockIdx.x * nJ * nJ, *Be = P.B + (size_t)blockIdx.x * nP * nJ, *Ce = P.C + (size_t)blockIdx.x * nP * nP;
    for (int idx = tid; idx < nJ * nJ; idx += 256) Ae[idx] = 0.0;    // cross-direction blocks are exactly zero (src/FEM.cpp:891-924)
    for (int d = 0; d < dim; ++d) {
        __syncthreads();
        // ---- RT basis of direction d and its reference divergence at every tensor point (src/FEM.cpp:377-453):
        // local order [lower faces nf | upper faces nf | bubbles ni]; x faces carry (eta, zeta), y faces (xi, zeta), z faces (xi, eta)
        for (int idx = tid; idx < 128 * PA; idx += 256) {
            const int Q = idx / PA, a = idx % PA;
            double v = 0.0, dv = 0.0;
            if (Q < nQ && a < nper) {
                const int qz = Q % nzl, qy = (Q / nzl) % nyl, qx = Q / (nzl * nyl);
                const int qs = d == 0 ? qx : (d == 1 ? qy : qz);
                const int q1 = d == 0 ? qy : qx, q2 = d == 2 ? qy : qz;
                int i = 0, j = 0, l = 0; bool bubble = false, upper = false;
                if (a < 2 * nf) {
                    const int f = a < nf ? a : a - nf; upper = a >= nf;
                    if (dim == 2) i = f; else if (dim == 3) { i = f % (k + 1); j = f / (k + 1); }
                } else {
                    bubble = true; const int b = a - 2 * nf;
                    if (dim == 1) l = b; else if (dim == 2) { l = b % k; i = b / k; } else { const int tr = b / k; l = b % k; i = tr % (k + 1); j = tr / (k + 1); }
                }
                double Pt = 1.0;
                if (dim >= 2) Pt *= t_P[i * 6 + q1];
                if (dim == 3) Pt *= t_P[j * 6 + q2];
                if (bubble) { v = t_b[l * 6 + qs] * Pt; dv = t_db[l * 6 + qs] * Pt; }
                else { v = (upper ? t_fR[qs] : t_fL[qs]) * Pt; dv = (upper ? 0.5 : -0.5) * Pt; }
            }
            Psi[idx] = v; Dv[idx] = dv;
        }
        __syncthreads();
        const double sA = invD * fac[d];
        if (MFMA) {
            // A_d: 3 x 3 tiles of 16 x 16 over nper <= 36 functions; B_d: 2 x 3 tiles (nP <= 27 rows); round-robin over the 4 wavefronts
            const int ta = (nper + 15) / 16, tp = (nP + 15) / 16;
            for (int t = wave; t < ta * ta + tp * ta; t += 4) {
                const bool isA = t < ta * ta;
                const int ti = isA ? t / ta : (t - ta * ta) / ta, tj = isA ? t % ta : (t - ta * ta) % ta;
                const asm_d4 acc = isA ? asm_tile_mfma(Psi, PA, ti * 16, Psi, PA, tj * 16, wq, nQ4, lane)
                                       : asm_tile_mfma(Phi, PP, ti * 16, Dv, PA, tj * 16, wq, nQ4, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ti * 16 + (lane >> 4) + 4 * r, col = tj * 16 + (lane & 15);
                    if (isA) { if (row < nper && col < nper) Ae[(size_t)(d * nper + row) * nJ + d * nper + col] = sA * acc[r]; }
                    else if (row < nP && col < nper) Be[(size_t)row * nJ + d * nper + col] = acc[r];
                }
            }
        } else {
            for (int idx = tid; idx < nper * nper; idx += 256) {     // A_loc[i][j] = (1/D) sum_q w psi_i psi_j factor_d (src/FEM.cpp:891-924)
                const int a = idx / nper, b = idx % nper;
                double s = 0.0;
                for (int Q = 0; Q < nQ; ++Q) s += Psi[Q * PA + a] * Psi[Q * PA + b] * wq[Q];
                Ae[(size_t)(d * nper + a) * nJ + d * nper + b] = sA * s;
            }
            for (int idx = tid; idx < nP * nper; idx += 256) {       // B_loc[p][j] = sum_q w phi_p div psi_j, no Jacobian (src/FEM.cpp:930-936)
                const int p = idx / nper, b = idx % nper;
                double s = 0.0;
                for (int Q = 0; Q < nQ; ++Q) s += Phi[Q * PP + p] * Dv[Q * PA + b] * wq[Q];
                Be[(size_t)p * nJ + d * nper + b] = s;
            }
        }
    }
    // ---- C_loc[p][r] = Sigma sum_q w detJ phi_p phi_r (src/FEM.cpp:941-949)
    const double sC = Sigma * detJ;
    if (MFMA) {
        const int tp = (nP + 15) / 16;
        for (int t = wave; t < tp * tp; t += 4) {
            const int ti = t / tp, tj = t % tp;
            const asm_d4 acc = asm_tile_mfma(Phi, PP, ti * 16, Phi, PP, tj * 16, wq, nQ4, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + (lane >> 4) + 4 * r, col = tj * 16 + (lane & 15);
                if (row < nP && col < nP) Ce[(size_t)row * nP + col] = sC * acc[r];
            }
        }
    } else {
        for (int idx = tid; idx < nP * nP; idx += 256) {
            const int p = idx / nP, r = idx % nP;
            double s = 0.0;
            for (int Q = 0; Q < nQ; ++Q) s += Phi[Q * PP + p] * Phi[Q * PP + r] * wq[Q];
            Ce[idx] = sC * s;
        }
    }
}

}  // namespace nf
