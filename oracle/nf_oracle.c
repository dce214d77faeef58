/*
 * nf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See nf_oracle.h.
 *
 * Single-threaded plain-C restatement of the reference hot path.  The global
 * matrix A is never stored as a general sparse matrix: RT face DOFs are shared
 * along grid lines only and A decouples by direction and transverse Legendre
 * mode, so in "chain order" (face, k bubbles, face, k bubbles, ...) it is
 * block-diagonal with half-bandwidth k+1.  The oracle assembles A from the
 * reference's local matrices into that banded storage (checking that nothing
 * falls outside the band) and replaces Eigen::SparseLU (src/solvers.cpp:163)
 * by an exact banded LDL^T -- mathematically the same A^-1.
 *
 * Deliberate deviations from the reference (documented in DESIGN.md):
 *  - unit tables (A-hat, B-hat, C-hat) are integrated once by the reference's
 *    quadrature and scaled per element instead of re-running
 *    the quadrature per element; nfo_local_matrices() is the literal version
 *    and tests compare both;
 *  - quadrature round-off entries of the unit tables (< 1e-13 relative) are
 *    zeroed, the reference keeps them when they exceed 1e-14 absolute;
 *  - n_phi < 200 or a DIRECT_* solver type: the reference forms S explicitly
 *    and calls Eigen solvers (src/solvers.cpp:259-509); here CG is run to
 *    1e-14 as the stand-in for the exact solve.
 */
#include "nf_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXQ 6
#define MAXHIST 4096

struct nfo {
    int dim, nx, ny, nz, ne;
    double *xb, *yb, *zb, *hx, *hy, *hz;
    int nxb, nyb, nzb;
    int k, m, ng;
    int nf, ni, nloc, nJloc, nper;          /* nper = 2nf+ni local J dofs per direction */
    long nJx, nJy, nJz, nJface, nJint, nJ, nPhi;
    int nq; double qp[MAXQ], qw[MAXQ];
    /* XS + solution (host) */
    double *D, *SigR, *NSF, *KSF, *Chi, *SRC, *SigS, *phi, *J, *phi_adj;
    double last_keff_adj; int has_valid_adjoint;
    /* CMFD (include/NeutFEM.hpp:119-143) */
    int cmfd_init; double cmfd_relax; double *Dt[3], *Dh[3];   /* per direction: ng * n_faces_dir */
    long nfc[3];
    int bc_set[8], bc_type[8]; double bc_val[8];
    double tol_keff, tol_flux, tol_L2; int max_outer, max_inner;
    int solver_type, solver_type_pushed;    /* NeutFEM.cpp:126 vs solvers.cpp:68 */
    double schur_tol; int schur_maxit;
    double last_keff; int has_valid_keff;
    int refactor_each;
    /* unit tables */
    double *Ahat[3];                        /* nper x nper per direction */
    double *Bhat;                           /* nloc x nJloc */
    double *Chat;                           /* nloc x nloc */
    /* built operators */
    int built;
    int bw;                                 /* half bandwidth = k+1 */
    long chain_base[3], chain_len[3];
    int *eJ;                                /* ne*nJloc : chain position of each local J dof */
    double **band;                          /* per group: nJ*(bw+1) factored band (L and D) */
    double **Aband;                         /* per group: unfactored copy (for refactor / diag) */
    double **Cd;                            /* per group: nPhi diagonal of C */
    double **Mf;                            /* per group: nPhi diagonal of M_fiss */
    double **Mchi;                          /* per group: nPhi diagonal of the chi-weighted mass matrix (adjoint) */
    double **Ms;                            /* ng*ng  : nPhi diagonal of M_scatter[g_to*ng+g_from] or NULL if empty */
    double **Sinv;                          /* diag cache per group */
    int diag_valid;
    double **Sfac; int **Spiv;              /* explicit Schur complement of a group, LU-factored (direct branch), or NULL */
    /* work */
    double *wt, *wu;                        /* nJ */
    /* stats */
    int last_outer, coarse_outer; long last_cg_total;
    double hist_k[MAXHIST], hist_dk[MAXHIST], hist_dphi[MAXHIST];
    double *hist_cg; int last_cg_its; double last_cg_res;
    /* NOT in the reference (nfo_set_void, see nf_oracle.h): cells cut out of the domain, with a Robin condition on their faces */
    unsigned char *voidm; double void_inv_alpha;
};

/* ---- include/FEM.hpp:82-123 ------------------------------------------------ */
static int gauss(int order, double *p, double *w)
{
    switch (order) {
    case 1: p[0] = 0.0; w[0] = 2.0; return 1;
    case 2: p[0] = -1.0 / sqrt(3.0); p[1] = 1.0 / sqrt(3.0); w[0] = w[1] = 1.0; return 2;
    case 3: p[0] = -sqrt(0.6); p[1] = 0.0; p[2] = sqrt(0.6);
            w[0] = 5.0 / 9.0; w[1] = 8.0 / 9.0; w[2] = 5.0 / 9.0; return 3;
    case 4: p[0] = -0.861136311594053; p[1] = -0.339981043584856; p[2] = 0.339981043584856; p[3] = 0.861136311594053;
            w[0] = 0.347854845137454; w[1] = 0.652145154862546; w[2] = 0.652145154862546; w[3] = 0.347854845137454; return 4;
    case 6: p[0] = -0.932469514203152; p[1] = -0.661209386466265; p[2] = -0.238619186083197;
            p[3] = 0.238619186083197; p[4] = 0.661209386466265; p[5] = 0.932469514203152;
            w[0] = 0.171324492379170; w[1] = 0.360761573048139; w[2] = 0.467913934572691;
            w[3] = 0.467913934572691; w[4] = 0.360761573048139; w[5] = 0.171324492379170; return 6;
    case 5:
    default: /* any other order silently becomes the 5-point rule (FEM.hpp:115-120) */
            p[0] = -0.906179845938664; p[1] = -0.538469310105683; p[2] = 0.0; p[3] = 0.538469310105683; p[4] = 0.906179845938664;
            w[0] = 0.236926885056189; w[1] = 0.478628670499366; w[2] = 0.568888888888889; w[3] = 0.478628670499366; w[4] = 0.236926885056189;
            return 5;
    }
}

/* ---- include/FEM.hpp:151-186 ----------------------------------------------- */
static double legP(int n, double xi)
{
    if (n == 0) return 1.0;
    if (n == 1) return xi;
    double a = 1.0, b = xi, c = 0.0;
    for (int k = 2; k <= n; ++k) { c = ((2 * k - 1) * xi * b - (k - 1) * a) / k; a = b; b = c; }
    return c;
}
static double legdP(int n, double xi)
{
    if (n == 0) return 0.0;
    if (n == 1) return 1.0;
    double den = xi * xi - 1.0;
    if (fabs(den) < 1e-14) {
        double s = (xi > 0) ? 1.0 : ((n % 2 == 0) ? 1.0 : -1.0);
        return s * n * (n + 1) / 2.0;
    }
    return n * (xi * legP(n, xi) - legP(n - 1, xi)) / den;
}

/* ---- RT / Pk basis at one point, src/FEM.cpp:377-671 ------------------------
 * Fills Jv[nJloc], dv[nJloc] in the local order [dir][lower nf | upper nf | bubbles ni]
 * (src/FEM.cpp:729-745) and pv[nloc]. */
static void basis_at(const nfo_t *h, double xi, double eta, double zeta, double *Jv, double *dv, double *pv)
{
    const int k = h->k, dim = h->dim, nf = h->nf, ni = h->ni, nper = h->nper;
    const double c[3] = { xi, eta, zeta };
    for (int d = 0; d < dim; ++d) {
        /* transverse coordinates: x-faces (eta,zeta), y-faces (xi,zeta), z-faces (xi,eta) (FEM.cpp:416-453) */
        double t1, t2;
        if (d == 0) { t1 = eta; t2 = zeta; } else if (d == 1) { t1 = xi; t2 = zeta; } else { t1 = xi; t2 = eta; }
        const double s = c[d];
        for (int f = 0; f < nf; ++f) {
            int i, j;
            if (dim == 1) { i = 0; j = 0; } else if (dim == 2) { i = f; j = 0; } else { i = f % (k + 1); j = f / (k + 1); }
            double Pt = 1.0;
            if (dim >= 2) Pt *= legP(i, t1);
            if (dim == 3) Pt *= legP(j, t2);
            Jv[d * nper + f] = 0.5 * (1.0 - s) * Pt;       dv[d * nper + f] = -0.5 * Pt;
            Jv[d * nper + nf + f] = 0.5 * (1.0 + s) * Pt;  dv[d * nper + nf + f] = 0.5 * Pt;
        }
        for (int b = 0; b < ni; ++b) {
            int l, i, j;
            if (dim == 1) { l = b; i = 0; j = 0; }
            else if (dim == 2) { l = b % k; i = b / k; j = 0; }
            else { int tr = b / k; l = b % k; i = tr % (k + 1); j = tr / (k + 1); }
            double Pt = 1.0;
            if (dim >= 2) Pt *= legP(i, t1);
            if (dim == 3) Pt *= legP(j, t2);
            double bub = 1.0 - s * s, Pl = legP(l, s), dPl = legdP(l, s);
            Jv[d * nper + 2 * nf + b] = bub * Pl * Pt;
            dv[d * nper + 2 * nf + b] = (-2.0 * s * Pl + bub * dPl) * Pt;
        }
    }
    const int n = h->m + 1;
    for (int p = 0; p < h->nloc; ++p) {
        int i, j, kk;
        if (dim == 1) { i = p; j = 0; kk = 0; }
        else if (dim == 2) { i = p % n; j = p / n; kk = 0; }
        else { i = p % n; j = (p / n) % n; kk = p / (n * n); }
        double v = legP(i, xi);
        if (dim >= 2) v *= legP(j, eta);
        if (dim == 3) v *= legP(kk, zeta);
        pv[p] = v;
    }
}

static void elem_coords(const nfo_t *h, int e, int *ix, int *iy, int *iz)
{
    *iz = e / (h->nx * h->ny); int r = e % (h->nx * h->ny); *iy = r / h->nx; *ix = r % h->nx;
}

/* geometric factors of LocalMatrices::Compute, src/FEM.cpp:795-813 */
static void geom_factors(const nfo_t *h, int ix, int iy, int iz, double fac[3], double *detJ)
{
    double hx = h->hx[ix], hy = h->hy[iy], hz = h->hz[iz];
    double jx = hx / 2.0, jy = hy / 2.0, jz = hz / 2.0;
    fac[0] = fac[1] = fac[2] = 0.0;
    if (h->dim == 1) { *detJ = jx; fac[0] = hx / 2.0; }
    else if (h->dim == 2) { *detJ = jx * jy; fac[0] = hy / hx; fac[1] = hx / hy; }   /* 2D quirk kept */
    else { *detJ = jx * jy * jz; fac[0] = 2.0 * hx / (hy * hz); fac[1] = 2.0 * hy / (hx * hz); fac[2] = 2.0 * hz / (hx * hy); }
}

/* ---- literal LocalMatrices::Compute, src/FEM.cpp:748-953 -------------------- */
void nfo_local_matrices(const nfo_t *h, int e, double D, double Sigma, double *A, double *B, double *C)
{
    const int nJ = h->nJloc, nP = h->nloc, nper = h->nper, nq = h->nq, dim = h->dim;
    memset(A, 0, sizeof(double) * nJ * nJ);
    memset(B, 0, sizeof(double) * nP * nJ);
    memset(C, 0, sizeof(double) * nP * nP);
    int ix, iy, iz; elem_coords(h, e, &ix, &iy, &iz);
    double fac[3], detJ; geom_factors(h, ix, iy, iz, fac, &detJ);
    const double invD = 1.0 / D;
    double *Jv = (double *)malloc(sizeof(double) * (2 * nJ + nP)), *dv = Jv + nJ, *pv = dv + nJ;
    const int nyl = dim >= 2 ? nq : 1, nzl = dim == 3 ? nq : 1;
    for (int qx = 0; qx < nq; ++qx)
        for (int qy = 0; qy < nyl; ++qy)
            for (int qz = 0; qz < nzl; ++qz) {
                double xi = h->qp[qx], wx = h->qw[qx];
                double eta = dim >= 2 ? h->qp[qy] : 0.0, wy = dim >= 2 ? h->qw[qy] : 1.0;
                double zeta = dim == 3 ? h->qp[qz] : 0.0, wz = dim == 3 ? h->qw[qz] : 1.0;
                double w_base = wx * wy * wz, weight = w_base * detJ;
                basis_at(h, xi, eta, zeta, Jv, dv, pv);
                for (int d = 0; d < dim; ++d)
                    for (int i = d * nper; i < (d + 1) * nper; ++i)
                        for (int j = d * nper; j <= i; ++j) {
                            double c = invD * Jv[i] * Jv[j] * w_base * fac[d];
                            A[i * nJ + j] += c;
                            if (i != j) A[j * nJ + i] += c;
                        }
                for (int p = 0; p < nP; ++p)
                    for (int j = 0; j < nJ; ++j) B[p * nJ + j] += pv[p] * dv[j] * w_base;
                for (int i = 0; i < nP; ++i)
                    for (int j = 0; j <= i; ++j) {
                        double c = Sigma * pv[i] * pv[j] * weight;
                        C[i * nP + j] += c;
                        if (i != j) C[j * nP + i] += c;
                    }
            }
    free(Jv);
}

/* unit tables: same quadrature loops with invD = factor = detJ = Sigma = 1 */
static void unit_tables(nfo_t *h)
{
    const int nJ = h->nJloc, nP = h->nloc, nper = h->nper, nq = h->nq, dim = h->dim;
    for (int d = 0; d < 3; ++d) h->Ahat[d] = (double *)calloc((size_t)nper * nper, sizeof(double));
    h->Bhat = (double *)calloc((size_t)nP * nJ, sizeof(double));
    h->Chat = (double *)calloc((size_t)nP * nP, sizeof(double));
    double *Jv = (double *)malloc(sizeof(double) * (2 * nJ + nP)), *dv = Jv + nJ, *pv = dv + nJ;
    const int nyl = dim >= 2 ? nq : 1, nzl = dim == 3 ? nq : 1;
    for (int qx = 0; qx < nq; ++qx)
        for (int qy = 0; qy < nyl; ++qy)
            for (int qz = 0; qz < nzl; ++qz) {
                double xi = h->qp[qx], wx = h->qw[qx];
                double eta = dim >= 2 ? h->qp[qy] : 0.0, wy = dim >= 2 ? h->qw[qy] : 1.0;
                double zeta = dim == 3 ? h->qp[qz] : 0.0, wz = dim == 3 ? h->qw[qz] : 1.0;
                double w = wx * wy * wz;
                basis_at(h, xi, eta, zeta, Jv, dv, pv);
                for (int d = 0; d < dim; ++d)
                    for (int i = 0; i < nper; ++i)
                        for (int j = 0; j < nper; ++j)
                            h->Ahat[d][i * nper + j] += Jv[d * nper + i] * Jv[d * nper + j] * w;
                for (int p = 0; p < nP; ++p)
                    for (int j = 0; j < nJ; ++j) h->Bhat[p * nJ + j] += pv[p] * dv[j] * w;
                for (int i = 0; i < nP; ++i)
                    for (int j = 0; j < nP; ++j) h->Chat[i * nP + j] += pv[i] * pv[j] * w;
            }
    free(Jv);
    /* zero quadrature round-off (documented deviation) */
    for (int d = 0; d < dim; ++d) {
        double mx = 0; for (int i = 0; i < nper * nper; ++i) mx = fmax(mx, fabs(h->Ahat[d][i]));
        for (int i = 0; i < nper * nper; ++i) if (fabs(h->Ahat[d][i]) < 1e-13 * mx) h->Ahat[d][i] = 0.0;
    }
    { double mx = 0; for (int i = 0; i < nP * nJ; ++i) mx = fmax(mx, fabs(h->Bhat[i]));
      for (int i = 0; i < nP * nJ; ++i) if (fabs(h->Bhat[i]) < 1e-13 * mx) h->Bhat[i] = 0.0; }
    { double mx = 0; for (int i = 0; i < nP * nP; ++i) mx = fmax(mx, fabs(h->Chat[i]));
      for (int i = 0; i < nP * nP; ++i) if (fabs(h->Chat[i]) < 1e-13 * mx) h->Chat[i] = 0.0; }
}

/* ---- DOF numbering, src/FEM.cpp:264-334 ------------------------------------- */
static long JxFace(const nfo_t *h, int ix, int iy, int iz, int l)
{
    long f;
    if (h->dim == 1) f = ix; else if (h->dim == 2) f = (long)iy * (h->nx + 1) + ix;
    else f = (long)iz * h->ny * (h->nx + 1) + (long)iy * (h->nx + 1) + ix;
    return f * h->nf + l;
}
static long JyFace(const nfo_t *h, int ix, int iy, int iz, int l)
{
    long f;
    if (h->dim == 2) f = (long)iy * h->nx + ix; else f = (long)iz * (h->ny + 1) * h->nx + (long)iy * h->nx + ix;
    return h->nJx + f * h->nf + l;
}
static long JzFace(const nfo_t *h, int ix, int iy, int iz, int l)
{
    long f = (long)iz * h->ny * h->nx + (long)iy * h->nx + ix;
    return h->nJx + h->nJy + f * h->nf + l;
}
static long JInt(const nfo_t *h, int d, long e, int b)
{
    return h->nJface + (long)d * h->ne * h->ni + e * h->ni + b;
}

/* src/FEM.cpp:955-999 */
static void global_J(const nfo_t *h, int ix, int iy, int iz, long *idx)
{
    const int nf = h->nf, ni = h->ni; int n = 0;
    long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
    for (int f = 0; f < nf; ++f) idx[n++] = JxFace(h, ix, iy, iz, f);
    for (int f = 0; f < nf; ++f) idx[n++] = JxFace(h, ix + 1, iy, iz, f);
    for (int b = 0; b < ni; ++b) idx[n++] = JInt(h, 0, e, b);
    if (h->dim >= 2) {
        for (int f = 0; f < nf; ++f) idx[n++] = JyFace(h, ix, iy, iz, f);
        for (int f = 0; f < nf; ++f) idx[n++] = JyFace(h, ix, iy + 1, iz, f);
        for (int b = 0; b < ni; ++b) idx[n++] = JInt(h, 1, e, b);
    }
    if (h->dim == 3) {
        for (int f = 0; f < nf; ++f) idx[n++] = JzFace(h, ix, iy, iz, f);
        for (int f = 0; f < nf; ++f) idx[n++] = JzFace(h, ix, iy, iz + 1, f);
        for (int b = 0; b < ni; ++b) idx[n++] = JInt(h, 2, e, b);
    }
}
void nfo_global_J_indices(const nfo_t *h, int ix, int iy, int iz, int *idx)
{
    long t[3 * 27 * 3]; global_J(h, ix, iy, iz, t);
    for (int i = 0; i < h->nJloc; ++i) idx[i] = (int)t[i];
}
void nfo_global_phi_indices(const nfo_t *h, int ix, int iy, int iz, int *idx)
{
    long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
    for (int i = 0; i < h->nloc; ++i) idx[i] = (int)(e * h->nloc + i);
}

/* chain position of local J dof j of element (ix,iy,iz): direction d, transverse
 * mode a, position along the line.  Face dof a of face index c -> c*(k+1);
 * bubble (l,a) of cell c -> c*(k+1)+1+l. */
static long chain_pos(const nfo_t *h, int ix, int iy, int iz, int j)
{
    const int nf = h->nf, k = h->k, nper = h->nper;
    int d = j / nper, r = j % nper;
    int c, nd; long line;
    if (d == 0) { c = ix; nd = h->nx; line = (long)iz * h->ny + iy; }
    else if (d == 1) { c = iy; nd = h->ny; line = (long)iz * h->nx + ix; }
    else { c = iz; nd = h->nz; line = (long)iy * h->nx + ix; }
    long len = (long)(nd + 1) + (long)nd * k;
    int a, pos;
    if (r < nf) { a = r; pos = c * (k + 1); }
    else if (r < 2 * nf) { a = r - nf; pos = (c + 1) * (k + 1); }
    else { int b = r - 2 * nf; a = b / k; pos = c * (k + 1) + 1 + (b % k); }
    return h->chain_base[d] + (line * nf + a) * len + pos;
}

/* ---- constructor, src/NeutFEM.cpp:82-300, src/FEM.cpp:23-83,177-259 --------- */
nfo_t *nfo_create(int rt_order, int p_order, int ng, int nxb, const double *xb, int nyb, const double *yb, int nzb, const double *zb)
{
    if (nxb < 2 || ng < 1 || !xb) return NULL;                  /* at least one cell along x (src/FEM.cpp:23-60) */
    nfo_t *h = (nfo_t *)calloc(1, sizeof(nfo_t));
    h->nxb = nxb; h->nyb = nyb; h->nzb = nzb;
    h->xb = (double *)malloc(sizeof(double) * nxb); memcpy(h->xb, xb, sizeof(double) * nxb);
    h->yb = (double *)malloc(sizeof(double) * (nyb > 0 ? nyb : 1)); if (nyb > 0) memcpy(h->yb, yb, sizeof(double) * nyb);
    h->zb = (double *)malloc(sizeof(double) * (nzb > 0 ? nzb : 1)); if (nzb > 0) memcpy(h->zb, zb, sizeof(double) * nzb);
    h->nx = nxb - 1; h->ny = nyb > 1 ? nyb - 1 : 1; h->nz = nzb > 1 ? nzb - 1 : 1;
    h->dim = h->nz > 1 ? 3 : (h->ny > 1 ? 2 : 1);
    h->ne = h->nx * h->ny * h->nz;
    h->hx = (double *)malloc(sizeof(double) * h->nx); h->hy = (double *)malloc(sizeof(double) * h->ny); h->hz = (double *)malloc(sizeof(double) * h->nz);
    for (int i = 0; i < h->nx; ++i) h->hx[i] = xb[i + 1] - xb[i];
    if (h->dim >= 2) for (int i = 0; i < h->ny; ++i) h->hy[i] = yb[i + 1] - yb[i]; else h->hy[0] = 1.0;
    if (h->dim == 3) for (int i = 0; i < h->nz; ++i) h->hz[i] = zb[i + 1] - zb[i]; else h->hz[0] = 1.0;
    int k = rt_order < 2 ? rt_order : 2, m = p_order < 2 ? p_order : 2;
    if (k < m) m = k;                                          /* NeutFEM.cpp:149-169 */
    h->k = k; h->m = m; h->ng = ng;
    const int dim = h->dim;
    h->nloc = dim == 1 ? (m + 1) : dim == 2 ? (m + 1) * (m + 1) : (m + 1) * (m + 1) * (m + 1);
    h->nf = dim == 1 ? 1 : dim == 2 ? (k + 1) : (k + 1) * (k + 1);
    h->ni = dim == 1 ? k : dim == 2 ? k * (k + 1) : k * (k + 1) * (k + 1);
    h->nper = 2 * h->nf + h->ni; h->nJloc = dim * h->nper;
    h->nPhi = (long)h->ne * h->nloc;
    if (dim == 1) { h->nJx = (long)(h->nx + 1) * h->nf; h->nJy = h->nJz = 0; }
    else if (dim == 2) { h->nJx = (long)(h->nx + 1) * h->ny * h->nf; h->nJy = (long)h->nx * (h->ny + 1) * h->nf; h->nJz = 0; }
    else { h->nJx = (long)(h->nx + 1) * h->ny * h->nz * h->nf; h->nJy = (long)h->nx * (h->ny + 1) * h->nz * h->nf; h->nJz = (long)h->nx * h->ny * (h->nz + 1) * h->nf; }
    h->nJface = h->nJx + h->nJy + h->nJz;
    h->nJint = (long)h->ne * dim * h->ni; h->nJ = h->nJface + h->nJint;
    h->nq = gauss(2 * (k > m ? k : m) + 3, h->qp, h->qw);      /* NeutFEM.cpp:276 */
    const long ne = h->ne;
    h->D = (double *)malloc(sizeof(double) * ng * ne); h->SRC = (double *)calloc(ng * ne, sizeof(double));
    h->SigR = (double *)malloc(sizeof(double) * ng * ne); h->NSF = (double *)calloc(ng * ne, sizeof(double));
    h->KSF = (double *)calloc(ng * ne, sizeof(double)); h->Chi = (double *)calloc(ng * ne, sizeof(double));
    h->SigS = (double *)calloc((size_t)ng * ng * ne, sizeof(double));
    for (long i = 0; i < ng * ne; ++i) { h->D[i] = 1.0; h->SigR[i] = 0.01; }
    if (ng > 0) for (long e = 0; e < ne; ++e) h->Chi[e] = 1.0;
    h->phi = (double *)malloc(sizeof(double) * ng * h->nPhi); h->J = (double *)calloc(ng * h->nJ, sizeof(double));
    for (long i = 0; i < ng * h->nPhi; ++i) h->phi[i] = 1.0;
    h->phi_adj = (double *)malloc(sizeof(double) * ng * h->nPhi);
    for (long i = 0; i < ng * h->nPhi; ++i) h->phi_adj[i] = 1.0;
    h->last_keff_adj = 1.0; h->has_valid_adjoint = 0;
    h->cmfd_init = 0; h->cmfd_relax = 1.0;
    h->nfc[0] = (long)(h->nx + 1) * h->ny * h->nz; h->nfc[1] = (long)h->nx * (h->ny + 1) * h->nz; h->nfc[2] = (long)h->nx * h->ny * (h->nz + 1);
    h->tol_keff = h->tol_flux = h->tol_L2 = 1e-5; h->max_outer = 200; h->max_inner = 1000;
    h->solver_type = 6; h->solver_type_pushed = 0;             /* BICGSTAB shown, DIRECT_LU used (quirk 11) */
    h->schur_tol = 1e-10; h->schur_maxit = 1000;               /* solvers.cpp:67-76 */
    h->last_keff = 1.0; h->has_valid_keff = 0;
    unit_tables(h);
    h->bw = k + 1;
    long base = 0;
    for (int d = 0; d < dim; ++d) {
        int nd = d == 0 ? h->nx : d == 1 ? h->ny : h->nz;
        long nlines = (long)h->ne / nd;
        h->chain_base[d] = base; h->chain_len[d] = (long)(nd + 1) + (long)nd * k;
        base += nlines * h->nf * h->chain_len[d];
    }
    if (base != h->nJ) { fprintf(stderr, "nf_oracle: chain count %ld != n_J %ld\n", base, h->nJ); abort(); }
    h->hist_cg = (double *)calloc((size_t)MAXHIST * (ng > 0 ? ng : 1), sizeof(double));
    return h;
}

static void free_built(nfo_t *h)
{
    if (!h->built) return;
    for (int g = 0; g < h->ng; ++g) { free(h->band[g]); free(h->Aband[g]); free(h->Cd[g]); free(h->Mf[g]); free(h->Mchi[g]); if (h->Sinv && h->Sinv[g]) free(h->Sinv[g]);
                                      if (h->Sfac) { free(h->Sfac[g]); free(h->Spiv[g]); } }
    free(h->Sfac); free(h->Spiv); h->Sfac = NULL; h->Spiv = NULL;
    for (int i = 0; i < h->ng * h->ng; ++i) free(h->Ms[i]);
    free(h->band); free(h->Aband); free(h->Cd); free(h->Mf); free(h->Mchi); free(h->Ms); free(h->Sinv);
    free(h->eJ); free(h->wt); free(h->wu);
    h->built = 0;
}
void nfo_destroy(nfo_t *h)
{
    if (!h) return;
    free_built(h);
    free(h->xb); free(h->yb); free(h->zb); free(h->hx); free(h->hy); free(h->hz);
    free(h->D); free(h->SRC); free(h->SigR); free(h->NSF); free(h->KSF); free(h->Chi); free(h->SigS); free(h->phi); free(h->J); free(h->phi_adj);
    for (int d = 0; d < 3; ++d) free(h->Ahat[d]);
    free(h->Bhat); free(h->Chat); free(h->hist_cg);
    for (int d = 0; d < 3; ++d) { free(h->Dt[d]); free(h->Dh[d]); }
    free(h->voidm);
    free(h);
}

/* NOT in the reference.  Cells with mask != 0 are cut out of the domain: they contribute nothing to A and B, and every face they
 * share with a kept cell carries the boundary term A(f,f) += I_f * inv_alpha, i.e. J.n = alpha phi there (the reference's own form of
 * a boundary term, ApplyDirichletToA, with 1/alpha in the place of its 2 D).  Used by tests/test_oracle.py to compute the IAEA-2D
 * benchmark AS SPECIFIED (vacuum condition J.n = 0.4692 phi on the stepped core boundary) next to the drivers' variant of it
 * (blank assemblies filled with reflector, "Dirichlet" on the 380 cm box), which explains the drivers' offset from the literature k. */
void nfo_set_void(nfo_t *h, const unsigned char *mask, double inv_alpha)
{
    free(h->voidm); h->voidm = NULL;
    if (mask) { h->voidm = (unsigned char *)malloc(h->ne); memcpy(h->voidm, mask, h->ne); h->void_inv_alpha = inv_alpha; }
}
static int is_void(const nfo_t *h, long e) { return h->voidm && h->voidm[e]; }

long nfo_info(const nfo_t *h, const char *key)
{
#define K(s, v) if (!strcmp(key, s)) return (long)(v)
    K("dim", h->dim); K("nx", h->nx); K("ny", h->ny); K("nz", h->nz); K("ne", h->ne); K("ng", h->ng);
    K("k", h->k); K("m", h->m); K("nf", h->nf); K("ni", h->ni); K("nloc", h->nloc); K("nJloc", h->nJloc);
    K("n_phi", h->nPhi); K("n_J", h->nJ); K("n_Jx", h->nJx); K("n_Jy", h->nJy); K("n_Jz", h->nJz); K("nq", h->nq);
    K("last_outer", h->last_outer); K("last_cg_total", h->last_cg_total); K("coarse_outer", h->coarse_outer);
    K("last_cg_its", h->last_cg_its);
#undef K
    return -1;
}
double *nfo_array(nfo_t *h, const char *name, long *n)
{
    const long ne = h->ne, ng = h->ng;
#define A(s, p, len) if (!strcmp(name, s)) { if (n) *n = (len); return (p); }
    A("D", h->D, ng * ne) A("SigR", h->SigR, ng * ne) A("NSF", h->NSF, ng * ne) A("KSF", h->KSF, ng * ne)
    A("Chi", h->Chi, ng * ne) A("SRC", h->SRC, ng * ne) A("SigS", h->SigS, ng * ng * ne)
    A("phi", h->phi, ng * h->nPhi) A("J", h->J, ng * h->nJ) A("phi_adj", h->phi_adj, ng * h->nPhi)
    A("hist_k", h->hist_k, h->last_outer) A("hist_dk", h->hist_dk, h->last_outer) A("hist_dphi", h->hist_dphi, h->last_outer)
    A("hist_cg", h->hist_cg, (long)h->last_outer * ng)
#undef A
    if (n) *n = 0;
    return NULL;
}

void nfo_set_bc(nfo_t *h, int attr, int type, double value) { if (attr >= 0 && attr < 8) { h->bc_set[attr] = 1; h->bc_type[attr] = type; h->bc_val[attr] = value; } }
void nfo_set_tol(nfo_t *h, double a, double b, double c, int mo, int mi)
{
    h->tol_keff = a; h->tol_flux = b; h->tol_L2 = c; h->max_outer = mo; h->max_inner = mi;
    h->schur_tol = b; h->schur_maxit = mi;                     /* NeutFEM.cpp:334 */
}
void nfo_set_linear_solver(nfo_t *h, int type) { h->solver_type = type; h->solver_type_pushed = 1; }
void nfo_reset_flux(nfo_t *h)
{
    for (long i = 0; i < h->ng * h->nPhi; ++i) { h->phi[i] = 1.0; h->phi_adj[i] = 1.0; }
    memset(h->J, 0, sizeof(double) * h->ng * h->nJ);
    h->has_valid_keff = 0; h->has_valid_adjoint = 0;
}
double nfo_last_keff_adjoint(const nfo_t *h) { return h->last_keff_adj; }
void nfo_set_refactor_each_solve(nfo_t *h, int on) { h->refactor_each = on; }
double nfo_last_keff(const nfo_t *h) { return h->last_keff; }

/* src/NeutFEM.cpp:2338-2347 */
static int boundary_attr(int dim, int dir, int upper)
{
    if (dim == 1) return upper ? 2 : 1;
    if (dim == 2) { if (dir == 0) return upper ? 2 : 1; return upper ? 3 : 4; }
    if (dir == 0) return upper ? 4 : 3;
    if (dir == 1) return upper ? 5 : 6;
    return upper ? 2 : 1;
}
static int is_dirichlet(const nfo_t *h, int dir, int upper)
{
    int a = boundary_attr(h->dim, dir, upper);
    return h->bc_set[a] && h->bc_type[a] == NFO_BC_DIRICHLET;
}
/* src/NeutFEM.cpp:1458-1489 */
static double boundary_face_integral(const nfo_t *h, int lf, double area)
{
    const int k = h->k;
    if (h->dim == 1) return 1.0;
    if (h->dim == 2) return 2.0 * (2.0 / (2.0 * lf + 1.0)) / area;
    int a = lf % (k + 1), b = lf / (k + 1);
    return 4.0 * (2.0 / (2.0 * a + 1.0)) * (2.0 / (2.0 * b + 1.0)) / area;
}
static double face_area(const nfo_t *h, int ix, int iy, int iz, int dir)
{
    if (dir == 0) return h->hy[iy] * h->hz[iz];
    if (dir == 1) return h->hx[ix] * h->hz[iz];
    return h->hx[ix] * h->hy[iy];
}

/* banded LDL^T in place: band[p*(bw+1)+q] = A[p][p-q] -> L[p][p-q] (q>0), D[p] (q=0) */
static int band_factor(double *band, long n, int bw)
{
    const int w = bw + 1;
    for (long p = 0; p < n; ++p) {
        int qmax = p < bw ? (int)p : bw;
        for (int q = qmax; q >= 1; --q) {
            long j = p - q;                      /* L[p][j] */
            double s = band[p * w + q];
            int rmax = (int)(j < (long)(bw - q) ? j : (bw - q));   /* r = j - t, t=1..rmax, need p-(j-t) <= bw */
            for (int t = 1; t <= rmax; ++t) {
                s -= band[p * w + (q + t)] * band[j * w + t];   /* (L*D)[p][r] * L[j][r] */
            }
            band[p * w + q] = s;                 /* temporarily L*D */
        }
        double d = band[p * w];
        for (int q = 1; q <= qmax; ++q) {
            long j = p - q;
            double ld = band[p * w + q];
            double l = ld / band[j * w];
            d -= ld * l;
            band[p * w + q] = l;
        }
        band[p * w] = d;
        if (!(d > 0.0)) return -1;
    }
    return 0;
}
static void band_solve(const double *band, long n, int bw, double *x)
{
    const int w = bw + 1;
    for (long p = 0; p < n; ++p) {
        int qmax = p < bw ? (int)p : bw; double s = x[p];
        for (int q = 1; q <= qmax; ++q) s -= band[p * w + q] * x[p - q];
        x[p] = s;
    }
    for (long p = 0; p < n; ++p) x[p] /= band[p * w];
    for (long p = n - 1; p >= 0; --p) {
        double v = x[p]; int qmax = p < bw ? (int)p : bw;
        for (int q = 1; q <= qmax; ++q) x[p - q] -= band[p * w + q] * v;
    }
}

/* ---- BuildMatrices, src/NeutFEM.cpp:402-457 (+1036-1302, 1328-1456) ---------- */
int nfo_build(nfo_t *h)
{
    free_built(h);
    const int ng = h->ng, nJl = h->nJloc, nP = h->nloc, nper = h->nper, dim = h->dim, bw = h->bw, w = bw + 1;
    const long ne = h->ne, nJ = h->nJ;
    h->band = (double **)calloc(ng, sizeof(double *)); h->Aband = (double **)calloc(ng, sizeof(double *));
    h->Cd = (double **)calloc(ng, sizeof(double *)); h->Mf = (double **)calloc(ng, sizeof(double *)); h->Mchi = (double **)calloc(ng, sizeof(double *));
    h->Ms = (double **)calloc((size_t)ng * ng, sizeof(double *)); h->Sinv = (double **)calloc(ng, sizeof(double *));
    h->Sfac = (double **)calloc(ng, sizeof(double *)); h->Spiv = (int **)calloc(ng, sizeof(int *));
    h->eJ = (int *)malloc(sizeof(int) * ne * nJl);
    h->wt = (double *)malloc(sizeof(double) * nJ); h->wu = (double *)malloc(sizeof(double) * nJ);
    h->built = 1; h->diag_valid = 0; h->cmfd_init = 0;             /* NeutFEM.cpp:454-456 */
    for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
        long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
        for (int j = 0; j < nJl; ++j) h->eJ[e * nJl + j] = (int)chain_pos(h, ix, iy, iz, j);
    }
    for (int g = 0; g < ng; ++g) {
        double *A = (double *)calloc((size_t)nJ * w, sizeof(double));
        /* AssembleA(g): NeutFEM.cpp:1036-1076, entries |a| <= 1e-14 dropped (:1064) */
        for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
            long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
            if (is_void(h, e)) continue;
            double fac[3], detJ; geom_factors(h, ix, iy, iz, fac, &detJ);
            double invD = 1.0 / h->D[g * ne + e];
            const int *ej = h->eJ + e * nJl;
            for (int d = 0; d < dim; ++d)
                for (int i = 0; i < nper; ++i) for (int j = 0; j <= i; ++j) {
                    double v = invD * h->Ahat[d][i * nper + j] * fac[d];
                    if (!(fabs(v) > 1e-14)) continue;
                    long p = ej[d * nper + i], q = ej[d * nper + j];
                    if (p < q) { long t = p; p = q; q = t; }
                    if (p - q > bw) { fprintf(stderr, "nf_oracle: A entry outside chain band\n"); return -2; }
                    A[p * w + (p - q)] += v;
                }
        }
        /* ApplyDirichletToA(g): NeutFEM.cpp:1328-1456 : A(f,f) += I_f * 2 * D */
        for (int d = 0; d < dim; ++d) for (int up = 0; up < 2; ++up) {
            if (!is_dirichlet(h, d, up)) continue;
            int n1, n2;
            if (d == 0) { n1 = h->ny; n2 = h->nz; } else if (d == 1) { n1 = h->nx; n2 = h->nz; } else { n1 = h->nx; n2 = h->ny; }
            for (int b2 = 0; b2 < n2; ++b2) for (int b1 = 0; b1 < n1; ++b1) {
                int ix, iy, iz;
                if (d == 0) { ix = up ? h->nx - 1 : 0; iy = b1; iz = b2; }
                else if (d == 1) { ix = b1; iy = up ? h->ny - 1 : 0; iz = b2; }
                else { ix = b1; iy = b2; iz = up ? h->nz - 1 : 0; }
                long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
                double D = h->D[g * ne + e], fa = face_area(h, ix, iy, iz, d);
                for (int f = 0; f < h->nf; ++f) {
                    long p = h->eJ[e * nJl + d * nper + (up ? h->nf : 0) + f];
                    A[p * w] += boundary_face_integral(h, f, fa) * 2.0 * D;
                }
            }
        }
        if (h->voidm) {                                              /* nfo_set_void: Robin term on kept|void faces, identity on DOFs no kept cell touches */
            for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
                long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
                if (!is_void(h, e)) continue;
                const int nn[3] = { h->nx, h->ny, h->nz }, ii[3] = { ix, iy, iz }; const long st[3] = { 1, h->nx, (long)h->nx * h->ny };
                for (int d = 0; d < dim; ++d) for (int up = 0; up < 2; ++up) {
                    const int has = up ? ii[d] + 1 < nn[d] : ii[d] > 0;
                    const long en = has ? e + (up ? st[d] : -st[d]) : -1;
                    if (has && !is_void(h, en)) {
                        int jx = ix, jy = iy, jz = iz; if (d == 0) jx += up ? 1 : -1; else if (d == 1) jy += up ? 1 : -1; else jz += up ? 1 : -1;
                        const double fa = face_area(h, jx, jy, jz, d);
                        for (int f = 0; f < h->nf; ++f) A[(long)h->eJ[e * nJl + d * nper + (up ? h->nf : 0) + f] * w] += boundary_face_integral(h, f, fa) * h->void_inv_alpha;
                    }
                }
            }
            for (long p = 0; p < nJ; ++p) if (A[p * w] == 0.0) A[p * w] = 1.0;
        }
        h->Aband[g] = A;
        h->band[g] = (double *)malloc(sizeof(double) * nJ * w);
        memcpy(h->band[g], A, sizeof(double) * nJ * w);
        if (band_factor(h->band[g], nJ, bw)) { fprintf(stderr, "nf_oracle: A not SPD (group %d)\n", g); return -1; }
        /* AssembleC(g): NeutFEM.cpp:1163-1202 ; C-hat is diagonal after the 1e-14 drop */
        h->Cd[g] = (double *)calloc(h->nPhi, sizeof(double));
        h->Mf[g] = (double *)calloc(h->nPhi, sizeof(double));
        h->Mchi[g] = (double *)calloc(h->nPhi, sizeof(double));
        for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
            long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
            double fac[3], detJ; geom_factors(h, ix, iy, iz, fac, &detJ);
            double vol = h->hx[ix] * h->hy[iy] * h->hz[iz];
            double sig = h->SigR[g * ne + e], nsf = h->NSF[g * ne + e];
            for (int p = 0; p < nP; ++p) {
                double c = sig * detJ * h->Chat[p * nP + p];
                h->Cd[g][e * nP + p] = fabs(c) > 1e-14 ? c : 0.0;
            }
            /* M_chi = AssembleWeightedMassMatrix(chi_g): NeutFEM.cpp:432-439,1495-1529 (always through C_loc, also for P0) */
            { double chi = h->Chi[g * ne + e];
              if (!(fabs(chi) < 1e-14))
                  for (int p = 0; p < nP; ++p) { double c = chi * detJ * h->Chat[p * nP + p]; h->Mchi[g][e * nP + p] = fabs(c) > 1e-14 ? c : 0.0; } }
            /* AssembleFissionMatrix: NeutFEM.cpp:1204-1252 */
            if (h->m == 0) { if (fabs(nsf) > 1e-14) h->Mf[g][e] = nsf * vol; }
            else if (!(fabs(nsf) < 1e-14))
                for (int p = 0; p < nP; ++p) { double c = nsf * detJ * h->Chat[p * nP + p]; h->Mf[g][e * nP + p] = fabs(c) > 1e-14 ? c : 0.0; }
        }
        /* AssembleScatteringMatrix(gp -> g): NeutFEM.cpp:1254-1302 */
        for (int gp = 0; gp < ng; ++gp) {
            const double *ss = h->SigS + ((long)g * ng + gp) * ne;
            double *M = (double *)calloc(h->nPhi, sizeof(double)); long nnz = 0;
            for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
                long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
                double s = ss[e];
                if (h->m == 0) { if (fabs(s) > 1e-14) { M[e] = s * h->hx[ix] * h->hy[iy] * h->hz[iz]; ++nnz; } }
                else if (!(fabs(s) < 1e-14)) {
                    double fac[3], detJ; geom_factors(h, ix, iy, iz, fac, &detJ);
                    for (int p = 0; p < nP; ++p) { double c = s * detJ * h->Chat[p * nP + p]; if (fabs(c) > 1e-14) { M[e * nP + p] = c; ++nnz; } }
                }
            }
            if (nnz == 0) { free(M); M = NULL; }
            h->Ms[g * ng + gp] = M;
        }
    }
    return 0;
}

/* t (chain order) = B^T x ; uses B-hat (B_loc has no geometry, FEM.cpp:930-936), entries <=1e-14 dropped */
static void apply_BT(const nfo_t *h, const double *x, double *t)
{
    const int nJl = h->nJloc, nP = h->nloc; const long ne = h->ne;
    memset(t, 0, sizeof(double) * h->nJ);
    for (long e = 0; e < ne; ++e) {
        if (is_void(h, e)) continue;
        const int *ej = h->eJ + e * nJl;
        for (int p = 0; p < nP; ++p) {
            double xv = x[e * nP + p];
            const double *Bp = h->Bhat + p * nJl;
            for (int j = 0; j < nJl; ++j) if (fabs(Bp[j]) > 1e-14) t[ej[j]] += Bp[j] * xv;
        }
    }
}
static void apply_B_add(const nfo_t *h, const double *u, double *y)
{
    const int nJl = h->nJloc, nP = h->nloc; const long ne = h->ne;
    for (long e = 0; e < ne; ++e) {
        if (is_void(h, e)) continue;
        const int *ej = h->eJ + e * nJl;
        for (int p = 0; p < nP; ++p) {
            const double *Bp = h->Bhat + p * nJl; double s = 0.0;
            for (int j = 0; j < nJl; ++j) if (fabs(Bp[j]) > 1e-14) s += Bp[j] * u[ej[j]];
            y[e * nP + p] += s;
        }
    }
}

/* SchurSolver::SchurProduct, src/solvers.cpp:535-547 */
void nfo_schur_apply(nfo_t *h, int g, const double *x, double *y)
{
    apply_BT(h, x, h->wt);
    band_solve(h->band[g], h->nJ, h->bw, h->wt);
    for (long i = 0; i < h->nPhi; ++i) y[i] = h->Cd[g][i] * x[i];
    apply_B_add(h, h->wt, y);
}

/* SchurSolver::SolveSchurImplicit, src/solvers.cpp:577-636 */
static int cg_solve(nfo_t *h, int g, const double *rhs, double *phi, double tol, int maxit)
{
    const long n = h->nPhi;
    double *r = (double *)malloc(sizeof(double) * n * 3), *p = r + n, *Ap = p + n;
    double rr = 0.0;
    for (long i = 0; i < n; ++i) { phi[i] = 0.0; r[i] = rhs[i]; p[i] = rhs[i]; rr += rhs[i] * rhs[i]; }
    const double rhs_norm = sqrt(rr), tol_sq = tol * tol * rhs_norm * rhs_norm;
    int its = 0;
    for (int k = 0; k < maxit; ++k) {
        nfo_schur_apply(h, g, p, Ap);
        double pAp = 0.0; for (long i = 0; i < n; ++i) pAp += p[i] * Ap[i];
        if (fabs(pAp) < 1e-30) break;
        double alpha = rr / pAp, rrn = 0.0;
        for (long i = 0; i < n; ++i) { phi[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; rrn += r[i] * r[i]; }
        its = k + 1;
        if (rrn < tol_sq) { rr = rrn; break; }
        double beta = rrn / rr;
        for (long i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];
        rr = rrn;
    }
    h->last_cg_its = its; h->last_cg_res = sqrt(rr) / rhs_norm;
    free(r);
    return its;
}

/* Explicit branch of the Schur solver (src/solvers.cpp:114-124: direct solver types, a solver type that was never pushed --
 * SchurSolver's own default is DIRECT_LU, :68 -- or n_phi < 200).  FormSchurComplement (:259-310): S = C + B (A^-1 B^T) column by
 * column, one A solve per column of B^T, entries of A^-1 B^T with |v| <= 1e-14 dropped (:291).  PrepareSolver / SolveSchurExplicit
 * (:330-509): Eigen's SparseLU / SimplicialLDLT / SimplicialLLT of S are exact solves; a dense LU with partial pivoting stands for all
 * three.  For the iterative types at n_phi < 200 the reference runs Eigen's CG / BiCGSTAB on the explicit S to tol_flux (and asserts on
 * the BiCGSTAB ones); the exact solve stands for those too -- it differs from such an iterate by at most the tolerance.
 * The factorisation is cached per group until the next BuildMatrices (the reference redoes it in every SetMatrices, :149-179).
 * Dense storage: meshes beyond NFO_DIRECT_MAX unknowns per group keep the CG-to-1e-14 stand-in below. */
#define NFO_DIRECT_MAX 6000
static int schur_explicit_factor(nfo_t *h, int g)
{
    if (h->Sfac[g]) return 0;
    const long n = h->nPhi, nJ = h->nJ;
    double *S = (double *)calloc((size_t)n * n, sizeof(double)), *e = (double *)calloc(n, sizeof(double));
    double *col = (double *)malloc(sizeof(double) * nJ), *y = (double *)malloc(sizeof(double) * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    for (long j = 0; j < n; ++j) {
        e[j] = 1.0;
        apply_BT(h, e, col);                                   /* column j of B^T (chain order) */
        band_solve(h->band[g], nJ, h->bw, col);                /* A x = B^T[:, j] */
        for (long i = 0; i < nJ; ++i) if (!(fabs(col[i]) > 1e-14)) col[i] = 0.0;
        memset(y, 0, sizeof(double) * n);
        apply_B_add(h, col, y);
        y[j] += h->Cd[g][j];
        for (long i = 0; i < n; ++i) S[i * n + j] = y[i];
        e[j] = 0.0;
    }
    int rc = 0;
    for (long k = 0; k < n && !rc; ++k) {                      /* LU, partial pivoting, row-major in place */
        long p = k; double best = fabs(S[k * n + k]);
        for (long i = k + 1; i < n; ++i) if (fabs(S[i * n + k]) > best) { best = fabs(S[i * n + k]); p = i; }
        piv[k] = (int)p;
        if (best == 0.0) { rc = -1; break; }
        if (p != k) for (long c = 0; c < n; ++c) { double t = S[k * n + c]; S[k * n + c] = S[p * n + c]; S[p * n + c] = t; }
        const double inv = 1.0 / S[k * n + k];
        for (long i = k + 1; i < n; ++i) {
            const double l = S[i * n + k] * inv;
            if (l == 0.0) continue;
            S[i * n + k] = l;
            double *ri = S + i * n; const double *rk = S + k * n;
            for (long c = k + 1; c < n; ++c) ri[c] -= l * rk[c];
        }
    }
    free(e); free(col); free(y);
    if (rc) { free(S); free(piv); fprintf(stderr, "nf_oracle: explicit Schur complement is singular (group %d)\n", g); return rc; }
    h->Sfac[g] = S; h->Spiv[g] = piv;
    return 0;
}
static void schur_explicit_solve(const nfo_t *h, int g, const double *rhs, double *phi)
{
    const long n = h->nPhi; const double *S = h->Sfac[g]; const int *piv = h->Spiv[g];
    for (long i = 0; i < n; ++i) phi[i] = rhs[i];
    for (long k = 0; k < n; ++k) { long p = piv[k]; if (p != k) { double t = phi[k]; phi[k] = phi[p]; phi[p] = t; } }   /* P b (whole rows were swapped) */
    for (long k = 0; k < n; ++k) { const double v = phi[k]; if (v != 0.0) for (long i = k + 1; i < n; ++i) phi[i] -= S[i * n + k] * v; }
    for (long k = n - 1; k >= 0; --k) { double v = phi[k]; const double *rk = S + k * n; for (long c = k + 1; c < n; ++c) v -= rk[c] * phi[c]; phi[k] = v / rk[k]; }
}

/* SchurSolver::Solve, src/solvers.cpp:203-240 (+ SetMatrices :149-179 when refactor_each) */
int nfo_solve_group(nfo_t *h, int g, const double *rhs, double *phi, double *J)
{
    if (h->refactor_each) {
        memcpy(h->band[g], h->Aband[g], sizeof(double) * h->nJ * (h->bw + 1));
        band_factor(h->band[g], h->nJ, h->bw);
    }
    int direct = !h->solver_type_pushed || h->solver_type <= 2 || h->nPhi < 200;   /* solvers.cpp:114-124 */
    int its;
    if (direct && h->nPhi <= NFO_DIRECT_MAX && schur_explicit_factor(h, g) == 0) {
        schur_explicit_solve(h, g, rhs, phi); its = 1;            /* last_iterations_ = 1, solvers.cpp:447 */
        h->last_cg_its = 1; h->last_cg_res = 0.0;
    } else
        its = direct ? cg_solve(h, g, rhs, phi, 1e-14, (int)(20 * h->nPhi + 50))   /* beyond NFO_DIRECT_MAX: CG stands in (<= n steps in exact arithmetic) */
                     : cg_solve(h, g, rhs, phi, h->schur_tol, h->schur_maxit);
    if (J) {                                                   /* J = -A^-1 B^T phi, solvers.cpp:227-228 */
        apply_BT(h, phi, h->wt);
        band_solve(h->band[g], h->nJ, h->bw, h->wt);
        const int nJl = h->nJloc; long idx[3 * 27 * 3];
        for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
            long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
            global_J(h, ix, iy, iz, idx);
            for (int j = 0; j < nJl; ++j) J[idx[j]] = -h->wt[h->eJ[e * nJl + j]];
        }
    }
    return its;
}

/* BuildDiagonalSchurCache, src/NeutFEM.cpp:483-597 */
static void build_diag_cache(nfo_t *h)
{
    if (h->k != 0 || h->m != 0 || h->diag_valid) return;
    const int ng = h->ng, nJl = h->nJloc, w = h->bw + 1; const long ne = h->ne;
    for (int g = 0; g < ng; ++g) {
        if (!h->Sinv[g]) h->Sinv[g] = (double *)malloc(sizeof(double) * ne);
        for (long e = 0; e < ne; ++e) {
            double S = h->Cd[g][e];
            for (int j = 0; j < nJl; ++j) {
                double B = fabs(h->Bhat[j]) > 1e-14 ? h->Bhat[j] : 0.0;
                double Aff = h->Aband[g][(long)h->eJ[e * nJl + j] * w];
                if (fabs(Aff) > 1e-14) S += B * B / Aff;
            }
            h->Sinv[g][e] = fabs(S) > 1e-14 ? 1.0 / S : 0.0;
        }
    }
    h->diag_valid = 1;
}
const double *nfo_diag_cache(nfo_t *h, int g) { build_diag_cache(h); return h->diag_valid ? h->Sinv[g] : NULL; }

/* SolveDiagonalSchur, src/NeutFEM.cpp:607-634  (J = + A_ff^-1 B^T phi : sign differs from the full path) */
static void solve_diag(nfo_t *h, int g, const double *rhs, double *phi, double *J)
{
    const long ne = h->ne; const int nJl = h->nJloc, w = h->bw + 1;
    for (long e = 0; e < ne; ++e) phi[e] = h->Sinv[g][e] * rhs[e];
    apply_BT(h, phi, h->wt);
    long idx[18];
    for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
        long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
        global_J(h, ix, iy, iz, idx);
        for (int j = 0; j < nJl; ++j) {
            long p = h->eJ[e * nJl + j]; double Aff = h->Aband[g][p * w];
            J[idx[j]] = fabs(Aff) < 1e-14 ? 0.0 : h->wt[p] / Aff;
        }
    }
}

/* ChebyshevAccel, src/solvers.cpp:664-756 */
typedef struct { int nmax, it; double sigma; double a[64], b[64]; double *p0, *p1; long n; } cheb_t;
static void cheb_init(cheb_t *c, int nmax, double sigma, long n)
{
    c->nmax = nmax; c->it = 0; c->sigma = sigma; c->n = n; c->p0 = c->p1 = NULL;
    double G = acosh(2. / sigma - 1.);
    c->a[0] = c->b[0] = 0.; c->a[1] = 2. / (2. - sigma); c->b[1] = 0.;
    for (int k = 2; k < nmax; ++k) { c->a[k] = cosh((k - 1) * G) / cosh(k * G); c->b[k] = cosh((k - 2) * G) / cosh(k * G); }
}
static void cheb_apply(cheb_t *c, double *phi)
{
    const long n = c->n;
    if (c->it == c->nmax) { c->it = 0; free(c->p0); free(c->p1); c->p0 = c->p1 = NULL; }
    if (c->it == 0) { c->p0 = (double *)malloc(sizeof(double) * n); memcpy(c->p0, phi, sizeof(double) * n); ++c->it; }
    else if (c->it == 1) {
        c->p1 = (double *)malloc(sizeof(double) * n);
        for (long i = 0; i < n; ++i) { c->p1[i] = c->p0[i] + c->a[1] * (phi[i] - c->p0[i]); phi[i] = c->p1[i]; }
        ++c->it;
    } else {
        double *nw = (double *)malloc(sizeof(double) * n);
        const double ca = (4. / c->sigma) * c->a[c->it], cb = c->b[c->it];
        for (long i = 0; i < n; ++i) { nw[i] = c->p1[i] + ca * (phi[i] - c->p1[i]) + cb * (c->p1[i] - c->p0[i]); phi[i] = nw[i]; }
        free(c->p0); c->p0 = c->p1; c->p1 = nw; ++c->it;
    }
}
static void cheb_free(cheb_t *c) { free(c->p0); free(c->p1); }

/* ---- CMFD acceleration, src/NeutFEM.cpp:662-1017 (x-direction D-hat only, as the reference) ---------------------- */
void nfo_set_cmfd_relaxation(nfo_t *h, double omega) { h->cmfd_relax = omega; }
static long cmfd_face(const nfo_t *h, int d, int ix, int iy, int iz)
{
    if (d == 0) return (long)iz * h->ny * (h->nx + 1) + (long)iy * (h->nx + 1) + ix;
    if (d == 1) return (long)iz * (h->ny + 1) * h->nx + (long)iy * h->nx + ix;
    return (long)iz * h->ny * h->nx + (long)iy * h->nx + ix;
}
/* InitializeCMFD + ComputeDtildeCoefficients (:662-821) */
static void cmfd_initialize(nfo_t *h)
{
    if (h->cmfd_init) return;
    const int ng = h->ng; const long ne = h->ne;
    for (int d = 0; d < 3; ++d) {
        free(h->Dt[d]); free(h->Dh[d]);
        h->Dt[d] = (double *)calloc((size_t)ng * h->nfc[d], sizeof(double));
        h->Dh[d] = (double *)calloc((size_t)ng * h->nfc[d], sizeof(double));
    }
    for (int g = 0; g < ng; ++g)
        for (int d = 0; d < h->dim; ++d) {
            const int nd = d == 0 ? h->nx : d == 1 ? h->ny : h->nz;
            const double *brk = d == 0 ? h->xb : d == 1 ? h->yb : h->zb;
            const int n0 = d == 0 ? nd + 1 : h->nx, n1 = d == 1 ? nd + 1 : h->ny, n2 = d == 2 ? nd + 1 : h->nz;
            for (int iz = 0; iz < n2; ++iz) for (int iy = 0; iy < n1; ++iy) for (int ix = 0; ix < n0; ++ix) {
                const int c = d == 0 ? ix : d == 1 ? iy : iz;
                int lo[3] = { ix, iy, iz }, hi[3] = { ix, iy, iz };
                double v;
                if (c == 0 || c == nd) {
                    lo[d] = c == 0 ? 0 : nd - 1;
                    long e = (long)lo[2] * h->nx * h->ny + (long)lo[1] * h->nx + lo[0];
                    double D = h->D[g * ne + e];
                    double dx = brk[c == 0 ? 1 : nd] - brk[c == 0 ? 0 : nd - 1];
                    v = 2.0 * D / dx;
                } else {
                    lo[d] = c - 1; hi[d] = c;
                    long eL = (long)lo[2] * h->nx * h->ny + (long)lo[1] * h->nx + lo[0], eR = (long)hi[2] * h->nx * h->ny + (long)hi[1] * h->nx + hi[0];
                    double DL = h->D[g * ne + eL], DR = h->D[g * ne + eR];
                    double dL = brk[c] - brk[c - 1], dR = brk[c + 1] - brk[c];
                    v = 2.0 * DL * DR / (DL * dR + DR * dL);
                }
                h->Dt[d][g * h->nfc[d] + cmfd_face(h, d, ix, iy, iz)] = v;
            }
        }
    h->cmfd_init = 1;
}
/* UpdateDhatCoefficients (:823-869): x faces only */
static void cmfd_update_dhat(nfo_t *h)
{
    const int dpe = h->nloc;
    for (int g = 0; g < h->ng; ++g) {
        const double *phi = h->phi + g * h->nPhi, *J = h->J + g * h->nJ;
        for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix <= h->nx; ++ix) {
            long f = cmfd_face(h, 0, ix, iy, iz);
            double Jn = J[JxFace(h, ix, iy, iz, 0)], pd;
            long eL = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix - 1, eR = eL + 1;
            if (ix == 0) pd = -phi[eR * dpe]; else if (ix == h->nx) pd = phi[eL * dpe]; else pd = phi[eL * dpe] - phi[eR * dpe];
            h->Dh[0][g * h->nfc[0] + f] = fabs(pd) > 1e-14 ? Jn / pd - h->Dt[0][g * h->nfc[0] + f] : 0.0;
        }
    }
}
/* y = M_cmfd p (7-point operator of ApplyCMFDCorrection, :893-972), matrix-free */
static void cmfd_matvec(const nfo_t *h, int g, const double *diag, const double *p, double *y)
{
    const long nxy = (long)h->nx * h->ny;
    for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
        long e = (long)iz * nxy + (long)iy * h->nx + ix;
        double s = diag[e] * p[e];
        for (int d = 0; d < h->dim; ++d) {
            const int c = d == 0 ? ix : d == 1 ? iy : iz, nd = d == 0 ? h->nx : d == 1 ? h->ny : h->nz;
            const long st = d == 0 ? 1 : d == 1 ? h->nx : nxy;
            const double A = face_area(h, ix, iy, iz, d);
            int u[3] = { ix, iy, iz }; u[d] += 1;
            const long fl = g * h->nfc[d] + cmfd_face(h, d, ix, iy, iz), fu = g * h->nfc[d] + cmfd_face(h, d, u[0], u[1], u[2]);
            if (c > 0) s -= (h->Dt[d][fl] + h->Dh[d][fl]) * A * p[e - st];
            if (c < nd - 1) s -= (h->Dt[d][fu] + h->Dh[d][fu]) * A * p[e + st];
        }
        y[e] = s;
    }
}
/* ApplyCMFDCorrection (:871-1017).  The linear solve is Eigen::ConjugateGradient<SpMat, Lower|Upper> with its default
 * DiagonalPreconditioner, tolerance 1e-8, 100 iterations, x0 = 0 -- restated from Eigen 3.4's published algorithm
 * (Eigen/src/IterativeLinearSolvers/ConjugateGradient.h: conjugate_gradient()). */
static void cmfd_correction(nfo_t *h, int g, const double *tf, double keff, double *corr)
{
    const long ne = h->ne; const int dpe = h->nloc;
    double *diag = (double *)malloc(sizeof(double) * ne * 7), *rhs = diag + ne, *x = rhs + ne, *r = x + ne, *pp = r + ne, *z = pp + ne, *tmp = z + ne;
    for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
        long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
        double dg = h->Cd[g][e * dpe];
        for (int d = 0; d < h->dim; ++d) {
            int u[3] = { ix, iy, iz }; u[d] += 1;
            const long fl = g * h->nfc[d] + cmfd_face(h, d, ix, iy, iz), fu = g * h->nfc[d] + cmfd_face(h, d, u[0], u[1], u[2]);
            dg += ((h->Dt[d][fl] + h->Dh[d][fl]) + (h->Dt[d][fu] + h->Dh[d][fu])) * face_area(h, ix, iy, iz, d);
        }
        diag[e] = dg;
        rhs[e] = h->Chi[g * ne + e] * tf[e * dpe] / keff;
    }
    double rhs2 = 0.0; for (long e = 0; e < ne; ++e) { x[e] = 0.0; r[e] = rhs[e]; rhs2 += rhs[e] * rhs[e]; }
    if (rhs2 > 0.0) {
        const double thr = fmax(1e-8 * 1e-8 * rhs2, 2.2250738585072014e-308);
        double rn2 = rhs2;
        if (!(rn2 < thr)) {
            double absNew = 0.0;
            for (long e = 0; e < ne; ++e) { pp[e] = (diag[e] != 0.0 ? 1.0 / diag[e] : 1.0) * r[e]; absNew += r[e] * pp[e]; }   /* DiagonalPreconditioner: m_invdiag * b */
            for (int i = 0; i < 100; ++i) {
                cmfd_matvec(h, g, diag, pp, tmp);
                double pt = 0.0; for (long e = 0; e < ne; ++e) pt += pp[e] * tmp[e];
                const double alpha = absNew / pt;
                rn2 = 0.0;
                for (long e = 0; e < ne; ++e) { x[e] += alpha * pp[e]; r[e] -= alpha * tmp[e]; rn2 += r[e] * r[e]; }
                if (rn2 < thr) break;
                double absOld = absNew; absNew = 0.0;
                for (long e = 0; e < ne; ++e) { z[e] = (diag[e] != 0.0 ? 1.0 / diag[e] : 1.0) * r[e]; absNew += r[e] * z[e]; }
                const double beta = absNew / absOld;
                for (long e = 0; e < ne; ++e) pp[e] = z[e] + beta * pp[e];
            }
        }
    }
    const double om = h->cmfd_relax;
    for (long e = 0; e < ne; ++e) {
        double ratio = 1.0, pc = h->phi[g * h->nPhi + e * dpe];
        if (fabs(pc) > 1e-14) { ratio = x[e] / pc; ratio = fmax(0.5, fmin(2.0, ratio)); }
        for (int d = 0; d < dpe; ++d) corr[e * dpe + d] = om * ratio + (1.0 - om) * 1.0;
    }
    free(diag);
}

/* probes: D-tilde / D-hat of (group, direction); one UpdateDhat + ApplyCMFDCorrection on the current phi / J */
long nfo_cmfd_coefficients(nfo_t *h, int g, int dir, double *dt, double *dh)
{
    cmfd_initialize(h);
    for (long f = 0; f < h->nfc[dir]; ++f) { if (dt) dt[f] = h->Dt[dir][g * h->nfc[dir] + f]; if (dh) dh[f] = h->Dh[dir][g * h->nfc[dir] + f]; }
    return h->nfc[dir];
}
void nfo_cmfd_probe(nfo_t *h, int g, const double *tf, double keff, double *corr)
{
    cmfd_initialize(h);
    cmfd_update_dhat(h);
    cmfd_correction(h, g, tf, keff, corr);
}

/* SolveKeff, src/NeutFEM.cpp:1627-1815 */
static double solve_keff_impl(nfo_t *h, int use_coarse_init, const int *factors, int nfactors, int use_diag, int use_cmfd);
double nfo_solve_keff(nfo_t *h, int use_coarse_init, const int *factors, int nfactors, int use_diag)
{
    return solve_keff_impl(h, use_coarse_init, factors, nfactors, use_diag, 0);
}
double nfo_solve_keff_cmfd(nfo_t *h, int use_coarse_init, const int *factors, int nfactors, int use_diag, int use_cmfd)
{
    return solve_keff_impl(h, use_coarse_init, factors, nfactors, use_diag, use_cmfd);
}
static double solve_keff_impl(nfo_t *h, int use_coarse_init, const int *factors, int nfactors, int use_diag, int use_cmfd)
{
    const int ng = h->ng; const long nP = h->nPhi, ne = h->ne, nJ = h->nJ; const int dpe = h->nloc;
    if (use_diag && !(h->k == 0 && h->m == 0)) use_diag = 0;
    if (use_diag) build_diag_cache(h);
    if (use_cmfd) cmfd_initialize(h);                            /* :1655-1658 */
    double keff = h->has_valid_keff ? h->last_keff : 1.0;
    h->coarse_outer = 0;
    if (use_coarse_init && nfactors > 0) {
        double *fc = (double *)malloc(sizeof(double) * ng * nP);
        keff = nfo_solve_coarse(h, factors, nfactors, fc);
        memcpy(h->phi, fc, sizeof(double) * ng * nP); free(fc);
    }
    cheb_t acc; cheb_init(&acc, 15, 0.98, ng * nP);
    double *tf = (double *)malloc(sizeof(double) * nP * 3), *rhs = tf + nP, *sol = rhs + nP;
    double *old = (double *)malloc(sizeof(double) * ng * nP);
    double *Jt = (double *)malloc(sizeof(double) * nJ);
    h->last_outer = 0; h->last_cg_total = 0;
    for (int it = 0; it < h->max_outer; ++it) {
        memcpy(old, h->phi, sizeof(double) * ng * nP);
        memset(tf, 0, sizeof(double) * nP);
        for (int g = 0; g < ng; ++g) for (long i = 0; i < nP; ++i) tf[i] += h->Mf[g][i] * h->phi[g * nP + i];
        double prod_old = 0.0; for (long i = 0; i < nP; ++i) prod_old += tf[i];
        for (int g = 0; g < ng; ++g) {
            memset(rhs, 0, sizeof(double) * nP);
            /* BuildFissionRHS, NeutFEM.cpp:1539-1561 */
            const double inv_k = 1.0 / keff;
            if (dpe == 1) for (long e = 0; e < ne; ++e) rhs[e] += inv_k * (h->Chi[g * ne + e] * tf[e]);
            else for (long e = 0; e < ne; ++e) {
                double cv = h->Chi[g * ne + e] * inv_k;
                if (fabs(cv) < 1e-14) continue;
                for (int d = 0; d < dpe; ++d) rhs[e * dpe + d] += cv * tf[e * dpe + d];
            }
            for (int gp = 0; gp < ng; ++gp) {
                if (gp == g) continue;
                const double *M = h->Ms[g * ng + gp];
                if (!M) continue;
                for (long i = 0; i < nP; ++i) rhs[i] += M[i] * h->phi[gp * nP + i];
            }
            int its = 0;
            if (use_diag) solve_diag(h, g, rhs, sol, Jt);
            else its = nfo_solve_group(h, g, rhs, sol, Jt);
            memcpy(h->phi + g * nP, sol, sizeof(double) * nP);
            memcpy(h->J + g * nJ, Jt, sizeof(double) * nJ);
            if (it < MAXHIST) h->hist_cg[(long)it * ng + g] = its;
            h->last_cg_total += its;
        }
        if (use_cmfd && it >= 2) {                                /* :1750-1761 */
            cmfd_update_dhat(h);
            for (int g = 0; g < ng; ++g) {
                cmfd_correction(h, g, tf, keff, sol);
                for (long i = 0; i < nP; ++i) h->phi[g * nP + i] *= sol[i];
            }
        }
        double prod_new = 0.0;
        for (int g = 0; g < ng; ++g) for (long i = 0; i < nP; ++i) prod_new += h->Mf[g][i] * h->phi[g * nP + i];
        const double keff_new = keff * (prod_new / prod_old);
        const double dk = fabs(keff_new - keff);
        if (it >= 1) keff = keff_new;
        double nsq = 0.0, dsq = 0.0;
        for (long i = 0; i < ng * nP; ++i) { nsq += h->phi[i] * h->phi[i]; double d = h->phi[i] - old[i]; dsq += d * d; }
        const double dphi = sqrt(dsq / nsq), norm = sqrt(nsq);
        if (norm > 1e-14) for (long i = 0; i < ng * nP; ++i) h->phi[i] /= norm;
        if (!use_cmfd && it >= 2) cheb_apply(&acc, h->phi);
        if (it < MAXHIST) { h->hist_k[it] = keff; h->hist_dk[it] = dk; h->hist_dphi[it] = dphi; }
        h->last_outer = it + 1;
        if (dk < h->tol_keff && dphi < h->tol_flux) break;
    }
    cheb_free(&acc); free(tf); free(old); free(Jt);
    h->has_valid_keff = 1; h->last_keff = keff;
    return keff;
}

/* SolveAdjoint, src/NeutFEM.cpp:1877-2082 (+ BuildFissionRHSAdjoint :1568-1589, SolveGroupInternalAdjoint :2107-2126) */
double nfo_solve_adjoint(nfo_t *h, int normalize_to_direct, int use_direct_keff)
{
    const int ng = h->ng; const long nP = h->nPhi, ne = h->ne; const int dpe = h->nloc;
    double keff = 1.0;
    if (use_direct_keff && h->has_valid_keff) keff = h->last_keff;
    double *pa = h->phi_adj;
    { double n0 = sqrt((double)(ng * nP)); for (long i = 0; i < ng * nP; ++i) pa[i] = 1.0 / n0; }   /* setConstant(1) / norm */
    cheb_t acc; cheb_init(&acc, 15, 0.98, ng * nP);
    double *tca = (double *)malloc(sizeof(double) * nP * 3), *rhs = tca + nP, *sol = rhs + nP;
    double *old = (double *)malloc(sizeof(double) * ng * nP);
    double *nsft = (double *)calloc(ne, sizeof(double));
    for (long e = 0; e < ne; ++e) for (int g = 0; g < ng; ++g) nsft[e] += h->NSF[g * ne + e];
    h->last_outer = 0; h->last_cg_total = 0;
    for (int it = 0; it < h->max_outer; ++it) {
        memcpy(old, pa, sizeof(double) * ng * nP);
        memset(tca, 0, sizeof(double) * nP);
        for (int g = 0; g < ng; ++g) for (long i = 0; i < nP; ++i) tca[i] += h->Mchi[g][i] * pa[g * nP + i];
        double prod_old = 0.0; for (long e = 0; e < ne; ++e) prod_old += nsft[e] * tca[e * dpe];
        for (int g = 0; g < ng; ++g) {
            memset(rhs, 0, sizeof(double) * nP);
            const double inv_k = 1.0 / keff;
            if (dpe == 1) for (long e = 0; e < ne; ++e) rhs[e] += inv_k * (h->NSF[g * ne + e] * tca[e]);
            else for (long e = 0; e < ne; ++e) {
                double nv = h->NSF[g * ne + e] * inv_k;
                if (fabs(nv) < 1e-14) continue;
                for (int d = 0; d < dpe; ++d) rhs[e * dpe + d] += nv * tca[e * dpe + d];
            }
            for (int gp = 0; gp < ng; ++gp) {
                if (gp == g) continue;
                const double *M = h->Ms[gp * ng + g];                       /* transposed block index (:1944-1950) */
                if (!M) continue;
                for (long i = 0; i < nP; ++i) rhs[i] += M[i] * pa[gp * nP + i];
            }
            int its = nfo_solve_group(h, g, rhs, sol, NULL);
            memcpy(pa + g * nP, sol, sizeof(double) * nP);
            if (it < MAXHIST) h->hist_cg[(long)it * ng + g] = its;
            h->last_cg_total += its;
        }
        memset(tca, 0, sizeof(double) * nP);
        for (int g = 0; g < ng; ++g) for (long i = 0; i < nP; ++i) tca[i] += h->Mchi[g][i] * pa[g * nP + i];
        double prod_new = 0.0; for (long e = 0; e < ne; ++e) prod_new += nsft[e] * tca[e * dpe];
        double keff_new = keff, dk;
        if (!use_direct_keff || !h->has_valid_keff) {
            if (fabs(prod_old) > 1e-14 && it > 0) keff_new = keff * (prod_new / prod_old);
            dk = fabs(keff_new - keff); keff = keff_new;
        } else dk = 0.0;
        double nsq = 0.0, dsq = 0.0;
        for (long i = 0; i < ng * nP; ++i) { nsq += pa[i] * pa[i]; double d = pa[i] - old[i]; dsq += d * d; }
        const double dphi = sqrt(dsq) / sqrt(nsq), norm = sqrt(nsq);
        if (norm > 1e-14) for (long i = 0; i < ng * nP; ++i) pa[i] /= norm;
        if (!use_direct_keff && it >= 5) cheb_apply(&acc, pa);
        if (it < MAXHIST) { h->hist_k[it] = keff; h->hist_dk[it] = dk; h->hist_dphi[it] = dphi; }
        h->last_outer = it + 1;
        int conv = dphi < h->tol_flux;
        if (!use_direct_keff) conv = conv && (dk < h->tol_keff);
        if (conv) break;
    }
    if (normalize_to_direct && h->has_valid_keff) {                         /* <phi, phi+> = 1 (:2020-2066) */
        double ip = 0.0; const int n = h->m + 1;
        for (int g = 0; g < ng; ++g) for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
            long e = (long)iz * h->nx * h->ny + (long)iy * h->nx + ix;
            double vol = h->hx[ix] * h->hy[iy] * h->hz[iz];
            for (int d = 0; d < dpe; ++d) {
                int i, j, kk;
                if (h->dim == 1) { i = d; j = 0; kk = 0; } else if (h->dim == 2) { i = d % n; j = d / n; kk = 0; } else { i = d % n; j = (d / n) % n; kk = d / (n * n); }
                double w = (2.0 / (2.0 * i + 1.0)) / 2.0;
                if (h->dim >= 2) w *= (2.0 / (2.0 * j + 1.0)) / 2.0;
                if (h->dim >= 3) w *= (2.0 / (2.0 * kk + 1.0)) / 2.0;
                ip += h->phi[g * nP + e * dpe + d] * pa[g * nP + e * dpe + d] * vol * w;
            }
        }
        if (fabs(ip) > 1e-14) for (long i = 0; i < ng * nP; ++i) pa[i] /= ip;
    }
    cheb_free(&acc); free(tca); free(old); free(nsft);
    h->has_valid_adjoint = 1; h->last_keff_adj = keff;
    return keff;
}

/* SolveCoarse, src/NeutFEM.cpp:2380-2611 */
double nfo_solve_coarse(nfo_t *h, const int *factors, int nfactors, double *phi_out)
{
    const int ng = h->ng, dim = h->dim; const long nef = h->ne, nPf = h->nPhi;
    if (nfactors <= 0) { memcpy(phi_out, h->phi, sizeof(double) * ng * nPf); return 1.0; }
    int rx = nfactors > 0 ? (factors[0] > 1 ? factors[0] : 1) : 1;
    int ry = (nfactors > 1 && dim >= 2) ? (factors[1] > 1 ? factors[1] : 1) : 1;
    int rz = (nfactors > 2 && dim >= 3) ? (factors[2] > 1 ? factors[2] : 1) : 1;
    if (h->nx % rx || h->ny % ry || h->nz % rz) { memcpy(phi_out, h->phi, sizeof(double) * ng * nPf); return 1.0; }
    const int nxc = h->nx / rx, nyc = h->ny / ry, nzc = h->nz / rz; const long nec = (long)nxc * nyc * nzc;
    double *xc = (double *)malloc(sizeof(double) * (nxc + 1)), *yc = (double *)malloc(sizeof(double) * (nyc + 1)), *zc = (double *)malloc(sizeof(double) * (nzc + 1));
    for (int i = 0; i <= nxc; ++i) xc[i] = h->xb[i * rx];
    if (dim >= 2) for (int j = 0; j <= nyc; ++j) yc[j] = h->yb[j * ry]; else yc[0] = 0.0;
    if (dim >= 3) for (int k = 0; k <= nzc; ++k) zc[k] = h->zb[k * rz]; else zc[0] = 0.0;
    nfo_t *c = nfo_create(0, 0, ng, nxc + 1, xc, dim >= 2 ? nyc + 1 : 1, yc, dim >= 3 ? nzc + 1 : 1, zc);
    free(xc); free(yc); free(zc);
    nfo_set_linear_solver(c, h->solver_type);
    nfo_set_tol(c, h->tol_keff * 10.0, h->tol_flux * 10.0, h->tol_L2, h->max_outer / 2, h->max_inner);
    for (int a = 0; a < 8; ++a) if (h->bc_set[a]) nfo_set_bc(c, a, h->bc_type[a], h->bc_val[a]);
    for (int g = 0; g < ng; ++g)
        for (int kz = 0; kz < nzc; ++kz) for (int ky = 0; ky < nyc; ++ky) for (int kx = 0; kx < nxc; ++kx) {
            const long ec = (long)kz * nyc * nxc + (long)ky * nxc + kx;
            double vt = 0, sD = 0, sR = 0, sN = 0, sK = 0, sC = 0; double sS[64]; for (int q = 0; q < ng; ++q) sS[q] = 0.0;
            for (int sz = 0; sz < rz; ++sz) for (int sy = 0; sy < ry; ++sy) for (int sx = 0; sx < rx; ++sx) {
                int ixf = kx * rx + sx, iyf = ky * ry + sy, izf = kz * rz + sz;
                long ef = (long)izf * h->ny * h->nx + (long)iyf * h->nx + ixf;
                double vol = h->xb[ixf + 1] - h->xb[ixf];
                if (dim >= 2) vol *= h->yb[iyf + 1] - h->yb[iyf];
                if (dim >= 3) vol *= h->zb[izf + 1] - h->zb[izf];
                vt += vol; sD += vol * h->D[g * nef + ef]; sR += vol * h->SigR[g * nef + ef];
                sN += vol * h->NSF[g * nef + ef]; sK += vol * h->KSF[g * nef + ef]; sC += vol * h->Chi[g * nef + ef];
                for (int gp = 0; gp < ng; ++gp) sS[gp] += vol * h->SigS[((long)g * ng + gp) * nef + ef];
            }
            c->D[g * nec + ec] = sD / vt; c->SigR[g * nec + ec] = sR / vt; c->NSF[g * nec + ec] = sN / vt;
            c->KSF[g * nec + ec] = sK / vt; c->Chi[g * nec + ec] = sC / vt;
            for (int gp = 0; gp < ng; ++gp) c->SigS[((long)g * ng + gp) * nec + ec] = sS[gp] / vt;
        }
    nfo_build(c);
    double kc = nfo_solve_keff(c, 0, NULL, 0, 0);
    h->coarse_outer = c->last_outer;
    memset(phi_out, 0, sizeof(double) * ng * nPf);
    const int dpe = h->nloc;
    for (int g = 0; g < ng; ++g)
        for (int iz = 0; iz < h->nz; ++iz) for (int iy = 0; iy < h->ny; ++iy) for (int ix = 0; ix < h->nx; ++ix) {
            long ef = (long)iz * h->ny * h->nx + (long)iy * h->nx + ix;
            long ec = (long)(iz / rz) * nyc * nxc + (long)(iy / ry) * nxc + ix / rx;
            phi_out[g * nPf + ef * dpe] = c->phi[g * nec + ec];
        }
    nfo_destroy(c);
    return kc;
}
