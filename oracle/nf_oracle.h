/*
 * nf_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the jujuC31/NeutFEM hot path (RTk-Pm assembly ->
 * per-group Schur-complement solve -> multigroup power iteration).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the shipped HIP path never links or calls it.
 *
 * PARITY STATUS.  Checked against every fixture the reference's drivers hold for this path -- the literature
 * k_ref scalars and the two published assembly-power tables (tests/iaea2d/iaea2d.py:479-504,
 * tests/koeberg2d/koeberg2d.py:553-576): RT0-P0 converges onto them with the mesh (IAEA-2D 8x8: -0.7 pcm, 1.4 % max
 * assembly power; tests/test_oracle.py).  That is discretisation accuracy.  At ROUNDING level the oracle is
 * "parity unpinned" by the reference's own tests: the reference ships no vectors for phi / J / iteration counts and
 * cannot be built here (Eigen3 absent).  There it is pinned by (i) closed-form local matrices and
 * (ii) an independent numpy/scipy explicit-sparse restatement (oracle/ref_scipy.py).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef NF_ORACLE_H
#define NF_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct nfo nfo_t;

/* include/NeutFEM.hpp:51-57 */
enum { NFO_BC_DIRICHLET = 0, NFO_BC_NEUMANN = 1, NFO_BC_MIRROR = 2, NFO_BC_ROBIN = 3, NFO_BC_PERIODIC = 4 };

/* src/NeutFEM.cpp:82-300 (constructor defaults) */
nfo_t *nfo_create(int rt_order, int p_order, int ng,
                  int nxb, const double *xb, int nyb, const double *yb, int nzb, const double *zb);
void nfo_destroy(nfo_t *h);

/* integer properties: "dim","nx","ny","nz","ne","ng","k","m","nf","ni","nloc","nJloc",
 * "n_phi","n_J","n_Jx","n_Jy","n_Jz","nq","last_outer","last_cg_total","coarse_outer" */
long nfo_info(const nfo_t *h, const char *key);
/* host arrays (owned by the handle, writable): "D","SigR","NSF","KSF","Chi","SRC","SigS",
 * "phi","J","phi_adj","hist_k","hist_dk","hist_dphi","hist_cg" ; *n receives the length */
double *nfo_array(nfo_t *h, const char *name, long *n);

void nfo_set_bc(nfo_t *h, int attr, int type, double value);                 /* src/NeutFEM.cpp:337-345 */
void nfo_set_tol(nfo_t *h, double tol_keff, double tol_flux, double tol_L2,
                 int max_outer, int max_inner);                               /* src/NeutFEM.cpp:327-335 */
void nfo_set_linear_solver(nfo_t *h, int type);                               /* src/NeutFEM.cpp:322-325 */
void nfo_reset_flux(nfo_t *h);                                                /* src/NeutFEM.cpp:347-354 */
void nfo_set_refactor_each_solve(nfo_t *h, int on);  /* mimic solvers.cpp:163 cost (timing only) */

/* literal quadrature local matrices, src/FEM.cpp:748-953.  A: nJloc^2, B: nloc*nJloc, C: nloc^2 (row-major) */
void nfo_local_matrices(const nfo_t *h, int e, double D, double Sigma, double *A, double *B, double *C);
/* src/FEM.cpp:955-999 / :1001-1008 */
void nfo_global_J_indices(const nfo_t *h, int ix, int iy, int iz, int *idx);
void nfo_global_phi_indices(const nfo_t *h, int ix, int iy, int iz, int *idx);

/* NOT in the reference: cut the cells with mask != 0 (ne bytes, NULL = none) out of the domain and put J.n = phi / inv_alpha on the
 * faces they share with kept cells (before nfo_build).  Only the Schur path honours it; see nf_oracle.c. */
void nfo_set_void(nfo_t *h, const unsigned char *mask, double inv_alpha);
int nfo_build(nfo_t *h);                                                      /* src/NeutFEM.cpp:402-457 */
/* y = C x + B A^-1 B^T x for group g, src/solvers.cpp:535-547 */
void nfo_schur_apply(nfo_t *h, int g, const double *x, double *y);
/* SchurSolver::Solve, src/solvers.cpp:203-240 ; returns CG iterations */
int nfo_solve_group(nfo_t *h, int g, const double *rhs, double *phi, double *J);
/* src/NeutFEM.cpp:1627-1815 */
double nfo_solve_keff(nfo_t *h, int use_coarse_init, const int *factors, int nfactors,
                      int use_diagonal_solver);
/* same with use_cmfd (CMFD acceleration, src/NeutFEM.cpp:662-1017,1750-1761); nfo_set_cmfd_relaxation = set_cmfd_relaxation */
double nfo_solve_keff_cmfd(nfo_t *h, int use_coarse_init, const int *factors, int nfactors,
                           int use_diagonal_solver, int use_cmfd);
void nfo_set_cmfd_relaxation(nfo_t *h, double omega);
/* probes: D-tilde/D-hat (src/NeutFEM.cpp:723-869; returns the face count) and one UpdateDhat + ApplyCMFDCorrection
 * (:823-1017) on the handle's current phi / J with the given total fission source (n_phi) and k */
long nfo_cmfd_coefficients(nfo_t *h, int g, int dir, double *dtilde, double *dhat);
void nfo_cmfd_probe(nfo_t *h, int g, const double *total_fiss, double keff, double *corr);
/* src/NeutFEM.cpp:2380-2611 ; phi_out has ng*n_phi entries */
double nfo_solve_coarse(nfo_t *h, const int *factors, int nfactors, double *phi_out);
/* src/NeutFEM.cpp:483-597 ; returns S_inv for group g (ne entries) or NULL */
const double *nfo_diag_cache(nfo_t *h, int g);
double nfo_last_keff(const nfo_t *h);
/* src/NeutFEM.cpp:1877-2082 ; the adjoint flux is the array "phi_adj" */
double nfo_solve_adjoint(nfo_t *h, int normalize_to_direct, int use_direct_keff);
double nfo_last_keff_adjoint(const nfo_t *h);

#ifdef __cplusplus
}
#endif
#endif
