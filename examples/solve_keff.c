/* Plain-C consumer of the drop-in boundary (include/neutfem_hip.h): builds a small homogeneous 3D problem,
 * runs BuildMatrices + SolveKeff on the MI355X and prints k-eff.  Compiled as C99 by tests/test_boundary.py
 * (header hygiene: no C++ in the ABI); on a machine without a HIP device it reports the library's error and exits 2.
 *
 *   gcc -std=c99 -Iinclude examples/solve_keff.c -Lneutfem_amd/lib -lneutfem_hip -Wl,-rpath,$PWD/neutfem_amd/lib -o solve_keff
 */
#include <stdio.h>
#include <stdlib.h>
#include "neutfem_hip.h"

#define N 24

int main(void)
{
    double brk[N + 1];
    const long ne = (long)N * N * N;
    double *D = malloc(sizeof(double) * 2 * ne), *SigR = malloc(sizeof(double) * 2 * ne), *NSF = malloc(sizeof(double) * 2 * ne);
    double *Chi = malloc(sizeof(double) * 2 * ne), *SigS = calloc(4 * ne, sizeof(double));
    nf_handle h = NULL;
    nf_keff_opts o;
    double k = 0.0; int n_outer = 0, a;
    long e;

    for (a = 0; a <= N; ++a) brk[a] = 5.0 * a;                  /* 120 cm cube, 5 cm cells */
    for (e = 0; e < ne; ++e) {                                    /* two groups, down-scatter 1 -> 2 */
        D[e] = 1.5; D[ne + e] = 0.4;
        SigR[e] = 0.03; SigR[ne + e] = 0.08;
        NSF[e] = 0.005; NSF[ne + e] = 0.135;
        Chi[e] = 1.0; Chi[ne + e] = 0.0;
        SigS[(1 * 2 + 0) * ne + e] = 0.02;                       /* [g_to * ng + g_from] */
    }
    if (nf_create(0, 0, 2, N + 1, brk, N + 1, brk, N + 1, brk, 0, &h) != NF_OK) {
        fprintf(stderr, "nf_create: %s\n", nf_last_error());
        return 2;
    }
    for (a = 1; a <= 6; ++a) nf_set_bc(h, a, NF_BC_DIRICHLET);
    if (nf_upload_xs(h, D, SigR, NSF, Chi, SigS) != NF_OK || nf_build(h) != NF_OK) {
        fprintf(stderr, "build: %s\n", nf_last_error());
        return 1;
    }
    o.tol_keff = 1e-7; o.tol_flux = 1e-6; o.max_outer = 300; o.max_inner = 1000;
    o.use_coarse_init = 1; o.coarse_factors[0] = o.coarse_factors[1] = o.coarse_factors[2] = 2; o.n_coarse_factors = 3;
    o.use_diagonal_solver = 0; o.solver_type = 6; o.solver_type_pushed = 1; o.profile = 0; o.use_cmfd = 0;
    if (nf_solve_keff(h, &o, &k, &n_outer) != NF_OK) {
        fprintf(stderr, "solve: %s\n", nf_last_error());
        return 1;
    }
    printf("k-eff = %.8f after %d outer iterations\n", k, n_outer);
    nf_destroy(h);
    free(D); free(SigR); free(NSF); free(Chi); free(SigS);
    return 0;
}
