/*
 * neutfem_hip.h -- C ABI of the MI355X (gfx950) hot path of neutfem_amd.
 *
 * This is the drop-in boundary: plain C types, opaque handle, int status
 * (0 = ok, <0 = error; text via nf_last_error()).  No exceptions, no torch or
 * Eigen types cross it.  Each entry point cites the reference interface it
 * replaces (paths relative to the reference repository jujuC31/NeutFEM).
 * Host pointers are marked _host, device pointers _dev (fp64 everywhere).
 *
 * Layouts (identical to the reference's flat arrays, include/NeutFEM.hpp:365-388):
 *   cell e = iz*nx*ny + iy*nx + ix                       (src/FEM.cpp:89-91)
 *   XS      [g*N + e]            SigS [(g_to*ng + g_from)*N + e]
 *   phi     [g*n_phi + e*n_loc + l]                      (n_loc = 1 for P0)   host arrays (nf_set_phi/nf_get_phi)
 *   *_dev vectors (nf_schur_apply, nf_solve_group): device DOF order [l*N + e] (moment-major; same thing for P0)
 *   J       [g*n_J + f]   faces x | y | z | bubbles      (src/FEM.cpp:264-334)
 */
#ifndef NEUTFEM_HIP_H
#define NEUTFEM_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct nf_solver *nf_handle;

enum { NF_OK = 0, NF_ERR_ARG = -1, NF_ERR_NO_DEVICE = -2, NF_ERR_HIP = -3, NF_ERR_UNSUPPORTED = -4,
       NF_ERR_STATE = -5, NF_ERR_NUMERIC = -6,
       NF_ERR_REMOTE = -7,   /* multi-rank team: another rank hit an error inside a solve; every rank returns at the same iteration */
       NF_ERR_COMM = -8 };   /* multi-rank team: a collective did not complete within NEUTFEM_COMM_TIMEOUT_S (a peer is gone or stuck).  The team is
                              * dead afterwards: its streams may hold collectives that never finish, so every further collective call on it
                              * returns NF_ERR_COMM at once and nf_destroy releases the host objects WITHOUT waiting for or freeing device
                              * resources.  The caller must end the process with a non-zero code (a fresh exit -- never an exec of another
                              * program from this process) */
/* BCType, include/NeutFEM.hpp:51-57 */
enum { NF_BC_DIRICHLET = 0, NF_BC_NEUMANN = 1, NF_BC_MIRROR = 2, NF_BC_ROBIN = 3, NF_BC_PERIODIC = 4 };

const char *nf_last_error(void);
/* number of visible HIP devices (0 when there is none); never throws */
int nf_device_count(void);

/* NeutFEM::NeutFEM (src/NeutFEM.cpp:82-300) + CartesianMesh (src/FEM.cpp:23-83) + FESpace
 * (src/FEM.cpp:177-259): mesh from break arrays (a 1-entry y/z array = inactive dimension),
 * RT/P orders clamped like the reference, device = HIP ordinal. */
int nf_create(int rt_order, int p_order, int ng,
              int nxb, const double *xb_host, int nyb, const double *yb_host, int nzb, const double *zb_host,
              int device, nf_handle *out);
int nf_destroy(nf_handle h);

/* ---- multi-GPU: z-slab decomposition (no reference counterpart; the reference is single-process) ------------
 * A slab is an nf_handle over the z-planes [k0,k1) of the global mesh: same x/y breaks, zb_slab = the slab's
 * own k1-k0+1 z-breaks, interface_below/above = 1 where another slab continues the mesh (that side is then not
 * a domain boundary).  Slabs living in one process on one device are chained with nf_link_slabs (bottom to top);
 * slabs of other processes are reached through RCCL: rank 0 calls nf_comm_unique_id, the 128 bytes are broadcast
 * by the launcher (torch.distributed / MPI / a file), every rank calls nf_comm_init.  Rank r must hold the slabs
 * just above rank r-1's.  After that nf_solve_keff / nf_time_schur_apply on any slab of the team run the whole
 * team; nf_set_phi / nf_get_phi / nf_upload_xs / nf_build stay per slab.
 * Collective calls (every rank, same order): nf_comm_init, nf_solve_keff, nf_solve_adjoint, nf_get_J, nf_build_diagonal_cache,
 * nf_team_schur_apply, nf_time_schur_apply.  Decisions that could differ between ranks (slab too thin for the
 * separator sweeps, coarse factors that do not divide a slab) are all-reduced first, so all ranks return the same
 * error.  Any RTk-Pm order; slabs need >= 4 z-planes (>= ~30 for the single-exchange fast path).
 * NEUTFEM_RCCL_LIB overrides the RCCL library path (tests use a host-staged stand-in to run several ranks on one GPU). */
int nf_create_slab(int rt_order, int p_order, int ng,
                   int nxb, const double *xb_host, int nyb, const double *yb_host, int nzb_slab, const double *zb_slab_host,
                   int interface_below, int interface_above, int device, nf_handle *out);
int nf_link_slabs(nf_handle *handles, int n);
int nf_comm_unique_id(void *id128_host);
int nf_comm_init(nf_handle h, const void *id128_host, int nranks, int rank);
/* what carries the data path: *comm_ranks = ncclCommCount of the live communicator (0: none, -1: the library lacks the entry),
 * lib_path = file the RCCL symbols were resolved from ("" without a communicator).  bench.py prints both (rccl_ranks, transport). */
int nf_comm_info(nf_handle h, int *comm_ranks, char *lib_path_host, size_t len);
/* diagnostic: grouped ncclSend/ncclRecv to the own rank on the comm stream + all-reduce(max) through the loaded RCCL */
int nf_comm_selftest(nf_handle h);
/* Schur apply on every local slab: x_dev[i] / y_dev[i] = device vectors of local slab i */
int nf_team_schur_apply(nf_handle h, int g, const double *const *x_dev, double *const *y_dev);

/* sizes: "dim","nx","ny","nz","ne","ng","n_phi","n_J","n_loc","last_outer","last_cg_total",
 * "coarse_outer","device","n_local_slabs","n_ranks","rank","cg_reductions" (cross-rank reductions per CG iteration of the last CG solve on a
 * team: 1 single-reduction CG, 2 reference recurrence, 0 undivided mesh),"vec_reduce","xchg_comm" ; returns -1 for an unknown key */
long nf_info(nf_handle h, const char *key);

/* NeutFEM::SetBC (src/NeutFEM.cpp:337-345): attr per BoundaryID (include/NeutFEM.hpp:73-91).
 * Only DIRICHLET changes the operator (src/NeutFEM.cpp:1328-1456); the rest is natural/ignored. */
int nf_set_bc(nf_handle h, int attr, int bc_type);

/* Public XS members D_data_, SigR_data_, NSF_data_, Chi_data_, SigS_data_
 * (include/NeutFEM.hpp:373-388) -> device.  Host arrays in the reference layouts. */
int nf_upload_xs(nf_handle h, const double *D_host, const double *SigR_host, const double *NSF_host,
                 const double *Chi_host, const double *SigS_host);

/* NeutFEM::BuildMatrices (src/NeutFEM.cpp:402-457): AssembleA/B/C, ApplyDirichletToA,
 * AssembleFissionMatrix, AssembleScatteringMatrix as closed-form per-cell coefficients, plus the
 * factorisation SchurSolver::SetMatrices does per solve (A_lu_solver_.compute, src/solvers.cpp:163),
 * done ONCE here as per-grid-line LDL^T.  Invalidates the diagonal cache, keeps the warm start. */
int nf_build(nf_handle h);

/* SchurSolver::SchurProduct (src/solvers.cpp:535-547): y = C_g x + B A_g^-1 B^T x. */
int nf_schur_apply(nf_handle h, int g, const double *x_dev, double *y_dev);

/* SchurSolver::SolveSchurImplicit (src/solvers.cpp:577-636): CG from x0 = 0, stop when
 * ||r||^2 < tol^2 ||b||^2 or after maxit iterations. its/res may be NULL. */
int nf_solve_group(nf_handle h, int g, const double *rhs_dev, double *phi_dev, double tol, int maxit,
                   int *its, double *res);

/* NeutFEM::BuildDiagonalSchurCache (src/NeutFEM.cpp:483-597); S_inv for group g is copied to
 * sinv_host (N doubles) when not NULL. */
int nf_build_diagonal_cache(nf_handle h);
int nf_get_diagonal_cache(nf_handle h, int g, double *sinv_host);

typedef struct nf_keff_opts {
    double tol_keff, tol_flux;      /* NeutFEM::SetTolerance, src/NeutFEM.cpp:327-335 */
    int max_outer, max_inner;
    int use_coarse_init;            /* SolveKeff arguments, src/wrapper.cpp:598-603 */
    int coarse_factors[3];
    int n_coarse_factors;
    int use_diagonal_solver;
    int solver_type;                /* LinearSolverType 0..9 (include/solvers.hpp:176-190) */
    int solver_type_pushed;         /* 0: set_linear_solver never called -> DIRECT_LU */
    int profile;                    /* 1: bracket every Schur-apply pass with HIP events */
    int use_cmfd;                   /* SolveKeff's 4th argument: CMFD correction from outer 2, replaces Chebyshev (:1750-1761,1786) */
} nf_keff_opts;

/* NeutFEM::SolveKeff(bool, vector<int>, bool, bool) (src/NeutFEM.cpp:1627-1815) incl. SolveCoarse
 * (src/NeutFEM.cpp:2380-2611), ChebyshevAccel (src/solvers.cpp:664-756) and SolveGroupInternal
 * (src/NeutFEM.cpp:2084-2105).  State kept between calls exactly like the reference:
 * current flux (nf_set_phi/nf_get_phi) and has_valid_keff_/last_keff_direct_. */
int nf_solve_keff(nf_handle h, const nf_keff_opts *opts, double *keff, int *n_outer);

/* No reference counterpart (the reference is one process): outer iterations the running or last nf_solve_keff has completed
 * (the `it` of the loop at src/NeutFEM.cpp:1694).  May be called from another thread while nf_solve_keff runs -- the watchdog
 * of a multi-rank job tells a slow solve from one whose peers are gone.  On a multi-rank team a rank that fails inside a solve
 * makes every rank return (the failing one with its own code, the others with NF_ERR_REMOTE); a collective that does not complete
 * within NEUTFEM_COMM_TIMEOUT_S seconds (default 120) returns NF_ERR_COMM. */
int nf_progress(nf_handle h, long *outers_done);
/* The progress line of the outer loop (src/NeutFEM.cpp:1791-1796: "It n : k = ... dk = ... dphi = ..." every 5th iteration) WHILE the solve
 * runs: fn(user, it, keff, dk, dphi) is called on the calling thread after every outer iteration of the host-driven loop -- the path every
 * mesh beyond ~28 k unknowns per group takes, where a solve lasts seconds to minutes.  The in-kernel paths (resident, one-XCD, diagonal device
 * loop: milliseconds) have no host between their outers; their lines come from nf_get_history afterwards.  fn = NULL removes it. */
typedef void (*nf_progress_fn)(void *user, int outer, double keff, double dk, double dphi);
int nf_set_progress_callback(nf_handle h, nf_progress_fn fn, void *user);

/* CMFD acceleration (src/NeutFEM.cpp:662-1017, include/NeutFEM.hpp:119-143,232-235): NeutFEM::InitializeCMFD
 * (D-tilde for every direction, D-hat = 0; idempotent until the next nf_build), SetCMFDRelaxation, and a probe that
 * downloads D-tilde / D-hat of (group, direction) in the reference's face numbering (either pointer may be NULL).
 * The correction itself runs inside nf_solve_keff when opts.use_cmfd is set.  On a slab team nf_initialize_cmfd is collective
 * (the D-tilde of an interface z face needs the neighbouring slab's edge cells) and the PCG exchanges one plane per interface
 * and iteration. */
int nf_initialize_cmfd(nf_handle h);
int nf_set_cmfd_relaxation(nf_handle h, double omega);
int nf_get_cmfd_coefficients(nf_handle h, int g, int dir, double *dtilde_host, double *dhat_host);

/* NeutFEM::SolveAdjoint(normalize_to_direct, use_direct_keff) (src/NeutFEM.cpp:1877-2082, BuildFissionRHSAdjoint
 * :1568-1589): tolerances / solver type from opts; the adjoint flux is fetched with nf_get_phi_adj (host layout of phi). */
int nf_solve_adjoint(nf_handle h, const nf_keff_opts *opts, int normalize_to_direct, int use_direct_keff, double *keff_adj, int *n_outer);
int nf_get_phi_adj(nf_handle h, double *phi_adj_host);

/* NeutFEM::SolveCoarse (src/NeutFEM.cpp:2380-2611): returns k_coarse and the prolonged flux
 * (ng*n_phi doubles, host) without touching the fine solution. */
int nf_solve_coarse(nf_handle h, const nf_keff_opts *opts, double *k_coarse, double *phi_host);
/* the two halves of SolveCoarse on their own (undivided meshes): nf_coarsen returns a BUILT RT0-P0 handle on the mesh merged by
 * (rx, ry, rz) with block-mean cross sections (src/NeutFEM.cpp:2409-2556; the caller solves and destroys it), nf_prolong
 * injects the coarse handle's current flux into the fine handle's (piecewise constant, higher moments zero, :2585-2606) */
int nf_coarsen(nf_handle h, int rx, int ry, int rz, nf_handle *coarse);
int nf_prolong(nf_handle coarse, nf_handle fine);

/* Sol_Phi_ / Sol_J_ (include/NeutFEM.hpp:380-388) and NeutFEM::ResetFlux (src/NeutFEM.cpp:347-354) */
int nf_set_phi(nf_handle h, const double *phi_host);
int nf_get_phi(nf_handle h, double *phi_host);
int nf_get_J(nf_handle h, double *J_host);   /* on a slab: the slab's own faces (shared interface planes appear in both neighbours); collective */
int nf_reset_flux(nf_handle h);
int nf_set_warm_state(nf_handle h, int has_valid_keff, double last_keff);
int nf_get_warm_state(nf_handle h, int *has_valid_keff, double *last_keff);

/* per-outer history of the last nf_solve_keff: k, dk, dphi (n_outer each), cg (n_outer*ng) */
int nf_get_history(nf_handle h, double *k, double *dk, double *dphi, int *cg, int cap_outer);

/* profiling: kernels timed with HIP events on the solver's stream (opts.profile / nf_time_schur_apply).
 * name in {"schur_x","schur_y","schur_z","schur_apply","schur_z1"}: number of timed launches and their total ms. */
int nf_profile_get(nf_handle h, const char *name, long *count, double *total_ms);
int nf_profile_reset(nf_handle h);
/* the same counters, with the iteration counts of the last solve, as one JSON object */
int nf_timers(nf_handle h, char *json_buf, size_t len);
/* times `reps` back-to-back Schur applies on group g (random x) with HIP events; average ms per apply */
int nf_time_schur_apply(nf_handle h, int g, int reps, double *avg_ms);
/* LocalMatrices::Compute(e, D, Sigma) (src/FEM.cpp:748-953) on the device, literally: the dense A_loc (n_Jloc x n_Jloc), B_loc
 * (n_loc x n_Jloc) and C_loc (n_loc x n_loc) of `n_elems` elements of group g by the reference's tensor Gauss quadrature (order
 * 2 max(k, m) + 3 with its 7 -> 5-point fallback, include/FEM.hpp:115-120), row-major like GetA / GetB / GetC, D = D_g(e),
 * Sigma = SigR_g(e); local DOF order of src/FEM.cpp:729-745.  One element per workgroup, quadrature points and basis values
 * staged in LDS.  The solver itself never forms these matrices (closed forms); this entry exists to check them
 * against the quadrature on the device and to measure the one dense contraction of the code base in both forms:
 * variant 0 = fp64 FMA, 1 = v_mfma_f64_16x16x4_f64.  reps >= 1 timed launches, average ms in *avg_ms (may be NULL).
 * Needs nf_upload_xs (not nf_build). */
int nf_local_matrices(nf_handle h, int g, int n_elems, const int *elems_host, double *A_host, double *B_host, double *C_host,
                      int variant, int reps, double *avg_ms);
/* HBM microbenchmark: `reps` device-to-device streaming copies of `bytes` (read + write counted) -> GB/s; the
 * roofline is reported against the 8 TB/s spec and against this measured figure */
int nf_time_device_copy(nf_handle h, size_t bytes, int reps, double *gbps);

/* tuning knobs (no reference counterpart):
 *   "s_tx" (0 auto, 8, 16, 32, 64) columns per block and "s_seg" (0 auto, 4, 8, 16, 32) cells per register segment of the y/z line
 *   kernels; "s_wsmin" segments per line from which whole wavefronts scan the segment summaries (default 64);
 *   "xcd" gives each XCD one contiguous range of tiles: bit 0 the y passes, bit 1 the z passes; default -1 = the y passes of meshes
 *   beyond "nt_min_cells" (256^3: y pass 133 -> 123 us; the z passes lose with it, smaller meshes see nothing);
 *   "cg_batch" CG iterations launched between host checks of the device-side stop flag (0 = automatic);
 *   "cg_fuse" (default 1) folds x += alpha p, p = r + beta p into the next pass that reads p (bit-identical iterates);
 *   "cg_lean" (default 1) lets the consumer of a reduction derive the CG scalars itself: no finalize launches on undivided meshes of at
 *   most "cg_lean_max_cells" cells (default 4 Mi; "cg_lean_grid" = blocks of the residual update), no k_cg_logic launches on slab teams;
 *   "cg_fuse3" (default 1) runs the x, y and z passes of an apply as ONE launch (two launches per CG iteration) on undivided meshes of
 *   at most "cg_fuse3_max_cells" cells (default 400 000: above, the four-launch lean iteration is faster);
 *   "cg_xcd" (default 1) runs every CG solve of an undivided mesh (any order, x lines of at most 128 cells) with "cg_xcd_min_cells" ..
 *   "cg_xcd_max_cells" unknowns per group (default 2000 .. 28000) as ONE launch on the workgroups of XCD "cg_xcd_id" (default 0; 8..15 name none: the solve falls back,
 *   for tests), "cg_xcd_groups" (default 32) workgroups per XCD being launched; nf_info "xcd_solves" counts its solves, "xcd_refused"
 *   the times its workgroups did not assemble and the launch path took over; "keff_xcd" (default 1) runs the whole SolveKeff of such a mesh
 *   (iterative full-Schur path, no CMFD) in one launch of the same kind (nf_info "last_path" 3);
 *   "resident" (default 1) runs the whole SolveKeff of an undivided mesh with at most "resident_max_dofs" flux DOFs per group (default
 *   2500) in one workgroup and one launch; "resident_lds" (default 1) keeps its CG vectors and factors in LDS as far as they fit;
 *   "resident_serial" (default 1): meshes whose moments, factors and directions' contributions all fit in LDS run one lane per
 *   (direction, transverse mode, line) with serial sweeps instead of the segmented scans (nf_info "last_resident_serial"), up to
 *   "resident_serial_max_dofs" flux DOFs per group (default 5120, the structural limit; "resident_max_dofs" lowers it too);
 *   "resident_two_sided" (default 1): lines of at least 4 cells are swept by two lanes that meet in the middle;
 *   "direct_max_dofs" (default 6000, at most 8192): explicit-S branch with a dense S^-1 up to this many flux DOFs per group, beyond it
 *   CG to 1e-14 stands in (nf_info "direct_standin_unconverged" counts group solves that did not get there);
 *   "host_pub" (default 1): the host reads the CG scalars and the outer iteration's sums from a mapped host page that a one-thread
 *   kernel fills (polled), not through a device-to-host copy and a stream drain;
 *   "prof_every" (default 8): a profiled solve (nf_keff_opts::profile) brackets every n-th Schur apply with events, not each one;
 *   "nt_loads" (default 1): on undivided RT0-P0 meshes of more than "nt_min_cells" cells (default 8 000 000: from there on the streams no longer live in the
 *   256 MB memory-side cache between launches) the direction passes read their streams with non-temporal loads;
 *   "outer_dev" (default 1) keeps the outer loop of the diagonal-Schur path on the device (undivided mesh, no CMFD);
 *   "sep_fold" (default 1): slab teams form the separator values inside the accumulation pass of the z lines (no k_separators launch);
 *   "cg_single_reduce" (default -1 = while the largest slab of the team has at most "cg_single_reduce_max_cells" cells, default unlimited; 1 always,
 *   0 never): RT0-P0 slab teams run the CG with ONE cross-rank reduction per iteration (p.q, q.q, r.q and the measured |r|^2 in one all-reduce
 *   of five doubles; r -= alpha q rides in the endpoint pass of the z lines; nf_info "cg_reductions" = 1) instead of the reference
 *   recurrence's two ("vec_reduce" then picks between all-reducing the block partials themselves and k_finalize + scalars);
 *   "endpoint_weights" (default 1): that CG's endpoint pass forms the chain-end responses of every z line as weighted sums of the line's cells
 *   (weights measured once per BuildMatrices with the chain solve itself: nz launches per group, 16 B per cell and group of HBM) instead of
 *   solving the chain -- a streaming kernel with the deferred CG update in the same sweep (nf_info "endpoint_weights");
 *   "xchg_comm" (default 0): the interface planes travel on a communicator of their own (created collectively before the next solve;
 *   nf_info "xchg_comm") instead of sharing the one the all-reduces use.
 * nf_info keys beyond the mesh sizes: "last_path" (0 host-driven outer loop, 1 diagonal device loop, 2 resident kernel, 3 one-XCD kernel),
 * "last_direct" (0 CG as configured, 1 dense S^-1, 2 CG to 1e-14 standing in). */
int nf_set_option(nf_handle h, const char *key, long value);

/* raw device-memory helpers so callers without torch can drive the *_dev entry points */
/* free / total HBM of a device (hipMemGetInfo): sizing of decompositions, leak checks */
int nf_mem_info(int device, size_t *free_bytes, size_t *total_bytes);
int nf_dev_alloc(nf_handle h, size_t bytes, void **ptr_dev);
int nf_dev_free(nf_handle h, void *ptr_dev);
int nf_memcpy_h2d(nf_handle h, void *dst_dev, const void *src_host, size_t bytes);
int nf_memcpy_d2h(nf_handle h, void *dst_host, const void *src_dev, size_t bytes);
int nf_synchronize(nf_handle h);
/* the hipStream_t every kernel of this handle is launched on */
void *nf_stream(nf_handle h);

#ifdef __cplusplus
}
#endif
#endif
