// neutfem_hip.hip -- C-ABI implementation (include/neutfem_hip.h) over the gfx950 kernels.
//
// Host logic mirrors the reference's control flow (NeutFEM::SolveKeff / SolveCoarse /
// SchurSolver::Solve) while every vector stays resident in HBM; per outer iteration the host
// reads back 4 doubles, per CG solve one small struct every few iterations.
//
// Multi-GPU: the mesh is cut into z-slabs.  A "team" is the set of slabs of one global problem that
// live in this process (one per GPU in production; several on one GPU in the loopback used for
// testing) plus an RCCL communicator to the slabs of the other processes.  Every phase of the solve
// loops over the local slabs; the only data-path communication is (i) one plane of fp64 per slab
// interface per Schur apply (partition method for the z-line solves) and (ii) all-reduces of 1-3
// scalars, all enqueued on the team's stream -- the host never waits for them.
#include "../../include/neutfem_hip.h"
#include "nf_kernels.h"
#include "nf_assembly.h"

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace nf;

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...)
{
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf; return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(NF_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NFCHK(x) do { int r_ = (x); if (r_ != NF_OK) return r_; } while (0)

static const int RED_GRID = 1024;          // fixed grid of the streaming/reduction kernels
static const int MAX_LOCAL_SLABS = 16;

// event-timed launches of one pass.  Launches queued behind a converged CG solve exit at once (CgScalars::done): they are no
// work and must not dilute the average, and an occasional preempted launch must not distort it either, so the figures are
// taken over the samples of at least a quarter of the MEDIAN duration.
struct ProfSlot {
    std::vector<float> samples;
    void stats(long *count, double *ms, long *skipped) const
    {
        *count = 0; *ms = 0.0; *skipped = 0;
        if (samples.empty()) return;
        std::vector<float> v(samples); std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        const double thr = 0.25 * v[v.size() / 2];
        for (float x : samples) { if (x < thr) ++*skipped; else { ++*count; *ms += x; } }
    }
};

// ---- RCCL, resolved lazily with dlopen so that single-GPU use never loads it -------------------
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;          // optional: used when a peer stops answering (comm_timeout_s)
    int (*CommCount)(const ncclComm_t, int *) = nullptr;   // optional: nf_comm_info reports what the LIVE communicator says, not what the launcher claimed
    char path[512] = { 0 };                          // file the symbols were actually resolved from (dladdr)
};
static Rccl g_rccl;
// NEUTFEM_TRACE_COMM=1: one line on stderr per collective this rank issues (debugging the order of collectives across ranks)
static bool trace_comm() { static int on = -1; if (on < 0) { const char *e = getenv("NEUTFEM_TRACE_COMM"); on = e && atoi(e) ? 1 : 0; } return on == 1; }
#define TRACE_COMM(...) do { if (trace_comm()) { fprintf(stderr, "[comm] " __VA_ARGS__); fputc('\n', stderr); } } while (0)
static const int NCCL_DOUBLE = 8, NCCL_SUM = 0, NCCL_MAX = 2;   // ncclFloat64 / ncclSum / ncclMax in rccl.h
#define NCCLCHK(x) do { int r_ = (x); if (r_ != 0) return fail(NF_ERR_HIP, "%s failed: %s", #x, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); } while (0)

static int rccl_load()
{
    if (g_rccl.lib) return NF_OK;
    // the ROCm install this library was built against first; symbols are taken from THIS handle only (dlsym), so a
    // second RCCL copy that a framework may have mapped (e.g. the one bundled with PyTorch) is never mixed in
    const char *env = getenv("NEUTFEM_RCCL_LIB");
    const char *names[] = { env ? env : "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so" };
    for (const char *n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_rccl.lib) break; }
    if (!g_rccl.lib) return fail(NF_ERR_HIP, "cannot load librccl.so: %s", dlerror());
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(g_rccl.lib, name); if (!g_rccl.field) return fail(NF_ERR_HIP, "librccl.so lacks %s", name); } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    *(void **)(&g_rccl.CommAbort) = dlsym(g_rccl.lib, "ncclCommAbort");
    *(void **)(&g_rccl.CommCount) = dlsym(g_rccl.lib, "ncclCommCount");
    Dl_info di;
    if (dladdr((void *)g_rccl.AllReduce, &di) && di.dli_fname) snprintf(g_rccl.path, sizeof g_rccl.path, "%s", di.dli_fname);
    return NF_OK;
}

struct nf_solver;
struct nf_team {
    std::vector<nf_solver *> slabs;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;   // interface planes travel here while the x / y passes run on `stream`
    // small slabs (a rank's share at strong-scaling sizes): a pass on 2 M cells does not fill the chip, so the y pass runs BESIDE the x
    // pass on a stream of its own, into its own vector; the accumulation pass of the z lines adds the two (SlabArgs::yadd)
    hipStream_t y_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int opt_xy_overlap = 1; long xy_overlap_max_cells = 6L << 20;
    hipEvent_t ev_z1 = nullptr, ev_xchg = nullptr;
    int nproc = 1, rank = 0;
    ncclComm_t comm = nullptr;
    // option "xchg_comm": the interface planes travel on a communicator of their own (created collectively in team_prepare), so that the
    // first contact with RCCL on several GPUs can A/B "one communicator driven from two streams" against "one communicator per stream"
    ncclComm_t comm_x = nullptr; int opt_xchg_comm = 0;
    // single-reduction CG on slab teams (Cg1 in nf_kernels.h): one all-reduce per CG iteration instead of two
    // -1 (default): slabs of at most cg1_max_cells cells on every rank (the largest slab of the TEAM decides, all-reduced in team_prepare).  With the
    // chain-solve endpoint pass the form cost 4 us per rank and iteration at 2 M cells per slab and 40 us at 16.8 M -- more than an all-reduce -- and
    // was limited to small slabs; with the weighted-sum endpoint pass (k_endpoint_w) it is FASTER than the two-reduction route in device time alone
    // at every size measured (256^3 as 2 / 4 / 8 slabs: -6.5 / -7.9 / -5.5 %, 512^3 as 8: -3.2 %; profiles/r04_d_*), so the limit is out of the way.
    int opt_cg1 = -1, last_cg_reductions = 0; long cg1_max_cells = 1L << 40, team_max_cells = 0;
    int opt_endpoint_w = 1, last_endpoint_w = 0;   // its endpoint pass as two weighted sums per line (k_endpoint_w) instead of a chain solve
    bool rccl_reduce = false;       // scalar reductions go through ncclAllReduce (nproc > 1, or forced for testing)
    double *d_partials = nullptr; long partial_stride = 0, slab_cap = 0;
    CgScalars *d_cg = nullptr;
    double *d_out = nullptr;        // 4 doubles read by the host each outer
    double *d_red = nullptr;        // 8 doubles: process-local sums awaiting the all-reduce, each followed by this rank's error flag
    // A rank that fails locally inside a solve must not leave the others waiting in the next collective:
    // it raises d_errsrc (k_finalize appends it to every reduction that crosses ranks; the consumers stop every rank with
    // CgScalars::err), keeps issuing the collectives of the schedule without the compute ("poisoned"), and returns its own error
    // once the flag has come back; the other ranks return NF_ERR_REMOTE at the same iteration.
    double *d_errsrc = nullptr;     // 1 double, 0 = fine
    // Vector reduce (multi-rank, one slab per rank, equal slab shapes): the block partials of p.q and |r|^2 are all-reduced as they are
    // (a few KB: still latency-bound) into d_vec and the consuming kernels sum them, every block redundantly, like the lean CG of an
    // undivided mesh -- the two one-block k_finalize launches per CG iteration leave the critical path (7 -> 5 kernels per rank).
    double *d_vec = nullptr; long vec_stride = 0;          // 2 rows: all-reduced p.q partials, all-reduced |r|^2 partials (+ flag slot each)
    int vec_cnt_pq = 0, vec_cnt_rr = 0; bool vec_ok = false; int opt_vec_reduce = 1, last_vec_reduce = 0;
    bool dead = false;              // a collective timed out (NF_ERR_COMM): the streams may hold operations that never complete -- no further solve, no stream waits at teardown
    bool dry = false;               // launch functions only report their partial counts
    bool poisoned = false; int poison_rc = 0; char poison_msg[256] = { 0 };
    int xchg_in_apply = 0;          // interface exchanges issued since the current Schur apply began
    long cg_iter_total = 0;         // CG iterations launched since the team was created (NEUTFEM_INJECT_FAIL=<rank>:<iteration>)
    int inject_rank = -1; long inject_iter = -1;
    int inject_outer = -1; char inject_where = 0;   // NEUTFEM_INJECT_FAIL=<rank>:o<n> / :e<n> (start / end of outer n, outside the CG loop), :a (next exchange-only collective)
    double comm_timeout_s = 120.0;  // multi-rank teams: a stream that does not drain for this long means a peer is gone (NEUTFEM_COMM_TIMEOUT_S)
    nf_progress_fn progress_fn = nullptr; void *progress_user = nullptr;   // nf_set_progress_callback
    volatile long outers_done = 0;  // completed outer iterations of the running / last SolveKeff (bench.py's watchdog polls it from another thread)
    bool linked_ready = false;      // separator diagonals exchanged
    int sep_sweeps = 0;             // Jacobi sweeps on the separator system (0: slabs thick enough for it to be diagonal to rounding)
    std::vector<int> last_its;
    // stats of the last SolveKeff
    int last_outer = 0, coarse_outer = 0; long last_cg_total = 0;
    std::vector<double> hist_k, hist_dk, hist_dphi; std::vector<int> hist_cg;
    int has_valid_keff = 0; double last_keff = 1.0;
    // profiling
    HostPub *h_pub = nullptr, *d_pub = nullptr; unsigned long long pub_seq = 0; int opt_pub = 1;   // low-latency scalar readback (k_publish)
    bool profile = false; long prof_tick = 0; int prof_every = 8;   // event-timed launches: every prof_every-th Schur apply of a profiled solve
    std::map<std::string, ProfSlot> prof;
    struct Ev { hipEvent_t a, b; int slot; };
    std::vector<Ev> ev_pending; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
    int cg_batch = 0;
    // 192^3 (7.1 M cells): 229 -> 233 us per CG iteration with them, 224^3 (11.2 M): 389 -> 356, 256^3: 599 -> 550
    int opt_nt_loads = 1; long nt_min_cells = 8000000;   // streaming (non-temporal) loads in the y / z passes of undivided meshes larger than this
    int opt_fuse = 1, opt_xcd = -1, opt_outer_dev = 1, opt_lean = 1, opt_lean_grid = RED_GRID, opt_sepfold = 1;
    long lean_max_cells = 4L << 20;                     // above that the redundant partial sums of 16 k x-pass blocks cost what the two tiny kernels cost
    OuterState *d_ost = nullptr; double *d_hist = nullptr; int hist_cap = 0;   // device-resident outer loop (diagonal path)
    int opt_s_tx = 0, opt_s_seg = 0, opt_wsmin = 0;       // tuning overrides (nf_set_option)
    int opt_x_p2 = 0;                                     // x pass inside CG: two load phases (k_schur_x P2); measured no gain, off
    int opt_split_dot = 1;                                // big undivided RT0-P0 meshes: per-pass shares of p.q (team_schur_apply)
    int opt_s_long_dirs = 3;                              // directions that may take the chunked kernel: bit 0 = y, bit 1 = z
    int s_long_min_y = 255;                               // y lines longer than this take the chunked kernel (s_long_min: z lines)
    int opt_s_long = -1, s_long_min = 256;                // chunked long-line pass (k_schur_c): -1 auto (lines longer than s_long_min), 0 off, 1 always
    size_t lds_limit = 160 * 1024;                        // dynamic LDS a block may ask for (hipDeviceAttributeMaxSharedMemoryPerBlock at team creation)
    // fused-direction CG (two launches per iteration) up to this many cells.  Measured crossover against the four-launch lean path after
    // the round-2 latency work: 64^3 19.1 vs 27.2 us per CG iteration, 80^3 39.1 vs 34.5, 128^3 77.7 vs 70.9, 160^3 221 vs 150
    int opt_fuse3 = 1; long fuse3_max_cells = 400000;
    // whole CG solve in one launch on the workgroups of one XCD (k_cg_xcd): RT0-P0 meshes between the one-workgroup resident kernel and
    // xcd_max_cells.  One XCD has an eighth of the chip's compute units and L2 (4 MiB): beyond that the launch path wins back.
    int opt_cgx = 1, opt_keffx = 1, xcd_id = 0, xcd_groups = 32, last_xcd = 0; long xcd_min_cells = 2000, xcd_max_cells = 28000, xcd_solves = 0, xcd_refused = 0;
    XcdState *d_xcd = nullptr; double *d_xpart = nullptr;
    int opt_resident = 1, opt_resident_lds = 1, opt_resident_serial = 1, opt_resident_two_sided = 1, last_resident_serial = 0; long resident_max_dofs = 2500, resident_serial_max_dofs = 5120;  // whole SolveKeff in one workgroup (k_resident_keff) up to this many flux DOFs per group
    int *d_hist_cg = nullptr; int hist_cg_cap = 0; ResidentOut *d_rout = nullptr;
    int last_path = 0;                                    // 0 host-driven outer loop, 1 diagonal device loop, 2 resident kernel, 3 one-XCD kernel (nf_info "last_path")
    long direct_max_dofs = 6000;                          // explicit-S branch with a dense S^-1 up to this many flux DOFs per group (the oracle's own limit: exact on both sides over the same range; 288 MB per group at 6000)
    int last_direct = 0;                                  // the last solve used: 0 CG as configured, 1 dense S^-1, 2 CG to 1e-14 standing in
    long standin_unconverged = 0;                         // group solves of the stand-in that ended above 1e-14
    // coarse twin of the team (SolveCoarse): built on the first coarse-mesh start and kept until the cross sections, the boundary
    // conditions or the team change -- creating and factoring it costs more than solving it on the benchmark meshes
    std::vector<nf_solver *> cc; int cc_f[3] = {0, 0, 0};
};

struct nf_solver {
    int device = 0;
    nf_team *team = nullptr;
    int slab_index = 0;
    int if_lo = 0, if_hi = 0;                            // interface with the slab below / above (z)
    // mesh (of this slab)
    int dim = 1, nx = 1, ny = 1, nz = 1, ng = 1, k = 0, m = 0;
    int n1 = 1, nloc = 1, nb = 0;                        // P moments per axis, per cell; bubble moments present in P_m
    long N = 0, nphi = 0, nJ = 0, nJx = 0, nJy = 0, nJz = 0;   // nphi = nloc * N DOFs per group (device layout [p][e])
    std::vector<double> xb, yb, zb, hx, hy, hz;
    double *d_hx = nullptr, *d_hy = nullptr, *d_hz = nullptr, *d_xb = nullptr, *d_yb = nullptr, *d_zb = nullptr;
    int bc_set[8] = {0}, bc_type[8] = {0};
    // XS on device (reference layouts)
    double *d_D = nullptr, *d_SigR = nullptr, *d_NSF = nullptr, *d_Chi = nullptr;
    std::vector<double *> d_SigS;          // ng*ng blocks [g_to*ng+g_from], nullptr when all |s| <= 1e-14
    bool xs_uploaded = false, built = false, diag_valid = false;
    // operators
    double *d_Cd = nullptr, *d_Mf = nullptr, *d_Mchi = nullptr;   // ng*nphi diagonals: C, fission, chi-weighted mass (adjoint)
    double *d_phi_adj = nullptr;                        // adjoint flux, ng*nphi (allocated by the first adjoint solve)
    int has_valid_adjoint = 0; double last_keff_adj = 1.0;
    std::vector<double *> d_Ms;                          // ng*ng
    const double **d_Ms_tab = nullptr;                   // the same ng*ng pointers on the device (resident kernel)
    double *d_L[3] = {nullptr, nullptr, nullptr}, *d_DR[3] = {nullptr, nullptr, nullptr}, *d_D0[3] = {nullptr, nullptr, nullptr};
    long nlines[3] = {0, 0, 0};
    double *d_Sinv = nullptr;
    double *d_Sdense = nullptr; bool dense_valid = false;   // explicit-S branch: S^-1 per group, ng * nphi^2, column-major
    // slab interfaces (partition method for the z lines), all per z-line
    double *d_alo = nullptr, *d_ahi = nullptr, *d_hlo = nullptr, *d_hhi = nullptr, *d_gfl = nullptr;   // ng * nlines[2]
    double *d_sinv_lo = nullptr, *d_sinv_hi = nullptr;  // ng * nlines[2]
    double *d_clo = nullptr, *d_chi = nullptr, *d_rlo = nullptr, *d_rhi = nullptr, *d_ulo = nullptr, *d_uhi = nullptr;
    double *d_ctlo = nullptr, *d_cthi = nullptr, *d_elo = nullptr, *d_ehi = nullptr, *d_relo = nullptr, *d_rehi = nullptr;   // separator sweeps (thin slabs)
    double *d_Wlo = nullptr, *d_Whi = nullptr; bool w_valid = false; int sr_cnt3 = 0;   // endpoint functionals of the z lines (k_endpoint_w), ng * N each; |r|^2 partials of its last launch
    // state
    double *d_phi = nullptr, *d_raw = nullptr;          // current iterate / raw group solutions, ng*N
    double *d_p0 = nullptr, *d_p1 = nullptr;            // Chebyshev history
    double *d_tf = nullptr, *d_rhs = nullptr, *d_r = nullptr, *d_p = nullptr, *d_q = nullptr;
    double *d_p2 = nullptr, *d_qy = nullptr, *d_qz = nullptr;   // fused-direction CG (k_apply3): second buffer of the p pair, y / z outputs
    bool raw_valid = false, raw_is_diag = false;
    CgFuse fuse = { nullptr, nullptr, nullptr };        // set by cg_solve around the applies of a fused CG (k_schur_x / k_schur_s mode 1)
    CgLean lean = { nullptr, nullptr, 0, 0, 0 };        // set by cg_solve per iteration of a lean CG (k_schur_x consumes the |r|^2 partials)
    CgLean lean_z1 = { nullptr, nullptr, 0, 0, 0 };     // slab teams: the endpoint pass of the z lines consumes the all-reduced |r|^2
    Cg1 cg1 = { nullptr, nullptr, 0 };                  // slab teams, single-reduction CG (set by cg_solve per iteration): the z passes take their SR instantiations
    bool zw_dot = false;                                // this apply: the y / z passes emit T_a sum z_f w_f as their share of x.y (team_schur_apply, split)
    hipStream_t pass_stream = nullptr;                  // this launch goes to another stream than the team's (x || y on small slabs)
    bool pass_noacc = false; const double *pass_yadd = nullptr;   // SlabArgs::noacc / yadd of this launch
    double *d_Jz = nullptr; bool jz_valid = false;      // slabs: z currents of the last solve, ng * nJz face DOFs (nf_get_J)
    double *d_Jzb = nullptr;                            // slabs, RT1+: z bubbles of the local cells, ng * N * ni
    // CMFD (include/NeutFEM.hpp:119-143): D~ / D^ per direction (ng * faces), PCG work vectors, scalars
    bool cmfd_init = false, cmfd_iface = false; double cmfd_relax = 1.0;
    double *d_Dt[3] = {nullptr, nullptr, nullptr}, *d_Dh[3] = {nullptr, nullptr, nullptr}; long nfc[3] = {0, 0, 0};
    double *d_cm = nullptr, *d_cmJ = nullptr; CmfdScalars *d_cmsc = nullptr;
    int cmfd_last_its = 0;
};

const char *nf_last_error(void) { return g_err.c_str(); }

static void coarse_cache_drop(nf_team *T)
{
    if (!T || T->cc.empty()) return;
    std::string keep = g_err;
    std::vector<nf_solver *> cc; cc.swap(T->cc);
    for (int i = (int)cc.size() - 1; i >= 0; --i) if (cc[i]) nf_destroy(cc[i]);
    g_err = keep;
}

int nf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

template <class T> static int dalloc(T **p, size_t n)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIPCHK(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return NF_OK;
}
template <class T> static void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }
// scratch device buffer of one function call: released on every return path
template <class T> struct DevTmp { T *p = nullptr; ~DevTmp() { if (p) (void)hipFree(p); } DevTmp() = default; DevTmp(const DevTmp &) = delete; DevTmp &operator=(const DevTmp &) = delete; };

static const char *SLOT_NAMES[5] = { "schur_x", "schur_y", "schur_z", "schur_apply", "schur_z1" };

static Geom make_geom(const nf_solver *S)
{
    Geom G; G.dim = S->dim; G.nx = S->nx; G.ny = S->ny; G.nz = S->nz; G.k = S->k; G.hx = S->d_hx; G.hy = S->d_hy; G.hz = S->d_hz;
    G.T0 = (double)(1 << (S->dim - 1));
    // face block of the 1-D element matrix after condensing the k bubbles (exact values of the reference's Gauss
    // quadrature, src/FEM.cpp:891-924): M^FF - M^Fb (M^bb)^-1 M^bF
    if (S->k == 0) { G.aLL = 2.0 / 3.0; G.aLR = 1.0 / 3.0; }
    else if (S->k == 1) { G.aLL = 1.0 / 4.0; G.aLR = -1.0 / 12.0; }
    else { G.aLL = 2.0 / 15.0; G.aLR = 1.0 / 30.0; }
    // GetBoundaryAttribute, src/NeutFEM.cpp:2338-2347
    for (int d = 0; d < 3; ++d) {
        int lo, hi;
        if (S->dim == 1) { lo = 1; hi = 2; }
        else if (S->dim == 2) { if (d == 0) { lo = 1; hi = 2; } else { lo = 4; hi = 3; } }
        else { if (d == 0) { lo = 3; hi = 4; } else if (d == 1) { lo = 6; hi = 5; } else { lo = 1; hi = 2; } }
        G.dir_lo[d] = S->bc_set[lo] && S->bc_type[lo] == NF_BC_DIRICHLET;
        G.dir_hi[d] = S->bc_set[hi] && S->bc_type[hi] == NF_BC_DIRICHLET;
    }
    if (S->if_lo) G.dir_lo[2] = 0;             // a slab interface is not a domain boundary
    if (S->if_hi) G.dir_hi[2] = 0;
    return G;
}

static int grid_for(long n, int block = 256, int cap = RED_GRID)
{
    long g = (n + block - 1) / block; if (g < 1) g = 1; if (g > cap) g = cap; return (int)g;
}

// transverse Legendre modes of one direction that couple to phi: a in [0,m]^(dim-1)
static int n_modes(const nf_solver *S) { int n = 1; for (int t = 1; t < S->dim; ++t) n *= S->n1; return n; }
// moment index p = i_x + n1 i_y + n1^2 i_z of (along-index i, transverse mode `mode`) for direction d, and T_a
static int moment_index(const nf_solver *S, int d, int mode, int i, double *Ta)
{
    const int n1 = S->n1;
    int idx[3] = {0, 0, 0}; double ta = 1.0; int q = mode;
    for (int t = 0; t < S->dim; ++t) {
        if (t == d) { idx[t] = i; continue; }
        idx[t] = q % n1; q /= n1;
        ta *= 2.0 / (2.0 * idx[t] + 1.0);
    }
    if (Ta) *Ta = ta;
    return idx[0] + n1 * idx[1] + n1 * n1 * idx[2];
}
// per-pass constants + moment pointers (ModeArgs) for direction d, mode `mode`; xb/yb: SoA vectors of nphi doubles
static ModeArgs mode_args(const nf_solver *S, int g, int d, int mode, const double *xb, double *yb)
{
    ModeArgs ma; memset(&ma, 0, sizeof ma);
    double Ta = 1.0;
    for (int i = 0; i <= S->nb; ++i) {
        const int p = moment_index(S, d, mode, i, &Ta);
        ma.x[i] = xb + (long)p * S->N; ma.y[i] = yb + (long)p * S->N; ma.Cd[i] = S->d_Cd + (long)g * S->nphi + (long)p * S->N;
    }
    ma.Ta = Ta; ma.dir = d; ma.D = S->d_D + (long)g * S->N;
    // bubble l: eL/eR = M^Fb_{L/R,l} / M^bb_l, Gc = int P_{l+1} d/dxi[(1-xi^2) P_l], iM = 1 / M^bb_l  (src/FEM.cpp:377-620)
    ma.eL[0] = 5.0 / 8.0; ma.eR[0] = 5.0 / 8.0; ma.Gc[0] = -4.0 / 3.0; ma.iM[0] = 15.0 / 16.0;
    ma.eL[1] = -7.0 / 8.0; ma.eR[1] = 7.0 / 8.0; ma.Gc[1] = -4.0 / 5.0; ma.iM[1] = 105.0 / 16.0;
    return ma;
}
// every transverse mode of direction d relative to mode 0 (ModeTab): T_a and moment offsets
static ModeTab mode_tab(const nf_solver *S, int d)
{
    ModeTab mt; memset(&mt, 0, sizeof mt);
    mt.n = n_modes(S);
    for (int m = 0; m < mt.n; ++m)
        for (int i = 0; i <= S->nb; ++i) {
            double Ta = 1.0;
            const int p = moment_index(S, d, m, i, &Ta), p0 = moment_index(S, d, 0, i, nullptr);
            mt.Ta[m] = Ta; mt.doff[m][i] = (long)(p - p0) * S->N;
        }
    return mt;
}

// ---- team management ---------------------------------------------------------------------------
static long slab_partial_need(const nf_solver *S)
{
    // the fused-direction launch (k_apply3) writes one partial per x, y and z block of the same launch
    return (std::max<long>(RED_GRID, S->nlines[0] + (long)((S->nx + 7) / 8) * ((long)S->ny + S->nz)) + 16) * n_modes(S);
}
static int team_alloc(nf_team *T)
{
    long cap = 0;
    for (auto *S : T->slabs) cap = std::max(cap, slab_partial_need(S));
    T->slab_cap = cap; T->partial_stride = cap * (long)T->slabs.size();
    NFCHK(dalloc(&T->d_partials, (size_t)T->partial_stride * 4));
    if (!T->d_cg) NFCHK(dalloc(&T->d_cg, 1));
    if (!T->d_out) NFCHK(dalloc(&T->d_out, 8));
    if (!T->d_red) NFCHK(dalloc(&T->d_red, 8));
    if (!T->d_errsrc) { NFCHK(dalloc(&T->d_errsrc, 1)); HIPCHK(hipMemset(T->d_errsrc, 0, sizeof(double))); }
    // NEUTFEM_INJECT_FAIL=<rank>:<iteration>[:<min local DOFs per cell>] (tests): the optional third field keeps the injection away from
    // the RT0-P0 coarse twin of a higher-order team, so that the non-lean reduction route of RT1 / RT2 teams is the one that is hit
    if (const char *e = getenv("NEUTFEM_INJECT_FAIL")) {
        int r = -1, ml = 0; long it = -1; char w = 0; int on = -1;
        if (sscanf(e, "%d:%c%d", &r, &w, &on) >= 2 && (w == 'o' || w == 'e' || w == 'a')) { T->inject_rank = r; T->inject_where = w; T->inject_outer = on; }
        else {
            const int nf = sscanf(e, "%d:%ld:%d", &r, &it, &ml);
            if (nf >= 2 && (nf < 3 || T->slabs.empty() || T->slabs[0]->nloc >= ml)) { T->inject_rank = r; T->inject_iter = it; }
        }
    }
    if (const char *e = getenv("NEUTFEM_COMM_TIMEOUT_S")) { const double v = atof(e); if (v > 0) T->comm_timeout_s = v; }
    T->last_its.assign(64, 0);
    return NF_OK;
}
static void team_free(nf_team *T)
{
    if (!T) return;
    (void)hipSetDevice(T->device);
    if (T->dead) {
        // after NF_ERR_COMM the streams hold collectives whose peers are gone: waiting for them, destroying them or the communicator can
        // block for ever.  Everything of this team is left to the process exit, which the caller owes anyway (include/neutfem_hip.h).
        delete T;
        return;
    }
    if (T->stream) (void)hipStreamSynchronize(T->stream);
    for (auto &e : T->ev_pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : T->ev_free) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (T->comm_x && g_rccl.CommDestroy) g_rccl.CommDestroy(T->comm_x);
    if (T->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(T->comm);
    if (T->d_xcd) (void)hipFree(T->d_xcd);
    dfree(T->d_xpart);
    dfree(T->d_partials); dfree(T->d_cg); dfree(T->d_out); dfree(T->d_red); dfree(T->d_errsrc); dfree(T->d_vec); dfree(T->d_ost); dfree(T->d_hist); dfree(T->d_hist_cg); dfree(T->d_rout);
    if (T->h_pub) (void)hipHostFree(T->h_pub);
    if (T->comm_stream) { (void)hipStreamSynchronize(T->comm_stream); (void)hipStreamDestroy(T->comm_stream); }
    if (T->y_stream) { (void)hipStreamSynchronize(T->y_stream); (void)hipStreamDestroy(T->y_stream); }
    if (T->ev_fork) (void)hipEventDestroy(T->ev_fork);
    if (T->ev_join) (void)hipEventDestroy(T->ev_join);
    if (T->ev_z1) (void)hipEventDestroy(T->ev_z1);
    if (T->ev_xchg) (void)hipEventDestroy(T->ev_xchg);
    if (T->stream) (void)hipStreamDestroy(T->stream);
    delete T;
}
static PartSegs segs_for(const nf_team *T, const std::vector<int> &counts)
{
    PartSegs ps; ps.n = (int)T->slabs.size();
    for (int i = 0; i < ps.n; ++i) { ps.off[i] = (int)(i * T->slab_cap); ps.cnt[i] = counts[i]; }
    return ps;
}

static int create_impl(int rt_order, int p_order, int ng, int nxb, const double *xb, int nyb, const double *yb, int nzb,
                       const double *zb, int if_lo, int if_hi, int device, nf_handle *out)
{
    if (!out || !xb || nxb < 2 || ng < 1 || ng > 64) return fail(NF_ERR_ARG, "nf_create: bad arguments");
    int k = std::min(rt_order, 2), m = std::min(p_order, 2);
    if (k < m) m = k;                                            // src/NeutFEM.cpp:149-169
    int ndev = nf_device_count();
    if (ndev <= 0) return fail(NF_ERR_NO_DEVICE, "nf_create: no HIP device visible (the gfx950 path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(NF_ERR_ARG, "nf_create: device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    nf_solver *S = new nf_solver();
    S->device = device; S->k = k; S->m = m; S->ng = ng; S->if_lo = if_lo; S->if_hi = if_hi;
    S->xb.assign(xb, xb + nxb);
    if (nyb > 1) S->yb.assign(yb, yb + nyb); else S->yb.assign(1, nyb == 1 && yb ? yb[0] : 0.0);
    if (nzb > 1) S->zb.assign(zb, zb + nzb); else S->zb.assign(1, nzb == 1 && zb ? zb[0] : 0.0);
    S->nx = nxb - 1; S->ny = nyb > 1 ? nyb - 1 : 1; S->nz = nzb > 1 ? nzb - 1 : 1;
    S->dim = S->nz > 1 ? 3 : (S->ny > 1 ? 2 : 1);               // src/FEM.cpp:33-35
    if ((if_lo || if_hi) && (S->dim != 3 || S->nz < 3)) { delete S; return fail(NF_ERR_ARG, "a slab needs a 3D mesh with at least 3 z-planes"); }
    // a 3D mesh reads y_breaks(i+1) - y_breaks(i) (src/FEM.cpp:43-47): one y cell still needs its two breaks (Eigen asserts there)
    if (S->dim == 3 && (nyb < 2 || !yb)) { delete S; return fail(NF_ERR_ARG, "nf_create: a 3D mesh needs at least 2 y breaks (got %d)", nyb); }
    S->N = (long)S->nx * S->ny * S->nz;
    S->n1 = m + 1; S->nloc = 1; for (int t = 0; t < S->dim; ++t) S->nloc *= S->n1;
    S->nphi = S->N * S->nloc; S->nb = std::min(k, m);
    S->hx.resize(S->nx); S->hy.assign(S->ny, 1.0); S->hz.assign(S->nz, 1.0);
    for (int i = 0; i < S->nx; ++i) S->hx[i] = xb[i + 1] - xb[i];
    if (S->dim >= 2) for (int i = 0; i < S->ny; ++i) S->hy[i] = yb[i + 1] - yb[i];
    if (S->dim == 3) for (int i = 0; i < S->nz; ++i) S->hz[i] = zb[i + 1] - zb[i];
    for (auto *h : { &S->hx, &S->hy, &S->hz })
        for (double v : *h)
            if (!(v > 0.0) || !std::isfinite(v)) { delete S; return fail(NF_ERR_ARG, "nf_create: mesh breaks must be finite and strictly increasing (cell width %g)", v); }
    {                                                            // src/FEM.cpp:177-259
        int nf = 1, ni = k; for (int t = 1; t < S->dim; ++t) { nf *= k + 1; ni *= k + 1; }
        S->nJx = (long)(S->nx + 1) * S->ny * S->nz * nf;
        S->nJy = S->dim >= 2 ? (long)S->nx * (S->ny + 1) * S->nz * nf : 0;
        S->nJz = S->dim == 3 ? (long)S->nx * S->ny * (S->nz + 1) * nf : 0;
        S->nJ = S->nJx + S->nJy + S->nJz + S->N * S->dim * ni;
    }
    S->nlines[0] = (long)S->ny * S->nz; S->nlines[1] = (long)S->nx * S->nz; S->nlines[2] = (long)S->nx * S->ny;
    nf_team *T = new nf_team();
    T->device = device; T->slabs.push_back(S); S->team = T; S->slab_index = 0;
    {   // dynamic LDS a workgroup may ask for on this device (gfx950: 160 KiB); the LDS-resident paths plan against it and step aside when it is smaller
        int a = 0, b = 0;
        if (hipDeviceGetAttribute(&a, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) a = 0;
        if (hipDeviceGetAttribute(&b, hipDeviceAttributeSharedMemPerBlockOptin, device) != hipSuccess) b = 0;
        (void)hipGetLastError();
        T->lds_limit = (size_t)std::max(std::max(a, b), 64 * 1024);
    }
    const char *cb = getenv("NEUTFEM_CG_BATCH"); T->cg_batch = cb ? atoi(cb) : 0;
    *out = S;
    int rc = NF_OK;
    // NEUTFEM_OPTS="key=value,key=value": nf_set_option on every handle at creation (A/B runs of programs that do not expose the options,
    // e.g. bench.py with slab teams); unknown keys are reported once on stderr
    if (const char *e = getenv("NEUTFEM_OPTS")) {
        std::string all(e); size_t pos = 0;
        while (pos < all.size()) {
            const size_t c = all.find(',', pos); const std::string kv = all.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos && nf_set_option(S, kv.substr(0, eq).c_str(), atol(kv.c_str() + eq + 1)) != NF_OK) fprintf(stderr, "NEUTFEM_OPTS: %s\n", nf_last_error());
            if (c == std::string::npos) break;
            pos = c + 1;
        }
    }
    auto up = [&](double **d, const std::vector<double> &h) {
        if (rc != NF_OK) return;
        rc = dalloc(d, h.size());
        if (rc == NF_OK && hipMemcpy(*d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(NF_ERR_HIP, "nf_create: upload failed");
    };
    if (hipStreamCreateWithFlags(&T->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(NF_ERR_HIP, "hipStreamCreate failed");
    if (rc == NF_OK && (if_lo || if_hi)) {
        if (hipStreamCreateWithFlags(&T->comm_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&T->ev_z1, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&T->ev_xchg, hipEventDisableTiming) != hipSuccess ||
            hipStreamCreateWithFlags(&T->y_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&T->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&T->ev_join, hipEventDisableTiming) != hipSuccess) rc = fail(NF_ERR_HIP, "comm stream / event creation failed");
    }
    up(&S->d_hx, S->hx); up(&S->d_hy, S->hy); up(&S->d_hz, S->hz); up(&S->d_xb, S->xb); up(&S->d_yb, S->yb); up(&S->d_zb, S->zb);
    const size_t NN = (size_t)S->nphi * ng;
    if (rc == NF_OK) rc = dalloc(&S->d_phi, NN);
    if (rc == NF_OK) rc = dalloc(&S->d_raw, NN);
    if (rc == NF_OK) rc = dalloc(&S->d_tf, S->nphi);
    if (rc == NF_OK) rc = dalloc(&S->d_rhs, S->nphi);
    if (rc == NF_OK) rc = dalloc(&S->d_r, S->nphi);
    if (rc == NF_OK) rc = dalloc(&S->d_p, S->nphi);
    if (rc == NF_OK) rc = dalloc(&S->d_q, S->nphi);
    if (rc == NF_OK && (if_lo || if_hi)) {
        const size_t nl = (size_t)S->nlines[2] * n_modes(S);          // exchange planes: [transverse mode][line]
        double **arrs[] = { &S->d_clo, &S->d_chi, &S->d_rlo, &S->d_rhi, &S->d_ulo, &S->d_uhi, &S->d_ctlo, &S->d_cthi, &S->d_elo, &S->d_ehi, &S->d_relo, &S->d_rehi };
        for (auto a : arrs) if (rc == NF_OK) { rc = dalloc(a, nl); if (rc == NF_OK) (void)hipMemset(*a, 0, nl * sizeof(double)); }
    }
    if (rc == NF_OK) rc = team_alloc(T);
    if (rc != NF_OK) { nf_destroy(S); *out = nullptr; return rc; }
    S->d_SigS.assign((size_t)ng * ng, nullptr); S->d_Ms.assign((size_t)ng * ng, nullptr);
    return nf_reset_flux(S);
}

int nf_create(int rt_order, int p_order, int ng, int nxb, const double *xb, int nyb, const double *yb, int nzb,
              const double *zb, int device, nf_handle *out)
{
    return create_impl(rt_order, p_order, ng, nxb, xb, nyb, yb, nzb, zb, 0, 0, device, out);
}
int nf_create_slab(int rt_order, int p_order, int ng, int nxb, const double *xb, int nyb, const double *yb, int nzb,
                   const double *zb_slab, int interface_below, int interface_above, int device, nf_handle *out)
{
    return create_impl(rt_order, p_order, ng, nxb, xb, nyb, yb, nzb, zb_slab, interface_below ? 1 : 0, interface_above ? 1 : 0, device, out);
}

int nf_destroy(nf_handle S)
{
    if (!S) return NF_OK;
    (void)hipSetDevice(S->device);
    nf_team *T = S->team;
    if (T && T->dead) {                                          // see team_free: nothing of a dead team is waited for or freed (hipFree drains the device)
        std::vector<nf_solver *> cc; cc.swap(T->cc);
        for (auto *c : cc) if (c) nf_destroy(c);
        T->slabs.erase(std::remove(T->slabs.begin(), T->slabs.end(), S), T->slabs.end());
        if (T->slabs.empty()) team_free(T);
        delete S;
        return NF_OK;
    }
    coarse_cache_drop(T);
    if (T && T->stream) (void)hipStreamSynchronize(T->stream);
    for (int d = 0; d < 3; ++d) { dfree(S->d_Dt[d]); dfree(S->d_Dh[d]); }
    dfree(S->d_cm); dfree(S->d_cmJ); dfree(S->d_cmsc);
    dfree(S->d_hx); dfree(S->d_hy); dfree(S->d_hz); dfree(S->d_xb); dfree(S->d_yb); dfree(S->d_zb);
    dfree(S->d_D); dfree(S->d_SigR); dfree(S->d_NSF); dfree(S->d_Chi);
    for (auto &p : S->d_SigS) dfree(p);
    for (auto &p : S->d_Ms) dfree(p);
    dfree(S->d_Ms_tab);
    dfree(S->d_Cd); dfree(S->d_Mf); dfree(S->d_Mchi); dfree(S->d_phi_adj); dfree(S->d_Sinv); dfree(S->d_Sdense);
    for (int d = 0; d < 3; ++d) { dfree(S->d_L[d]); dfree(S->d_DR[d]); dfree(S->d_D0[d]); }
    dfree(S->d_alo); dfree(S->d_ahi); dfree(S->d_hlo); dfree(S->d_hhi); dfree(S->d_gfl); dfree(S->d_sinv_lo); dfree(S->d_sinv_hi);
    dfree(S->d_clo); dfree(S->d_chi); dfree(S->d_rlo); dfree(S->d_rhi); dfree(S->d_ulo); dfree(S->d_uhi);
    dfree(S->d_Wlo); dfree(S->d_Whi);
    dfree(S->d_Jz); dfree(S->d_Jzb); dfree(S->d_ctlo); dfree(S->d_cthi); dfree(S->d_elo); dfree(S->d_ehi); dfree(S->d_relo); dfree(S->d_rehi);
    dfree(S->d_phi); dfree(S->d_raw); dfree(S->d_p0); dfree(S->d_p1);
    dfree(S->d_tf); dfree(S->d_rhs); dfree(S->d_r); dfree(S->d_p); dfree(S->d_q); dfree(S->d_p2); dfree(S->d_qy); dfree(S->d_qz);
    if (T) {
        T->slabs.erase(std::remove(T->slabs.begin(), T->slabs.end(), S), T->slabs.end());
        for (size_t i = 0; i < T->slabs.size(); ++i) T->slabs[i]->slab_index = (int)i;
        if (T->slabs.empty()) team_free(T);
    }
    delete S;
    return NF_OK;
}

// link n slabs (ordered bottom to top, same device, adjacent ones sharing an interface) into one team
int nf_link_slabs(nf_handle *handles, int n)
{
    if (!handles || n < 1 || n > MAX_LOCAL_SLABS) return fail(NF_ERR_ARG, "nf_link_slabs: 1..%d slabs", MAX_LOCAL_SLABS);
    nf_solver *S0 = handles[0];
    if (!S0) return fail(NF_ERR_ARG, "nf_link_slabs: null handle");
    HIPCHK(hipSetDevice(S0->device));
    for (int i = 0; i < n; ++i) {
        nf_solver *S = handles[i];
        if (!S || S->team->slabs.size() != 1) return fail(NF_ERR_STATE, "nf_link_slabs: handle %d is null or already linked", i);
        if (S->device != S0->device || S->nx != S0->nx || S->ny != S0->ny || S->ng != S0->ng)
            return fail(NF_ERR_ARG, "nf_link_slabs: slab %d does not match slab 0 (device, nx, ny, groups)", i);
        if (i > 0 && (!S->if_lo || !handles[i - 1]->if_hi)) return fail(NF_ERR_ARG, "nf_link_slabs: slabs %d/%d lack the shared interface flag", i - 1, i);
    }
    nf_team *T = S0->team;
    for (int i = 0; i < n; ++i) coarse_cache_drop(handles[i]->team);
    for (int i = 1; i < n; ++i) {
        nf_solver *S = handles[i];
        team_free(S->team);
        S->team = T; S->slab_index = i; T->slabs.push_back(S);
    }
    T->linked_ready = false;
    return team_alloc(T);
}

int nf_comm_unique_id(void *id128)
{
    if (!id128) return fail(NF_ERR_ARG, "nf_comm_unique_id: null buffer");
    NFCHK(rccl_load());
    ncclUniqueId id; NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return NF_OK;
}
int nf_comm_init(nf_handle S, const void *id128, int nranks, int rank)
{
    if (!S || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(NF_ERR_ARG, "nf_comm_init: bad arguments");
    nf_team *T = S->team;
    const bool force = getenv("NEUTFEM_FORCE_RCCL") != nullptr;   // 1-rank communicator: exercises the RCCL reduce path on one GPU
    if (nranks == 1 && !force) { T->nproc = 1; T->rank = 0; return NF_OK; }
    NFCHK(rccl_load());
    HIPCHK(hipSetDevice(T->device));
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    NCCLCHK(g_rccl.CommInitRank(&T->comm, nranks, id, rank));
    T->nproc = nranks; T->rank = rank; T->linked_ready = false; T->rccl_reduce = true;
    return NF_OK;
}

// What carries the data path of this team: ranks of the live communicator as the library itself counts them (ncclCommCount; 0 = no
// communicator, -1 = the library has no such entry) and the file the RCCL symbols were resolved from.
int nf_comm_info(nf_handle S, int *comm_ranks, char *lib_path, size_t len)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    nf_team *T = S->team;
    if (comm_ranks) {
        *comm_ranks = 0;
        if (T->comm) { *comm_ranks = -1; int n = 0; if (g_rccl.CommCount && g_rccl.CommCount(T->comm, &n) == 0) *comm_ranks = n; }
    }
    if (lib_path && len) snprintf(lib_path, len, "%s", T->comm ? g_rccl.path : "");
    return NF_OK;
}

// Diagnostic: the point-to-point calls of the plane exchange against the loaded RCCL, on this rank alone -- a grouped
// ncclSend / ncclRecv pair to and from the own rank on the comm stream, ordered against the main stream with the same
// events the Schur apply uses, followed by an all-reduce(max).  Verifies the dlsym'd signatures and the stream / event
// choreography with the real library where only one GPU is available.
int nf_comm_selftest(nf_handle S)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    nf_team *T = S->team;
    if (!T->comm) return fail(NF_ERR_STATE, "nf_comm_selftest: no communicator (call nf_comm_init first)");
    HIPCHK(hipSetDevice(T->device));
    const size_t n = 1 << 16;
    double *a = nullptr, *b = nullptr;
    NFCHK(dalloc(&a, n)); if (dalloc(&b, n) != NF_OK) { dfree(a); return NF_ERR_HIP; }
    hipStream_t cs = T->comm_stream ? T->comm_stream : T->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreateWithFlags(&e0, hipEventDisableTiming); (void)hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    int rc = NF_OK;
    hipLaunchKernelGGL(k_fill_pattern, dim3(64), dim3(256), 0, T->stream, a, (long)n);
    (void)hipMemsetAsync(b, 0, n * sizeof(double), T->stream);
    (void)hipEventRecord(e0, T->stream); (void)hipStreamWaitEvent(cs, e0, 0);
    if (g_rccl.GroupStart() != 0 || g_rccl.Send(a, n, NCCL_DOUBLE, T->rank, T->comm, cs) != 0 ||
        g_rccl.Recv(b, n, NCCL_DOUBLE, T->rank, T->comm, cs) != 0 || g_rccl.GroupEnd() != 0) rc = fail(NF_ERR_HIP, "nf_comm_selftest: grouped ncclSend/ncclRecv failed");
    (void)hipEventRecord(e1, cs); (void)hipStreamWaitEvent(T->stream, e1, 0);
    double w = 3.25;
    if (rc == NF_OK) {
        (void)hipMemcpyAsync(T->d_red, &w, sizeof w, hipMemcpyHostToDevice, T->stream);
        if (g_rccl.AllReduce(T->d_red, T->d_red, 1, NCCL_DOUBLE, NCCL_MAX, T->comm, T->stream) != 0) rc = fail(NF_ERR_HIP, "nf_comm_selftest: ncclAllReduce(max) failed");
        (void)hipMemcpyAsync(&w, T->d_red, sizeof w, hipMemcpyDeviceToHost, T->stream);
    }
    // the vector reduce's call form: out-of-place sum of 1024 partials + 1 flag slot (b <- sum over ranks of a[0 .. 1025))
    const size_t nv = 1025;
    if (rc == NF_OK && g_rccl.AllReduce(a, b + n - nv, nv, NCCL_DOUBLE, NCCL_SUM, T->comm, T->stream) != 0) rc = fail(NF_ERR_HIP, "nf_comm_selftest: ncclAllReduce(sum, 1025 doubles) failed");
    std::vector<double> ha(n), hb(n);
    if (hipStreamSynchronize(T->stream) != hipSuccess) rc = fail(NF_ERR_HIP, "nf_comm_selftest: stream failed");
    if (rc == NF_OK) {
        (void)hipMemcpy(ha.data(), a, n * sizeof(double), hipMemcpyDeviceToHost); (void)hipMemcpy(hb.data(), b, n * sizeof(double), hipMemcpyDeviceToHost);
        bool vec_ok = true;                                        // every rank sends the same pattern: the sum is nproc times it
        for (size_t i = 0; i < nv && vec_ok; ++i) vec_ok = hb[n - nv + i] == (double)T->nproc * ha[i];
        if (memcmp(ha.data(), hb.data(), (n - nv) * sizeof(double)) != 0) rc = fail(NF_ERR_NUMERIC, "nf_comm_selftest: received plane differs from the sent one");
        else if (!vec_ok) rc = fail(NF_ERR_NUMERIC, "nf_comm_selftest: all-reduce(sum) of a 1025-double vector returned wrong values");
        else if (T->nproc == 1 && w != 3.25) rc = fail(NF_ERR_NUMERIC, "nf_comm_selftest: all-reduce(max) returned %g", w);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    dfree(a); dfree(b);
    return rc;
}

long nf_info(nf_handle S, const char *key)
{
    if (!S || !key) return -1;
    nf_team *T = S->team;
#define K(s, v) if (!strcmp(key, s)) return (long)(v)
    K("dim", S->dim); K("nx", S->nx); K("ny", S->ny); K("nz", S->nz); K("ne", S->N); K("ng", S->ng);
    K("n_phi", S->nphi); K("n_J", S->nJ); K("n_loc", S->nloc); K("rt_order", S->k); K("p_order", S->m); K("last_outer", T->last_outer);
    K("last_cg_total", T->last_cg_total); K("coarse_outer", T->coarse_outer); K("device", S->device);
    K("last_path", T->last_path); K("last_resident_serial", T->last_resident_serial); K("last_direct", T->last_direct); K("direct_standin_unconverged", T->standin_unconverged); K("n_local_slabs", T->slabs.size()); K("n_ranks", T->nproc); K("rank", T->rank); K("vec_reduce", T->last_vec_reduce); K("cg_reductions", T->last_cg_reductions); K("endpoint_weights", T->last_endpoint_w); K("xchg_comm", T->comm_x ? 1 : 0); K("last_xcd", T->last_xcd); K("xcd_solves", T->xcd_solves); K("xcd_refused", T->xcd_refused);
#undef K
    return -1;
}

int nf_set_bc(nf_handle S, int attr, int bc_type)
{
    if (!S || attr < 0 || attr >= 8) return fail(NF_ERR_ARG, "nf_set_bc: bad attribute %d", attr);
    S->bc_set[attr] = 1; S->bc_type[attr] = bc_type; S->dense_valid = false;
    coarse_cache_drop(S->team);
    return NF_OK;
}

int nf_upload_xs(nf_handle S, const double *D, const double *SigR, const double *NSF, const double *Chi, const double *SigS)
{
    if (!S || !D || !SigR || !NSF || !Chi || !SigS) return fail(NF_ERR_ARG, "nf_upload_xs: null pointer");
    HIPCHK(hipSetDevice(S->device));
    hipStream_t st = S->team->stream;
    const size_t NN = (size_t)S->N * S->ng, B = NN * sizeof(double);
    // the device copies are overwritten from here on: whatever was built from the old ones is stale until the next nf_build,
    // also when this upload is refused below (a refused upload leaves the handle un-built, never half-valid)
    S->xs_uploaded = false; S->built = false; S->diag_valid = false; S->cmfd_init = false; S->cmfd_iface = false; S->dense_valid = false;
    coarse_cache_drop(S->team);
    NFCHK(dalloc(&S->d_D, NN)); NFCHK(dalloc(&S->d_SigR, NN)); NFCHK(dalloc(&S->d_NSF, NN)); NFCHK(dalloc(&S->d_Chi, NN));
    HIPCHK(hipMemcpyAsync(S->d_D, D, B, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(S->d_SigR, SigR, B, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(S->d_NSF, NSF, B, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(S->d_Chi, Chi, B, hipMemcpyHostToDevice, st));
    const int ng = S->ng;
    for (int i = 0; i < ng * ng; ++i) {
        const double *blk = SigS + (size_t)i * S->N;
        bool nz = false;                                          // src/NeutFEM.cpp:1265 : |sigs| > 1e-14
        for (long e = 0; e < S->N; ++e) if (!(std::fabs(blk[e]) <= 1e-14)) { nz = true; break; }   // NaN counts as non-empty (flagged below)
        if (!nz) { dfree(S->d_SigS[i]); continue; }              // an empty block adds exact zeros: skipping it is equivalent (:1722)
        NFCHK(dalloc(&S->d_SigS[i], S->N));
        HIPCHK(hipMemcpyAsync(S->d_SigS[i], blk, S->N * sizeof(double), hipMemcpyHostToDevice, st));
    }
    // non-finite cross sections or D == 0 would only surface later as a NaN eigenvalue: refuse them here (checked on the device)
    int *d_flags = nullptr, flags = 0;
    NFCHK(dalloc(&d_flags, 1));
    (void)hipMemsetAsync(d_flags, 0, sizeof(int), st);
    const int gchk = grid_for((long)NN, 256, 4096);
    hipLaunchKernelGGL(k_check_xs, dim3(gchk), dim3(256), 0, st, S->d_D, (long)NN, 1, 0, d_flags);
    hipLaunchKernelGGL(k_check_xs, dim3(gchk), dim3(256), 0, st, S->d_SigR, (long)NN, 0, 1, d_flags);
    hipLaunchKernelGGL(k_check_xs, dim3(gchk), dim3(256), 0, st, S->d_NSF, (long)NN, 0, 2, d_flags);
    hipLaunchKernelGGL(k_check_xs, dim3(gchk), dim3(256), 0, st, S->d_Chi, (long)NN, 0, 3, d_flags);
    for (int i = 0; i < ng * ng; ++i)
        if (S->d_SigS[i]) hipLaunchKernelGGL(k_check_xs, dim3(grid_for(S->N, 256, 4096)), dim3(256), 0, st, S->d_SigS[i], S->N, 0, 4, d_flags);
    hipError_t ce = hipMemcpyAsync(&flags, d_flags, sizeof(int), hipMemcpyDeviceToHost, st);
    if (ce == hipSuccess) ce = hipStreamSynchronize(st);
    dfree(d_flags);
    if (ce != hipSuccess) return fail(NF_ERR_HIP, "nf_upload_xs: %s", hipGetErrorString(ce));
    if (flags) {
        static const char *names[5] = { "D (non-finite or zero)", "SigR", "NSF", "Chi", "SigS" };
        std::string which;
        for (int b = 0; b < 5; ++b) if (flags & (1 << b)) { if (!which.empty()) which += ", "; which += names[b]; }
        return fail(NF_ERR_ARG, "nf_upload_xs: invalid cross sections: %s", which.c_str());
    }
    S->xs_uploaded = true; S->built = false;
    return NF_OK;
}

int nf_build(nf_handle S)
{
    if (!S) return fail(NF_ERR_ARG, "nf_build: null handle");
    if (!S->xs_uploaded) return fail(NF_ERR_STATE, "nf_build: call nf_upload_xs first");
    HIPCHK(hipSetDevice(S->device));
    coarse_cache_drop(S->team);
    hipStream_t st = S->team->stream;
    const int ng = S->ng; const long N = S->N, NP = S->nphi; const size_t NN = (size_t)N * ng;
    NFCHK(dalloc(&S->d_Cd, (size_t)NP * ng)); NFCHK(dalloc(&S->d_Mf, (size_t)NP * ng)); NFCHK(dalloc(&S->d_Mchi, (size_t)NP * ng));
    for (int d = 0; d < S->dim; ++d) {
        NFCHK(dalloc(&S->d_L[d], NN)); NFCHK(dalloc(&S->d_DR[d], NN)); NFCHK(dalloc(&S->d_D0[d], (size_t)S->nlines[d] * ng));
    }
    ChatArgs ch; ch.nloc = S->nloc;                               // C-hat_pp = prod 2/(2 i_t + 1), Legendre mass (include/FEM.hpp:197-200)
    for (int p = 0; p < S->nloc; ++p) {
        int q = p; double c = 1.0;
        for (int t = 0; t < S->dim; ++t) { c *= 2.0 / (2.0 * (q % S->n1) + 1.0); q /= S->n1; }
        ch.c[p] = c;
    }
    const bool slab = S->if_lo || S->if_hi;
    if (slab) {
        const size_t nl = (size_t)S->nlines[2] * ng;
        NFCHK(dalloc(&S->d_alo, nl)); NFCHK(dalloc(&S->d_ahi, nl)); NFCHK(dalloc(&S->d_hlo, nl)); NFCHK(dalloc(&S->d_hhi, nl));
        NFCHK(dalloc(&S->d_gfl, nl)); NFCHK(dalloc(&S->d_sinv_lo, nl)); NFCHK(dalloc(&S->d_sinv_hi, nl));
    }
    Geom G = make_geom(S);
    const int gN = grid_for(N, 256, 65535);
    for (int g = 0; g < ng; ++g) {
        hipLaunchKernelGGL(k_cell_coef, dim3(gN), dim3(256), 0, st, S->d_SigR + g * N, S->d_Cd + g * NP, S->d_hx, S->d_hy, S->d_hz, S->nx, S->ny, N, 0, S->dim, ch);
        hipLaunchKernelGGL(k_cell_coef, dim3(gN), dim3(256), 0, st, S->d_NSF + g * N, S->d_Mf + g * NP, S->d_hx, S->d_hy, S->d_hz, S->nx, S->ny, N, 1, S->dim, ch);
        hipLaunchKernelGGL(k_cell_coef, dim3(gN), dim3(256), 0, st, S->d_Chi + g * N, S->d_Mchi + g * NP, S->d_hx, S->d_hy, S->d_hz, S->nx, S->ny, N, 2, S->dim, ch);   // M_chi (:432-439)
        for (int gp = 0; gp < ng; ++gp) {
            const int i = g * ng + gp;
            if (!S->d_SigS[i]) { dfree(S->d_Ms[i]); continue; }
            NFCHK(dalloc(&S->d_Ms[i], NP));
            hipLaunchKernelGGL(k_cell_coef, dim3(gN), dim3(256), 0, st, S->d_SigS[i], S->d_Ms[i], S->d_hx, S->d_hy, S->d_hz, S->nx, S->ny, N, 1, S->dim, ch);
        }
        for (int d = 0; d < S->dim; ++d) {
            const long nl = S->nlines[d];
            SlabOut so = { nullptr, nullptr, nullptr, nullptr, nullptr };
            int lo = 0, hi = 0;
            if (d == 2 && slab) {
                lo = S->if_lo; hi = S->if_hi;
                so.alo = S->d_alo + g * nl; so.ahi = S->d_ahi + g * nl; so.hlo = S->d_hlo + g * nl; so.hhi = S->d_hhi + g * nl; so.gfl = S->d_gfl + g * nl;
            }
            hipLaunchKernelGGL(k_factor_lines, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, st, G, d, S->d_D + g * N,
                               S->d_L[d] + g * N, S->d_DR[d] + g * N, S->d_D0[d] + g * nl, nl, lo, hi, so);
        }
    }
    {
        std::vector<const double *> tab(S->d_Ms.begin(), S->d_Ms.end());
        NFCHK(dalloc(&S->d_Ms_tab, tab.size()));
        HIPCHK(hipMemcpyAsync(S->d_Ms_tab, tab.data(), tab.size() * sizeof(double *), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));                         // `tab` is host stack memory
    }
    // Chebyshev history of the power iteration (two multigroup vectors): allocated here, not inside the outer loop
    if (!S->d_p0) { NFCHK(dalloc(&S->d_p0, (size_t)NP * ng)); NFCHK(dalloc(&S->d_p1, (size_t)NP * ng)); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    S->built = true; S->diag_valid = false; S->cmfd_init = false; S->cmfd_iface = false; S->dense_valid = false;   // src/NeutFEM.cpp:454-456
    S->team->linked_ready = false;
    return NF_OK;
}

// Host wait for a stream with low wake-up latency: hipStreamSynchronize sleeps on an interrupt, which costs 20-30 us per wait -- as much
// as two CG iterations of a 27 k-cell mesh, once per group solve and once per outer iteration.  Poll for a while first; long waits
// (big meshes) fall through to the blocking call, which also reports errors.
static hipError_t stream_wait(hipStream_t st)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0;; ++i) {
        const hipError_t q = hipStreamQuery(st);
        if (q != hipErrorNotReady) return q == hipSuccess ? hipSuccess : hipStreamSynchronize(st);
        if ((i & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    return hipStreamSynchronize(st);
}

// Wait for the solver's stream.  On a multi-rank team the stream holds collectives: if it does not drain within comm_timeout_s a peer
// is gone or stuck, and blocking for ever would pin this rank (and, under torchrun, the whole node) -- abort the communicator and
// return NF_ERR_COMM instead.
static int team_stream_wait(nf_team *T, hipStream_t st)
{
    if (T->nproc <= 1) { HIPCHK(stream_wait(st)); return NF_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    for (long i = 0;; ++i) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return NF_OK;
        if (q != hipErrorNotReady) { HIPCHK(q); }
        if (i > 2000) usleep(i > 20000 ? 200 : 20);
        if ((i & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > T->comm_timeout_s) {
            if (g_rccl.CommAbort && T->comm_x) { (void)g_rccl.CommAbort(T->comm_x); }
            if (g_rccl.CommAbort && T->comm) { (void)g_rccl.CommAbort(T->comm); }
            T->comm = nullptr; T->comm_x = nullptr; T->dead = true;   // with or without ncclCommAbort the queued collectives may never finish: the team is unusable from here on
            for (auto *c : T->cc) if (c && c->team) { c->team->comm = nullptr; c->team->comm_x = nullptr; c->team->dead = true; }   // the coarse twin borrowed the same communicator
            return fail(NF_ERR_COMM, "rank %d: a collective did not complete within %.0f s (NEUTFEM_COMM_TIMEOUT_S): a peer rank is gone or stuck", T->rank, T->comm_timeout_s);
        }
    }
}

// Read the CG scalars and / or nout doubles of `out` back to the host: through the mapped page when available, else D2H copy + wait.
static bool pub_ready(nf_team *T)                                 // the mapped page exists (created on first use)
{
    if (T->opt_pub && !T->h_pub) {
        void *hp = nullptr;
        if (hipHostMalloc(&hp, sizeof(HostPub), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) { T->h_pub = (HostPub *)hp; T->d_pub = (HostPub *)dp; memset(hp, 0, sizeof(HostPub)); }
            else (void)hipHostFree(hp);
        }
        if (!T->h_pub) { (void)hipGetLastError(); T->opt_pub = 0; }
    }
    return T->opt_pub && T->h_pub;
}
static int pub_wait(nf_team *T, unsigned long long seq, CgScalars *h_cg, double *h_out, int nout)   // a kernel of the stream publishes `seq`
{
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (int i = 0;; ++i) {
        if (__atomic_load_n(&T->h_pub->seq, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
        if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    if (!seen) {                                                  // long wait (big mesh) or an error: block, then look again
        NFCHK(team_stream_wait(T, T->stream));
        if (__atomic_load_n(&T->h_pub->seq, __ATOMIC_ACQUIRE) != seq) return fail(NF_ERR_HIP, "scalar readback: the device never published sequence %llu", seq);
    }
    if (h_cg) *h_cg = T->h_pub->cg;
    for (int i = 0; i < nout; ++i) h_out[i] = T->h_pub->out[i];
    return NF_OK;
}
static int readback(nf_team *T, const CgScalars *d_cg, CgScalars *h_cg, const double *d_out, double *h_out, int nout)
{
    hipStream_t st = T->stream;
    if (pub_ready(T)) {
        const unsigned long long seq = ++T->pub_seq;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d_cg, d_out, nout, T->d_pub, seq);
        return pub_wait(T, seq, h_cg, h_out, nout);
    }
    if (h_cg) HIPCHK(hipMemcpyAsync(h_cg, d_cg, sizeof *h_cg, hipMemcpyDeviceToHost, st));
    if (nout > 0) HIPCHK(hipMemcpyAsync(h_out, d_out, nout * sizeof(double), hipMemcpyDeviceToHost, st));
    NFCHK(team_stream_wait(T, st));
    return NF_OK;
}

// ---- profiling helpers -----------------------------------------------------------------------
static void prof_begin(nf_team *T, int slot, hipEvent_t *a, hipEvent_t *b)
{
    if (T->ev_free.empty()) {
        hipEvent_t x, y; (void)hipEventCreate(&x); (void)hipEventCreate(&y); T->ev_free.push_back({x, y});
    }
    auto pr = T->ev_free.back(); T->ev_free.pop_back();
    *a = pr.first; *b = pr.second;
    (void)hipEventRecord(*a, T->stream);
    T->ev_pending.push_back({*a, *b, slot});
}
static void prof_collect(nf_team *T)
{
    for (auto &e : T->ev_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) T->prof[SLOT_NAMES[e.slot]].samples.push_back(ms);
        T->ev_free.push_back({e.a, e.b});
    }
    T->ev_pending.clear();
}

// ---- communication between slabs -----------------------------------------------------------------
// exchange one plane per interface: own c_hi goes up (becomes the upper slab's r_lo), own c_lo goes down.
// which = 0: the per-apply contributions (d_clo/d_chi -> d_rlo/d_rhi); which = 2: the separator-sweep couplings
// (d_elo/d_ehi -> d_relo/d_rehi); which = 1: the separator-diagonal halves of
// group g (d_hlo/d_hhi -> d_rlo/d_rhi, build time).
static int exchange_planes(nf_team *T, int which, int g, hipStream_t st)
{
    const int ns = (int)T->slabs.size();
    ++T->xchg_in_apply;
    auto send_lo = [&](nf_solver *S) { return which == 0 ? S->d_clo : which == 2 ? S->d_elo : S->d_hlo + (size_t)g * S->nlines[2]; };
    auto send_hi = [&](nf_solver *S) { return which == 0 ? S->d_chi : which == 2 ? S->d_ehi : S->d_hhi + (size_t)g * S->nlines[2]; };
    auto recv_lo = [&](nf_solver *S) { return which == 2 ? S->d_relo : S->d_rlo; };
    auto recv_hi = [&](nf_solver *S) { return which == 2 ? S->d_rehi : S->d_rhi; };
    for (int i = 0; i + 1 < ns; ++i) {                            // interfaces between local slabs
        nf_solver *A = T->slabs[i], *B = T->slabs[i + 1];
        const size_t bytes = (size_t)A->nlines[2] * (which == 1 ? 1 : n_modes(A)) * sizeof(double);
        HIPCHK(hipMemcpyAsync(recv_lo(B), send_hi(A), bytes, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(recv_hi(A), send_lo(B), bytes, hipMemcpyDeviceToDevice, st));
    }
    nf_solver *bot = T->slabs.front(), *top = T->slabs.back();
    if (T->nproc > 1 && (bot->if_lo || top->if_hi)) {
        const size_t cnt = (size_t)bot->nlines[2] * (which == 1 ? 1 : n_modes(bot));
        TRACE_COMM("rank %d exchange which=%d count=%zu lo=%d hi=%d poisoned=%d", T->rank, which, cnt, bot->if_lo, top->if_hi, (int)T->poisoned);
        ncclComm_t cx = T->comm_x ? T->comm_x : T->comm;
        NCCLCHK(g_rccl.GroupStart());
        if (bot->if_lo) {
            NCCLCHK(g_rccl.Send(send_lo(bot), cnt, NCCL_DOUBLE, T->rank - 1, cx, st));
            NCCLCHK(g_rccl.Recv(recv_lo(bot), cnt, NCCL_DOUBLE, T->rank - 1, cx, st));
        }
        if (top->if_hi) {
            NCCLCHK(g_rccl.Send(send_hi(top), cnt, NCCL_DOUBLE, T->rank + 1, cx, st));
            NCCLCHK(g_rccl.Recv(recv_hi(top), cnt, NCCL_DOUBLE, T->rank + 1, cx, st));
        }
        NCCLCHK(g_rccl.GroupEnd());
    } else if (bot->if_lo || top->if_hi) {
        return fail(NF_ERR_STATE, "slab %s has an interface but no neighbour: link the slabs (nf_link_slabs) or init the communicator (nf_comm_init)",
                    bot->if_lo ? "bottom" : "top");
    }
    return NF_OK;
}

static int launch_s(nf_solver *S, int d, int g, const ModeArgs &ma, const Geom &G, int last, double *partials, const CgScalars *cg, int *nparts, int zmode);
// build-time: S_red = own half + neighbour's half for every separator; checks that separators decouple
static int team_prepare(nf_team *T)
{
    if (T->linked_ready) return NF_OK;
    if (T->opt_xchg_comm && T->nproc > 1 && T->comm && !T->comm_x) {
        // collective: rank 0 draws a second unique id and hands it to the others through the first communicator (an all-reduce(sum) of
        // its 128 bytes as doubles, zeros from everyone else)
        ncclUniqueId id2; memset(&id2, 0, sizeof id2);
        double v[128]; for (double &x : v) x = 0.0;
        if (T->rank == 0) { NCCLCHK(g_rccl.GetUniqueId(&id2)); for (int i = 0; i < 128; ++i) v[i] = (double)(unsigned char)id2.internal[i]; }
        DevTmp<double> d; NFCHK(dalloc(&d.p, 128));
        HIPCHK(hipMemcpyAsync(d.p, v, sizeof v, hipMemcpyHostToDevice, T->stream));
        NCCLCHK(g_rccl.AllReduce(d.p, d.p, 128, NCCL_DOUBLE, NCCL_SUM, T->comm, T->stream));
        HIPCHK(hipMemcpyAsync(v, d.p, sizeof v, hipMemcpyDeviceToHost, T->stream));
        NFCHK(team_stream_wait(T, T->stream));
        for (int i = 0; i < 128; ++i) id2.internal[i] = (char)(unsigned char)v[i];
        NCCLCHK(g_rccl.CommInitRank(&T->comm_x, T->nproc, id2, T->rank));
        TRACE_COMM("rank %d: second communicator for the plane exchange", T->rank);
    }
    bool any = false;
    for (auto *S : T->slabs) { if (!S->built) return fail(NF_ERR_STATE, "every slab must be built (nf_build) before solving"); any |= S->if_lo || S->if_hi; }
    if (any) {
        const int ng = T->slabs[0]->ng;
        for (int g = 0; g < ng; ++g) {
            NFCHK(exchange_planes(T, 1, g, T->stream));
            for (auto *S : T->slabs) {
                const long nl = S->nlines[2];
                const unsigned gr = (unsigned)((nl + 255) / 256);
                if (S->if_lo) hipLaunchKernelGGL(k_sred_inv, dim3(gr), dim3(256), 0, T->stream, S->d_hlo + g * nl, S->d_rlo, S->d_sinv_lo + g * nl, nl, 0);
                if (S->if_hi) hipLaunchKernelGGL(k_sred_inv, dim3(gr), dim3(256), 0, T->stream, S->d_hhi + g * nl, S->d_rhi, S->d_sinv_hi + g * nl, nl, 1);
            }
            NFCHK(team_stream_wait(T, T->stream));
        }
        // separators must decouple through a slab: |a_lo a_hi (T_II^-1)[first,last]| * sqrt(sinv_lo sinv_hi) <= 1e-15.
        // The verdict is taken on the maximum over ALL ranks, so that every process of a decomposed run refuses together
        // (a single refusing rank would leave the others blocked in their next collective).
        double worst = 0.0; int worst_nz = 0;
        for (auto *S : T->slabs) {
            if (!(S->if_lo && S->if_hi)) continue;
            const size_t nl = (size_t)S->nlines[2] * ng;
            std::vector<double> gfl(nl), slo(nl), shi(nl);
            HIPCHK(hipMemcpy(gfl.data(), S->d_gfl, nl * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(slo.data(), S->d_sinv_lo, nl * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(shi.data(), S->d_sinv_hi, nl * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nl; ++i) {
                const double w = std::fabs(gfl[i]) * std::sqrt(std::fabs(slo[i] * shi[i]));
                if (w > worst) { worst = w; worst_nz = S->nz; }
            }
        }
        if (T->rccl_reduce && T->nproc > 1) {
            HIPCHK(hipMemcpyAsync(T->d_red, &worst, sizeof(double), hipMemcpyHostToDevice, T->stream));
            NCCLCHK(g_rccl.AllReduce(T->d_red, T->d_red, 1, NCCL_DOUBLE, NCCL_MAX, T->comm, T->stream));
            HIPCHK(hipMemcpyAsync(&worst, T->d_red, sizeof(double), hipMemcpyDeviceToHost, T->stream));
            NFCHK(team_stream_wait(T, T->stream));                      // a collective sits on the stream: bounded wait (NEUTFEM_COMM_TIMEOUT_S)
        }
        // Jacobi on the separator system contracts by <= 2 * worst per sweep (two neighbours): sweeps until the neglected term is
        // below 1e-16 of the solution; none for thick slabs (>= ~30 planes), refused beyond 8 (slabs of fewer than ~4 planes)
        T->sep_sweeps = 0;
        if (worst > 1e-15) {
            const double rho = 2.0 * worst;
            int m = rho < 1.0 ? (int)std::ceil(std::log(1e-16) / std::log(rho)) - 1 : 99;
            if (m < 1) m = 1;
            if (m > 8) {
                if (worst_nz) return fail(NF_ERR_UNSUPPORTED, "slab of %d z-planes is too thin: separator coupling %.2e needs %d sweeps (limit 8; use slabs of >= 4 planes)", worst_nz, worst, m);
                return fail(NF_ERR_UNSUPPORTED, "a slab on another rank is too thin: separator coupling %.2e needs %d sweeps (limit 8; use slabs of >= 4 planes)", worst, m);
            }
            T->sep_sweeps = m;
        }
    }
    // vector reduce: partial counts of the accumulation pass (z lines, mode 2) and of k_cg_rupdate, equal on every rank?  EVERY rank of a
    // multi-rank team takes part in this all-reduce, eligible or not (an ineligible rank contributes zeros, which makes max != -min
    // and switches the path off everywhere): a collective behind a rank-local condition would be a hang.
    T->vec_ok = false;
    T->team_max_cells = 0; for (auto *X : T->slabs) T->team_max_cells = std::max(T->team_max_cells, X->N);
    if (T->rccl_reduce && T->nproc > 1) {
        nf_solver *S = T->slabs[0];
        const bool eligible = T->slabs.size() == 1 && S->nloc == 1 && S->dim == 3 && (S->if_lo || S->if_hi);
        int np = 0;
        if (eligible) {
            T->dry = true;
            const int rc = launch_s(S, 2, 0, mode_args(S, 0, 2, 0, S->d_p, S->d_q), make_geom(S), 1, T->d_partials, nullptr, &np, 2);
            T->dry = false;
            if (rc != NF_OK) np = 0;                              // not an error here: the scalar route serves
        }
        const int nr = eligible ? grid_for(S->nphi) : 0;
        double v[5] = { (double)np, -(double)np, (double)nr, -(double)nr, (double)T->team_max_cells };
        HIPCHK(hipMemcpyAsync(T->d_red, v, sizeof v, hipMemcpyHostToDevice, T->stream));
        NCCLCHK(g_rccl.AllReduce(T->d_red, T->d_red, 5, NCCL_DOUBLE, NCCL_MAX, T->comm, T->stream));
        HIPCHK(hipMemcpyAsync(v, T->d_red, sizeof v, hipMemcpyDeviceToHost, T->stream));
        NFCHK(team_stream_wait(T, T->stream));                      // a collective sits on the stream: bounded wait (NEUTFEM_COMM_TIMEOUT_S)
        T->team_max_cells = (long)v[4];                              // every rank takes the same decision about the CG form (cg_solve)
        T->vec_cnt_pq = np; T->vec_cnt_rr = nr;
        T->vec_stride = T->slab_cap + 2;
        T->vec_ok = eligible && v[0] == -v[1] && v[2] == -v[3] && np > 0 && np < T->slab_cap && nr > 0 && nr < T->slab_cap;
        if (T->vec_ok && !T->d_vec) { NFCHK(dalloc(&T->d_vec, (size_t)T->vec_stride * 2)); HIPCHK(hipMemset(T->d_vec, 0, (size_t)T->vec_stride * 2 * sizeof(double))); }
    }
    // endpoint functionals for the single-reduction CG (k_endpoint_w): the response of c_lo / c_hi to the unit vector of every plane, measured with
    // the endpoint pass itself -- nz launches per group, once per BuildMatrices
    for (auto *S : T->slabs) S->w_valid = false;
    if (T->opt_endpoint_w && T->opt_cg1 != 0 && (T->opt_cg1 > 0 || T->team_max_cells <= T->cg1_max_cells))
        for (auto *S : T->slabs) {
            if (!(S->if_lo || S->if_hi) || S->nb != 0 || S->dim != 3 || S->nloc != 1) continue;
            const size_t NG = (size_t)S->N * S->ng; const long nxy = S->nlines[2];
            if (!S->d_Wlo) { NFCHK(dalloc(&S->d_Wlo, NG)); NFCHK(dalloc(&S->d_Whi, NG)); }
            HIPCHK(hipMemsetAsync(S->d_Wlo, 0, NG * sizeof(double), T->stream)); HIPCHK(hipMemsetAsync(S->d_Whi, 0, NG * sizeof(double), T->stream));
            HIPCHK(hipMemsetAsync(S->d_p, 0, (size_t)S->N * sizeof(double), T->stream));
            const Geom G = make_geom(S);
            const unsigned gf = (unsigned)grid_for(nxy);
            for (int g = 0; g < S->ng; ++g)
                for (int k = 0; k < S->nz; ++k) {
                    double *plane = S->d_p + (size_t)k * nxy;
                    hipLaunchKernelGGL(k_fill_const, dim3(gf), dim3(256), 0, T->stream, plane, nxy, 1.0);
                    NFCHK(launch_s(S, 2, g, mode_args(S, g, 2, 0, S->d_p, S->d_q), G, 0, nullptr, nullptr, nullptr, 1));
                    if (S->if_lo) HIPCHK(hipMemcpyAsync(S->d_Wlo + (size_t)g * S->N + (size_t)k * nxy, S->d_clo, (size_t)nxy * sizeof(double), hipMemcpyDeviceToDevice, T->stream));
                    if (S->if_hi) HIPCHK(hipMemcpyAsync(S->d_Whi + (size_t)g * S->N + (size_t)k * nxy, S->d_chi, (size_t)nxy * sizeof(double), hipMemcpyDeviceToDevice, T->stream));
                    hipLaunchKernelGGL(k_fill_const, dim3(gf), dim3(256), 0, T->stream, plane, nxy, 0.0);
                }
            HIPCHK(hipStreamSynchronize(T->stream));
            HIPCHK(hipGetLastError());
            S->w_valid = true;
        }
    T->linked_ready = true;
    return NF_OK;
}

// process-local sum of the partials (+ all-reduce over ranks) + the scalar logic of `op`
// want_flag (FIN_SUM on a multi-rank team): the all-reduced error flags of the ranks land in out[nq], next to the sums (0 elsewhere)
static int team_finalize(nf_team *T, int op, const std::vector<int> &counts, int nq, double *out, double tol, int maxit, bool want_flag = false)
{
    PartSegs ps = segs_for(T, counts);
    if (!T->rccl_reduce) {
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, T->stream, op, T->d_partials, ps, T->partial_stride, nq, T->d_cg, out, tol, maxit, 0, T->d_red, (const double *)nullptr);
    } else {
        // the sums of this rank followed by its error flag: one all-reduce of nq + 1 doubles (CgScalars::err)
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, T->stream, op, T->d_partials, ps, T->partial_stride, nq, T->d_cg, out, tol, maxit, 1, T->d_red, (const double *)T->d_errsrc);
        TRACE_COMM("rank %d allreduce finalize op=%d count=%d poisoned=%d", T->rank, op, nq + 1, (int)T->poisoned);
        NCCLCHK(g_rccl.AllReduce(T->d_red, T->d_red, (size_t)nq + 1, NCCL_DOUBLE, NCCL_SUM, T->comm, T->stream));
        hipLaunchKernelGGL(k_cg_logic, dim3(1), dim3(1), 0, T->stream, op, T->d_red, nq, T->d_cg, out, tol, maxit, want_flag ? 2 : 1);
    }
    return NF_OK;
}

// A rank-local failure on a multi-rank team OUTSIDE the CG loop (an allocation, a refused launch between two collectives of the outer
// iteration): the rank raises its flag and carries on with the schedule instead of returning -- its peers are about to enter the next
// collective.  The flag travels in the reductions that exist anyway: the next CG solve ends on every rank at its first reduction
// (cg_logic, FIN_RHS), the per-outer reduction hands the flags to the host (team_finalize, want_flag).  Returns true when the caller
// must return `code` at once (single process).
static bool team_poison(nf_team *T, int code)
{
    if (code == NF_OK) return false;
    if (T->nproc <= 1) return true;
    if (!T->poisoned) {
        T->poisoned = true; T->poison_rc = code; snprintf(T->poison_msg, sizeof T->poison_msg, "%s", nf_last_error());
        const double one = 1.0;
        (void)hipMemcpyAsync(T->d_errsrc, &one, sizeof one, hipMemcpyHostToDevice, T->stream);
        (void)hipStreamSynchronize(T->stream);
        (void)hipGetLastError();
    }
    return false;
}
static int team_poison_result(nf_team *T)                         // the poisoned rank's own error, once every rank has stopped
{
    const int code = T->poison_rc; char msg[256]; snprintf(msg, sizeof msg, "%s", T->poison_msg);
    T->poisoned = false;
    (void)hipMemsetAsync(T->d_errsrc, 0, sizeof(double), T->stream);
    return fail(code, "%s", msg);
}
// Exchange-only collectives (nf_build_diagonal_cache, nf_get_J, nf_initialize_cmfd: planes travel, nothing is reduced): every rank does
// its fallible local work first (allocations), then the verdicts are all-reduced, and only if every rank is fine is the first plane
// posted -- a rank that failed would otherwise return while its neighbours wait in ncclRecv until the timeout.
static int team_verdict(nf_team *T, int local_rc, const char *what)
{
    if (T->nproc > 1 && T->rank == T->inject_rank && T->inject_where == 'a') { T->inject_where = 0; local_rc = fail(NF_ERR_HIP, "injected failure on rank %d before %s (NEUTFEM_INJECT_FAIL)", T->rank, what); }
    if (!(T->rccl_reduce && T->nproc > 1)) return local_rc;
    std::string keep = local_rc != NF_OK ? g_err : std::string();
    double v = local_rc != NF_OK ? 1.0 : 0.0;
    HIPCHK(hipMemcpyAsync(T->d_red, &v, sizeof v, hipMemcpyHostToDevice, T->stream));
    TRACE_COMM("rank %d allreduce verdict before %s", T->rank, what);
    NCCLCHK(g_rccl.AllReduce(T->d_red, T->d_red, 1, NCCL_DOUBLE, NCCL_MAX, T->comm, T->stream));
    HIPCHK(hipMemcpyAsync(&v, T->d_red, sizeof v, hipMemcpyDeviceToHost, T->stream));
    NFCHK(team_stream_wait(T, T->stream));
    if (local_rc != NF_OK) { g_err = keep; return local_rc; }
    if (v != 0.0) return fail(NF_ERR_REMOTE, "another rank of the team reported an error before %s; no rank started it", what);
    return NF_OK;
}

// lean CG on slab teams: process-local sum of the partials into red[0] (+ all-reduce over ranks); the kernels that consume the
// total derive the CG scalars themselves (CgLean with count < 0), so no k_cg_logic launch follows
static int team_reduce(nf_team *T, const std::vector<int> &counts, double *red)
{
    PartSegs ps = segs_for(T, counts);
    // red[0] = the sum, red[1] = this rank's error flag: both travel in the one all-reduce (the consumers read red[1], CgLean count < 0)
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, T->stream, (int)FIN_SUM, T->d_partials, ps, T->partial_stride, 1, T->d_cg, red, 0.0, 0, 1, red, (const double *)T->d_errsrc);
    if (T->rccl_reduce) { TRACE_COMM("rank %d allreduce reduce count=2 poisoned=%d", T->rank, (int)T->poisoned); NCCLCHK(g_rccl.AllReduce(red, red, 2, NCCL_DOUBLE, NCCL_SUM, T->comm, T->stream)); }
    return NF_OK;
}

// single-reduction CG: the four rows of block partials (p.q, q.q, r.q from the accumulation pass, |r|^2 from the endpoint pass; the same
// count per slab in every row) -> red[0..3], this rank's error flag -> red[4], one all-reduce of five doubles
static int team_reduce_sr(nf_team *T, const std::vector<int> &counts)
{
    PartSegs ps = segs_for(T, counts);
    std::vector<int> c3(counts);                                 // row 3: the endpoint pass's own block count where k_endpoint_w ran (0 on a poisoned rank)
    for (size_t i = 0; i < c3.size(); ++i) if (T->slabs[i]->sr_cnt3 > 0 && counts[i] > 0) c3[i] = T->slabs[i]->sr_cnt3;
    PartSegs ps3 = segs_for(T, c3);
    hipLaunchKernelGGL(k_reduce_rows, dim3(1), dim3(256), 0, T->stream, (const double *)T->d_partials, ps, ps3, T->partial_stride, T->d_red, (const double *)T->d_errsrc);
    if (T->rccl_reduce) { TRACE_COMM("rank %d allreduce single-reduction count=5 poisoned=%d", T->rank, (int)T->poisoned); NCCLCHK(g_rccl.AllReduce(T->d_red, T->d_red, 5, NCCL_DOUBLE, NCCL_SUM, T->comm, T->stream)); }
    return NF_OK;
}

// Opt a kernel in to `lds` bytes of dynamic LDS (above the 64 KiB a launch gets by default).  Asked once per kernel and size class,
// not on every solve; a refusal (a device or driver with less LDS than planned for) is not an error of the solve: the caller steps
// back to the host-driven path.
static bool lds_opt_in(const void *fn, size_t lds)
{
    if (lds <= 64 * 1024) return true;
    // the attribute belongs to the CURRENT device's copy of the function: a process that holds handles on two devices must ask on each
    static std::map<std::pair<int, const void *>, size_t> granted;
    static std::mutex mu;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find({ dev, fn });
    if (it != granted.end() && it->second >= lds) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return false; }
    granted[{ dev, fn }] = lds;
    return true;
}
// ---- Schur apply -----------------------------------------------------------------------------
template <int NCH, int NB>
static void launch_x_t(nf_solver *S, int g, const ModeArgs &ma, const Geom &G, int lpl_log2, int first, int last,
                       double *partials, const CgScalars *cg, unsigned grid)
{
    const long N = S->N;
    const bool vec = (S->nx % 2 == 0);
    hipStream_t st = S->team->stream;
    const double *L = S->d_L[0] + g * N, *DR = S->d_DR[0] + g * N, *D0 = S->d_D0[0] + g * S->nlines[0];
    const CgFuse fz = (S->if_lo || S->if_hi) ? CgFuse{ nullptr, nullptr, nullptr } : S->fuse;   // slabs fuse in their endpoint pass instead
    const ModeTab mt = mode_tab(S, 0);
    const dim3 gr(grid, (unsigned)mt.n);                          // all transverse modes in one launch
    const bool nt = NB == 0 && vec && S->team->opt_nt_loads && N > S->team->nt_min_cells;   // streaming loads beyond the caches (per slab on teams)
    // inside CG: two load phases (schur_x_task, P2): 116 instead of 140 VGPRs at four chunks per lane, four waves per SIMD instead of
    // three -- and no faster (512^3: 1548 vs 1529 us, 256^3: 185.9 vs 186.3; profiles/r03_e_ab_cg.txt).  Kept as an option, off.
    const bool p2 = NB == 0 && NCH >= 2 && fz.p && S->team->opt_x_p2 == 1;
    if (nt && p2) hipLaunchKernelGGL((k_schur_x<2, NCH, true, NB, NB == 0, NB == 0 && (NCH >= 2)>), gr, dim3(256), 0, st, ma, mt, G, L, DR, D0, S->nx, S->ny, S->nlines[0], lpl_log2,
                               first | ((S->team->opt_xcd >= 0 && (S->team->opt_xcd & 4)) ? 2 : 0), last, partials, cg, fz, S->lean);
    else if (nt) hipLaunchKernelGGL((k_schur_x<2, NCH, true, NB, NB == 0>), gr, dim3(256), 0, st, ma, mt, G, L, DR, D0, S->nx, S->ny, S->nlines[0], lpl_log2,
                               first | ((S->team->opt_xcd >= 0 && (S->team->opt_xcd & 4)) ? 2 : 0), last, partials, cg, fz, S->lean);
    else if (vec) hipLaunchKernelGGL((k_schur_x<2, NCH, true, NB>), gr, dim3(256), 0, st, ma, mt, G, L, DR, D0, S->nx, S->ny, S->nlines[0], lpl_log2, first, last, partials, cg, fz, S->lean);
    else hipLaunchKernelGGL((k_schur_x<2, NCH, false, NB>), gr, dim3(256), 0, st, ma, mt, G, L, DR, D0, S->nx, S->ny, S->nlines[0], lpl_log2, first, last, partials, cg, fz, S->lean);
}
template <int NB>
static int launch_x_nb(nf_solver *S, int g, const ModeArgs &ma, const Geom &G, int last, double *partials, const CgScalars *cg, int *nparts)
{
    const int K = 2;
    int lanes = (S->nx + K - 1) / K, lpl_log2 = 0;
    while ((1 << lpl_log2) < lanes && lpl_log2 < 6) ++lpl_log2;
    const int LPL = 1 << lpl_log2, LPW = 64 / LPL;
    const int nch = (S->nx + LPL * K - 1) / (LPL * K);
    const unsigned grid = (unsigned)((S->nlines[0] + 4 * LPW - 1) / (4 * LPW));
    if (nparts) *nparts = (int)grid * n_modes(S);
    if (nch <= 1) launch_x_t<1, NB>(S, g, ma, G, lpl_log2, 1, last, partials, cg, grid);
    else if (nch <= 2) launch_x_t<2, NB>(S, g, ma, G, lpl_log2, 1, last, partials, cg, grid);
    else if (nch <= 4) launch_x_t<4, NB>(S, g, ma, G, lpl_log2, 1, last, partials, cg, grid);
    else if (nch <= 8) launch_x_t<8, NB>(S, g, ma, G, lpl_log2, 1, last, partials, cg, grid);
    else if (nch <= 16 && NB == 0) launch_x_t<16, 0>(S, g, ma, G, lpl_log2, 1, last, partials, cg, grid);   // long lines: more registers per lane, lower occupancy
    else if (nch <= 32 && NB == 0) launch_x_t<32, 0>(S, g, ma, G, lpl_log2, 1, last, partials, cg, grid);
    else return fail(NF_ERR_UNSUPPORTED, "nx = %d exceeds the x-line kernel limit (%d cells)", S->nx, NB == 0 ? 4096 : 1024);
    return NF_OK;
}
static int launch_x(nf_solver *S, int g, const ModeArgs &ma, const Geom &G, int last, double *partials, const CgScalars *cg, int *nparts)
{
    if (S->nb == 0) return launch_x_nb<0>(S, g, ma, G, last, partials, cg, nparts);
    if (S->nb == 1) return launch_x_nb<1>(S, g, ma, G, last, partials, cg, nparts);
    return launch_x_nb<2>(S, g, ma, G, last, partials, cg, nparts);
}

// Does direction d (1 = y, 2 = z) of this mesh take the chunked long-line kernel (k_schur_c)?  Plain lines of RT0-P0 meshes beyond
// s_long_min cells (s_long: -1 = from 257 cells per line, where k_schur_s would drop below 32 columns; 0 = never; 1 = whenever the
// shape allows), one group's array below 4 GiB (32-bit byte offsets inside), at most 1024 threads and the device's LDS.
struct ChunkPlan { bool ok = false; int NS = 0, TX = 0; size_t lds = 0; };
static ChunkPlan chunk_plan(const nf_solver *S, int d)
{
    ChunkPlan P; const nf_team *T = S->team;
    const int n = d == 1 ? S->ny : S->nz;
    if (S->nb != 0 || d < 1 || d >= S->dim || (d == 2 && (S->if_lo || S->if_hi)) || !((T->opt_s_long_dirs >> (d - 1)) & 1)) return P;
    // y lines already from 256 cells (32 columns x 512 threads, two blocks per CU: y 128 -> 120 us inside CG at 256^3, 510 -> 502 us per
    // iteration; nothing at 192^3, a loss at 128^3); z lines from 257: the last pass would need the split dot product, which eats its
    // 6 us at 256^3 (profiles/r03_i_ab_cg_chunked_y.txt)
    const int nmin = d == 1 ? T->s_long_min_y : T->s_long_min;
    if (!(T->opt_s_long == 1 || (T->opt_s_long < 0 && n > nmin)) || (size_t)S->N * sizeof(double) >= (1ull << 32)) return P;
    P.NS = (n + 15) / 16;                                        // segments of 8 cells per chunk, two chunks
    // 32 columns: 512-thread blocks at 256-cell lines (two per CU: one block's load phase overlaps the other's scans) beat 64 columns
    // x 1024 threads there (y 113 vs 117 us, z 116 vs 127 at 256^3, profiles/r03_h_ab_256.txt); at 512 cells 32 columns need the full 1024
    P.TX = T->opt_s_tx ? T->opt_s_tx : 32;
    while (P.TX > 8 && P.TX * P.NS > 1024) P.TX >>= 1;
    while (P.TX > 8 && P.TX / 2 >= S->nx) P.TX >>= 1;
    P.lds = (size_t)(3 * P.TX * P.NS + 2 * P.TX + 16 + 2 * 8 * P.NS * P.TX) * sizeof(double);
    P.ok = P.TX * P.NS <= 1024 && P.lds <= T->lds_limit;
    // up to 256 cells per line the one-chunk kernel has 32 columns too, and the chunked one only wins through two blocks per CU
    // overlapping: that needs tiles to overlap with.  A 256 x 256 x 32 slab has 256 of them -- one per CU -- and loses (y 22.8 vs 20.8 us,
    // 8-slab loopback 983 vs 971 us per CG iteration, profiles/r03_n_thin_slab_tiles.txt); 256^3 has 2048 and wins.
    if (P.ok && n <= 256 && T->opt_s_long < 1) {
        const long tiles = (long)((S->nx + P.TX - 1) / P.TX) * (d == 1 ? S->nz : S->ny);
        if (tiles < 1024) P.ok = false;
    }
    return P;
}

// zmode: 0 = plain line kernel (y lines, or z lines of an undivided mesh); 1 / 2 = slab chain passes (z lines)
static int launch_s(nf_solver *S, int d, int g, const ModeArgs &ma, const Geom &G, int last, double *partials, const CgScalars *cg, int *nparts, int zmode)
{
    nf_team *T = S->team;
    const long N = S->N;
    const int n = d == 1 ? S->ny : S->nz;
    const long nxy = (long)S->nx * S->ny;
    const long sl = d == 1 ? S->nx : nxy, ostride = d == 1 ? nxy : S->nx;
    const int nouter = d == 1 ? S->nz : S->ny;
    // long plain lines of RT0-P0 meshes: two chunks per block, twice the tile width (k_schur_c).  Its share of x.y comes in the z.w
    // form only: inside CG it needs the split dot product of team_schur_apply
    const ChunkPlan cp = zmode == 0 ? chunk_plan(S, d) : ChunkPlan();
    if (cp.ok && (!(last && partials) || S->zw_dot)) {
        const int NS = cp.NS, TXc = cp.TX; const size_t ldsc = cp.lds;
        {
            dim3 grid((unsigned)((S->nx + TXc - 1) / TXc), (unsigned)nouter), block((unsigned)((TXc * NS + 63) / 64 * 64));
            const double *L = S->d_L[d] + g * N, *DR = S->d_DR[d] + g * N, *D0 = S->d_D0[d] + g * S->nlines[d];
            const bool nt = T->opt_nt_loads && S->N > T->nt_min_cells;
            const int xcd = T->opt_xcd >= 0 ? (T->opt_xcd >> (d - 1)) & 1 : (d == 1 && nt);
            bool launched = false;                                    // false: the device refused the LDS -> the one-chunk kernel below
#define NF_C(DIRV, NTV) do { if (lds_opt_in((const void *)k_schur_c<DIRV, NTV>, ldsc)) { \
            hipLaunchKernelGGL((k_schur_c<DIRV, NTV>), grid, block, ldsc, T->stream, ma.x[0], ma.y[0], ma.Ta, L, DR, D0, n, sl, ostride, S->nx, TXc, NS, last, partials, cg, xcd); \
            launched = hipGetLastError() == hipSuccess; } } while (0)   /* a refused launch (LDS, block size) falls through to the one-chunk kernel */
            if (d == 1) { if (nt) NF_C(1, true); else NF_C(1, false); } else { if (nt) NF_C(2, true); else NF_C(2, false); }
#undef NF_C
            if (launched) { if (nparts) *nparts = (int)(grid.x * grid.y); return NF_OK; }
        }
    }
    // 8-cell segments up to 1024 cells per line (16 / 32 cells spill to scratch: 1.4x slower even though TX drops to 8 at 1024)
    // higher orders: 4-cell segments up to 256 cells per line; beyond, RT1 keeps 8 (148 B / lane of scratch and still faster: y 97 vs 122 us on
    // 48 x 512 x 48), RT2 takes 4 up to 512 cells (8 columns per block, no scratch: y 352 vs 383 us, z 412 vs 431; profiles/r04_g_higher_orders.txt)
    int SEG = T->opt_s_seg ? T->opt_s_seg : (S->nb > 0 ? ((n <= 256 || (S->nb == 2 && n <= 512)) ? 4 : 8) : (n <= 1024 ? 8 : (n <= 2048 ? 16 : 32)));
    int NSEG = (n + SEG - 1) / SEG;
    if (NSEG > 128) return fail(NF_ERR_UNSUPPORTED, "line length %d exceeds the segmented kernel limit", n);
    int TX = T->opt_s_tx ? T->opt_s_tx : 64;
    const int tmax = zmode != 0 ? 512 : 1024;                     // slab variants: 512 threads (register budget, see k_schur_s)
    while (TX > 8 && TX * NSEG > tmax) TX >>= 1;
    if (TX * NSEG > tmax) return fail(NF_ERR_UNSUPPORTED, "line length %d needs more than %d threads per block", n, tmax);
    while (TX > 8 && TX / 2 >= S->nx) TX >>= 1;                   // narrow meshes
    const ModeTab mt = mode_tab(S, d);
    // whole wavefronts: the reductions use data-parallel-primitive moves and read lane 63 (threads beyond TX * NSEG only keep the barriers company)
    dim3 grid((unsigned)((S->nx + TX - 1) / TX), (unsigned)nouter, (unsigned)mt.n), block((unsigned)((TX * NSEG + 63) / 64 * 64));
    if (nparts) *nparts = (int)(grid.x * grid.y * grid.z);
    if (T->dry) return NF_OK;
    const double *L = S->d_L[d] + g * N, *DR = S->d_DR[d] + g * N, *D0 = S->d_D0[d] + g * S->nlines[d];
    hipStream_t st = S->pass_stream ? S->pass_stream : T->stream;
    SlabArgs sa; memset(&sa, 0, sizeof sa);
    sa.noacc = S->pass_noacc ? 1 : 0; sa.yadd = S->pass_yadd;
    // XCD-contiguous tile order (k_schur_s): bit 0 = y passes, bit 1 = z passes; -1 (default) = the y passes of meshes in the streaming
    // regime -- the 8 x tiles of a row set then run on one XCD back to back (256^3: y 133 -> 123 us, 501 -> 493 us per CG iteration;
    // z passes lose 10 us with it; neutral to -0.5 % at 96^3 ... 192^3)
    // z passes of SMALL slabs (at most 6 Mi cells): the XCD-contiguous order by default too (8-slab loopback at 256^3: accumulation pass 33.3 -> 30.7 us,
    // 848 -> 835 us per CG iteration; 4 slabs 800 -> 773; on 8.4 M / 16.8 M-cell slabs it loses: 93 -> 101 / 191 -> 201 us; profiles/r04_o_slab_options.txt)
    sa.xcd = T->opt_xcd >= 0 ? (T->opt_xcd >> (d - 1)) & 1 : ((d == 1 && T->opt_nt_loads && S->N > T->nt_min_cells) || ((zmode == 1 || zmode == 2) && S->N <= (6L << 20)));
    sa.wsmin = T->opt_wsmin;
    const size_t lds = (size_t)(4 * TX * (NSEG + 1) + TX + 32) * sizeof(double);   // + reduction scratch (block_sum: 8, block_sum3: 24 doubles at 512 threads)
    const CgFuse fz = (zmode == 1 && S->nloc == 1) ? S->fuse : CgFuse{ nullptr, nullptr, nullptr };
    const CgLean lz = (zmode == 1 && S->nloc == 1 && fz.p) ? S->lean_z1 : CgLean{ nullptr, nullptr, 0, 0, 0 };
    // undivided meshes beyond the caches (the classic path's sizes): the variant with streaming loads (SF doubles as that flag for !SLAB)
    const bool nt = zmode == 0 && S->nb == 0 && SEG == 8 && T->opt_nt_loads && S->N > T->nt_min_cells;
    const bool nts = zmode != 0 && zmode != 3 && S->nb == 0 && SEG == 8 && T->opt_nt_loads && S->N > T->nt_min_cells;   // the same for the z passes of a slab
    const bool zw = zmode == 0 && S->nb == 0 && S->zw_dot && last && partials;                                              // z.w form of the pass's share of x.y
    // single-reduction CG (Cg1): the endpoint pass (mode 1) consumes the reduction of the previous iteration and carries r -= alpha q, the
    // accumulation pass (mode 2) leaves p.q, q.q, r.q; |r|^2 comes from the endpoint pass (row 3 of the partial buffer)
    const bool sr = S->cg1.red != nullptr && S->nb == 0 && SEG == 8 && d == 2 && ((zmode == 1 && fz.p) || (zmode == 2 && last && partials));
    if (sr) {
        sa.sr = S->cg1; sa.sr_r = S->d_r; sa.sr_q = S->d_q; sa.sr_stride = T->partial_stride;
        sa.sr_part = T->d_partials + 3 * T->partial_stride + (long)S->slab_index * T->slab_cap;
    }
#define NF_S(SEGV, DIRV, SLABV, NBV) do { if (SLABV && NBV == 0 && SEGV == 8 && DIRV == 2 && sr && zmode == 1 && nts) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, SLABV && NBV == 0, SLABV && NBV == 0 && SEGV == 8 && DIRV == 2, false, SLABV && NBV == 0 && SEGV == 8 && DIRV == 2>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else if (SLABV && NBV == 0 && SEGV == 8 && DIRV == 2 && sr && zmode == 1) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, SLABV && NBV == 0, false, false, SLABV && NBV == 0 && SEGV == 8 && DIRV == 2>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else if (SLABV && NBV == 0 && SEGV == 8 && DIRV == 2 && sr && nts) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, false, SLABV && NBV == 0 && SEGV == 8 && DIRV == 2, false, SLABV && NBV == 0 && SEGV == 8 && DIRV == 2>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else if (SLABV && NBV == 0 && SEGV == 8 && DIRV == 2 && sr) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, false, false, false, SLABV && NBV == 0 && SEGV == 8 && DIRV == 2>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else if (SLABV && NBV == 0 && SEGV == 8 && DIRV == 2 && nts && fz.p) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, SLABV && NBV == 0, SLABV && NBV == 0 && SEGV == 8 && DIRV == 2>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else if (SLABV && NBV == 0 && SEGV == 8 && DIRV == 2 && nts) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, false, SLABV && NBV == 0 && SEGV == 8 && DIRV == 2>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else if (SLABV && NBV == 0 && fz.p) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, SLABV && NBV == 0>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else if (!SLABV && NBV == 0 && zw && SEGV == 8 && nt) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, !SLABV && NBV == 0 && SEGV == 8, false, !SLABV && NBV == 0>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else if (!SLABV && NBV == 0 && zw) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, false, false, !SLABV && NBV == 0>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else if (!SLABV && NBV == 0 && SEGV == 8 && nt) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, !SLABV && NBV == 0 && SEGV == 8>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); \
        else hipLaunchKernelGGL((k_schur_s<SEGV, DIRV, SLABV, NBV, false>), grid, block, lds, st, ma, mt, G, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, last, partials, cg, sa, fz, lz); } while (0)
#define NF_S_SEG(DIRV, SLABV, NBV) do { if (SEG == 4) NF_S(4, DIRV, SLABV, NBV); else if (SEG == 8) NF_S(8, DIRV, SLABV, NBV); \
        else if (SEG == 16 && NBV == 0) NF_S(16, DIRV, SLABV, 0); else if (SEG == 32 && NBV == 0) NF_S(32, DIRV, SLABV, 0); else return fail(NF_ERR_ARG, "bad s_seg"); } while (0)
    if (zmode != 0) {
        const long nl = S->nlines[2];
        sa.if_lo = S->if_lo; sa.if_hi = S->if_hi; sa.mode = zmode;
        sa.alo = S->d_alo + g * nl; sa.ahi = S->d_ahi + g * nl; sa.ulo = S->d_ulo; sa.uhi = S->d_uhi; sa.clo = S->d_clo; sa.chi = S->d_chi;
        if (zmode == 3) {                                           // current reconstruction: face DOFs and bubbles of every transverse mode
            int nfa = 1, ni = S->k; for (int t = 1; t < S->dim; ++t) { nfa *= S->k + 1; ni *= S->k + 1; }
            sa.nfa = nfa; sa.ni = ni;
            sa.jz = S->d_Jz + (size_t)g * S->nJz; sa.jzb = S->d_Jzb ? S->d_Jzb + (size_t)g * S->N * ni : nullptr;
            for (int m = 0; m < mt.n && m < 9; ++m) {                // RT transverse index of P mode m: a = sum a_t (k+1)^t (src/FEM.cpp:364-374)
                int amode = 0, q = m, mul = 1;
                for (int t = 0; t < S->dim; ++t) { if (t == d) continue; amode += (q % S->n1) * mul; q /= S->n1; mul *= S->k + 1; }
                sa.amode[m] = amode;
            }
        }
        if (zmode == 2 && T->sep_sweeps == 0 && T->opt_sepfold) {     // separators formed inside the pass (no k_separators launch)
            sa.fold = 1; sa.rlo = S->d_rlo; sa.rhi = S->d_rhi; sa.sinv_lo = S->d_sinv_lo + g * nl; sa.sinv_hi = S->d_sinv_hi + g * nl;
        }
        if (S->nb == 0) NF_S_SEG(2, true, 0); else if (S->nb == 1) NF_S_SEG(2, true, 1); else NF_S_SEG(2, true, 2);
    } else if (d == 1) {
        if (S->nb == 0) NF_S_SEG(1, false, 0); else if (S->nb == 1) NF_S_SEG(1, false, 1); else NF_S_SEG(1, false, 2);
    } else {
        if (S->nb == 0) NF_S_SEG(2, false, 0); else if (S->nb == 1) NF_S_SEG(2, false, 1); else NF_S_SEG(2, false, 2);
    }
#undef NF_S_SEG
#undef NF_S
    return NF_OK;
}

// Partition method, first half (slab teams): the endpoint pass of every slab (mode 1), the plane exchange and the separator
// values (+ Jacobi sweeps for thin slabs).  The exchange and the separator kernels run on the comm stream; ev_xchg marks
// their end -- the caller's z pass (or current reconstruction) waits for it, the x / y passes do not.
static int team_endpoint_phase(nf_team *T, int g, const std::vector<const double *> &xs, const std::vector<double *> &ys, const CgScalars *cg,
                               hipEvent_t after_z1, bool need_u = false)
{
    const int ns = (int)T->slabs.size();
    for (int i = 0; i < ns; ++i) {
        nf_solver *S = T->slabs[i];
        if (!(S->if_lo || S->if_hi)) continue;
        S->sr_cnt3 = 0;
        if (S->cg1.red && S->w_valid && T->opt_endpoint_w && S->fuse.p == xs[i]) {
            // single-reduction CG: the chain-end responses as weighted sums of the line's cells + the deferred CG update (k_endpoint_w)
            const dim3 gr((unsigned)((S->nx + 63) / 64), (unsigned)S->ny);
            if ((long)gr.x * gr.y <= T->slab_cap) {
                hipLaunchKernelGGL(k_endpoint_w, gr, dim3(256), 0, T->stream, S->fuse.p, const_cast<double *>(S->fuse.r), (const double *)S->d_q, S->fuse.xsol,
                                   (const double *)(S->d_Wlo + (size_t)g * S->N), (const double *)(S->d_Whi + (size_t)g * S->N), S->d_clo, S->d_chi,
                                   S->nx, S->ny, S->nz, S->if_lo, S->if_hi, 40, S->cg1, cg, T->d_partials + 3 * T->partial_stride + (long)S->slab_index * T->slab_cap);
                S->sr_cnt3 = (int)(gr.x * gr.y); T->last_endpoint_w = 1;
                continue;
            }
        }
        NFCHK(launch_s(S, 2, g, mode_args(S, g, 2, 0, xs[i], ys[i]), make_geom(S), 0, nullptr, cg, nullptr, 1));
    }
    if (after_z1) (void)hipEventRecord(after_z1, T->stream);
    HIPCHK(hipEventRecord(T->ev_z1, T->stream));
    HIPCHK(hipStreamWaitEvent(T->comm_stream, T->ev_z1, 0));
    NFCHK(exchange_planes(T, 0, 0, T->comm_stream));
    auto each_slab = [&](auto &&launch) {
        for (int i = 0; i < ns; ++i) {
            nf_solver *S = T->slabs[i];
            if (S->if_lo || S->if_hi) launch(S, S->nlines[2], dim3((unsigned)((S->nlines[2] * n_modes(S) + 255) / 256)));
        }
    };
    // thick slabs (no separator sweeps): the accumulation pass forms u = (c_below + c_above) S_red^-1 itself (SlabArgs::fold)
    if (need_u || T->sep_sweeps > 0 || !T->opt_sepfold)
        each_slab([&](nf_solver *S, long nl, dim3 gr) {
            hipLaunchKernelGGL(k_separators, gr, dim3(256), 0, T->comm_stream, S->d_clo, S->d_chi, S->d_rlo, S->d_rhi, S->d_sinv_lo + g * nl, S->d_sinv_hi + g * nl,
                               S->d_ulo, S->d_uhi, S->d_ctlo, S->d_cthi, nl, nl * n_modes(S), S->if_lo, S->if_hi, cg); });
    for (int sweep = 0; sweep < T->sep_sweeps; ++sweep) {
        each_slab([&](nf_solver *S, long nl, dim3 gr) {
            hipLaunchKernelGGL(k_sep_couple, gr, dim3(256), 0, T->comm_stream, S->d_gfl + g * nl, S->d_ulo, S->d_uhi, S->d_elo, S->d_ehi, nl, nl * n_modes(S), S->if_lo, S->if_hi, cg); });
        NFCHK(exchange_planes(T, 2, 0, T->comm_stream));
        each_slab([&](nf_solver *S, long nl, dim3 gr) {
            hipLaunchKernelGGL(k_sep_update, gr, dim3(256), 0, T->comm_stream, S->d_ctlo, S->d_cthi, S->d_elo, S->d_ehi, S->d_relo, S->d_rehi,
                               S->d_sinv_lo + g * nl, S->d_sinv_hi + g * nl, S->d_ulo, S->d_uhi, nl, nl * n_modes(S), S->if_lo, S->if_hi, cg); });
    }
    HIPCHK(hipEventRecord(T->ev_xchg, T->comm_stream));
    return NF_OK;
}

static bool team_is_single(const nf_team *T) { return T->slabs.size() == 1 && !T->slabs[0]->if_lo && !T->slabs[0]->if_hi; }

// y = S_g x on every local slab.  xs / ys: per-slab device pointers (nphi doubles, layout [p][e]).  With `want_dot` the
// passes of the last direction leave the block partials of x.y in the team buffer and counts[] receives the number per slab.
static int team_schur_apply(nf_team *T, int g, const std::vector<const double *> &xs, const std::vector<double *> &ys, bool want_dot,
                            const CgScalars *cg, std::vector<int> *counts)
{
    const int ns = (int)T->slabs.size();
    const int dim = T->slabs[0]->dim;
    bool any_if = false; for (auto *S : T->slabs) any_if |= S->if_lo || S->if_hi;
    hipEvent_t a, b, ta = nullptr, tb = nullptr;
    // event-timed: every prof_every-th apply (an event record keeps the launches around it from being dispatched back to back:
    // eight records per apply cost 36 us per CG iteration at 256^3, 6 % of the iteration they are meant to measure)
    const bool prof = T->profile && (T->prof_tick++ % T->prof_every == 0);
    if (prof) prof_begin(T, 3, &ta, &tb);
    if (any_if) {                                                 // partition method step 1 + interface exchange
        if (prof) prof_begin(T, 4, &a, &b);
        NFCHK(team_endpoint_phase(T, g, xs, ys, cg, prof ? b : nullptr));
    }
    // Split dot product (undivided RT0-P0 meshes outside the lean path, i.e. the big ones): every pass leaves the partials of ITS share
    // of x.y -- the x pass x.(C x + S_x x), the y / z passes T_a sum z_f w_f from their forward sweeps (schur_s_tile, ZW) -- one after
    // the other in the slab's segment of the partial buffer; the consumer (k_finalize) sums them all.  The last pass then needs x only
    // for its forward sweep.
    // Taken where a chunked long-line pass runs (it has no x left when its parked chunk comes back) or on request (split_dot = 2: tests);
    // elsewhere the last pass sums x_i y_i with x still in its registers, which is cheaper than three sets of partials (256^3: 504 vs
    // 531 us per CG iteration, profiles/r03_e_ab_cg.txt).
    bool split = want_dot && dim >= 2 && team_is_single(T) && T->slabs[0]->nb == 0 && !T->slabs[0]->lean.st && T->opt_split_dot;
    if (split && T->opt_split_dot < 2) split = chunk_plan(T->slabs[0], dim - 1).ok;      // only the LAST pass forms the dot product
    std::vector<int> totals(ns, 0);
    for (int d = 0; d < dim; ++d) {
        const int last = d == dim - 1;
        if (d == 2 && any_if) HIPCHK(hipStreamWaitEvent(T->stream, T->ev_xchg, 0));
        if (prof) prof_begin(T, d, &a, &b);
        // x || y (slab teams, RT0-P0, slabs small enough not to fill the chip): fork after the endpoint pass -- the new p is complete --,
        // the y pass of every local slab on y_stream into d_qy, join before the accumulation pass, which adds d_qy
        const bool xy = dim == 3 && any_if && T->opt_xy_overlap && T->slabs[0]->nb == 0 && T->y_stream && !prof;
        bool xy_all = xy; if (xy) for (auto *X : T->slabs) xy_all = xy_all && X->N <= T->xy_overlap_max_cells && !chunk_plan(X, 1).ok;
        if (d == 0 && xy_all) { HIPCHK(hipEventRecord(T->ev_fork, T->stream)); HIPCHK(hipStreamWaitEvent(T->y_stream, T->ev_fork, 0)); }
        if (d == 2 && xy_all) { HIPCHK(hipEventRecord(T->ev_join, T->y_stream)); HIPCHK(hipStreamWaitEvent(T->stream, T->ev_join, 0)); }
        for (int i = 0; i < ns; ++i) {
            nf_solver *S = T->slabs[i];
            const Geom G = make_geom(S);
            if (xy_all && !S->d_qy) NFCHK(dalloc(&S->d_qy, S->nphi));
            {
                double *part = (want_dot && (last || split)) ? T->d_partials + i * T->slab_cap + totals[i] : nullptr;
                const ModeArgs ma = mode_args(S, g, d, 0, xs[i], ys[i]);     // mode 0; the kernels derive the others (ModeTab)
                int np = 0;
                S->zw_dot = split;
                if (d == 0) NFCHK(launch_x(S, g, ma, G, last || split, part, cg, &np));
                else if (d == 2 && (S->if_lo || S->if_hi)) {
                    S->pass_yadd = xy_all ? S->d_qy : nullptr;
                    const int rz = launch_s(S, 2, g, ma, G, last, part, cg, &np, 2);
                    S->pass_yadd = nullptr;
                    NFCHK(rz);
                } else if (d == 1 && xy_all) {
                    S->pass_stream = T->y_stream; S->pass_noacc = true;
                    const int ry = launch_s(S, 1, g, mode_args(S, g, 1, 0, xs[i], S->d_qy), G, 0, nullptr, cg, &np, 0);
                    S->pass_stream = nullptr; S->pass_noacc = false;
                    NFCHK(ry);
                } else NFCHK(launch_s(S, d, g, ma, G, last || split, part, cg, &np, 0));
                S->zw_dot = false;
                if (part) totals[i] += np;
            }
            if (counts && last) (*counts)[i] = totals[i];
        }
        if (prof) (void)hipEventRecord(b, T->stream);
    }
    if (prof) (void)hipEventRecord(tb, T->stream);
    return NF_OK;
}


int nf_schur_apply(nf_handle S, int g, const double *x_dev, double *y_dev)
{
    if (!S || !x_dev || !y_dev || g < 0 || g >= S->ng) return fail(NF_ERR_ARG, "nf_schur_apply: bad arguments");
    if (!S->built) return fail(NF_ERR_STATE, "nf_schur_apply: call nf_build first");
    nf_team *T = S->team;
    if (!team_is_single(T)) return fail(NF_ERR_STATE, "nf_schur_apply works on an undivided mesh; use nf_team_schur_apply for slabs");
    HIPCHK(hipSetDevice(S->device));
    NFCHK(team_schur_apply(T, g, { x_dev }, { y_dev }, false, nullptr, nullptr));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(T->stream));
    if (T->profile) prof_collect(T);
    return NF_OK;
}

// team variant: x/y pointer arrays with one entry per local slab (N_slab doubles each, device)
int nf_team_schur_apply(nf_handle S, int g, const double *const *x_dev, double *const *y_dev)
{
    if (!S || !x_dev || !y_dev || g < 0 || g >= S->ng) return fail(NF_ERR_ARG, "nf_team_schur_apply: bad arguments");
    if (S->team->dead) return fail(NF_ERR_COMM, "this team lost a collective (NF_ERR_COMM) and is unusable: destroy the handles and end the process with a non-zero code (never re-exec)");
    nf_team *T = S->team;
    HIPCHK(hipSetDevice(T->device));
    NFCHK(team_prepare(T));
    std::vector<const double *> xs(x_dev, x_dev + T->slabs.size()); std::vector<double *> ys(y_dev, y_dev + T->slabs.size());
    NFCHK(team_schur_apply(T, g, xs, ys, false, nullptr, nullptr));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(T->stream));
    if (T->profile) prof_collect(T);
    return NF_OK;
}

#ifdef NF_STAMPS
static long long *g_stamps = nullptr;
extern "C" int nf_debug_stamps(long long *out48) { if (!g_stamps) return -1; return hipMemcpy(out48, g_stamps, 48 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2; }
#endif
// ---- fused-direction apply (k_apply3): geometry of the launch ------------------------------------
struct Fuse3Plan { bool ok = false; int nch = 1, vec = 1, seg = 8, B = 256, nblocks = 0; size_t lds = 0; Apply3 A; };
static Fuse3Plan fuse3_plan(const nf_solver *S)
{
    Fuse3Plan P; memset(&P.A, 0, sizeof P.A);
    const int K = 2, modes = n_modes(S);
    int lanes = (S->nx + K - 1) / K, lpl_log2 = 0;
    while ((1 << lpl_log2) < lanes && lpl_log2 < 6) ++lpl_log2;
    const int LPL = 1 << lpl_log2, LPW = 64 / LPL;
    const int nch = (S->nx + LPL * K - 1) / (LPL * K);
    if (nch > 4) return P;                                       // x lines beyond 512 cells: registers of a 512-thread block
    P.nch = nch <= 1 ? 1 : nch <= 2 ? 2 : 4; P.vec = S->nx % 2 == 0;
    const int nmax = std::max(S->dim >= 2 ? S->ny : 1, S->dim == 3 ? S->nz : 1);
    P.seg = S->nb > 0 ? (nmax <= 256 ? 4 : 8) : 8;
    int B = 256;
    for (int r = 0; r < 2; ++r) {
        if (S->dim < r + 2) continue;
        const int n = r == 0 ? S->ny : S->nz;
        const int NSEG = (n + P.seg - 1) / P.seg;
        if (NSEG > 64) return P;
        // small meshes: narrow tiles, so that the launch spreads over many CUs (one CU pulls ~10 B/cycle; measured with in-kernel
        // stamps at 38x38x19: 64-column tiles spent 4 us of a 6 us block in their load phase); big ones: wide rows for HBM
        int TX = S->N <= (1L << 16) ? 8 : S->N <= (1L << 18) ? 16 : S->N <= (1L << 20) ? 32 : 64;
        while (TX > 8 && TX * NSEG > 512) TX >>= 1;
        while (TX > 8 && TX / 2 >= S->nx) TX >>= 1;
        const long nxy = (long)S->nx * S->ny;
        P.A.n[r] = n; P.A.TX[r] = TX; P.A.NSEG[r] = NSEG; P.A.gx[r] = (S->nx + TX - 1) / TX; P.A.gy[r] = r == 0 ? S->nz : S->ny;
        P.A.sl[r] = r == 0 ? S->nx : nxy; P.A.ostride[r] = r == 0 ? nxy : S->nx;
        B = std::max(B, (TX * NSEG + 63) / 64 * 64);
    }
    P.B = B;
    const int nw = B / 64;
    P.A.ntask_x = (int)((S->nlines[0] + LPW - 1) / LPW); P.A.lpl_log2 = lpl_log2;
    P.A.nbx = (int)(((long)P.A.ntask_x * modes + nw - 1) / nw);
    P.A.nby = S->dim >= 2 ? P.A.gx[0] * P.A.gy[0] * modes : 0;
    P.A.nbz = S->dim == 3 ? P.A.gx[1] * P.A.gy[1] * modes : 0;
    P.nblocks = P.A.nbx + P.A.nby + P.A.nbz;
    P.lds = (size_t)(4 * B + 320 + 16) * sizeof(double);
    P.A.stamps = nullptr;
#ifdef NF_STAMPS
    { static long long *d_st = nullptr; if (!d_st) (void)hipMalloc((void **)&d_st, 48 * sizeof(long long)); P.A.stamps = d_st; g_stamps = d_st; }
#endif
    P.ok = true;
    return P;
}
// one k_apply3 launch: q_d = S_d p for every direction; pin = the vector the iteration reads, pout = where the new p goes
static int launch_apply3(nf_solver *S, int g, const Fuse3Plan &P, const double *pin, double *pout, double *xsol, const CgLean &lean)
{
    nf_team *T = S->team; hipStream_t st = T->stream;
    const long N = S->N;
    const Geom G = make_geom(S);
    ModeArgs ma[3]; ModeTab mt[3];
    double *qd[3] = { S->d_q, S->d_qy, S->d_qz };
    for (int d = 0; d < 3; ++d) {
        const int dd = d < S->dim ? d : 0;
        ma[d] = mode_args(S, g, dd, 0, pin, qd[d < S->dim ? d : 0]); mt[d] = mode_tab(S, dd);
    }
    const double *L[3], *DR[3], *D0[3];
    for (int d = 0; d < 3; ++d) { const int dd = d < S->dim ? d : 0; L[d] = S->d_L[dd] + g * N; DR[d] = S->d_DR[dd] + g * N; D0[d] = S->d_D0[dd] + g * S->nlines[dd]; }
    const CgFuse fz = { const_cast<double *>(pin), S->d_r, xsol, pout };
#define NF_A3(NCHV, VECV, NBV, SEGV) hipLaunchKernelGGL((k_apply3<NCHV, VECV, NBV, SEGV>), dim3((unsigned)P.nblocks), dim3((unsigned)P.B), P.lds, st, ma[0], ma[1], ma[2], mt[0], mt[1], mt[2], G, \
        L[0], DR[0], D0[0], L[1], DR[1], D0[1], L[2], DR[2], D0[2], S->nx, S->ny, S->nlines[0], P.A, T->d_partials, T->d_cg, fz, lean)
#define NF_A3_SEG(NCHV, VECV, NBV) do { if (NBV == 0 || P.seg == 8) NF_A3(NCHV, VECV, NBV, 8); else NF_A3(NCHV, VECV, NBV, (NBV == 0 ? 8 : 4)); } while (0)
#define NF_A3_NB(NCHV, VECV) do { if (S->nb == 0) NF_A3_SEG(NCHV, VECV, 0); else if (S->nb == 1) NF_A3_SEG(NCHV, VECV, 1); else NF_A3_SEG(NCHV, VECV, 2); } while (0)
#define NF_A3_VEC(NCHV) do { if (P.vec) NF_A3_NB(NCHV, true); else NF_A3_NB(NCHV, false); } while (0)
    if (P.nch == 1) NF_A3_VEC(1); else if (P.nch == 2) NF_A3_VEC(2); else NF_A3_VEC(4);
#undef NF_A3_VEC
#undef NF_A3_NB
#undef NF_A3_SEG
#undef NF_A3
    return NF_OK;
}

// ---- whole CG solve on one XCD (k_cg_xcd) ---------------------------------------------------------
// CgScalars hold the outcome of FIN_RHS, r = p = rhs, x = 0.  One launch; the kernel publishes the final scalars to the host itself.
static bool xcd_eligible(const nf_team *T, const nf_solver *S, const Fuse3Plan &P)
{
    if (!T->opt_cgx || !P.ok || T->profile) return false;
    if (S->nb == 0 ? (S->nloc != 1 || n_modes(S) != 1) : (P.seg != 4 || n_modes(S) > 9 || S->nb > 2)) return false;   // P0 flux: one unknown per cell; bubble moments: the SEG = 4 tiles
    if (S->nphi < T->xcd_min_cells || S->nphi > T->xcd_max_cells) return false;                                      // the window counts unknowns per group
    for (int r = 0; r < 2; ++r) if (S->dim >= r + 2 && (P.A.NSEG[r] >= 64 || P.A.TX[r] * P.A.NSEG[r] > 448)) return false;   // no wavefront scan in the packed tiles; a tile fits beside the other roles
    return S->nx <= 128;                                          // x lines: at most two chunks per lane (128 VGPRs at 1024 threads)
}
// geometry, factors (of group g) and CG vectors of a k_cg_xcd / k_keff_xcd launch; *nch_out: chunks per lane of the x lines
static int xcd_fill(nf_solver *S, int g, const Fuse3Plan &P, XcdArgs &A, int *nch_out)
{
    nf_team *T = S->team;
    const long N = S->N;
    if (!T->d_xcd) { HIPCHK(hipMalloc((void **)&T->d_xcd, sizeof(XcdState))); NFCHK(dalloc(&T->d_xpart, 512)); HIPCHK(hipMemset(T->d_xpart, 0, 512 * sizeof(double))); }
    HIPCHK(hipMemsetAsync(T->d_xcd, 0, sizeof(XcdState), T->stream));
    memset(&A, 0, sizeof A);
    A.G = make_geom(S);
    double *qd[3] = { S->d_q, S->d_qy, S->d_qz };
    for (int d = 0; d < 3; ++d) {
        const int dd = d < S->dim ? d : 0;
        A.ma[d] = mode_args(S, g, dd, 0, S->d_p, qd[dd]);
        A.L[d] = S->d_L[dd] + g * N; A.DR[d] = S->d_DR[dd] + g * N; A.D0[d] = S->d_D0[dd] + g * S->nlines[dd]; A.q[d] = qd[dd];
    }
    A.nx = S->nx; A.ny = S->ny; A.dim = S->dim; A.nlines_x = S->nlines[0]; A.N = S->nphi;
    A.nmodes = n_modes(S);
    for (int d = 0; d < 3; ++d) A.mt[d] = mode_tab(S, d < S->dim ? d : 0);
    // x lines: few wavefronts matter more here than short scans (the roles share the 12 wavefronts of a workgroup, and a second round
    // costs a whole memory round trip): two chunks per lane from 33 cells on -- 38 cells = 16 lanes x 2 cells x 2 chunks, four lines
    // per wavefront, where the launch path takes 32 lanes and two lines
    int lanes = (S->nx + 1) / 2, lpl_log2 = 0;
    while ((1 << lpl_log2) < lanes && lpl_log2 < 5) ++lpl_log2;
    int nch = (lanes + (1 << lpl_log2) - 1) >> lpl_log2;
    if (nch == 1 && lpl_log2 == 5) {
        // 32 lanes, one chunk (the shorter chain) if the three roles of a workgroup still fit its wavefronts in one round
        const int Pn = T->xcd_groups, tasks = (int)((S->nlines[0] + 1) / 2) * A.nmodes;
        int waves = (tasks + Pn - 1) / Pn;
        for (int r = 0; r < 2; ++r) if (S->dim >= r + 2) waves += (((P.A.gx[r] * P.A.gy[r] * A.nmodes + Pn - 1) / Pn) * P.A.TX[r] * P.A.NSEG[r] + 63) / 64;
        if (waves > XCD_THREADS / 64) { lpl_log2 = 4; nch = 2; }
    }
    if (nch > 2) return fail(NF_ERR_STATE, "k_cg_xcd: x lines of %d cells", S->nx);
    A.lpl_log2 = lpl_log2; A.ntask_x = (int)((S->nlines[0] + (64 >> lpl_log2) - 1) / (64 >> lpl_log2));
    for (int r = 0; r < 2; ++r) { A.n[r] = P.A.n[r]; A.TX[r] = P.A.TX[r]; A.NSEG[r] = P.A.NSEG[r]; A.gx[r] = P.A.gx[r]; A.gy[r] = P.A.gy[r]; A.sl[r] = P.A.sl[r]; A.ostride[r] = P.A.ostride[r]; }
    if (S->dim < 2) { A.TX[0] = A.NSEG[0] = 1; } if (S->dim < 3) { A.TX[1] = A.NSEG[1] = 1; }
    A.pA = S->d_p; A.pB = S->d_p2; A.r = S->d_r;
    A.part = T->d_xpart; A.st = T->d_xcd; A.xcc = T->xcd_id;
    *nch_out = nch;
    return NF_OK;
}
static const size_t XCD_LDS = (size_t)(5 * 1024 + 64 + 32) * sizeof(double);
static int launch_cg_xcd(nf_solver *S, int g, const Fuse3Plan &P, double *xsol, unsigned long long seq)
{
    nf_team *T = S->team; hipStream_t st = T->stream;
    XcdArgs A; int nch = 1;
    NFCHK(xcd_fill(S, g, P, A, &nch));
    A.xsol = xsol; A.cg = T->d_cg; A.hp = T->d_pub; A.seq = seq;
    const size_t lds = XCD_LDS;
    const unsigned G = 8u * (unsigned)T->xcd_groups;
#define NF_XCD(NCHV, VECV, NBV, SEGV) do { if (!lds_opt_in((const void *)k_cg_xcd<NCHV, VECV, NBV, SEGV>, lds)) return fail(NF_ERR_HIP, "k_cg_xcd: %zu bytes of LDS refused", lds); \
        hipLaunchKernelGGL((k_cg_xcd<NCHV, VECV, NBV, SEGV>), dim3(G), dim3(XCD_THREADS), lds, st, A); } while (0)
#define NF_XCD_NB(NCHV, VECV) do { if (S->nb == 0) NF_XCD(NCHV, VECV, 0, 8); else if (S->nb == 1) NF_XCD(NCHV, VECV, 1, 4); else NF_XCD(NCHV, VECV, 2, 4); } while (0)
    if (nch == 1) { if (P.vec) NF_XCD_NB(1, true); else NF_XCD_NB(1, false); }
    else { if (P.vec) NF_XCD_NB(2, true); else NF_XCD_NB(2, false); }
#undef NF_XCD_NB
#undef NF_XCD
    HIPCHK(hipGetLastError());
    return NF_OK;
}

// ---- CG (SchurSolver::SolveSchurImplicit, src/solvers.cpp:577-636) -----------------------------
// rhs / x: per-slab pointers
// inited: k_group_rhs has already written x = 0, r = p = rhs and the |rhs|^2 partials (one launch less per group solve)
static int cg_solve(nf_team *T, int g, const std::vector<const double *> &rhs, const std::vector<double *> &x, double tol, int maxit,
                    int *its_out, double *res_out, bool inited = false)
{
    const int ns = (int)T->slabs.size();
    std::vector<int> gcnt(ns), acnt(ns);
    std::vector<const double *> ps(ns); std::vector<double *> qs(ns);
    for (int i = 0; i < ns; ++i) {
        nf_solver *S = T->slabs[i];
        gcnt[i] = grid_for(S->nphi); ps[i] = S->d_p; qs[i] = S->d_q;
        if (!inited) hipLaunchKernelGGL(k_cg_init, dim3(gcnt[i]), dim3(256), 0, T->stream, rhs[i], x[i], S->d_r, S->d_p, S->nphi, T->d_partials + i * T->slab_cap);
    }
    NFCHK(team_finalize(T, FIN_RHS, gcnt, 1, T->d_out, tol, maxit));
    CgScalars sc; memset(&sc, 0, sizeof sc);
    int launched = 0;
    // first batch: what the previous solve of this group needed, plus two (an iteration launched past convergence costs five
    // early-exit kernels, ~10 us; a second host check costs a D2H copy and a stream drain, ~50 us)
    int batch = T->cg_batch > 0 ? T->cg_batch : (T->last_its[g] > 0 ? T->last_its[g] + 2 : 1), grow = 2;
    // fused variant (RT0-P0, undivided mesh): x_sol / p updates ride in the next x pass (k_schur_x, CgFuse)
    // (undivided mesh: in the x pass; slab teams: in the endpoint pass of the z lines, the first pass to read p)
    const bool fused = T->opt_fuse != 0 && (team_is_single(T) || T->slabs[0]->nloc == 1);   // undivided: any order (x pass); slab teams: P0 (z endpoint pass)
    for (int i = 0; i < ns; ++i) T->slabs[i]->fuse = fused ? CgFuse{ T->slabs[i]->d_p, T->slabs[i]->d_r, x[i] } : CgFuse{ nullptr, nullptr, nullptr };
    // lean variant on top of the fused one (undivided mesh, no RCCL): no k_finalize launches, see CgLean.  Row 0 of the partial
    // buffer holds the p.q partials of the last direction pass, row 1 the |r|^2 partials of k_cg_rupdate.
    const bool lean = fused && T->opt_lean && team_is_single(T) && !T->rccl_reduce && T->slabs[0]->N <= T->lean_max_cells;
    // every block of the next x pass sums the |r|^2 partials; fewer partials (cg_lean_grid) cost k_cg_rupdate more than they save (measured)
    const int gru = lean ? grid_for(T->slabs[0]->nphi, 256, T->opt_lean_grid) : 0;
    double *row1 = T->d_partials + T->partial_stride;
    const CgLean no_lean = { nullptr, nullptr, 0, 0, 0 };
    int rc = NF_OK;
    // lean variant for slab teams (fused, RT0-P0): the endpoint pass of the z lines and k_cg_rupdate consume the all-reduced
    // totals (d_red[2] = |r|^2, d_red[0] = p.q, each followed by the ranks' error flags) and derive beta / alpha and the stop tests themselves: no k_cg_logic launches
    const bool tlean = fused && T->opt_lean && !team_is_single(T);
    // single-reduction variant of it (Cg1): one reduction per iteration; needs the 8-cell-segment z passes on every local slab
    bool sr = tlean && (T->opt_cg1 > 0 || (T->opt_cg1 < 0 && T->team_max_cells <= T->cg1_max_cells));
    for (auto *S : T->slabs) sr = sr && S->dim == 3 && S->nb == 0 && (T->opt_s_seg == 0 || T->opt_s_seg == 8) && S->nz <= 1024;
    T->last_cg_reductions = team_is_single(T) ? 0 : (sr ? 1 : 2); T->last_endpoint_w = 0;
    // fused-direction variant on top of the lean one (small / medium meshes): two launches per iteration, see k_apply3
    Fuse3Plan f3;
    nf_solver *S0 = T->slabs[0];
    if (lean && T->opt_fuse3 && S0->N <= T->fuse3_max_cells) f3 = fuse3_plan(S0);
    if (f3.ok && f3.nblocks > T->slab_cap) f3.ok = false;
    if (f3.ok) {
        if (!S0->d_p2) NFCHK(dalloc(&S0->d_p2, S0->nphi));
        if (S0->dim >= 2 && !S0->d_qy) NFCHK(dalloc(&S0->d_qy, S0->nphi));
        if (S0->dim == 3 && !S0->d_qz) NFCHK(dalloc(&S0->d_qz, S0->nphi));
    }
    T->last_xcd = 0;
    if (f3.ok && xcd_eligible(T, S0, f3) && pub_ready(T)) {
        // the whole solve in one launch on one XCD (k_cg_xcd); the kernel hands the final scalars to the host itself
        const unsigned long long seq = ++T->pub_seq;
        for (int i = 0; i < ns; ++i) T->slabs[i]->fuse = CgFuse{ nullptr, nullptr, nullptr };
        NFCHK(launch_cg_xcd(S0, g, f3, x[0], seq));
        NFCHK(pub_wait(T, seq, &sc, nullptr, 0));
        if (sc.err == 4) {
            // a grid barrier timed out in the middle of the solve (a participant stalled for longer than the bound: another process on
            // the GPU, a debugger): x, r and p are half-way, but the right-hand side is intact -- start the solve again on the launch
            // path (the kernels have all left by now: the flag that ended one ended the others) and keep this solver off the one-XCD path
            T->opt_cgx = 0; ++T->xcd_refused;
            HIPCHK(hipStreamSynchronize(T->stream));
            for (int i = 0; i < ns; ++i) {
                nf_solver *S = T->slabs[i];
                hipLaunchKernelGGL(k_cg_init, dim3(gcnt[i]), dim3(256), 0, T->stream, rhs[i], x[i], S->d_r, S->d_p, S->nphi, T->d_partials + i * T->slab_cap);
            }
        }
        if (sc.err != 3 && sc.err != 4) {
            T->last_xcd = 1; ++T->xcd_solves;
            HIPCHK(hipGetLastError());
            if (!std::isfinite(sc.rr)) return fail(NF_ERR_NUMERIC, "CG produced a non-finite residual (group %d)", g);
            T->last_its[g] = sc.its;
            if (its_out) *its_out = sc.its;
            if (res_out) *res_out = sc.rhs_norm > 0 ? std::sqrt(sc.rr) / sc.rhs_norm : 0.0;
            return NF_OK;
        }
        // the workgroups did not assemble (nothing placed on the XCD, or it is busy): no vector has been touched -- this solve and the
        // following ones of this solver go through the launch path
        if (sc.err == 3) { T->opt_cgx = 0; ++T->xcd_refused; }
        for (int i = 0; i < ns; ++i) T->slabs[i]->fuse = fused ? CgFuse{ T->slabs[i]->d_p, T->slabs[i]->d_r, x[i] } : CgFuse{ nullptr, nullptr, nullptr };
        NFCHK(team_finalize(T, FIN_RHS, gcnt, 1, T->d_out, tol, maxit));
        memset(&sc, 0, sizeof sc);
    }
    // A local failure on a multi-rank team (a refused launch, a failed HIP call) must not end this rank's part of the schedule: the
    // other ranks are about to wait in the collectives of this batch.  The rank raises its error flag (it rides in every reduction
    // that crosses ranks, CgScalars::err), stops launching compute, keeps issuing exactly the collectives the schedule holds, and
    // returns when the flag has stopped every rank -- at the same iteration everywhere.
    const bool multi = T->nproc > 1;
    bool any_if = false; for (auto *S : T->slabs) any_if |= S->if_lo || S->if_hi;
    // vector reduce (see nf_team::d_vec): row 0 of the partial buffer = this rank's p.q partials + flag slot, row 1 = |r|^2 partials + flag slot
    const bool vred = tlean && !sr && multi && ns == 1 && T->vec_ok && T->opt_vec_reduce;
    double *send_pq = T->d_partials, *send_rr = T->d_partials + T->partial_stride;
    double *vec_pq = T->d_vec, *vec_rr = vred ? T->d_vec + T->vec_stride : nullptr;
    T->last_vec_reduce = vred ? 1 : 0;
    if (vred) {                                                   // other kernels use these rows between solves: clear the two flag slots
        HIPCHK(hipMemsetAsync(send_pq + T->vec_cnt_pq, 0, sizeof(double), T->stream));
        HIPCHK(hipMemsetAsync(send_rr + T->vec_cnt_rr, 0, sizeof(double), T->stream));
    }
    auto bad = [&](int code) -> bool {                            // true: leave the loop (single process); false: carry on (poisoned or fine)
        if (code == NF_OK) return false;
        if (!multi) { rc = code; return true; }
        if (!T->poisoned) {
            T->poisoned = true; T->poison_rc = code; snprintf(T->poison_msg, sizeof T->poison_msg, "%s", nf_last_error());
            const double one = 1.0;
            (void)hipMemcpyAsync(T->d_errsrc, &one, sizeof one, hipMemcpyHostToDevice, T->stream);
            if (vred) {                                           // the flag slots of the two vectors this rank sends from now on
                (void)hipMemcpyAsync(send_pq + T->vec_cnt_pq, &one, sizeof one, hipMemcpyHostToDevice, T->stream);
                (void)hipMemcpyAsync(send_rr + T->vec_cnt_rr, &one, sizeof one, hipMemcpyHostToDevice, T->stream);
            }
            (void)hipStreamSynchronize(T->stream);               // `one` lives on this stack frame; the stream may also have to recover from the failed call
            (void)hipGetLastError();
        }
        return false;
    };
    while (launched < maxit && rc == NF_OK) {
        int nb = std::min(batch, maxit - launched);
        for (int it = 0; it < nb && rc == NF_OK; ++it) {
            const int index = launched + it;                      // iteration number within this solve
            const long global_it = T->cg_iter_total++;
            if (f3.ok) {
                // p lives in a pair of buffers: iteration 0 reads A = d_p as k_cg_init left it; iteration i >= 1 reads the p of
                // iteration i-1 (A for odd i, B for even i) and writes r + beta p into the other one
                double *A = S0->d_p, *B = S0->d_p2;
                double *pin = (index == 0 || (index & 1)) ? A : B, *pout = pin == A ? B : A;
                hipEvent_t ta = nullptr, tb = nullptr;
                const bool prof = T->profile && (T->prof_tick++ % T->prof_every == 0);
                if (prof) prof_begin(T, 3, &ta, &tb);
                rc = launch_apply3(S0, g, f3, pin, pout, x[0], CgLean{ T->d_cg, row1, gru, index & 1, index == 0 ? 1 : 0 });
                if (prof) (void)hipEventRecord(tb, T->stream);
                hipLaunchKernelGGL(k_cg_rupdate3, dim3(gru), dim3(256), 0, T->stream, S0->d_r, (const double *)S0->d_q, (const double *)(S0->dim >= 2 ? S0->d_qy : nullptr),
                                   (const double *)(S0->dim == 3 ? S0->d_qz : nullptr), S0->nphi, T->d_cg, row1, CgLean{ T->d_cg, T->d_partials, f3.nblocks, index & 1, 0 });
                continue;
            }
            if (lean) T->slabs[0]->lean = CgLean{ T->d_cg, row1, gru, index & 1, index == 0 ? 1 : 0 };
            if (sr) for (auto *S : T->slabs) S->cg1 = Cg1{ T->d_red, T->d_cg, index };
            else if (tlean) for (auto *S : T->slabs) S->lean_z1 = vred ? CgLean{ T->d_cg, vec_rr, T->vec_cnt_rr, index & 1, index == 0 ? 1 : 0, T->vec_cnt_rr }
                                                                   : CgLean{ T->d_cg, T->d_red + 2, -1, index & 1, index == 0 ? 1 : 0 };
            T->xchg_in_apply = 0;
            int ra = NF_OK;
            if (multi && T->rank == T->inject_rank && global_it == T->inject_iter)
                { ra = fail(NF_ERR_HIP, "injected failure on rank %d at CG iteration %ld (NEUTFEM_INJECT_FAIL)", T->rank, global_it); TRACE_COMM("rank %d INJECT at %ld", T->rank, global_it); }
            else if (!T->poisoned) ra = team_schur_apply(T, g, ps, qs, true, T->d_cg, &acnt);
            if (lean) T->slabs[0]->lean = no_lean;
            if (tlean) for (auto *S : T->slabs) { S->lean_z1 = no_lean; S->cg1 = Cg1{ nullptr, nullptr, 0 }; }
            if (bad(ra)) break;
            if (T->poisoned && any_if) {
                // the interface exchanges this apply still owes its neighbours: one per apply + one per separator sweep (team_endpoint_phase)
                // with the stream dependencies of team_endpoint_phase: the exchange waits for the solver's stream (ev_z1), the reductions
                // wait for the exchange (ev_xchg).  Without the first one the comm stream runs iterations ahead, and a transport whose
                // operations of one process share a progress thread (the stand-in does; RCCL's proxy has the same shape) deadlocks: this
                // rank blocks in the NEXT iteration's receive while its peers still wait for it in THIS iteration's all-reduce.
                if (T->xchg_in_apply == 0) { (void)hipEventRecord(T->ev_z1, T->stream); (void)hipStreamWaitEvent(T->comm_stream, T->ev_z1, 0); }
                for (int k = T->xchg_in_apply; k < 1 + T->sep_sweeps; ++k) (void)exchange_planes(T, k == 0 ? 0 : 2, 0, T->comm_stream);
                (void)hipEventRecord(T->ev_xchg, T->comm_stream); (void)hipStreamWaitEvent(T->stream, T->ev_xchg, 0);
                for (int i = 0; i < ns; ++i) acnt[i] = 0;
            }
            if (sr) {                                             // the one reduction of this iteration; its consumer is the next endpoint pass
                if (bad(team_reduce_sr(T, acnt))) break;
                continue;
            }
            if (vred) {
                // the partial vectors themselves cross the ranks; their consumers sum them (CgLean with a count and a flag slot)
                auto allred = [&](double *send, double *recv, int cnt) -> int { NCCLCHK(g_rccl.AllReduce(send, recv, (size_t)cnt + 1, NCCL_DOUBLE, NCCL_SUM, T->comm, T->stream)); return NF_OK; };
                if (!T->poisoned && acnt[0] != T->vec_cnt_pq) { if (bad(fail(NF_ERR_STATE, "vector reduce: the accumulation pass left %d partials, %d were agreed", acnt[0], T->vec_cnt_pq))) break; }
                if (bad(allred(send_pq, vec_pq, T->vec_cnt_pq))) break;
                if (!T->poisoned)
                    hipLaunchKernelGGL(k_cg_rupdate, dim3(gcnt[0]), dim3(256), 0, T->stream, T->slabs[0]->d_r, T->slabs[0]->d_q, T->slabs[0]->nphi, T->d_cg, send_rr,
                                       CgLean{ T->d_cg, vec_pq, T->vec_cnt_pq, index & 1, 0, T->vec_cnt_pq });
                if (bad(allred(send_rr, vec_rr, T->vec_cnt_rr))) break;
                continue;
            }
            if (tlean) {
                if (bad(team_reduce(T, acnt, T->d_red))) break;   // d_red[0] = p.q, d_red[1] = error flags
                if (!T->poisoned)
                    for (int i = 0; i < ns; ++i) {
                        nf_solver *S = T->slabs[i];
                        hipLaunchKernelGGL(k_cg_rupdate, dim3(gcnt[i]), dim3(256), 0, T->stream, S->d_r, S->d_q, S->nphi, T->d_cg, T->d_partials + i * T->slab_cap,
                                           CgLean{ T->d_cg, T->d_red, -1, index & 1, 0 });
                    }
                if (bad(team_reduce(T, gcnt, T->d_red + 2))) break;   // d_red[2] = |r|^2, d_red[3] = error flags
                continue;
            }
            if (!lean && bad(team_finalize(T, FIN_PAP, acnt, 1, T->d_out, 0.0, 0))) break;
            if (!T->poisoned)
                for (int i = 0; i < ns; ++i) {
                    nf_solver *S = T->slabs[i];
                    if (lean) hipLaunchKernelGGL(k_cg_rupdate, dim3(gru), dim3(256), 0, T->stream, S->d_r, S->d_q, S->nphi, T->d_cg, row1,
                                                 CgLean{ T->d_cg, T->d_partials, acnt[0], index & 1, 0 });
                    else if (fused) hipLaunchKernelGGL(k_cg_rupdate, dim3(gcnt[i]), dim3(256), 0, T->stream, S->d_r, S->d_q, S->nphi, T->d_cg, T->d_partials + i * T->slab_cap, no_lean);
                    else hipLaunchKernelGGL(k_cg_update, dim3(gcnt[i]), dim3(256), 0, T->stream, x[i], S->d_r, S->d_p, S->d_q, S->nphi, T->d_cg, T->d_partials + i * T->slab_cap);
                }
            if (lean) continue;
            if (bad(team_finalize(T, FIN_RR, gcnt, 1, T->d_out, 0.0, 0))) break;
            if (fused || T->poisoned) continue;
            for (int i = 0; i < ns; ++i) {
                nf_solver *S = T->slabs[i];
                hipLaunchKernelGGL(k_cg_pupdate, dim3(gcnt[i]), dim3(256), 0, T->stream, S->d_p, S->d_r, S->nphi, T->d_cg);
            }
        }
        if (rc != NF_OK) break;
        launched += nb;
        // lean: the stop tests of the batch's last iteration have not been evaluated yet (the next x pass would do it)
        if (lean && pub_ready(T)) {                               // the kernel that evaluates the stop tests hands the scalars to the host itself
            const unsigned long long seq = ++T->pub_seq;
            hipLaunchKernelGGL(k_cg_lean_rr, dim3(1), dim3(256), 0, T->stream, CgLean{ T->d_cg, row1, gru, launched & 1, 0 }, T->d_pub, seq);
            { const int rw = pub_wait(T, seq, &sc, nullptr, 0); if (rw != NF_OK) { rc = rw; break; } }
        } else {
            if (lean) hipLaunchKernelGGL(k_cg_lean_rr, dim3(1), dim3(256), 0, T->stream, CgLean{ T->d_cg, row1, gru, launched & 1, 0 }, (HostPub *)nullptr, 0ULL);
            if (sr) hipLaunchKernelGGL(k_cg1_logic, dim3(1), dim3(64), 0, T->stream, Cg1{ T->d_red, T->d_cg, launched }, (HostPub *)nullptr, 0ULL);
            else if (tlean) hipLaunchKernelGGL(k_cg_lean_rr, dim3(1), dim3(256), 0, T->stream, vred ? CgLean{ T->d_cg, vec_rr, T->vec_cnt_rr, launched & 1, 0, T->vec_cnt_rr }
                                                                                                 : CgLean{ T->d_cg, T->d_red + 2, -1, launched & 1, 0 }, (HostPub *)nullptr, 0ULL);
            { const int rw = readback(T, T->d_cg, &sc, nullptr, nullptr, 0); if (rw != NF_OK) { rc = rw; break; } }
        }
        if (sc.done) break;
        // after the first (predicted) batch grow geometrically: an iteration launched past convergence is five early-exit
        // kernels (~10 us), a host check is a D2H copy + stream drain (~50 us)
        batch = T->cg_batch > 0 ? T->cg_batch : (launched < 8 ? 1 : grow);
        if (launched >= 8) grow = std::min(64, 2 * grow);
    }
    for (int i = 0; i < ns; ++i) T->slabs[i]->fuse = CgFuse{ nullptr, nullptr, nullptr };
    NFCHK(rc);
    if (T->poisoned) {                                            // the flag has stopped every rank (or maxit did): this rank reports what hit it
        const int code = T->poison_rc; char msg[256]; snprintf(msg, sizeof msg, "%s", T->poison_msg);
        T->poisoned = false;
        (void)hipMemsetAsync(T->d_errsrc, 0, sizeof(double), T->stream);
        return fail(code, "%s", msg);
    }
    if (sc.err == 2) return fail(NF_ERR_REMOTE, "another rank of the team reported an error during the CG solve of group %d; every rank stopped at iteration %d", g, sc.its);
    if (launched == 0) {
        HIPCHK(hipMemcpyAsync(&sc, T->d_cg, sizeof sc, hipMemcpyDeviceToHost, T->stream));
        HIPCHK(hipStreamSynchronize(T->stream));
    }
    if (fused)
        for (int i = 0; i < ns; ++i) {
            const double *plast = T->slabs[i]->d_p;
            if (f3.ok && sc.its >= 2 && ((sc.its - 1) & 1)) plast = T->slabs[i]->d_p2;   // the direction of the last iteration (see the buffer pair above)
            hipLaunchKernelGGL(k_cg_flush, dim3(gcnt[i]), dim3(256), 0, T->stream, x[i], plast, T->slabs[i]->nphi, T->d_cg);
        }
    HIPCHK(hipGetLastError());
    if (T->profile) prof_collect(T);
    if (!std::isfinite(sc.rr)) return fail(NF_ERR_NUMERIC, "CG produced a non-finite residual (group %d)", g);
    T->last_its[g] = sc.its;
    if (its_out) *its_out = sc.its;
    if (res_out) *res_out = sc.rhs_norm > 0 ? std::sqrt(sc.rr) / sc.rhs_norm : 0.0;
    return NF_OK;
}

int nf_solve_group(nf_handle S, int g, const double *rhs_dev, double *phi_dev, double tol, int maxit, int *its, double *res)
{
    if (!S || !rhs_dev || !phi_dev || g < 0 || g >= S->ng) return fail(NF_ERR_ARG, "nf_solve_group: bad arguments");
    if (!S->built) return fail(NF_ERR_STATE, "nf_solve_group: call nf_build first");
    if (!team_is_single(S->team)) return fail(NF_ERR_STATE, "nf_solve_group works on an undivided mesh");
    HIPCHK(hipSetDevice(S->device));
    return cg_solve(S->team, g, { rhs_dev }, { phi_dev }, tol, maxit, its, res);
}

// ---- diagonal cache ----------------------------------------------------------------------------
// Whole team at once: on slabs the interface faces need the neighbour's edge-cell a2 (one plane per group, exchanged here).
int nf_build_diagonal_cache(nf_handle S)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    nf_team *T = S->team;
    for (auto *X : T->slabs) if (!X->built) return fail(NF_ERR_STATE, "nf_build_diagonal_cache: call nf_build first");
    if (S->k != 0 || S->m != 0) return NF_OK;                     // "non applicable (ordre > 0)", src/NeutFEM.cpp:484-487
    bool valid = true, any_if = false;
    for (auto *X : T->slabs) { valid &= X->diag_valid; any_if |= X->if_lo || X->if_hi; }
    if (valid) return NF_OK;
    HIPCHK(hipSetDevice(S->device));
    {
        int lrc = NF_OK;
        for (auto *X : T->slabs) if (lrc == NF_OK && !X->d_Sinv) lrc = dalloc(&X->d_Sinv, (size_t)X->N * X->ng);
        NFCHK(team_verdict(T, lrc, "the edge-plane exchange of nf_build_diagonal_cache"));
    }
    for (int g = 0; g < S->ng; ++g) {
        if (any_if) {
            for (auto *X : T->slabs) {
                const long nl = X->nlines[2];
                hipLaunchKernelGGL(k_edge_a2, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, T->stream, make_geom(X), X->d_D + g * X->N, X->d_clo, X->d_chi, nl);
            }
            NFCHK(exchange_planes(T, 0, 0, T->stream));
        }
        for (auto *X : T->slabs) {
            const long N = X->N;
            hipLaunchKernelGGL(k_diag_cache, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, T->stream, make_geom(X), X->d_D + g * N, X->d_Cd + g * N,
                               X->d_Sinv + g * N, N, X->if_lo ? X->d_rlo : (const double *)nullptr, X->if_hi ? X->d_rhi : (const double *)nullptr);
        }
        NFCHK(team_stream_wait(T, T->stream));                  // the exchange buffers are reused by the next group
    }
    HIPCHK(hipGetLastError());
    for (auto *X : T->slabs) X->diag_valid = true;
    return NF_OK;
}
int nf_get_diagonal_cache(nf_handle S, int g, double *sinv_host)
{
    if (!S || g < 0 || g >= S->ng || !sinv_host) return fail(NF_ERR_ARG, "nf_get_diagonal_cache: bad arguments");
    NFCHK(nf_build_diagonal_cache(S));
    if (!S->diag_valid) return fail(NF_ERR_UNSUPPORTED, "the diagonal cache exists for RT0-P0 only");
    HIPCHK(hipMemcpy(sinv_host, S->d_Sinv + g * S->N, S->N * sizeof(double), hipMemcpyDeviceToHost));
    return NF_OK;
}

// ---- state -------------------------------------------------------------------------------------
// host layout [g][e*nloc + p] (reference, src/FEM.cpp:321-334) <-> device layout [g][p][e]
static int phi_transfer(nf_solver *S, double *host, bool to_device, double *dev = nullptr)
{
    if (!dev) dev = S->d_phi;
    HIPCHK(hipSetDevice(S->device));
    hipStream_t st = S->team->stream;
    HIPCHK(hipStreamSynchronize(st));
    const size_t NN = (size_t)S->nphi * S->ng;
    if (S->nloc == 1) {
        if (to_device) HIPCHK(hipMemcpy(dev, host, NN * sizeof(double), hipMemcpyHostToDevice));
        else HIPCHK(hipMemcpy(host, dev, NN * sizeof(double), hipMemcpyDeviceToHost));
        return NF_OK;
    }
    DevTmp<double> tmp_; NFCHK(dalloc(&tmp_.p, NN)); double *tmp = tmp_.p;
    if (to_device) HIPCHK(hipMemcpy(tmp, host, NN * sizeof(double), hipMemcpyHostToDevice));
    for (int g = 0; g < S->ng; ++g) {
        const double *src = (to_device ? tmp : dev) + (size_t)g * S->nphi;
        double *dst = (to_device ? dev : tmp) + (size_t)g * S->nphi;
        hipLaunchKernelGGL(k_transpose_dofs, dim3(grid_for(S->nphi)), dim3(256), 0, st, src, dst, S->N, S->nloc, to_device ? 1 : 0);
    }
    HIPCHK(hipStreamSynchronize(st));
    if (!to_device) HIPCHK(hipMemcpy(host, tmp, NN * sizeof(double), hipMemcpyDeviceToHost));
    return NF_OK;
}
int nf_set_phi(nf_handle S, const double *phi)
{
    if (!S || !phi) return fail(NF_ERR_ARG, "nf_set_phi: bad arguments");
    return phi_transfer(S, const_cast<double *>(phi), true);
}
int nf_get_phi(nf_handle S, double *phi)
{
    if (!S || !phi) return fail(NF_ERR_ARG, "nf_get_phi: bad arguments");
    return phi_transfer(S, phi, false);
}
int nf_reset_flux(nf_handle S)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(S->device));
    std::vector<double> ones((size_t)S->nphi * S->ng, 1.0);     // Sol_Phi_ = 1 on every DOF, src/NeutFEM.cpp:347-354
    HIPCHK(hipMemcpy(S->d_phi, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
    S->team->has_valid_keff = 0; S->raw_valid = false;
    return NF_OK;
}
int nf_set_warm_state(nf_handle S, int v, double k) { if (!S) return fail(NF_ERR_ARG, "null handle"); S->team->has_valid_keff = v; S->team->last_keff = k; return NF_OK; }
int nf_get_warm_state(nf_handle S, int *v, double *k) { if (!S) return fail(NF_ERR_ARG, "null handle"); if (v) *v = S->team->has_valid_keff; if (k) *k = S->team->last_keff; return NF_OK; }

// z currents of every local slab from the raw group fluxes of the last solve: endpoint pass, exchange, separators, then the
// chain solve in "emit" mode (k_schur_s mode 3).  Cached until the next solve.
static int team_reconstruct_Jz(nf_team *T)
{
    bool all = true; for (auto *S : T->slabs) all &= S->jz_valid;
    if (all) return NF_OK;
    NFCHK(team_prepare(T));
    const int ns = (int)T->slabs.size(), ng = T->slabs[0]->ng;
    {
        int lrc = NF_OK;
        for (auto *S : T->slabs) {
            if (lrc == NF_OK && !S->d_Jz) lrc = dalloc(&S->d_Jz, (size_t)S->nJz * ng);
            if (lrc == NF_OK && S->k > 0 && !S->d_Jzb) { int ni = S->k; for (int t = 1; t < S->dim; ++t) ni *= S->k + 1; lrc = dalloc(&S->d_Jzb, (size_t)S->N * ni * ng); }
        }
        NFCHK(team_verdict(T, lrc, "the partition-method solve of nf_get_J"));
    }
    for (auto *S : T->slabs) {
        HIPCHK(hipMemsetAsync(S->d_Jz, 0, (size_t)S->nJz * ng * sizeof(double), T->stream));         // RT modes that see no phi moment stay 0
        if (S->k > 0) {
            int ni = S->k; for (int t = 1; t < S->dim; ++t) ni *= S->k + 1;
            HIPCHK(hipMemsetAsync(S->d_Jzb, 0, (size_t)S->N * ni * ng * sizeof(double), T->stream));
        }
    }
    std::vector<const double *> xs(ns); std::vector<double *> ys(ns);
    for (int g = 0; g < ng; ++g) {
        for (int i = 0; i < ns; ++i) { xs[i] = T->slabs[i]->d_raw + (size_t)g * T->slabs[i]->nphi; ys[i] = T->slabs[i]->d_q; }
        NFCHK(team_endpoint_phase(T, g, xs, ys, nullptr, nullptr, true));
        HIPCHK(hipStreamWaitEvent(T->stream, T->ev_xchg, 0));
        for (int i = 0; i < ns; ++i) {
            nf_solver *S = T->slabs[i];
            NFCHK(launch_s(S, 2, g, mode_args(S, g, 2, 0, xs[i], ys[i]), make_geom(S), 0, nullptr, nullptr, nullptr, 3));
        }
        NFCHK(team_stream_wait(T, T->stream));                  // the exchange buffers are reused by the next group
    }
    HIPCHK(hipGetLastError());
    for (auto *S : T->slabs) S->jz_valid = true;
    return NF_OK;
}

// z currents of every local slab after a diagonal-Schur solve (J_f = +(B^T phi)_f / A_ff, src/NeutFEM.cpp:620-633): per group
// one plane of edge-cell a2 and one of edge-cell phi per interface go to the neighbour; collective like team_reconstruct_Jz.
static int team_reconstruct_Jz_diag(nf_team *T)
{
    bool all = true; for (auto *S : T->slabs) all &= S->jz_valid;
    if (all) return NF_OK;
    NFCHK(team_prepare(T));
    const int ng = T->slabs[0]->ng;
    hipStream_t st = T->stream;
    {
        int lrc = NF_OK;
        for (auto *S : T->slabs) if (lrc == NF_OK && !S->d_Jz) lrc = dalloc(&S->d_Jz, (size_t)S->nJz * ng);
        NFCHK(team_verdict(T, lrc, "the edge-plane exchanges of nf_get_J (diagonal path)"));
    }
    for (int g = 0; g < ng; ++g) {
        // (1) a2 of the edge cells -> d_rlo / d_rhi ; kept in d_ctlo / d_cthi while the planes are reused for phi
        for (auto *S : T->slabs) {
            const long nl = S->nlines[2];
            hipLaunchKernelGGL(k_edge_a2, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, make_geom(S), S->d_D + g * S->N, S->d_clo, S->d_chi, nl);
        }
        NFCHK(exchange_planes(T, 0, 0, st));
        for (auto *S : T->slabs) {
            const size_t b = (size_t)S->nlines[2] * sizeof(double);
            HIPCHK(hipMemcpyAsync(S->d_ctlo, S->d_rlo, b, hipMemcpyDeviceToDevice, st)); HIPCHK(hipMemcpyAsync(S->d_cthi, S->d_rhi, b, hipMemcpyDeviceToDevice, st));
        }
        // (2) phi of the edge cells (first and last plane of the raw group flux)
        for (auto *S : T->slabs) {
            const size_t b = (size_t)S->nlines[2] * sizeof(double);
            const double *raw = S->d_raw + (size_t)g * S->nphi;
            HIPCHK(hipMemcpyAsync(S->d_clo, raw, b, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(S->d_chi, raw + (size_t)(S->nz - 1) * S->nlines[2], b, hipMemcpyDeviceToDevice, st));
        }
        NFCHK(exchange_planes(T, 0, 0, st));
        for (auto *S : T->slabs) {
            const long nl = S->nlines[2];
            hipLaunchKernelGGL(k_diag_Jz_slab, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, st, make_geom(S), S->d_D + g * S->N, S->d_raw + (size_t)g * S->nphi,
                               S->d_Jz + (size_t)g * S->nJz, nl, S->if_lo, S->if_hi, (const double *)S->d_rlo, (const double *)S->d_ctlo,
                               (const double *)S->d_rhi, (const double *)S->d_cthi);
        }
        NFCHK(team_stream_wait(T, st));                         // the exchange buffers are reused by the next group
    }
    HIPCHK(hipGetLastError());
    for (auto *S : T->slabs) S->jz_valid = true;
    return NF_OK;
}

int nf_get_J(nf_handle S, double *J_host)
{
    if (!S || !J_host) return fail(NF_ERR_ARG, "nf_get_J: bad arguments");
    if (S->team->dead) return fail(NF_ERR_COMM, "this team lost a collective (NF_ERR_COMM) and is unusable: destroy the handles and end the process with a non-zero code (never re-exec)");
    HIPCHK(hipSetDevice(S->device));
    hipStream_t st = S->team->stream;
    const long N = S->N, nJ = S->nJ;
    if (!S->raw_valid) { memset(J_host, 0, sizeof(double) * nJ * S->ng); return NF_OK; }   // Sol_J_ = 0 before any solve
    const bool slab = S->if_lo || S->if_hi;
    if (slab) {
        // z currents cross slabs: one partition-method solve per group on the raw group fluxes, for the whole team (collective:
        // every rank calls nf_get_J on its slabs in the same order; the first call after a solve does the work for all local slabs)
        if (S->raw_is_diag) NFCHK(team_reconstruct_Jz_diag(S->team));       // RT0-P0 by construction (the diagonal path exists for that order only)
        else NFCHK(team_reconstruct_Jz(S->team));
    }
    DevTmp<double> dJ_; NFCHK(dalloc(&dJ_.p, (size_t)nJ)); double *dJ = dJ_.p;
    Geom G = make_geom(S);
    int nfa = 1, ni = S->k; for (int t = 1; t < S->dim; ++t) { nfa *= S->k + 1; ni *= S->k + 1; }
    const long nJface = S->nJx + S->nJy + S->nJz;
    const long foff[3] = { 0, S->nJx, S->nJx + S->nJy };
    for (int g = 0; g < S->ng; ++g) {
        HIPCHK(hipMemsetAsync(dJ, 0, (size_t)nJ * sizeof(double), st));                     // modes that see no phi moment stay 0
        for (int d = 0; d < (slab ? 2 : S->dim); ++d)
            for (int mode = 0; mode < n_modes(S); ++mode) {
                ModeArgs ma = mode_args(S, g, d, mode, S->d_raw + (size_t)g * S->nphi, S->d_raw + (size_t)g * S->nphi);
                // transverse mode index in the RT numbering: a = sum a_t (k+1)^t over the transverse axes (FEM.cpp:364-374)
                int amode = 0, q = mode, mul = 1;
                for (int t = 0; t < S->dim; ++t) { if (t == d) continue; amode += (q % S->n1) * mul; q /= S->n1; mul *= S->k + 1; }
                hipLaunchKernelGGL(k_flux_to_J, dim3((unsigned)((S->nlines[d] + 63) / 64)), dim3(64), 0, st, G, ma, S->nb, amode, nfa, ni,
                                   S->d_D + g * N, S->d_L[d] + g * N, S->d_DR[d] + g * N, S->d_D0[d] + g * S->nlines[d],
                                   dJ + foff[d], dJ + nJface + (long)d * N * ni, S->nlines[d], S->raw_is_diag ? 1 : 0);
            }
        if (slab) {
            HIPCHK(hipMemcpyAsync(dJ + foff[2], S->d_Jz + (size_t)g * S->nJz, S->nJz * sizeof(double), hipMemcpyDeviceToDevice, st));
            if (ni > 0 && !S->raw_is_diag && S->d_Jzb)
                HIPCHK(hipMemcpyAsync(dJ + nJface + 2L * N * ni, S->d_Jzb + (size_t)g * N * ni, (size_t)N * ni * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(hipMemcpyAsync(J_host + (size_t)g * nJ, dJ, nJ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    HIPCHK(hipGetLastError());
    return NF_OK;
}

// ChebyshevAccel(15, 0.98) coefficient tables a_n, b_n (src/solvers.cpp:664-700), shared by the direct and adjoint iterations
static const int CHEB_NMAX = 15;
static const double CHEB_SIGMA = 0.98;
static void cheb_tables(double *ca, double *cbv)
{
    const double sigma = CHEB_SIGMA, Gm = std::acosh(2. / sigma - 1.);
    ca[0] = cbv[0] = 0.; ca[1] = 2. / (2. - sigma); cbv[1] = 0.;
    for (int i = 2; i < CHEB_NMAX; ++i) { ca[i] = std::cosh((i - 1) * Gm) / std::cosh(i * Gm); cbv[i] = std::cosh((i - 2) * Gm) / std::cosh(i * Gm); }
}

// ---- explicit-S branch (src/solvers.cpp:114-124, 259-509) ----------------------------------------
// FormSchurComplement + PrepareSolver once per BuildMatrices: column j of S = S e_j through the matrix-free apply (the
// reference solves A x = B^T e_j and multiplies by B: the same numbers up to its 1e-14 drop threshold), then S^-1 by n
// Gauss-Jordan steps on the whole chip.  Undivided meshes of at most direct_max_dofs unknowns per group.
static int dense_prepare(nf_team *T)
{
    nf_solver *S = T->slabs[0];
    if (S->dense_valid) return NF_OK;
    const int n = (int)S->nphi, ng = S->ng; const size_t nn = (size_t)n * n;
    hipStream_t st = T->stream;
    NFCHK(dalloc(&S->d_Sdense, nn * ng));
    DevTmp<double> work; NFCHK(dalloc(&work.p, nn));
    for (int g = 0; g < ng; ++g) {
        double *M = S->d_Sdense + (size_t)g * nn;
        for (int j = 0; j < n; ++j) {
            hipLaunchKernelGGL(k_unit_vector, dim3(grid_for(n)), dim3(256), 0, st, S->d_p, (long)n, (long)j);
            NFCHK(team_schur_apply(T, g, { S->d_p }, { M + (size_t)j * n }, false, nullptr, nullptr));
        }
        double *a = M, *b = work.p;
        const unsigned gr = (unsigned)((nn + 255) / 256);
        for (int k = 0; k < n; ++k) { hipLaunchKernelGGL(k_gj_step, dim3(gr), dim3(256), 0, st, (const double *)a, b, n, k); std::swap(a, b); }
        if (a != M) HIPCHK(hipMemcpyAsync(M, a, nn * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    HIPCHK(hipGetLastError());
    S->dense_valid = true;
    return NF_OK;
}
static void dense_solve(nf_team *T, int g, const double *rhs, double *sol)
{
    nf_solver *S = T->slabs[0];
    const int n = (int)S->nphi;
    hipLaunchKernelGGL(k_dense_matvec, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, T->stream, (const double *)(S->d_Sdense + (size_t)g * n * n), rhs, sol, n);
}

// ---- SolveCoarse (src/NeutFEM.cpp:2380-2611) ---------------------------------------------------
static int solve_keff_impl(nf_team *T, const nf_keff_opts *o, double *keff, int *n_outer);

// coarse twin of one slab (src/NeutFEM.cpp:2409-2556): every (rx, ry, rz) block of cells becomes one RT0-P0 cell, cross sections
// are arithmetic volume-weighted block means, boundary conditions and interface flags carry over; returned built.
static int coarsen_slab(nf_solver *S, int rx, int ry, int rz, nf_handle *out)
{
    const int dim = S->dim, ng = S->ng;
    hipStream_t st = S->team->stream;
    const int nxc = S->nx / rx, nyc = S->ny / ry, nzc = S->nz / rz;
    std::vector<double> xc(nxc + 1), yc(dim >= 2 ? nyc + 1 : 1), zc(dim >= 3 ? nzc + 1 : 1);
    for (int a = 0; a <= nxc; ++a) xc[a] = S->xb[a * rx];
    if (dim >= 2) for (int j = 0; j <= nyc; ++j) yc[j] = S->yb[j * ry]; else yc[0] = 0.0;
    if (dim >= 3) for (int kk = 0; kk <= nzc; ++kk) zc[kk] = S->zb[kk * rz]; else zc[0] = 0.0;
    nf_handle C = nullptr;
    NFCHK(create_impl(0, 0, ng, nxc + 1, xc.data(), (int)yc.size(), yc.data(), (int)zc.size(), zc.data(), S->if_lo, S->if_hi, S->device, &C));
    *out = C;
    for (int a = 0; a < 8; ++a) if (S->bc_set[a]) nf_set_bc(C, a, S->bc_type[a]);
    const long Nc = C->N;
    auto coarsen = [&](const double *fine, double **coarse, int nfields) -> int {
        NFCHK(dalloc(coarse, (size_t)Nc * nfields));
        hipLaunchKernelGGL(k_coarsen, dim3((unsigned)((Nc + 127) / 128)), dim3(128), 0, st, fine, *coarse, S->d_xb, S->d_yb, S->d_zb,
                           dim, S->nx, S->ny, S->nz, rx, ry, rz, nfields);
        return NF_OK;
    };
    NFCHK(coarsen(S->d_D, &C->d_D, ng)); NFCHK(coarsen(S->d_SigR, &C->d_SigR, ng));
    NFCHK(coarsen(S->d_NSF, &C->d_NSF, ng)); NFCHK(coarsen(S->d_Chi, &C->d_Chi, ng));
    for (int b = 0; b < ng * ng; ++b)
        if (S->d_SigS[b]) NFCHK(coarsen(S->d_SigS[b], &C->d_SigS[b], 1));        // mean of an all-zero block is zero
    HIPCHK(hipStreamSynchronize(st));
    C->xs_uploaded = true;
    return nf_build(C);
}

// Builds + solves the coarse problem on the device (every local slab coarsens its own planes; the coarse slabs form a
// team that shares the fine team's communicator); the prolonged flux of slab i is written to dsts[i] (ng*nphi).
static int coarse_init(nf_team *T, const nf_keff_opts *o, double *k_coarse, const std::vector<double *> &dsts, bool *done)
{
    *done = false;
    const int ns = (int)T->slabs.size();
    nf_solver *S0 = T->slabs[0];
    const int dim = S0->dim, ng = S0->ng;
    const int nf = o->n_coarse_factors;
    int rx = nf > 0 ? std::max(o->coarse_factors[0], 1) : 1;
    int ry = (nf > 1 && dim >= 2) ? std::max(o->coarse_factors[1], 1) : 1;
    int rz = (nf > 2 && dim >= 3) ? std::max(o->coarse_factors[2], 1) : 1;
    // :2402-2407 -> (1.0, Sol_Phi_).  On a decomposed mesh every slab must be divisible (a global decision is needed,
    // so indivisible local slabs are an error rather than a silent fallback on some ranks only).
    double bad = 0.0; const nf_solver *Sbad = nullptr;
    for (auto *S : T->slabs)
        if (S->nx % rx || S->ny % ry || S->nz % rz) { bad = 1.0; Sbad = S; }
    if (T->rccl_reduce && T->nproc > 1) {                         // one verdict for all ranks (a lone refusal would block the others)
        HIPCHK(hipMemcpyAsync(T->d_red, &bad, sizeof(double), hipMemcpyHostToDevice, T->stream));
        NCCLCHK(g_rccl.AllReduce(T->d_red, T->d_red, 1, NCCL_DOUBLE, NCCL_MAX, T->comm, T->stream));
        HIPCHK(hipMemcpyAsync(&bad, T->d_red, sizeof(double), hipMemcpyDeviceToHost, T->stream));
        NFCHK(team_stream_wait(T, T->stream));                      // a collective sits on the stream: bounded wait (NEUTFEM_COMM_TIMEOUT_S)
    }
    if (bad != 0.0) {
        if (ns == 1 && T->nproc == 1) return NF_OK;
        if (Sbad) return fail(NF_ERR_ARG, "coarse factors (%d,%d,%d) do not divide slab %d x %d x %d", rx, ry, rz, Sbad->nx, Sbad->ny, Sbad->nz);
        return fail(NF_ERR_ARG, "coarse factors (%d,%d,%d) do not divide a slab on another rank", rx, ry, rz);
    }
    std::vector<nf_handle> C(ns, nullptr);
    int rc = NF_OK;
    const bool reuse = (int)T->cc.size() == ns && T->cc_f[0] == rx && T->cc_f[1] == ry && T->cc_f[2] == rz;
    if (reuse) {
        for (int i = 0; i < ns && rc == NF_OK; ++i) { C[i] = T->cc[i]; rc = nf_reset_flux(C[i]); }   // a fresh coarse solver starts from phi = 1, k = 1 (:2458)
    } else {
        coarse_cache_drop(T);
        for (int i = 0; i < ns && rc == NF_OK; ++i) rc = coarsen_slab(T->slabs[i], rx, ry, rz, &C[i]);
        if (rc == NF_OK && ns > 1) rc = nf_link_slabs(C.data(), ns);
        if (rc == NF_OK) { T->cc = C; T->cc_f[0] = rx; T->cc_f[1] = ry; T->cc_f[2] = rz; }
        else { std::string keep = g_err; for (int i = ns - 1; i >= 0; --i) if (C[i]) nf_destroy(C[i]); g_err = keep; return rc; }
    }
    nf_team *CT = rc == NF_OK ? C[0]->team : nullptr;
    if (CT) {
        CT->comm = T->comm; CT->nproc = T->nproc; CT->rank = T->rank; CT->rccl_reduce = T->rccl_reduce; CT->linked_ready = false;
        // tuning options (nf_set_option) apply to the coarse solve as well
        CT->opt_cg1 = T->opt_cg1; CT->cg1_max_cells = T->cg1_max_cells; CT->opt_endpoint_w = T->opt_endpoint_w; CT->comm_x = T->comm_x; CT->opt_fuse = T->opt_fuse; CT->opt_lean = T->opt_lean; CT->opt_sepfold = T->opt_sepfold; CT->opt_lean_grid = T->opt_lean_grid; CT->lean_max_cells = T->lean_max_cells;
        CT->opt_fuse3 = T->opt_fuse3; CT->fuse3_max_cells = T->fuse3_max_cells; CT->opt_cgx = T->opt_cgx; CT->opt_keffx = T->opt_keffx; CT->xcd_min_cells = T->xcd_min_cells; CT->xcd_max_cells = T->xcd_max_cells; CT->xcd_id = T->xcd_id; CT->xcd_groups = T->xcd_groups; CT->opt_resident = T->opt_resident; CT->opt_resident_lds = T->opt_resident_lds; CT->opt_resident_serial = T->opt_resident_serial; CT->opt_resident_two_sided = T->opt_resident_two_sided; CT->resident_max_dofs = T->resident_max_dofs; CT->resident_serial_max_dofs = T->resident_serial_max_dofs;
        CT->cg_batch = T->cg_batch; CT->opt_outer_dev = T->opt_outer_dev; CT->direct_max_dofs = T->direct_max_dofs;
    }
    double kc = 1.0; int nout = 0;
    if (rc == NF_OK) {
        nf_keff_opts co = *o;                                    // :2460-2467
        co.tol_keff = o->tol_keff * 10.0; co.tol_flux = o->tol_flux * 10.0; co.max_outer = o->max_outer / 2;
        co.use_coarse_init = 0; co.n_coarse_factors = 0; co.use_diagonal_solver = 0; co.solver_type_pushed = 1; co.profile = 0; co.use_cmfd = 0;
        rc = solve_keff_impl(CT, &co, &kc, &nout);
    }
    if (rc == NF_OK) {
        T->coarse_outer = nout;
        for (int i = 0; i < ns; ++i) {
            nf_solver *S = T->slabs[i];
            (void)hipMemsetAsync(dsts[i], 0, (size_t)S->nphi * ng * sizeof(double), CT->stream);   // higher moments 0 (:2585-2606)
            hipLaunchKernelGGL(k_prolong, dim3((unsigned)((S->N + 255) / 256)), dim3(256), 0, CT->stream, C[i]->d_phi, dsts[i], S->nx, S->ny, S->nz, rx, ry, rz, ng, S->nphi);
        }
        if (hipStreamSynchronize(CT->stream) != hipSuccess) rc = fail(NF_ERR_HIP, "prolong failed");
    }
    if (CT && CT->dead) { T->dead = true; T->comm = nullptr; T->comm_x = nullptr; }   // a collective of the coarse solve timed out on the BORROWED communicator: the fine team lost it too
    if (CT) { CT->comm = nullptr; CT->comm_x = nullptr; CT->nproc = 1; CT->rccl_reduce = false; }   // borrowed for the solve only: never destroyed with the coarse team
    if (rc != NF_OK) { coarse_cache_drop(T); return rc; }
    *k_coarse = kc; *done = true;
    return NF_OK;
}

int nf_solve_coarse(nf_handle S, const nf_keff_opts *o, double *k_coarse, double *phi_host)
{
    if (!S || !o || !k_coarse || !phi_host) return fail(NF_ERR_ARG, "nf_solve_coarse: bad arguments");
    if (!S->built) return fail(NF_ERR_STATE, "nf_solve_coarse: call nf_build first");
    if (!team_is_single(S->team)) return fail(NF_ERR_UNSUPPORTED, "coarse-mesh initialisation is not available on a slab-decomposed mesh");
    HIPCHK(hipSetDevice(S->device));
    bool done = false; double kc = 1.0;
    if (o->n_coarse_factors > 0) NFCHK(coarse_init(S->team, o, &kc, { S->d_raw }, &done));
    S->raw_valid = false;
    if (done) std::swap(S->d_raw, S->d_phi);                     // reuse the layout-converting download
    int rc = phi_transfer(S, phi_host, false);
    if (done) std::swap(S->d_raw, S->d_phi);
    NFCHK(rc);
    *k_coarse = done ? kc : 1.0;
    return NF_OK;
}

// ---- CMFD (src/NeutFEM.cpp:662-1017) -------------------------------------------------------------
// InitializeCMFD: D~ for every direction, D^ = 0.  Idempotent until the next nf_build (is_initialized, :663).
static int cmfd_initialize(nf_solver *S)
{
    if (S->cmfd_init) return NF_OK;
    hipStream_t st = S->team->stream;
    const int ng = S->ng; const long N = S->N;
    S->nfc[0] = (long)(S->nx + 1) * S->ny * S->nz; S->nfc[1] = (long)S->nx * (S->ny + 1) * S->nz; S->nfc[2] = (long)S->nx * S->ny * (S->nz + 1);
    Geom G = make_geom(S);
    for (int d = 0; d < 3; ++d) {
        NFCHK(dalloc(&S->d_Dt[d], (size_t)S->nfc[d] * ng)); NFCHK(dalloc(&S->d_Dh[d], (size_t)S->nfc[d] * ng));
        HIPCHK(hipMemsetAsync(S->d_Dt[d], 0, (size_t)S->nfc[d] * ng * sizeof(double), st));
        HIPCHK(hipMemsetAsync(S->d_Dh[d], 0, (size_t)S->nfc[d] * ng * sizeof(double), st));
        if (d >= S->dim) continue;
        for (int g = 0; g < ng; ++g)
            hipLaunchKernelGGL(k_cmfd_dtilde, dim3((unsigned)((S->nfc[d] + 255) / 256)), dim3(256), 0, st, G, d, S->d_D + g * N, S->d_Dt[d] + g * S->nfc[d], S->nfc[d]);
    }
    if (!S->d_cm) NFCHK(dalloc(&S->d_cm, (size_t)N * 6));
    if (!S->d_cmJ) NFCHK(dalloc(&S->d_cmJ, (size_t)S->nJ));
    if (!S->d_cmsc) NFCHK(dalloc(&S->d_cmsc, 1));
    HIPCHK(hipGetLastError());
    S->cmfd_init = true;
    return NF_OK;
}
static int cmfd_initialize_team(nf_team *T);
int nf_initialize_cmfd(nf_handle S)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    if (!S->built) return fail(NF_ERR_STATE, "nf_initialize_cmfd: call nf_build first");
    HIPCHK(hipSetDevice(S->device));
    for (auto *X : S->team->slabs) if (!X->built) return fail(NF_ERR_STATE, "nf_initialize_cmfd: every slab of the team must be built first");
    NFCHK(cmfd_initialize_team(S->team));                        // collective on a multi-rank team (interface D-tilde)
    HIPCHK(hipStreamSynchronize(S->team->stream));
    return NF_OK;
}
// team-wide: every slab of the handle's team applies the same omega in k_cmfd_correct (on a multi-rank run every rank must set the same value)
int nf_set_cmfd_relaxation(nf_handle S, double omega) { if (!S) return fail(NF_ERR_ARG, "null handle"); for (auto *X : S->team->slabs) X->cmfd_relax = omega; return NF_OK; }
int nf_get_cmfd_coefficients(nf_handle S, int g, int dir, double *dtilde_host, double *dhat_host)
{
    if (!S || g < 0 || g >= S->ng || dir < 0 || dir > 2) return fail(NF_ERR_ARG, "nf_get_cmfd_coefficients: bad arguments");
    if (!S->cmfd_init) return fail(NF_ERR_STATE, "CMFD is not initialised");
    HIPCHK(hipSetDevice(S->device));
    HIPCHK(hipStreamSynchronize(S->team->stream));
    if (dtilde_host) HIPCHK(hipMemcpy(dtilde_host, S->d_Dt[dir] + g * S->nfc[dir], S->nfc[dir] * sizeof(double), hipMemcpyDeviceToHost));
    if (dhat_host) HIPCHK(hipMemcpy(dhat_host, S->d_Dh[dir] + g * S->nfc[dir], S->nfc[dir] * sizeof(double), hipMemcpyDeviceToHost));
    return NF_OK;
}

// UpdateDhatCoefficients for every group, then phi_g *= ApplyCMFDCorrection(g, phi_g, total_fiss, keff) (:1750-1761).
// phi = post-sweep group fluxes (d_raw), tf = d_tf of this outer, diag_sign = the J convention of the group solver.
static int cmfd_step(nf_solver *S, double keff, int use_diag)
{
    nf_team *T = S->team; hipStream_t st = T->stream;
    const int ng = S->ng; const long N = S->N, NP = S->nphi;
    Geom G = make_geom(S);
    int nfa = 1, ni = S->k; for (int t = 1; t < S->dim; ++t) { nfa *= S->k + 1; ni *= S->k + 1; }
    const long nJface = S->nJx + S->nJy + S->nJz;
    for (int g = 0; g < ng; ++g) {
        // Sol_J_ x-face mode 0 of group g (the only current D^ reads)
        ModeArgs ma = mode_args(S, g, 0, 0, S->d_raw + (size_t)g * NP, S->d_raw + (size_t)g * NP);
        hipLaunchKernelGGL(k_flux_to_J, dim3((unsigned)((S->nlines[0] + 63) / 64)), dim3(64), 0, st, G, ma, S->nb, 0, nfa, ni,
                           S->d_D + g * N, S->d_L[0] + g * N, S->d_DR[0] + g * N, S->d_D0[0] + g * S->nlines[0],
                           S->d_cmJ, S->d_cmJ + nJface, S->nlines[0], use_diag);
        hipLaunchKernelGGL(k_cmfd_dhat, dim3((unsigned)((S->nfc[0] + 255) / 256)), dim3(256), 0, st, G, S->d_raw + (size_t)g * NP, S->d_cmJ, nfa,
                           S->d_Dt[0] + g * S->nfc[0], S->d_Dh[0] + g * S->nfc[0], S->nfc[0]);
    }
    double *diag = S->d_cm, *x = diag + N, *r = x + N, *pp = r + N, *q = pp + N, *z = q + N;
    const int gN = grid_for(N);
    const long stride = T->partial_stride;
    S->cmfd_last_its = 0;
    for (int g = 0; g < ng; ++g) {
        CmfdFaces F;
        for (int d = 0; d < 3; ++d) { F.Dt[d] = S->d_Dt[d] + g * S->nfc[d]; F.Dh[d] = S->d_Dh[d] + g * S->nfc[d]; }
        hipLaunchKernelGGL(k_cmfd_setup, dim3(gN), dim3(256), 0, st, G, F, S->d_Cd + (size_t)g * NP, S->d_Chi + g * N, S->d_tf, 1.0 / keff,
                           diag, x, r, pp, N, T->d_partials, stride);
        hipLaunchKernelGGL(k_cmfd_logic, dim3(1), dim3(256), 0, st, 0, T->d_partials, gN, stride, S->d_cmsc);
        CmfdScalars hs; hs.done = 0; hs.its = 0;
        for (int base = 0; base < 100 && !hs.done; base += 10) {
            for (int i = 0; i < 10; ++i) {
                hipLaunchKernelGGL(k_cmfd_matvec, dim3(gN), dim3(256), 0, st, G, F, diag, pp, q, N, S->d_cmsc, T->d_partials, (const double *)nullptr, (const double *)nullptr);
                hipLaunchKernelGGL(k_cmfd_logic, dim3(1), dim3(256), 0, st, 1, T->d_partials, gN, stride, S->d_cmsc);
                hipLaunchKernelGGL(k_cmfd_update, dim3(gN), dim3(256), 0, st, diag, pp, q, x, r, z, N, S->d_cmsc, T->d_partials, stride);
                hipLaunchKernelGGL(k_cmfd_logic, dim3(1), dim3(256), 0, st, 2, T->d_partials, gN, stride, S->d_cmsc);
                hipLaunchKernelGGL(k_cmfd_pupdate, dim3(gN), dim3(256), 0, st, z, pp, N, S->d_cmsc);
            }
            HIPCHK(hipMemcpyAsync(&hs, S->d_cmsc, sizeof(hs), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        S->cmfd_last_its += hs.its;
        hipLaunchKernelGGL(k_cmfd_correct, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, x, S->d_raw + (size_t)g * NP, N, S->nloc, S->cmfd_relax);
    }
    HIPCHK(hipGetLastError());
    return NF_OK;
}

// CMFD on a slab team: every slab initialises its own faces; the D-tilde of an interface z face couples the edge cells of two slabs
// (one plane of D per group and the edge cell height travel to the neighbour, once per BuildMatrices).
static int cmfd_initialize_team(nf_team *T)
{
    bool done = true, any_if = false;
    for (auto *S : T->slabs) { done &= S->cmfd_init && S->cmfd_iface; any_if |= S->if_lo || S->if_hi; }
    if (done) return NF_OK;
    {
        int lrc = NF_OK;
        for (auto *S : T->slabs) if (lrc == NF_OK) lrc = cmfd_initialize(S);     // allocations + the slab-local D-tilde
        if (any_if) NFCHK(team_verdict(T, lrc, "the interface exchange of nf_initialize_cmfd")); else NFCHK(lrc);
    }
    if (any_if) {
        NFCHK(team_prepare(T));
        hipStream_t st = T->stream;
        const int ng = T->slabs[0]->ng;
        for (int g = 0; g < ng; ++g) {
            for (auto *S : T->slabs) {                            // edge planes of D
                const size_t b = (size_t)S->nlines[2] * sizeof(double);
                HIPCHK(hipMemcpyAsync(S->d_clo, S->d_D + (size_t)g * S->N, b, hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(S->d_chi, S->d_D + (size_t)g * S->N + (size_t)(S->nz - 1) * S->nlines[2], b, hipMemcpyDeviceToDevice, st));
            }
            NFCHK(exchange_planes(T, 0, 0, st));
            for (auto *S : T->slabs) {
                const size_t b = (size_t)S->nlines[2] * sizeof(double);
                HIPCHK(hipMemcpyAsync(S->d_ctlo, S->d_rlo, b, hipMemcpyDeviceToDevice, st)); HIPCHK(hipMemcpyAsync(S->d_cthi, S->d_rhi, b, hipMemcpyDeviceToDevice, st));
                const int gr = grid_for(S->nlines[2]);
                hipLaunchKernelGGL(k_fill_const, dim3(gr), dim3(256), 0, st, S->d_clo, S->nlines[2], S->hz.front());   // edge cell heights
                hipLaunchKernelGGL(k_fill_const, dim3(gr), dim3(256), 0, st, S->d_chi, S->nlines[2], S->hz.back());
            }
            NFCHK(exchange_planes(T, 0, 0, st));
            for (auto *S : T->slabs) {
                const long nl = S->nlines[2]; const unsigned gr = (unsigned)((nl + 255) / 256);
                double *Dt = S->d_Dt[2] + (size_t)g * S->nfc[2];
                if (S->if_lo) hipLaunchKernelGGL(k_cmfd_dtilde_iface, dim3(gr), dim3(256), 0, st, (const double *)(S->d_D + (size_t)g * S->N), (const double *)S->d_ctlo,
                                                 (const double *)S->d_rlo, S->hz.front(), Dt, nl, 1);
                if (S->if_hi) hipLaunchKernelGGL(k_cmfd_dtilde_iface, dim3(gr), dim3(256), 0, st, (const double *)(S->d_D + (size_t)g * S->N + (size_t)(S->nz - 1) * nl),
                                                 (const double *)S->d_cthi, (const double *)S->d_rhi, S->hz.back(), Dt + (size_t)S->nz * nl, nl, 0);
            }
            NFCHK(team_stream_wait(T, st));                     // the exchange planes are reused by the next group
        }
        HIPCHK(hipGetLastError());
    }
    for (auto *S : T->slabs) S->cmfd_iface = true;
    return NF_OK;
}

// UpdateDhat + ApplyCMFDCorrection on a slab team (:1750-1761): the PCG of cmfd_step with one plane of p per interface and
// iteration (7-point stencil across the cut) and team-wide dot products (process-local sums + all-reduce over ranks); the CG
// scalars live in the first local slab's CmfdScalars.  Same arithmetic per cell as the undivided solve.
static int cmfd_step_team(nf_team *T, double keff, int use_diag)
{
    hipStream_t st = T->stream;
    const int ns = (int)T->slabs.size(), ng = T->slabs[0]->ng;
    CmfdScalars *sc = T->slabs[0]->d_cmsc;
    std::vector<int> gN(ns);
    for (int i = 0; i < ns; ++i) gN[i] = grid_for(T->slabs[i]->N);
    auto reduce_logic = [&](int op, int nq) -> int {
        PartSegs ps = segs_for(T, gN);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, st, (int)FIN_SUM, T->d_partials, ps, T->partial_stride, nq, T->d_cg, T->d_red, 0.0, 0, 1, T->d_red, (const double *)nullptr);
        if (T->rccl_reduce) NCCLCHK(g_rccl.AllReduce(T->d_red, T->d_red, (size_t)nq, NCCL_DOUBLE, NCCL_SUM, T->comm, st));
        hipLaunchKernelGGL(k_cmfd_logic_tot, dim3(1), dim3(1), 0, st, op, (const double *)T->d_red, sc);
        return NF_OK;
    };
    for (auto *S : T->slabs) {                                    // D-hat: x faces only, slab-local (x lines do not cross slabs)
        const long N = S->N, NP = S->nphi;
        Geom G = make_geom(S);
        int nfa = 1, ni = S->k; for (int t = 1; t < S->dim; ++t) { nfa *= S->k + 1; ni *= S->k + 1; }
        const long nJface = S->nJx + S->nJy + S->nJz;
        for (int g = 0; g < ng; ++g) {
            ModeArgs ma = mode_args(S, g, 0, 0, S->d_raw + (size_t)g * NP, S->d_raw + (size_t)g * NP);
            hipLaunchKernelGGL(k_flux_to_J, dim3((unsigned)((S->nlines[0] + 63) / 64)), dim3(64), 0, st, G, ma, S->nb, 0, nfa, ni,
                               S->d_D + g * N, S->d_L[0] + g * N, S->d_DR[0] + g * N, S->d_D0[0] + g * S->nlines[0],
                               S->d_cmJ, S->d_cmJ + nJface, S->nlines[0], use_diag);
            hipLaunchKernelGGL(k_cmfd_dhat, dim3((unsigned)((S->nfc[0] + 255) / 256)), dim3(256), 0, st, G, S->d_raw + (size_t)g * NP, S->d_cmJ, nfa,
                               S->d_Dt[0] + g * S->nfc[0], S->d_Dh[0] + g * S->nfc[0], S->nfc[0]);
        }
        S->cmfd_last_its = 0;
    }
    for (int g = 0; g < ng; ++g) {
        for (int i = 0; i < ns; ++i) {
            nf_solver *S = T->slabs[i]; const long N = S->N;
            CmfdFaces F; for (int d = 0; d < 3; ++d) { F.Dt[d] = S->d_Dt[d] + g * S->nfc[d]; F.Dh[d] = S->d_Dh[d] + g * S->nfc[d]; }
            double *diag = S->d_cm, *x = diag + N, *r = x + N, *pp = r + N;
            hipLaunchKernelGGL(k_cmfd_setup, dim3(gN[i]), dim3(256), 0, st, make_geom(S), F, S->d_Cd + (size_t)g * S->nphi, S->d_Chi + g * N, S->d_tf, 1.0 / keff,
                               diag, x, r, pp, N, T->d_partials + i * T->slab_cap, T->partial_stride);
        }
        NFCHK(reduce_logic(0, 2));
        CmfdScalars hs; hs.done = 0; hs.its = 0;
        for (int base = 0; base < 100 && !hs.done; base += 10) {
            for (int it = 0; it < 10; ++it) {
                for (auto *S : T->slabs) {                        // edge planes of p to the neighbours
                    const size_t b = (size_t)S->nlines[2] * sizeof(double); const double *pp = S->d_cm + 3 * S->N;
                    HIPCHK(hipMemcpyAsync(S->d_clo, pp, b, hipMemcpyDeviceToDevice, st));
                    HIPCHK(hipMemcpyAsync(S->d_chi, pp + (size_t)(S->nz - 1) * S->nlines[2], b, hipMemcpyDeviceToDevice, st));
                }
                NFCHK(exchange_planes(T, 0, 0, st));
                for (int i = 0; i < ns; ++i) {
                    nf_solver *S = T->slabs[i]; const long N = S->N;
                    CmfdFaces F; for (int d = 0; d < 3; ++d) { F.Dt[d] = S->d_Dt[d] + g * S->nfc[d]; F.Dh[d] = S->d_Dh[d] + g * S->nfc[d]; }
                    double *diag = S->d_cm, *pp = diag + 3 * N, *q = diag + 4 * N;
                    hipLaunchKernelGGL(k_cmfd_matvec, dim3(gN[i]), dim3(256), 0, st, make_geom(S), F, (const double *)diag, (const double *)pp, q, N, (const CmfdScalars *)sc,
                                       T->d_partials + i * T->slab_cap, S->if_lo ? (const double *)S->d_rlo : (const double *)nullptr,
                                       S->if_hi ? (const double *)S->d_rhi : (const double *)nullptr);
                }
                NFCHK(reduce_logic(1, 1));
                for (int i = 0; i < ns; ++i) {
                    nf_solver *S = T->slabs[i]; const long N = S->N;
                    double *diag = S->d_cm, *x = diag + N, *r = x + N, *pp = r + N, *q = pp + N, *z = q + N;
                    hipLaunchKernelGGL(k_cmfd_update, dim3(gN[i]), dim3(256), 0, st, (const double *)diag, (const double *)pp, (const double *)q, x, r, z, N, (const CmfdScalars *)sc,
                                       T->d_partials + i * T->slab_cap, T->partial_stride);
                }
                NFCHK(reduce_logic(2, 2));
                for (int i = 0; i < ns; ++i) {
                    nf_solver *S = T->slabs[i]; const long N = S->N;
                    hipLaunchKernelGGL(k_cmfd_pupdate, dim3(gN[i]), dim3(256), 0, st, (const double *)(S->d_cm + 5 * N), S->d_cm + 3 * N, N, (const CmfdScalars *)sc);
                }
            }
            HIPCHK(hipMemcpyAsync(&hs, sc, sizeof(hs), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        for (auto *S : T->slabs) {
            S->cmfd_last_its += hs.its;
            hipLaunchKernelGGL(k_cmfd_correct, dim3((unsigned)((S->N + 255) / 256)), dim3(256), 0, st, (const double *)(S->d_cm + S->N), S->d_raw + (size_t)g * S->nphi, S->N, S->nloc, S->cmfd_relax);
        }
    }
    HIPCHK(hipGetLastError());
    return NF_OK;
}

// the two halves of SolveCoarse as separate entry points (undivided meshes): a built coarse handle, and the injection of its
// flux into the fine handle's current flux (piecewise constant, higher moments zero, :2585-2606)
int nf_coarsen(nf_handle S, int rx, int ry, int rz, nf_handle *coarse)
{
    if (!S || !coarse || rx < 1 || ry < 1 || rz < 1) return fail(NF_ERR_ARG, "nf_coarsen: bad arguments");
    if (!S->xs_uploaded) return fail(NF_ERR_STATE, "nf_coarsen: call nf_upload_xs first");
    if (!team_is_single(S->team)) return fail(NF_ERR_UNSUPPORTED, "nf_coarsen works on an undivided mesh (slab teams coarsen inside nf_solve_keff)");
    if (S->dim < 2) ry = 1;
    if (S->dim < 3) rz = 1;
    if (S->nx % rx || S->ny % ry || S->nz % rz) return fail(NF_ERR_ARG, "coarse factors (%d,%d,%d) do not divide the mesh %d x %d x %d", rx, ry, rz, S->nx, S->ny, S->nz);
    HIPCHK(hipSetDevice(S->device));
    *coarse = nullptr;
    int rc = coarsen_slab(S, rx, ry, rz, coarse);
    if (rc != NF_OK && *coarse) { std::string keep = g_err; nf_destroy(*coarse); *coarse = nullptr; g_err = keep; }
    return rc;
}
int nf_prolong(nf_handle C, nf_handle S)
{
    if (!C || !S) return fail(NF_ERR_ARG, "nf_prolong: null handle");
    if (C->ng != S->ng || C->dim != S->dim || C->nloc != 1 || C->nx < 1 || S->nx % C->nx || S->ny % C->ny || S->nz % C->nz || C->device != S->device)
        return fail(NF_ERR_ARG, "nf_prolong: %d x %d x %d is not a coarsening of %d x %d x %d (same device, groups, dimension; RT0-P0 coarse mesh)", C->nx, C->ny, C->nz, S->nx, S->ny, S->nz);
    HIPCHK(hipSetDevice(S->device));
    hipStream_t st = S->team->stream;
    HIPCHK(hipStreamSynchronize(C->team->stream));
    HIPCHK(hipMemsetAsync(S->d_phi, 0, (size_t)S->nphi * S->ng * sizeof(double), st));
    hipLaunchKernelGGL(k_prolong, dim3((unsigned)((S->N + 255) / 256)), dim3(256), 0, st, C->d_phi, S->d_phi, S->nx, S->ny, S->nz,
                       S->nx / C->nx, S->ny / C->ny, S->nz / C->nz, S->ng, S->nphi);
    HIPCHK(hipStreamSynchronize(st));
    return NF_OK;
}
// counters and timers of the last solve as one JSON object (profiling slots of nf_profile_get, iteration counts)
int nf_timers(nf_handle S, char *buf, size_t len)
{
    if (!S || !buf || len < 2) return fail(NF_ERR_ARG, "nf_timers: bad arguments");
    nf_team *T = S->team;
    std::string js = "{";
    char tmp[160];
    for (const char *nm : SLOT_NAMES) {
        long c = 0, sk = 0; double msum = 0.0;
        T->prof[nm].stats(&c, &msum, &sk);
        snprintf(tmp, sizeof tmp, "\"%s\": {\"count\": %ld, \"ms\": %.6f, \"skipped_noop\": %ld}, ", nm, c, msum, sk);
        js += tmp;
    }
    snprintf(tmp, sizeof tmp, "\"last_outer\": %d, \"coarse_outer\": %d, \"last_cg_total\": %ld, \"separator_sweeps\": %d}", T->last_outer, T->coarse_outer, T->last_cg_total, T->sep_sweeps);
    js += tmp;
    if (js.size() + 1 > len) return fail(NF_ERR_ARG, "nf_timers: buffer of %zu bytes is too small (%zu needed)", len, js.size() + 1);
    memcpy(buf, js.c_str(), js.size() + 1);
    return NF_OK;
}

// ---- diagonal path, outer loop on the device (undivided mesh, no CMFD) --------------------------------------------
static int solve_keff_diag_device(nf_team *T, const nf_keff_opts *o, double keff0, const double *ca, const double *cbv, double sigma,
                                  double *keff_out)
{
    nf_solver *S = T->slabs[0];
    const int ng = S->ng; const long N = S->N, NP = S->nphi, NT = NP * ng;
    hipStream_t st = T->stream;
    if (!S->d_p0) { NFCHK(dalloc(&S->d_p0, (size_t)NT)); NFCHK(dalloc(&S->d_p1, (size_t)NT)); }
    if (!T->d_ost) NFCHK(dalloc(&T->d_ost, 1));
    if (T->hist_cap < o->max_outer) { NFCHK(dalloc(&T->d_hist, (size_t)3 * o->max_outer + 8)); T->hist_cap = o->max_outer; }
    double *hk = T->d_hist, *hdk = hk + T->hist_cap, *hdp = hdk + T->hist_cap;
    OuterState h; memset(&h, 0, sizeof h);
    h.keff = keff0; h.tol_keff = o->tol_keff; h.tol_flux = o->tol_flux; h.max_outer = o->max_outer;
    h.ca1 = ca[1];
    for (int i = 2; i < 15; ++i) { h.a3[i] = (4. / sigma) * ca[i]; h.cb[i] = cbv[i]; }
    HIPCHK(hipMemcpyAsync(T->d_ost, &h, sizeof h, hipMemcpyHostToDevice, st));
    const int gN = grid_for(NP);
    const long stride = T->partial_stride;
    hipLaunchKernelGGL(k_fission, dim3(gN), dim3(256), 0, st, S->d_Mf, S->d_phi, ng, NP, S->d_tf, T->d_partials, (const double *)nullptr, 0L);
    ScatterArgs sa; sa.ng = ng;
    OuterState hs = h;
    int queued = 0;
    while (queued < o->max_outer) {
        const int nb = std::min(queued == 0 ? 8 : 32, o->max_outer - queued);
        for (int b = 0; b < nb; ++b) {
            for (int g = 0; g < ng; ++g) {
                for (int gp = 0; gp < 64; ++gp) sa.M[gp] = gp < ng ? S->d_Ms[g * ng + gp] : nullptr;
                hipLaunchKernelGGL(k_diag_group, dim3(gN), dim3(256), 0, st, sa, g, S->d_Chi + g * N, S->d_tf, S->d_raw, S->d_phi, S->d_Sinv + g * N,
                                   S->d_Mf + (size_t)g * NP, S->d_raw + (size_t)g * NP, NP, T->d_ost, T->d_partials, stride);
            }
            hipLaunchKernelGGL(k_outer_logic, dim3(1), dim3(256), 0, st, T->d_partials, gN, gN, stride, T->d_ost, hk, hdk, hdp);
            hipLaunchKernelGGL(k_normalize_fission, dim3(gN), dim3(256), 0, st, S->d_raw, S->d_phi, S->d_p0, S->d_p1, S->d_Mf, ng, NP, T->d_ost, S->d_tf, T->d_partials);
        }
        queued += nb;
        HIPCHK(hipMemcpyAsync(&hs, T->d_ost, sizeof hs, hipMemcpyDeviceToHost, st));
        HIPCHK(stream_wait(st));
        if (hs.done || hs.done_next) break;
    }
    HIPCHK(hipGetLastError());
    const int n = hs.it;
    T->hist_k.resize(n); T->hist_dk.resize(n); T->hist_dphi.resize(n); T->hist_cg.assign((size_t)n * ng, 0);
    if (n > 0) {
        HIPCHK(hipMemcpy(T->hist_k.data(), hk, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(T->hist_dk.data(), hdk, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(T->hist_dphi.data(), hdp, n * sizeof(double), hipMemcpyDeviceToHost));
    }
    T->last_outer = n; T->last_cg_total = 0;
    if (hs.done == 2) return fail(NF_ERR_NUMERIC, "power iteration diverged (outer %d: k=%g dphi=%g)", n - 1, T->hist_k[n - 1], T->hist_dphi[n - 1]);
    *keff_out = hs.keff;
    return NF_OK;
}

// ---- resident solve (k_resident_keff): whole power iteration of a small undivided mesh in one launch ------------------
static bool resident_plan(const nf_solver *S, ResidentArgs *A)
{
    const int K = 2;
    int lanes = (S->nx + K - 1) / K, lpl_log2 = 0;
    while ((1 << lpl_log2) < lanes && lpl_log2 < 6) ++lpl_log2;
    const int LPL = 1 << lpl_log2, LPW = 64 / LPL;
    if (S->nx > LPL * K) return false;                           // one chunk per x line (nx <= 128)
    const int seg = S->nb > 0 ? 4 : 8;
    for (int r = 0; r < 2; ++r) {
        if (S->dim < r + 2) { A->n[r] = 1; A->TX[r] = 8; A->NSEG[r] = 1; A->gx[r] = A->gy[r] = 0; A->sl[r] = A->ostride[r] = 0; continue; }
        const int n = r == 0 ? S->ny : S->nz;
        const int NSEG = (n + seg - 1) / seg;
        if (NSEG > 64) return false;
        int TX = 64;
        while (TX > 8 && TX * NSEG > 512) TX >>= 1;
        while (TX > 8 && TX / 2 >= S->nx) TX >>= 1;
        const long nxy = (long)S->nx * S->ny;
        A->n[r] = n; A->TX[r] = TX; A->NSEG[r] = NSEG; A->gx[r] = (S->nx + TX - 1) / TX; A->gy[r] = r == 0 ? S->nz : S->ny;
        A->sl[r] = r == 0 ? S->nx : nxy; A->ostride[r] = r == 0 ? nxy : S->nx;
    }
    A->lpl_log2 = lpl_log2; A->ntask_x = (int)((S->nlines[0] + LPW - 1) / LPW);
    return true;
}
// does the line-per-lane variant of the resident kernel take this mesh?  (the same arithmetic as the plan inside solve_keff_resident)
static bool resident_serial_fits(const nf_team *T, const nf_solver *S)
{
    if (!T->opt_resident_serial || !T->opt_resident_lds) return false;
    const long cap = (long)T->lds_limit / 8 - 32;
    const long Np = (long)(S->nx | 1) * S->ny * S->nz;
    long lines = 0; for (int d = 0; d < S->dim; ++d) lines += (S->nlines[d] + 1) & ~1L;
    if (S->nb == 0) { const long pitch = Np <= 1536 ? 1536 : 2560; return Np <= pitch && 64 + 16 + (1 + 3L * S->dim) * pitch + 3 * lines <= cap; }
    const long PC = (Np + 63) & ~63L, NPp = PC * S->nloc;
    return NPp <= 5120 && n_modes(S) <= 9 && S->nloc <= 27 && 64 + 16 + 176 + (1 + S->dim) * NPp + 2L * S->dim * PC + 3 * lines <= cap;
}
static const int NF_RESIDENT_UNAVAILABLE = 1;                     // positive: not an error code of the C ABI
static int solve_keff_resident(nf_team *T, const nf_keff_opts *o, double keff0, const double *ca, const double *cbv, double sigma,
                               double cg_tol, int cg_max, double *keff_out)
{
    nf_solver *S = T->slabs[0];
    const int ng = S->ng; hipStream_t st = T->stream;
    ResidentArgs A; memset(&A, 0, sizeof A);
    const bool planned = resident_plan(S, &A);
    if (!planned && !resident_serial_fits(T, S)) return fail(NF_ERR_STATE, "resident solve: mesh not eligible");
    A.G = make_geom(S); A.ng = ng; A.dim = S->dim; A.nmodes = n_modes(S); A.N = S->N; A.nphi = S->nphi;
    for (int d = 0; d < 3; ++d) {
        const int dd = d < S->dim ? d : 0;
        A.ma[d] = mode_args(S, 0, dd, 0, S->d_p, S->d_q); A.mt[d] = mode_tab(S, dd);
        A.L[d] = S->d_L[dd]; A.DR[d] = S->d_DR[dd]; A.D0[d] = S->d_D0[dd]; A.nlines[d] = S->nlines[dd];
    }
    A.Mf = S->d_Mf; A.Chi = S->d_Chi; A.Ms = S->d_Ms_tab;
    A.phi = S->d_phi; A.raw = S->d_raw; A.p0 = S->d_p0; A.p1 = S->d_p1; A.tf = S->d_tf; A.r = S->d_r; A.p = S->d_p; A.q = S->d_q;
    A.keff0 = keff0; A.tol_keff = o->tol_keff; A.tol_flux = o->tol_flux; A.cg_tol = cg_tol; A.cg_max = cg_max; A.max_outer = o->max_outer;
    A.ca1 = ca[1];
    for (int i = 2; i < 15; ++i) { A.a3[i] = (4. / sigma) * ca[i]; A.cb[i] = cbv[i]; }
    if (T->hist_cap < o->max_outer) { NFCHK(dalloc(&T->d_hist, (size_t)3 * o->max_outer + 8)); T->hist_cap = o->max_outer; }
#ifdef NF_STAMPS
    (void)hipMemsetAsync(T->d_hist + 3 * (size_t)o->max_outer, 0, 8 * sizeof(double), st);
#endif
    if (T->hist_cg_cap < o->max_outer * ng) { NFCHK(dalloc(&T->d_hist_cg, (size_t)o->max_outer * ng)); T->hist_cg_cap = o->max_outer * ng; }
    if (!T->d_rout) NFCHK(dalloc(&T->d_rout, 1));
    A.hist = T->d_hist; A.hist_cg = T->d_hist_cg; A.out = T->d_rout;
    // the kernel indexes its history with max_outer as the row length
    const int B = 512;
    // LDS plan: 160 KiB per workgroup; behind the scratch of the direction passes the CG vectors and this group's factors, in
    // priority order, as far as they fit (ResidentArgs::lds_mask)
    A.Cd0 = S->d_Cd; A.lds_mask = 0;
    // RT0-P0: the line-per-lane variant when everything its sweeps touch fits in LDS (p, the contribution of every direction, the
    // factors and first pivots); r, x_sol and the C diagonal follow as far as there is room
    const long cap = (long)T->lds_limit / 8 - 32;
    bool serial = false;
    if (S->nb == 0 && T->opt_resident_serial && T->opt_resident_lds && S->nphi <= T->resident_serial_max_dofs) {
        const long Np = (long)(S->nx | 1) * S->ny * S->nz;     // rows padded to an odd length (bank-conflict-free x lines)
        const int pitch = Np <= 1536 ? 1536 : 2560;              // multiples of 512 (cells per thread) and of 64 (ds_read2st64)
        long need = 64 + 16 + (1 + 3L * S->dim) * pitch;
        int slot = 0;
        for (int d = 0; d < S->dim; ++d) {
            const int nd = d == 0 ? S->nx : d == 1 ? S->ny : S->nz;
            A.tw[d] = (nd >= 4 && T->opt_resident_two_sided) ? 1 : 0;   // two lanes per line, meeting in the middle
            need += 3 * ((S->nlines[d] + 1) & ~1L); A.slot0[d] = slot; slot += (int)((S->nlines[d] * (A.tw[d] ? 2 : 1) + 63) / 64) * 64;
        }
        for (int d = S->dim; d < 4; ++d) A.slot0[d] = slot;
        if (Np <= pitch && need <= cap) {
            serial = true;
            const size_t lds = (size_t)need * sizeof(double);
#define NF_RES_SERIAL(P) do { if (!lds_opt_in((const void *)k_resident_keff<false, 0, P>, lds)) return NF_RESIDENT_UNAVAILABLE; \
            hipLaunchKernelGGL((k_resident_keff<false, 0, P>), dim3(1), dim3(B), lds, st, A); } while (0)
            if (pitch == 1536) NF_RES_SERIAL(1536); else NF_RES_SERIAL(2560);
#undef NF_RES_SERIAL
        }
    }
    // RT_k-P_m with bubble moments: the same idea, DOF = moment * PC + padded cell, one contribution vector per direction
    if (S->nb > 0 && T->opt_resident_serial && T->opt_resident_lds && S->nphi <= T->resident_serial_max_dofs) {
        const long Np = (long)(S->nx | 1) * S->ny * S->nz;
        const long PC = (Np + 63) & ~63L, NPp = PC * S->nloc;
        const int nm = n_modes(S);
        long need = 64 + 16 + 176 + (1 + S->dim) * NPp + 2L * S->dim * PC;
        int slot = 0;
        for (int d = 0; d < S->dim; ++d) {
            const int nd = d == 0 ? S->nx : d == 1 ? S->ny : S->nz;
            A.tw[d] = (nd >= 4 && T->opt_resident_two_sided) ? 1 : 0;
            need += 3 * ((S->nlines[d] + 1) & ~1L); A.slot0[d] = slot; slot += (int)((S->nlines[d] * nm * (A.tw[d] ? 2 : 1) + 63) / 64) * 64;
        }
        for (int d = S->dim; d < 4; ++d) A.slot0[d] = slot;
        if (NPp <= 5120 && nm <= 9 && S->nloc <= 27 && need <= cap) {
            serial = true;
            A.PC = (int)PC;
            memset(A.mom, 0, sizeof A.mom); memset(A.diagc, 0, sizeof A.diagc);
            const ModeArgs c = mode_args(S, 0, 0, 0, S->d_p, S->d_q);
            for (int d = 0; d < S->dim; ++d)
                for (int m = 0; m < nm; ++m) {
                    double Ta = 1.0;
                    for (int i = 0; i <= S->nb; ++i) A.mom[d][m][i] = moment_index(S, d, m, i, &Ta);
                    for (int l = 0; l < S->nb; ++l) A.diagc[A.mom[d][m][l + 1]][d] = Ta * c.Gc[l] * c.Gc[l] * c.iM[l];
                }
            const size_t lds = (size_t)need * sizeof(double);
#define NF_RES_HI(NBV) do { if (!lds_opt_in((const void *)k_resident_keff<false, NBV, -1>, lds)) return NF_RESIDENT_UNAVAILABLE; \
            hipLaunchKernelGGL((k_resident_keff<false, NBV, -1>), dim3(1), dim3(B), lds, st, A); } while (0)
            if (S->nb == 1) NF_RES_HI(1); else NF_RES_HI(2);
#undef NF_RES_HI
        }
    }
    T->last_resident_serial = serial ? 1 : 0;
    if (!serial && !planned) return fail(NF_ERR_STATE, "resident solve: mesh not eligible");
    if (!serial) {
    long used = 5 * B + 64 + 16;
    if (T->opt_resident_lds) {
        const long NP2 = (S->nphi + 1) & ~1L, N2 = (S->N + 1) & ~1L;
        const int bits[11] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10 };
        for (int b : bits) {
            if ((b == 6 || b == 7) && S->dim < 2) continue;
            if ((b == 8 || b == 9) && S->dim < 3) continue;
            const long need = (b <= 3 || b == 10) ? NP2 : N2;
            if (used + need <= cap) { used += need; A.lds_mask |= 1 << b; }
        }
    }
    const size_t lds = (size_t)used * sizeof(double);
#define NF_RES(VECV, NBV) do { if (!lds_opt_in((const void *)k_resident_keff<VECV, NBV>, lds)) return NF_RESIDENT_UNAVAILABLE; \
        hipLaunchKernelGGL((k_resident_keff<VECV, NBV>), dim3(1), dim3(B), lds, st, A); } while (0)
    const bool vec = S->nx % 2 == 0;
    if (S->nb == 0) { if (vec) NF_RES(true, 0); else NF_RES(false, 0); }
    else if (S->nb == 1) { if (vec) NF_RES(true, 1); else NF_RES(false, 1); }
    else { if (vec) NF_RES(true, 2); else NF_RES(false, 2); }
#undef NF_RES
    }
    HIPCHK(hipGetLastError());
    ResidentOut ro;
    HIPCHK(hipMemcpyAsync(&ro, T->d_rout, sizeof ro, hipMemcpyDeviceToHost, st));
    HIPCHK(stream_wait(st));
    const int n = ro.n_outer;
    T->hist_k.resize(n); T->hist_dk.resize(n); T->hist_dphi.resize(n); T->hist_cg.resize((size_t)n * ng);
    if (n > 0) {
        HIPCHK(hipMemcpy(T->hist_k.data(), T->d_hist, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(T->hist_dk.data(), T->d_hist + o->max_outer, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(T->hist_dphi.data(), T->d_hist + 2 * (size_t)o->max_outer, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(T->hist_cg.data(), T->d_hist_cg, (size_t)n * ng * sizeof(int), hipMemcpyDeviceToHost));
        for (int g = 0; g < ng; ++g) T->last_its[g] = T->hist_cg[(size_t)(n - 1) * ng + g];
    }
    T->last_outer = n; T->last_cg_total = ro.cg_total;
    if (ro.status == 2) return fail(NF_ERR_NUMERIC, "power iteration diverged (outer %d: k=%g dphi=%g)", n - 1, T->hist_k[n - 1], T->hist_dphi[n - 1]);
#ifdef NF_STAMPS
    { long long acc[5]; (void)hipMemcpy(acc, T->d_hist + 3 * (size_t)o->max_outer, sizeof acc, hipMemcpyDeviceToHost);
      if (acc[4] > 0) fprintf(stderr, "[resident stamps] %lld CG iterations: x pass %.0f, y/z passes %.0f, p.q reduction %.0f, r update + |r|^2 reduction %.0f cycles per iteration\n",
                              acc[4], (double)acc[0] / acc[4], (double)acc[1] / acc[4], (double)acc[2] / acc[4], (double)acc[3] / acc[4]); }
#endif
    *keff_out = ro.keff;
    return NF_OK;
}

// ---- whole SolveKeff on one XCD (k_keff_xcd): the meshes of k_cg_xcd, iterative full-Schur path ------------------------
// returns NF_RESIDENT_UNAVAILABLE when the workgroups did not assemble (nothing has been touched: the host-driven path takes over)
static int solve_keff_xcd(nf_team *T, const nf_keff_opts *o, const Fuse3Plan &P, double keff0, const double *ca, const double *cbv, double sigma,
                          double cg_tol, int cg_max, double *keff_out)
{
    nf_solver *S = T->slabs[0];
    const int ng = S->ng; hipStream_t st = T->stream;
    if (!S->d_p2) NFCHK(dalloc(&S->d_p2, S->nphi));
    if (S->dim >= 2 && !S->d_qy) NFCHK(dalloc(&S->d_qy, S->nphi));
    if (S->dim == 3 && !S->d_qz) NFCHK(dalloc(&S->d_qz, S->nphi));
    XcdArgs A; int nch = 1;
    NFCHK(xcd_fill(S, 0, P, A, &nch));
    XcdOuter O; memset(&O, 0, sizeof O);
    O.ng = ng; O.NP = S->nphi; O.N = S->N;
    O.Mf = S->d_Mf; O.Chi = S->d_Chi; O.Ms = S->d_Ms_tab;
    O.phi = S->d_phi; O.raw = S->d_raw; O.p0 = S->d_p0; O.p1 = S->d_p1; O.tf = S->d_tf;
    for (int d = 0; d < 3; ++d) O.nl[d] = S->nlines[d < S->dim ? d : 0];
    O.keff0 = keff0; O.tol_keff = o->tol_keff; O.tol_flux = o->tol_flux; O.cg_tol = cg_tol; O.cg_max = cg_max; O.max_outer = o->max_outer;
    O.ca1 = ca[1];
    for (int i = 2; i < 15; ++i) { O.a3[i] = (4. / sigma) * ca[i]; O.cb[i] = cbv[i]; }
    if (T->hist_cap < o->max_outer) { NFCHK(dalloc(&T->d_hist, (size_t)3 * o->max_outer + 8)); T->hist_cap = o->max_outer; }
    if (T->hist_cg_cap < o->max_outer * ng) { NFCHK(dalloc(&T->d_hist_cg, (size_t)o->max_outer * ng)); T->hist_cg_cap = o->max_outer * ng; }
    if (!T->d_rout) NFCHK(dalloc(&T->d_rout, 1));
    O.hist = T->d_hist; O.hist_cg = T->d_hist_cg; O.out = T->d_rout;
    const size_t lds = XCD_LDS;
    const unsigned G = 8u * (unsigned)T->xcd_groups;
#define NF_XK(NCHV, VECV, NBV, SEGV) do { if (!lds_opt_in((const void *)k_keff_xcd<NCHV, VECV, NBV, SEGV>, lds)) return NF_RESIDENT_UNAVAILABLE; \
        hipLaunchKernelGGL((k_keff_xcd<NCHV, VECV, NBV, SEGV>), dim3(G), dim3(XCD_THREADS), lds, st, A, O); } while (0)
#define NF_XK_NB(NCHV, VECV) do { if (S->nb == 0) NF_XK(NCHV, VECV, 0, 8); else if (S->nb == 1) NF_XK(NCHV, VECV, 1, 4); else NF_XK(NCHV, VECV, 2, 4); } while (0)
    if (nch == 1) { if (P.vec) NF_XK_NB(1, true); else NF_XK_NB(1, false); }
    else { if (P.vec) NF_XK_NB(2, true); else NF_XK_NB(2, false); }
#undef NF_XK_NB
#undef NF_XK
    HIPCHK(hipGetLastError());
    ResidentOut ro;
    HIPCHK(hipMemcpyAsync(&ro, T->d_rout, sizeof ro, hipMemcpyDeviceToHost, st));
    HIPCHK(stream_wait(st));
    if (ro.status == 3) { T->opt_keffx = 0; T->opt_cgx = 0; ++T->xcd_refused; return NF_RESIDENT_UNAVAILABLE; }
    if (ro.status == 4) { T->opt_keffx = 0; T->opt_cgx = 0; return fail(NF_ERR_HIP, "XCD-local SolveKeff (k_keff_xcd): a grid barrier timed out in outer iteration %d (a participant was lost); the path is now off for this solver", ro.n_outer); }
    const int n = ro.n_outer;
    T->hist_k.resize(n); T->hist_dk.resize(n); T->hist_dphi.resize(n); T->hist_cg.resize((size_t)n * ng);
    if (n > 0) {
        HIPCHK(hipMemcpy(T->hist_k.data(), T->d_hist, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(T->hist_dk.data(), T->d_hist + o->max_outer, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(T->hist_dphi.data(), T->d_hist + 2 * (size_t)o->max_outer, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(T->hist_cg.data(), T->d_hist_cg, (size_t)n * ng * sizeof(int), hipMemcpyDeviceToHost));
        for (int g = 0; g < ng; ++g) T->last_its[g] = T->hist_cg[(size_t)(n - 1) * ng + g];
    }
    T->last_outer = n; T->last_cg_total = ro.cg_total; T->xcd_solves += (long)n * ng; T->last_xcd = 1;
    if (ro.status == 2) return fail(NF_ERR_NUMERIC, "power iteration diverged (outer %d: k=%g dphi=%g)", n - 1, n > 0 ? T->hist_k[n - 1] : 0.0, n > 0 ? T->hist_dphi[n - 1] : 0.0);
    *keff_out = ro.keff;
    return NF_OK;
}

// ---- SolveKeff (src/NeutFEM.cpp:1627-1815) -----------------------------------------------------
static int solve_keff_impl(nf_team *T, const nf_keff_opts *o, double *keff_out, int *n_outer)
{
    const int ns = (int)T->slabs.size();
    nf_solver *S0 = T->slabs[0];
    const int ng = S0->ng;
    const bool single = team_is_single(T);
    NFCHK(team_prepare(T));
    int use_diag = (o->use_diagonal_solver && S0->k == 0 && S0->m == 0) ? 1 : 0;   // flag dropped for order > 0 (:1640-1644)
    if (use_diag) NFCHK(nf_build_diagonal_cache(S0));         // whole team; slabs exchange one edge plane per group
    const bool use_cmfd = o->use_cmfd != 0;
    if (use_cmfd) NFCHK(cmfd_initialize_team(T));                 // :1655-1658
    double keff = T->has_valid_keff ? T->last_keff : 1.0;        // :1662
    T->coarse_outer = 0;
    if (o->use_coarse_init && o->n_coarse_factors > 0) {          // :1665-1670
        bool done = false; double kc = 1.0;
        std::vector<double *> dsts; for (auto *S : T->slabs) dsts.push_back(S->d_phi);
        NFCHK(coarse_init(T, o, &kc, dsts, &done));
        keff = done ? kc : 1.0;
    }
    long Ntot = 0; for (auto *S : T->slabs) Ntot += S->nphi;
    // SchurSolver type: DIRECT_* or n_phi < 200 -> "exact" solve (CG to 1e-14 stands in)
    // SchurSolver::NeedsExplicitSchur (src/solvers.cpp:114-124): direct types, a type that was never pushed (quirk 11), n_phi < 200.
    // Undivided meshes up to direct_max_dofs get the real thing (dense S^-1, dense_prepare); beyond that and on slab teams CG to
    // 1e-14 stands in, bounded so that an ill-conditioned S (IAEA-3D: cond ~1e17) cannot run away; such solves are counted.
    const bool direct = !o->solver_type_pushed || o->solver_type <= 2 || (single && Ntot < 200);
    const bool dense = direct && single && !T->rccl_reduce && Ntot <= T->direct_max_dofs;
    const double cg_tol = direct ? 1e-14 : o->tol_flux;           // SetTolerance forwards tol_flux (:334)
    const int cg_max = direct ? (int)std::min<long>(20 * Ntot + 50, 200000L) : o->max_inner;   // CG ends in <= n steps in exact arithmetic
    T->last_direct = dense ? 1 : direct ? 2 : 0;
    if (dense) NFCHK(dense_prepare(T));
    // ChebyshevAccel(15, 0.98), src/solvers.cpp:664-700
    const int nmax = CHEB_NMAX; const double sigma = CHEB_SIGMA;
    double ca[16], cbv[16];
    cheb_tables(ca, cbv);
    int cheb_it = 0;
    T->hist_k.clear(); T->hist_dk.clear(); T->hist_dphi.clear(); T->hist_cg.clear();
    T->last_outer = 0; T->last_cg_total = 0;
    T->profile = o->profile != 0;
    if (use_diag && single && !use_cmfd && !T->rccl_reduce && T->opt_outer_dev && o->max_outer > 0) {
        NFCHK(solve_keff_diag_device(T, o, keff, ca, cbv, sigma, &keff));
        HIPCHK(hipStreamSynchronize(T->stream));
        T->profile = false; T->last_path = 1;
        S0->raw_valid = T->last_outer > 0; S0->raw_is_diag = true;
        T->has_valid_keff = 1; T->last_keff = keff;
        if (keff_out) *keff_out = keff;
        if (n_outer) *n_outer = T->last_outer;
        return NF_OK;
    }
    T->last_path = 0;
    {
        ResidentArgs probe;
        if (single && !use_diag && !use_cmfd && !direct && !T->rccl_reduce && T->opt_resident && o->max_outer > 0 &&
            ((S0->nphi <= T->resident_max_dofs && resident_plan(S0, &probe)) || (S0->nphi <= T->resident_serial_max_dofs && resident_serial_fits(T, S0)))) {
            const bool prof_req = T->profile;
            T->last_path = 2; T->profile = false;
            const int rr = solve_keff_resident(T, o, keff, ca, cbv, sigma, cg_tol, cg_max, &keff);
            if (rr != NF_RESIDENT_UNAVAILABLE) {
                NFCHK(rr);
                S0->raw_valid = T->last_outer > 0; S0->raw_is_diag = false; S0->jz_valid = false;
                T->has_valid_keff = 1; T->last_keff = keff;
                if (keff_out) *keff_out = keff;
                if (n_outer) *n_outer = T->last_outer;
                return NF_OK;
            }
            T->last_path = 0; T->profile = prof_req;              // the device refused the LDS the resident kernel plans with: host-driven path
        }
    }
    // mid-size meshes (2 k - 28 k unknowns per group, every order): the whole power iteration in one launch on one XCD (k_keff_xcd)
    if (single && !use_diag && !use_cmfd && !direct && !dense && !T->rccl_reduce && o->max_outer > 0 && T->opt_keffx && T->opt_fuse && T->opt_lean && T->opt_fuse3 &&
        S0->N <= T->fuse3_max_cells && S0->N <= T->lean_max_cells) {
        const Fuse3Plan f3 = fuse3_plan(S0);
        if (xcd_eligible(T, S0, f3)) {
            T->last_path = 3;
            const int rx = solve_keff_xcd(T, o, f3, keff, ca, cbv, sigma, cg_tol, cg_max, &keff);
            if (rx != NF_RESIDENT_UNAVAILABLE) {
                NFCHK(rx);
                S0->raw_valid = T->last_outer > 0; S0->raw_is_diag = false; S0->jz_valid = false;
                T->has_valid_keff = 1; T->last_keff = keff;
                if (keff_out) *keff_out = keff;
                if (n_outer) *n_outer = T->last_outer;
                return NF_OK;
            }
            T->last_path = 0;
        }
    }
    ScatterArgs sa; sa.ng = ng;
    double hout[5] = { 0, 0, 0, 0, 0 };
    std::vector<int> gN(ns), gT(ns);
    std::vector<const double *> rhs(ns); std::vector<double *> sol(ns);
    for (int i = 0; i < ns; ++i) { gN[i] = grid_for(T->slabs[i]->nphi); gT[i] = grid_for(T->slabs[i]->nphi * ng); }
    const bool multi = T->nproc > 1;
    // multi-rank teams: a rank-local failure between two collectives of the outer iteration poisons the rank instead of ending its part
    // of the schedule (team_poison); every rank then leaves at the same point -- the first reduction of the next CG solve, or the
    // per-outer reduction, whichever comes first
#define NF_OUTER_CHK(x) do { const int r_ = (x); if (team_poison(T, r_)) return r_; } while (0)
    for (int it = 0; it < o->max_outer; ++it) {
        if (multi && T->rank == T->inject_rank && T->inject_where == 'o' && it == T->inject_outer)
            NF_OUTER_CHK(fail(NF_ERR_HIP, "injected failure on rank %d at the start of outer iteration %d (NEUTFEM_INJECT_FAIL)", T->rank, it));
        // total_fiss and prod_old (:1700-1707)
        for (int i = 0; i < ns && !T->poisoned; ++i) {
            nf_solver *S = T->slabs[i];
            hipLaunchKernelGGL(k_fission, dim3(gN[i]), dim3(256), 0, T->stream, S->d_Mf, S->d_phi, ng, S->nphi, S->d_tf, T->d_partials + i * T->slab_cap, (const double *)nullptr, 0L);
        }
        NF_OUTER_CHK(team_finalize(T, FIN_SUM, gN, 1, T->d_out, 0.0, 0));
        for (int g = 0; g < ng; ++g) {
            for (int i = 0; i < ns && !T->poisoned; ++i) {
                nf_solver *S = T->slabs[i]; const long N = S->N, NP = S->nphi;
                for (int gp = 0; gp < 64; ++gp) sa.M[gp] = gp < ng ? S->d_Ms[g * ng + gp] : nullptr;
                double *dst = use_diag ? S->d_raw + g * NP : S->d_rhs;
                // CG path: the start of the solve (x = 0, r = p = rhs, |rhs|^2 partials) rides in the same launch (cg_solve(..., inited))
                const bool cgi = !use_diag && !dense;
                hipLaunchKernelGGL(k_group_rhs, dim3(gN[i]), dim3(256), 0, T->stream, sa, g, S->d_Chi + g * N, S->d_tf, 1.0 / keff, S->d_raw, S->d_phi,
                                   use_diag ? S->d_Sinv + g * N : (const double *)nullptr, dst, NP, N,
                                   cgi ? S->d_raw + g * NP : (double *)nullptr, cgi ? S->d_r : (double *)nullptr, cgi ? S->d_p : (double *)nullptr,
                                   cgi ? T->d_partials + i * T->slab_cap : (double *)nullptr);
                rhs[i] = S->d_rhs; sol[i] = S->d_raw + g * NP;
            }
            int its = 0; double res = 0.0;
            if (use_diag) { }
            else if (dense) { dense_solve(T, g, rhs[0], sol[0]); its = 1; }          // last_iterations_ = 1 (src/solvers.cpp:447)
            else {
                NFCHK(cg_solve(T, g, rhs, sol, cg_tol, cg_max, &its, &res, true));
                if (direct && !(res <= 1e-14)) ++T->standin_unconverged;
            }
            T->hist_cg.push_back(its); T->last_cg_total += its;
        }
        if (use_cmfd && it >= 2 && !T->poisoned) { if (single) NFCHK(cmfd_step(S0, keff, use_diag)); else NFCHK(cmfd_step_team(T, keff, use_diag)); }   // :1750-1761
        if (multi && T->rank == T->inject_rank && T->inject_where == 'e' && it == T->inject_outer)
            NF_OUTER_CHK(fail(NF_ERR_HIP, "injected failure on rank %d at the end of outer iteration %d (NEUTFEM_INJECT_FAIL)", T->rank, it));
        // prod_new, norms (:1766-1779)
        for (int i = 0; i < ns && !T->poisoned; ++i) {
            nf_solver *S = T->slabs[i];
            hipLaunchKernelGGL(k_outer_reduce, dim3(gT[i]), dim3(256), 0, T->stream, S->d_Mf, S->d_raw, S->d_phi, S->nphi * ng, T->d_partials + i * T->slab_cap, T->partial_stride);
        }
        NF_OUTER_CHK(team_finalize(T, FIN_SUM, gT, 3, T->d_out + 1, 0.0, 0, true));
        NFCHK(readback(T, nullptr, nullptr, T->d_out, hout, multi ? 5 : 4));
        if (multi && (T->poisoned || hout[4] != 0.0)) {            // the ranks' error flags came with the sums: every rank is here, at the same outer
            if (T->poisoned) return team_poison_result(T);
            return fail(NF_ERR_REMOTE, "another rank of the team reported an error during outer iteration %d; every rank stopped there", it);
        }
        const double prod_old = hout[0], prod_new = hout[1], nsq = hout[2], dsq = hout[3];
        const double keff_new = keff * (prod_new / prod_old);
        const double dk = std::fabs(keff_new - keff);
        if (it >= 1) keff = keff_new;                             // :1774
        const double dphi = std::sqrt(dsq / nsq), norm = std::sqrt(nsq);
        if (!std::isfinite(keff_new) || !std::isfinite(dphi))
            return fail(NF_ERR_NUMERIC, "power iteration diverged (outer %d: k=%g dphi=%g)", it, keff_new, dphi);
        // normalise + Chebyshev (:1780-1788, src/solvers.cpp:720-756)
        int mode = 0; double a = 0.0, b = 0.0;
        if (it >= 2 && !use_cmfd) {                               // :1786-1788
            if (cheb_it == nmax) cheb_it = 0;
            if (cheb_it == 0) { mode = 1; }
            else if (cheb_it == 1) { mode = 2; a = ca[1]; }
            else { mode = 3; a = (4. / sigma) * ca[cheb_it]; b = cbv[cheb_it]; }
        }
        for (int i = 0; i < ns; ++i) {
            nf_solver *S = T->slabs[i]; const long NT = S->nphi * ng;
            if (mode && !S->d_p0) { NF_OUTER_CHK(dalloc(&S->d_p0, (size_t)NT)); if (!T->poisoned) NF_OUTER_CHK(dalloc(&S->d_p1, (size_t)NT)); }
            if (T->poisoned) break;
            hipLaunchKernelGGL(k_normalize_cheb, dim3(gT[i]), dim3(256), 0, T->stream, S->d_raw, S->d_phi, S->d_p0, S->d_p1, NT, norm,
                               norm > 1e-14 ? 1 : 0, mode, a, b);
            if (mode == 3) std::swap(S->d_p0, S->d_p1);
        }
        if (it >= 2 && !use_cmfd) ++cheb_it;
        T->hist_k.push_back(keff); T->hist_dk.push_back(dk); T->hist_dphi.push_back(dphi);
        T->last_outer = it + 1;
        __atomic_store_n(&T->outers_done, (long)(it + 1), __ATOMIC_RELEASE);
        if (T->progress_fn) T->progress_fn(T->progress_user, it, keff, dk, dphi);
        if (dk < o->tol_keff && dphi < o->tol_flux && !T->poisoned) break;        // :1799-1802 (a poisoned rank stays in the schedule until the flag has gone round)
    }
#undef NF_OUTER_CHK
    if (T->poisoned) {                                            // poisoned in the last outer the loop ran: one more reduction takes the flag to everybody
        (void)team_finalize(T, FIN_SUM, gN, 1, T->d_out, 0.0, 0, true);
        (void)team_stream_wait(T, T->stream);
        return team_poison_result(T);
    }
    HIPCHK(hipStreamSynchronize(T->stream));
    HIPCHK(hipGetLastError());
    if (T->profile) prof_collect(T);
    T->profile = false;
    for (auto *S : T->slabs) { S->raw_valid = T->last_outer > 0; S->raw_is_diag = use_diag != 0; S->jz_valid = false; }
    T->has_valid_keff = 1; T->last_keff = keff;                   // :1808-1809
    if (keff_out) *keff_out = keff;
    if (n_outer) *n_outer = T->last_outer;
    return NF_OK;
}

int nf_solve_keff(nf_handle S, const nf_keff_opts *o, double *keff, int *n_outer)
{
    if (!S || !o) return fail(NF_ERR_ARG, "nf_solve_keff: bad arguments");
    if (S->team->dead) return fail(NF_ERR_COMM, "this team lost a collective (NF_ERR_COMM) and is unusable: destroy the handles and end the process with a non-zero code (never re-exec)");
    for (auto *X : S->team->slabs) if (!X->built) return fail(NF_ERR_STATE, "nf_solve_keff: call nf_build first");
    HIPCHK(hipSetDevice(S->device));
    __atomic_store_n(&S->team->outers_done, 0L, __ATOMIC_RELEASE);
    return solve_keff_impl(S->team, o, keff, n_outer);
}

int nf_set_progress_callback(nf_handle S, nf_progress_fn fn, void *user)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    S->team->progress_fn = fn; S->team->progress_user = user;
    return NF_OK;
}

// Outer iterations the running (or last) nf_solve_keff has completed on the host-driven path.  Safe to call from another thread
// while the solve runs: a watchdog (bench.py) uses it to tell a slow solve from one whose peers are gone.
int nf_progress(nf_handle S, long *outers_done)
{
    if (!S || !outers_done) return fail(NF_ERR_ARG, "nf_progress: bad arguments");
    *outers_done = __atomic_load_n(&S->team->outers_done, __ATOMIC_ACQUIRE);
    return NF_OK;
}

// ---- SolveAdjoint (src/NeutFEM.cpp:1877-2082) -----------------------------------------------------
// Literal control flow of the reference, including what makes it fragile (forward-ordered sweep on the transposed scatter,
// Chebyshev from outer 5 when k is free).  Works on slab teams (collective).
int nf_solve_adjoint(nf_handle S, const nf_keff_opts *o, int normalize_to_direct, int use_direct_keff, double *keff_adj, int *n_outer)
{
    if (!S || !o) return fail(NF_ERR_ARG, "nf_solve_adjoint: bad arguments");
    if (S->team->dead) return fail(NF_ERR_COMM, "this team lost a collective (NF_ERR_COMM) and is unusable: destroy the handles and end the process with a non-zero code (never re-exec)");
    nf_team *T = S->team;
    for (auto *X : T->slabs) if (!X->built) return fail(NF_ERR_STATE, "nf_solve_adjoint: call nf_build first");
    HIPCHK(hipSetDevice(T->device));
    NFCHK(team_prepare(T));
    const int ns = (int)T->slabs.size();
    nf_solver *S0 = T->slabs[0];
    const int ng = S0->ng;
    hipStream_t st = T->stream;
    double keff = 1.0;
    if (use_direct_keff && T->has_valid_keff) keff = T->last_keff;          // :1885-1889
    // global number of flux DOFs (the start vector is 1 / ||1||, :1891-1892): local sum, all-reduced on a multi-rank team
    double ntot = 0.0; long NPtot = 0;
    for (auto *X : T->slabs) { ntot += (double)X->nphi * ng; NPtot += X->nphi; }
    if (T->rccl_reduce && T->nproc > 1) {
        HIPCHK(hipMemcpyAsync(T->d_red, &ntot, sizeof(double), hipMemcpyHostToDevice, st));
        NCCLCHK(g_rccl.AllReduce(T->d_red, T->d_red, 1, NCCL_DOUBLE, NCCL_SUM, T->comm, st));
        HIPCHK(hipMemcpyAsync(&ntot, T->d_red, sizeof(double), hipMemcpyDeviceToHost, st));
        NFCHK(team_stream_wait(T, st));
    }
    std::vector<DevTmp<double>> nsft(ns);
    std::vector<int> gN(ns), gT(ns);
    std::vector<const double *> rhs(ns); std::vector<double *> sol(ns);
    for (int i = 0; i < ns; ++i) {
        nf_solver *X = T->slabs[i];
        const long NT = X->nphi * ng;
        gN[i] = grid_for(X->nphi); gT[i] = grid_for(NT);
        if (!X->d_phi_adj) NFCHK(dalloc(&X->d_phi_adj, (size_t)NT));
        hipLaunchKernelGGL(k_fill_const, dim3(gT[i]), dim3(256), 0, st, X->d_phi_adj, NT, 1.0 / std::sqrt(ntot));
        NFCHK(dalloc(&nsft[i].p, (size_t)X->N));
        hipLaunchKernelGGL(k_sum_groups, dim3(grid_for(X->N)), dim3(256), 0, st, X->d_NSF, nsft[i].p, X->N, ng);    // :1898-1905
    }
    const bool direct = !o->solver_type_pushed || o->solver_type <= 2 || (team_is_single(T) && NPtot < 200);
    const bool dense = direct && team_is_single(T) && !T->rccl_reduce && NPtot <= T->direct_max_dofs;   // S is symmetric: the same S^-1
    const double cg_tol = direct ? 1e-14 : o->tol_flux;
    const int cg_max = direct ? (int)std::min<long>(20 * NPtot + 50, 200000L) : o->max_inner;
    if (dense) NFCHK(dense_prepare(T));
    const int nmax = CHEB_NMAX; const double sigma = CHEB_SIGMA;
    double ca[16], cbv[16];
    cheb_tables(ca, cbv);
    int cheb_it = 0;
    T->hist_k.clear(); T->hist_dk.clear(); T->hist_dphi.clear(); T->hist_cg.clear();
    T->last_outer = 0; T->last_cg_total = 0;
    ScatterArgs sa; sa.ng = ng;
    double hout[4];
    int rc = NF_OK;
    // sum_e nsf_tot[e] * (M_chi v)[e, dof 0] over the team, result in `out` (:1912-1936)
    auto chi_product = [&](bool of_raw, double *out) -> int {
        for (int i = 0; i < ns; ++i) {
            nf_solver *X = T->slabs[i];
            hipLaunchKernelGGL(k_fission, dim3(gN[i]), dim3(256), 0, st, X->d_Mchi, of_raw ? X->d_raw : X->d_phi_adj, ng, X->nphi, X->d_tf,
                               T->d_partials + i * T->slab_cap, (const double *)nsft[i].p, X->N);
        }
        return team_finalize(T, FIN_SUM, gN, 1, out, 0.0, 0);
    };
    for (int it = 0; it < o->max_outer && rc == NF_OK; ++it) {
        if ((rc = chi_product(false, T->d_out)) != NF_OK) break;
        for (int g = 0; g < ng && rc == NF_OK; ++g) {
            for (int i = 0; i < ns; ++i) {
                nf_solver *X = T->slabs[i];
                for (int gp = 0; gp < 64; ++gp) sa.M[gp] = gp < ng ? X->d_Ms[gp * ng + g] : nullptr;       // transposed blocks (:1944-1950)
                hipLaunchKernelGGL(k_group_rhs, dim3(gN[i]), dim3(256), 0, st, sa, g, X->d_NSF + g * X->N, X->d_tf, 1.0 / keff, X->d_raw, X->d_phi_adj,
                                   (const double *)nullptr, X->d_rhs, X->nphi, X->N);
                rhs[i] = X->d_rhs; sol[i] = X->d_raw + g * X->nphi;
            }
            int its = 0;
            if (dense) { dense_solve(T, g, rhs[0], sol[0]); its = 1; }
            else rc = cg_solve(T, g, rhs, sol, cg_tol, cg_max, &its, nullptr);
            T->hist_cg.push_back(its); T->last_cg_total += its;
        }
        if (rc != NF_OK) break;
        // NB: chi_product(true) overwrites d_tf with M_chi raw; the next outer recomputes it from phi_adj first
        if ((rc = chi_product(true, T->d_out + 1)) != NF_OK) break;
        for (int i = 0; i < ns; ++i) {
            nf_solver *X = T->slabs[i];
            hipLaunchKernelGGL(k_outer_reduce, dim3(gT[i]), dim3(256), 0, st, X->d_Mchi, X->d_raw, X->d_phi_adj, X->nphi * ng, T->d_partials + i * T->slab_cap, T->partial_stride);
        }
        if ((rc = team_finalize(T, FIN_SUM, gT, 3, T->d_out + 2, 0.0, 0)) != NF_OK) break;     // d_out[2..4]: { unused, ||phi||^2, ||dphi||^2 }
        double h5[5];
        if (hipMemcpyAsync(h5, T->d_out, 5 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = fail(NF_ERR_HIP, "adjoint: read-back failed"); break;
        }
        hout[0] = h5[0]; hout[1] = h5[1];
        const double prod_old = hout[0], prod_new = hout[1], nsq = h5[3], dsq = h5[4];
        double keff_new = keff, dk;
        if (!use_direct_keff || !T->has_valid_keff) {                          // :1966-1975
            if (std::fabs(prod_old) > 1e-14 && it > 0) keff_new = keff * (prod_new / prod_old);
            dk = std::fabs(keff_new - keff); keff = keff_new;
        } else dk = 0.0;
        const double dphi = std::sqrt(dsq) / std::sqrt(nsq), norm = std::sqrt(nsq);
        if (!std::isfinite(keff) || !std::isfinite(dphi)) { rc = fail(NF_ERR_NUMERIC, "adjoint iteration diverged (outer %d: k=%g dphi=%g)", it, keff, dphi); break; }
        int mode = 0; double a = 0.0, b = 0.0;
        if (!use_direct_keff && it >= 5) {                                     // :1990-1992
            if (cheb_it == nmax) cheb_it = 0;
            if (cheb_it == 0) mode = 1; else if (cheb_it == 1) { mode = 2; a = ca[1]; } else { mode = 3; a = (4. / sigma) * ca[cheb_it]; b = cbv[cheb_it]; }
            ++cheb_it;
        }
        for (int i = 0; i < ns && rc == NF_OK; ++i) {
            nf_solver *X = T->slabs[i]; const long NT = X->nphi * ng;
            if (mode && !X->d_p0) { if ((rc = dalloc(&X->d_p0, (size_t)NT)) != NF_OK || (rc = dalloc(&X->d_p1, (size_t)NT)) != NF_OK) break; }
            hipLaunchKernelGGL(k_normalize_cheb, dim3(gT[i]), dim3(256), 0, st, X->d_raw, X->d_phi_adj, X->d_p0, X->d_p1, NT, norm, norm > 1e-14 ? 1 : 0, mode, a, b);
            if (mode == 3) std::swap(X->d_p0, X->d_p1);
        }
        if (rc != NF_OK) break;
        T->hist_k.push_back(keff); T->hist_dk.push_back(dk); T->hist_dphi.push_back(dphi);
        T->last_outer = it + 1;
        bool conv = dphi < o->tol_flux;
        if (!use_direct_keff) conv = conv && dk < o->tol_keff;
        if (conv) break;
    }
    if (rc == NF_OK && normalize_to_direct && T->has_valid_keff) {            // <phi, phi+> = 1 with vol * w = detJ * C-hat_pp (:2020-2066)
        std::vector<DevTmp<double>> mass(ns), one(ns);
        for (int i = 0; i < ns && rc == NF_OK; ++i) {
            nf_solver *X = T->slabs[i];
            rc = dalloc(&mass[i].p, (size_t)X->nphi); if (rc == NF_OK) rc = dalloc(&one[i].p, (size_t)X->N);
            if (rc != NF_OK) break;
            ChatArgs ch; ch.nloc = X->nloc;
            for (int p = 0; p < X->nloc; ++p) { int q = p; double c = 1.0; for (int t = 0; t < X->dim; ++t) { c *= 2.0 / (2.0 * (q % X->n1) + 1.0); q /= X->n1; } ch.c[p] = c; }
            hipLaunchKernelGGL(k_fill_const, dim3(grid_for(X->N)), dim3(256), 0, st, one[i].p, X->N, 1.0);
            hipLaunchKernelGGL(k_cell_coef, dim3(grid_for(X->N, 256, 65535)), dim3(256), 0, st, one[i].p, mass[i].p, X->d_hx, X->d_hy, X->d_hz, X->nx, X->ny, X->N, 2, X->dim, ch);
            hipLaunchKernelGGL(k_dot3, dim3(gT[i]), dim3(256), 0, st, X->d_phi, X->d_phi_adj, mass[i].p, X->nphi, ng, T->d_partials + i * T->slab_cap);
        }
        if (rc == NF_OK) rc = team_finalize(T, FIN_SUM, gT, 1, T->d_out, 0.0, 0);
        double ip = 0.0;
        if (rc == NF_OK && (hipMemcpyAsync(&ip, T->d_out, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
            rc = fail(NF_ERR_HIP, "adjoint: read-back failed");
        if (rc == NF_OK && std::fabs(ip) > 1e-14)
            for (int i = 0; i < ns; ++i) hipLaunchKernelGGL(k_scale, dim3(gT[i]), dim3(256), 0, st, T->slabs[i]->d_phi_adj, T->slabs[i]->nphi * ng, ip);
        (void)hipStreamSynchronize(st);
    }
    (void)hipStreamSynchronize(st);
    if (rc != NF_OK) return rc;
    HIPCHK(hipGetLastError());
    for (auto *X : T->slabs) { X->raw_valid = false; X->has_valid_adjoint = 1; X->last_keff_adj = keff; }
    if (keff_adj) *keff_adj = keff;
    if (n_outer) *n_outer = T->last_outer;
    return NF_OK;
}
int nf_get_phi_adj(nf_handle S, double *phi_host)
{
    if (!S || !phi_host) return fail(NF_ERR_ARG, "nf_get_phi_adj: bad arguments");
    if (!S->d_phi_adj) { for (long i = 0; i < S->nphi * S->ng; ++i) phi_host[i] = 1.0; return NF_OK; }   // Sol_Phi_adj_ = 1 (:226-235)
    return phi_transfer(S, phi_host, false, S->d_phi_adj);
}

int nf_get_history(nf_handle S, double *k, double *dk, double *dphi, int *cg, int cap)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    nf_team *T = S->team;
    const int n = std::min<int>(cap, T->last_outer);
    for (int i = 0; i < n; ++i) {
        if (k) k[i] = T->hist_k[i];
        if (dk) dk[i] = T->hist_dk[i];
        if (dphi) dphi[i] = T->hist_dphi[i];
        if (cg) for (int g = 0; g < S->ng; ++g) cg[i * S->ng + g] = T->hist_cg[i * S->ng + g];
    }
    return NF_OK;
}

int nf_profile_get(nf_handle S, const char *name, long *count, double *total_ms)
{
    if (!S || !name) return fail(NF_ERR_ARG, "nf_profile_get: bad arguments");
    auto it = S->team->prof.find(name);
    long c = 0, sk = 0; double ms = 0.0;
    if (it != S->team->prof.end()) it->second.stats(&c, &ms, &sk);
    if (count) *count = c;
    if (total_ms) *total_ms = ms;
    return NF_OK;
}
int nf_profile_reset(nf_handle S) { if (!S) return fail(NF_ERR_ARG, "null handle"); S->team->prof.clear(); return NF_OK; }

int nf_time_schur_apply(nf_handle S, int g, int reps, double *avg_ms)
{
    if (!S || g < 0 || g >= S->ng || reps < 1 || !avg_ms) return fail(NF_ERR_ARG, "nf_time_schur_apply: bad arguments");
    nf_team *T = S->team;
    for (auto *X : T->slabs) if (!X->built) return fail(NF_ERR_STATE, "nf_time_schur_apply: call nf_build first");
    HIPCHK(hipSetDevice(S->device));
    NFCHK(team_prepare(T));
    std::vector<const double *> xs; std::vector<double *> ys;
    for (auto *X : T->slabs) {
        hipLaunchKernelGGL(k_fill_pattern, dim3(grid_for(X->nphi)), dim3(256), 0, T->stream, X->d_p, X->nphi);
        xs.push_back(X->d_p); ys.push_back(X->d_q);
    }
    NFCHK(team_schur_apply(T, g, xs, ys, false, nullptr, nullptr));   // warm-up
    T->profile = true; const int every = T->prof_every; T->prof_every = 1;
    for (int i = 0; i < std::min(reps, 8); ++i) NFCHK(team_schur_apply(T, g, xs, ys, false, nullptr, nullptr));
    HIPCHK(hipStreamSynchronize(T->stream));
    prof_collect(T); T->profile = false; T->prof_every = every;
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, T->stream));
    for (int i = 0; i < reps; ++i) NFCHK(team_schur_apply(T, g, xs, ys, false, nullptr, nullptr));
    HIPCHK(hipEventRecord(b, T->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *avg_ms = ms / reps;
    return NF_OK;
}

// LocalMatrices::Compute on the device, literally (nf_assembly.h): A_loc, B_loc, C_loc of `n_elems` elements of group g by tensor
// quadrature, one element per workgroup.  variant 0 = plain fp64 FMA, 1 = v_mfma_f64_16x16x4_f64.  reps > 1 repeats the launch for timing.
int nf_local_matrices(nf_handle S, int g, int n_elems, const int *elems_host, double *A_host, double *B_host, double *C_host, int variant, int reps, double *avg_ms)
{
    if (!S || g < 0 || g >= S->ng || n_elems < 1 || !elems_host || !A_host || !B_host || !C_host || variant < 0 || variant > 1 || reps < 1)
        return fail(NF_ERR_ARG, "nf_local_matrices: bad arguments");
    if (!S->xs_uploaded) return fail(NF_ERR_STATE, "nf_local_matrices: call nf_upload_xs first");
    for (int i = 0; i < n_elems; ++i) if (elems_host[i] < 0 || elems_host[i] >= S->N) return fail(NF_ERR_ARG, "nf_local_matrices: element %d out of range", elems_host[i]);
    HIPCHK(hipSetDevice(S->device));
    nf_team *T = S->team; hipStream_t st = T->stream;
    AsmArgs P; memset(&P, 0, sizeof P);
    P.dim = S->dim; P.k = S->k; P.m = S->m;
    // Gauss rule of order 2 max(k, m) + 3 (src/NeutFEM.cpp:276): 3 points for RT0, 5 for RT1, and 7 -> the 5-point fallback for RT2
    // (include/FEM.hpp:115-120)
    if (2 * std::max(S->k, S->m) + 3 == 3) {
        P.nq = 3; const double a = std::sqrt(0.6);
        P.qp[0] = -a; P.qp[1] = 0.0; P.qp[2] = a; P.qw[0] = 5.0 / 9.0; P.qw[1] = 8.0 / 9.0; P.qw[2] = 5.0 / 9.0;
    } else {
        P.nq = 5;
        const double p5[5] = { -0.906179845938664, -0.538469310105683, 0.0, 0.538469310105683, 0.906179845938664 };
        const double w5[5] = { 0.236926885056189, 0.478628670499366, 0.568888888888889, 0.478628670499366, 0.236926885056189 };
        for (int i = 0; i < 5; ++i) { P.qp[i] = p5[i]; P.qw[i] = w5[i]; }
    }
    int nf = 1, ni = S->k; for (int t = 1; t < S->dim; ++t) { nf *= S->k + 1; ni *= S->k + 1; }
    const int nJ = S->dim * (2 * nf + ni); int nP = 1; for (int t = 0; t < S->dim; ++t) nP *= S->m + 1;
    P.nx = S->nx; P.ny = S->ny; P.nz = S->nz; P.hx = S->d_hx; P.hy = S->d_hy; P.hz = S->d_hz;
    P.D = S->d_D + (size_t)g * S->N; P.Sig = S->d_SigR + (size_t)g * S->N; P.n_elems = n_elems;
    DevTmp<double> dA, dB, dC; DevTmp<int> dEl; int *&d_el = dEl.p;
    HIPCHK(hipMalloc((void **)&dA.p, (size_t)n_elems * nJ * nJ * sizeof(double))); HIPCHK(hipMalloc((void **)&dB.p, (size_t)n_elems * nP * nJ * sizeof(double)));
    HIPCHK(hipMalloc((void **)&dC.p, (size_t)n_elems * nP * nP * sizeof(double)));
    HIPCHK(hipMalloc((void **)&d_el, (size_t)n_elems * sizeof(int)));
    int rc = NF_OK;
    if (hipMemcpy(d_el, elems_host, (size_t)n_elems * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) rc = fail(NF_ERR_HIP, "nf_local_matrices: upload failed");
    P.elems = d_el; P.A = dA.p; P.B = dB.p; P.C = dC.p;
    const size_t lds = (size_t)(18 + 6 + 6 + 12 + 12 + 128 + 2 * 128 * 48 + 128 * 32) * sizeof(double);
    const void *fn = variant ? (const void *)k_local_matrices<true> : (const void *)k_local_matrices<false>;
    if (rc == NF_OK && !lds_opt_in(fn, lds)) rc = fail(NF_ERR_UNSUPPORTED, "nf_local_matrices needs %zu bytes of LDS per workgroup", lds);
    hipEvent_t a = nullptr, b = nullptr;
    if (rc == NF_OK) {
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        for (int r = 0; r <= reps; ++r) {                         // one warm-up launch, then `reps` timed ones
            if (r == 1) (void)hipEventRecord(a, st);
            if (variant) hipLaunchKernelGGL(k_local_matrices<true>, dim3((unsigned)n_elems), dim3(256), lds, st, P);
            else hipLaunchKernelGGL(k_local_matrices<false>, dim3((unsigned)n_elems), dim3(256), lds, st, P);
        }
        (void)hipEventRecord(b, st);
        if (hipEventSynchronize(b) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(NF_ERR_HIP, "nf_local_matrices: kernel failed");
        float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b);
        if (avg_ms) *avg_ms = ms / reps;
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    }
    if (rc == NF_OK && (hipMemcpy(A_host, dA.p, (size_t)n_elems * nJ * nJ * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(B_host, dB.p, (size_t)n_elems * nP * nJ * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(C_host, dC.p, (size_t)n_elems * nP * nP * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(NF_ERR_HIP, "nf_local_matrices: download failed");
    return rc;
}

int nf_time_device_copy(nf_handle S, size_t bytes, int reps, double *gbps)
{
    if (!S || bytes < 16 || reps < 1 || !gbps) return fail(NF_ERR_ARG, "nf_time_device_copy: bad arguments");
    HIPCHK(hipSetDevice(S->device));
    hipStream_t st = S->team->stream;
    const long n2 = (long)(bytes / 16);
    nf_d2 *a = nullptr, *b = nullptr;
    NFCHK(dalloc(&a, (size_t)n2)); if (dalloc(&b, (size_t)n2) != NF_OK) { dfree(a); return NF_ERR_HIP; }
    (void)hipMemsetAsync(a, 0, (size_t)n2 * 16, st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0.0; hipError_t e = hipSuccess;
    int ncu = 256; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, S->device);
    const bool verbose = getenv("NEUTFEM_COPY_VERBOSE") != nullptr;
    // best of: plain / non-temporal copy kernels with 4 (or 8) 16-byte loads in flight per lane at 16 ... 256 blocks per CU or one chunk
    // per block, and the runtime's own device-to-device copy (measured on one box, GB/s: 16 per CU 5424 / 5249 nt, 64 per CU 5622 / 5886 nt,
    // 8 in flight 5344 - 5638, hipMemcpyDtoD 4778; round 3's kernel with 8-byte halves: 4790)
    struct V { int kind, per_cu; };                               // kind 0 plain x4, 1 nt x4, 2 nt x8, 3 hipMemcpy ; per_cu 0 = one chunk per block
    const V vs[] = { {0, 16}, {0, 64}, {0, 256}, {0, 0}, {1, 16}, {1, 64}, {1, 256}, {1, 0}, {2, 64}, {2, 0}, {3, 0} };
    const int NV = (int)(sizeof vs / sizeof vs[0]);
    for (int variant = 0; variant < NV && e == hipSuccess; ++variant) {
        const V v = vs[variant];
        const int U = v.kind == 2 ? 8 : 4;
        const long full = (n2 + 256L * U - 1) / (256L * U);
        const int grid = (int)std::min<long>(v.per_cu ? (long)ncu * v.per_cu : full, full);
        auto once = [&]() {
            if (v.kind == 3) { (void)hipMemcpyAsync(b, a, (size_t)n2 * 16, hipMemcpyDeviceToDevice, st); return; }
            if (v.kind == 0) hipLaunchKernelGGL((k_copy<false, 4>), dim3(grid), dim3(256), 0, st, (const nf_d2 *)a, b, n2);
            else if (v.kind == 1) hipLaunchKernelGGL((k_copy<true, 4>), dim3(grid), dim3(256), 0, st, (const nf_d2 *)a, b, n2);
            else hipLaunchKernelGGL((k_copy<true, 8>), dim3(grid), dim3(256), 0, st, (const nf_d2 *)a, b, n2);
        };
        once();
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) once();
        (void)hipEventRecord(e1, st);
        e = hipEventSynchronize(e1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        const double gb = ms > 0.f ? 2.0 * (double)n2 * 16.0 * reps / (ms * 1e-3) / 1e9 : 0.0;   // read + write
        if (verbose) fprintf(stderr, "[copy] variant %2d (%s, grid %d): %.0f GB/s\n", variant,
                             v.kind == 3 ? "hipMemcpyDtoD" : v.kind == 0 ? "plain, 4 in flight" : v.kind == 1 ? "nt, 4 in flight" : "nt, 8 in flight", v.kind == 3 ? 0 : grid, gb);
        if (e == hipSuccess) best = std::max(best, gb);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    dfree(a); dfree(b);
    if (e != hipSuccess) return fail(NF_ERR_HIP, "nf_time_device_copy: %s", hipGetErrorString(e));
    *gbps = best;
    return NF_OK;
}

int nf_set_option(nf_handle S, const char *key, long value)
{
    if (!S || !key) return fail(NF_ERR_ARG, "nf_set_option: bad arguments");
    nf_team *T = S->team;
    // tuning overrides of the y/z line kernel: 0 = automatic; the partial-sum buffer is sized for >= 8 columns per block
    // (slab_partial_need) and the kernel is instantiated for these segment lengths only
    if (!strcmp(key, "s_tx")) {
        if (value != 0 && value != 8 && value != 16 && value != 32 && value != 64) return fail(NF_ERR_ARG, "nf_set_option: s_tx must be 0 (auto), 8, 16, 32 or 64");
        T->opt_s_tx = (int)value;
        if (T->nproc > 1) T->linked_ready = false;               // the partial counts agreed for the vector reduce depend on the tile shape: agree again
    } else if (!strcmp(key, "s_seg")) {
        if (value != 0 && value != 4 && value != 8 && value != 16 && value != 32) return fail(NF_ERR_ARG, "nf_set_option: s_seg must be 0 (auto), 4, 8, 16 or 32");
        T->opt_s_seg = (int)value;
        if (T->nproc > 1) T->linked_ready = false;
    }
    else if (!strcmp(key, "s_pair")) { /* retired: the two-columns-per-thread variant lost to occupancy */ }
    else if (!strcmp(key, "s_wsmin")) T->opt_wsmin = (int)std::max(0L, std::min(100000L, value));
    else if (!strcmp(key, "vec_reduce")) T->opt_vec_reduce = value != 0;
    else if (!strcmp(key, "cg_single_reduce")) { T->opt_cg1 = value < 0 ? -1 : (value != 0); T->linked_ready = false; }
    else if (!strcmp(key, "cg_single_reduce_max_cells")) { T->cg1_max_cells = std::max(0L, value); T->linked_ready = false; }
    else if (!strcmp(key, "endpoint_weights")) { T->opt_endpoint_w = value != 0; T->linked_ready = false; }
    else if (!strcmp(key, "xchg_comm")) { T->opt_xchg_comm = value != 0; if (T->nproc > 1) T->linked_ready = false; }
    else if (!strcmp(key, "xy_overlap")) T->opt_xy_overlap = value != 0;
    else if (!strcmp(key, "xy_overlap_max_cells")) T->xy_overlap_max_cells = std::max(0L, value);
    else if (!strcmp(key, "split_dot")) T->opt_split_dot = (int)std::max(0L, std::min(2L, value));   // 0 never, 1 where a chunked pass runs, 2 always (big undivided RT0-P0 meshes)
    else if (!strcmp(key, "x_two_phase")) T->opt_x_p2 = value < 0 ? -1 : (value > 0 ? 1 : 0);
    else if (!strcmp(key, "s_long")) T->opt_s_long = value < 0 ? -1 : (value > 0 ? 1 : 0);
    else if (!strcmp(key, "s_long_dirs")) T->opt_s_long_dirs = (int)(value & 3);
    else if (!strcmp(key, "s_long_min_y")) T->s_long_min_y = (int)std::max(1L, std::min(1000000L, value));
    else if (!strcmp(key, "s_long_min")) T->s_long_min = (int)std::max(1L, std::min(1000000L, value));
    else if (!strcmp(key, "cg_batch")) T->cg_batch = (int)value;
    else if (!strcmp(key, "cg_fuse")) T->opt_fuse = value != 0;
    else if (!strcmp(key, "xcd")) T->opt_xcd = value < 0 ? -1 : (int)(value & 7);
    else if (!strcmp(key, "host_pub")) T->opt_pub = value != 0;
    else if (!strcmp(key, "prof_every")) { if (value < 1) return fail(NF_ERR_ARG, "prof_every must be >= 1"); T->prof_every = (int)value; }
    else if (!strcmp(key, "nt_loads")) T->opt_nt_loads = value != 0;
    else if (!strcmp(key, "nt_min_cells")) T->nt_min_cells = value;
    else if (!strcmp(key, "outer_dev")) T->opt_outer_dev = value != 0;
    else if (!strcmp(key, "cg_lean")) T->opt_lean = value != 0;
    else if (!strcmp(key, "sep_fold")) T->opt_sepfold = value != 0;
    else if (!strcmp(key, "cg_lean_max_cells")) T->lean_max_cells = value;
    else if (!strcmp(key, "resident")) T->opt_resident = value != 0;
    else if (!strcmp(key, "resident_lds")) T->opt_resident_lds = value != 0;
    else if (!strcmp(key, "resident_serial")) T->opt_resident_serial = value != 0;
    else if (!strcmp(key, "resident_two_sided")) T->opt_resident_two_sided = value != 0;
    else if (!strcmp(key, "resident_max_dofs")) { T->resident_max_dofs = value; T->resident_serial_max_dofs = std::min<long>(value, 5120); }   // one knob caps both variants
    else if (!strcmp(key, "resident_serial_max_dofs")) T->resident_serial_max_dofs = value;
    else if (!strcmp(key, "direct_max_dofs")) T->direct_max_dofs = std::max(0L, std::min(8192L, value));
    else if (!strcmp(key, "cg_fuse3")) T->opt_fuse3 = value != 0;
    else if (!strcmp(key, "cg_xcd")) T->opt_cgx = value != 0;
    else if (!strcmp(key, "keff_xcd")) T->opt_keffx = value != 0;
    else if (!strcmp(key, "cg_xcd_min_cells")) T->xcd_min_cells = value;
    else if (!strcmp(key, "cg_xcd_max_cells")) T->xcd_max_cells = value;
    else if (!strcmp(key, "cg_xcd_id")) T->xcd_id = (int)(value & 15);   // 8..15: no such XCD -- nobody registers, the solve falls back (tests)
    else if (!strcmp(key, "cg_xcd_groups")) T->xcd_groups = (int)std::max<long>(1, std::min<long>(value, 64));
    else if (!strcmp(key, "cg_fuse3_max_cells")) T->fuse3_max_cells = value;
    else if (!strcmp(key, "cg_lean_grid")) T->opt_lean_grid = (int)std::max(1L, std::min(1024L, value));
    else return fail(NF_ERR_ARG, "nf_set_option: unknown key %s", key);
    return NF_OK;
}

// ---- raw memory helpers ------------------------------------------------------------------------
int nf_mem_info(int device, size_t *free_bytes, size_t *total_bytes)
{
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return fail(NF_ERR_ARG, "nf_mem_info: device %d out of range", device); }
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return NF_OK;
}
int nf_dev_alloc(nf_handle S, size_t bytes, void **p)
{
    if (!S || !p) return fail(NF_ERR_ARG, "nf_dev_alloc: bad arguments");
    HIPCHK(hipSetDevice(S->device)); HIPCHK(hipMalloc(p, std::max<size_t>(bytes, 8))); return NF_OK;
}
int nf_dev_free(nf_handle S, void *p) { if (!S) return fail(NF_ERR_ARG, "null handle"); HIPCHK(hipSetDevice(S->device)); if (p) HIPCHK(hipFree(p)); return NF_OK; }
int nf_memcpy_h2d(nf_handle S, void *dst, const void *src, size_t bytes)
{
    if (!S || !dst || !src) return fail(NF_ERR_ARG, "nf_memcpy_h2d: bad arguments");
    HIPCHK(hipSetDevice(S->device)); HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return NF_OK;
}
int nf_memcpy_d2h(nf_handle S, void *dst, const void *src, size_t bytes)
{
    if (!S || !dst || !src) return fail(NF_ERR_ARG, "nf_memcpy_d2h: bad arguments");
    HIPCHK(hipSetDevice(S->device)); HIPCHK(hipStreamSynchronize(S->team->stream)); HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return NF_OK;
}
int nf_synchronize(nf_handle S) { if (!S) return fail(NF_ERR_ARG, "null handle"); HIPCHK(hipSetDevice(S->device)); HIPCHK(hipStreamSynchronize(S->team->stream)); HIPCHK(hipDeviceSynchronize()); return NF_OK; }
void *nf_stream(nf_handle S) { return S ? (void *)S->team->stream : nullptr; }

// diagnostic (not part of the C ABI): the partial / stamp buffer of k_cg_xcd (doubles 384.. hold s_memrealtime sums in -DNF_XSTAMPS builds)
extern "C" int nf_debug_xcd_buffer(void *h, double *out, int n, int coarse)
{
    nf_solver *S = (nf_solver *)h; if (!S || !S->team) return NF_ERR_ARG;
    nf_team *T = S->team;
    (void)coarse;
    if (!T->d_xpart) { for (int i = 0; i < n; ++i) out[i] = 0.0; return NF_OK; }
    HIPCHK(hipMemcpy(out, T->d_xpart, sizeof(double) * std::min(n, 512), hipMemcpyDeviceToHost));
    return NF_OK;
}
