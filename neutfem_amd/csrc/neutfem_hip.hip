// neutfem_hip.hip -- C-ABI implementation (include/neutfem_hip.h) over the gfx950 kernels.
//
// Host logic mirrors the reference's control flow (NeutFEM::SolveKeff / SolveCoarse /
// SchurSolver::Solve) while every vector stays resident in HBM; per outer iteration the host
// reads back 4 doubles, per CG solve one small struct every few iterations.
#include "../../include/neutfem_hip.h"
#include "nf_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
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

struct ProfSlot { long count = 0; double ms = 0.0; };

struct nf_solver {
    int device = 0;
    hipStream_t stream = nullptr;
    // mesh
    int dim = 1, nx = 1, ny = 1, nz = 1, ng = 1, k = 0, m = 0;
    long N = 0, nJ = 0, nJx = 0, nJy = 0, nJz = 0;
    std::vector<double> xb, yb, zb, hx, hy, hz;
    double *d_hx = nullptr, *d_hy = nullptr, *d_hz = nullptr, *d_xb = nullptr, *d_yb = nullptr, *d_zb = nullptr;
    int bc_set[8] = {0}, bc_type[8] = {0};
    // XS on device (reference layouts)
    double *d_D = nullptr, *d_SigR = nullptr, *d_NSF = nullptr, *d_Chi = nullptr;
    std::vector<double *> d_SigS;          // ng*ng blocks [g_to*ng+g_from], nullptr when all |s| <= 1e-14
    bool xs_uploaded = false, built = false, diag_valid = false;
    // operators
    double *d_Cd = nullptr, *d_Mf = nullptr;            // ng*N
    std::vector<double *> d_Ms;                          // ng*ng
    double *d_L[3] = {nullptr, nullptr, nullptr}, *d_DR[3] = {nullptr, nullptr, nullptr}, *d_D0[3] = {nullptr, nullptr, nullptr};
    long nlines[3] = {0, 0, 0};
    double *d_Sinv = nullptr;
    // state
    double *d_phi = nullptr, *d_raw = nullptr;          // current iterate / raw group solutions, ng*N
    double *d_p0 = nullptr, *d_p1 = nullptr;            // Chebyshev history
    double *d_tf = nullptr, *d_rhs = nullptr, *d_r = nullptr, *d_p = nullptr, *d_q = nullptr;
    double *d_partials = nullptr; long partial_stride = 0;
    CgScalars *d_cg = nullptr;
    double *d_out = nullptr;                            // 4 doubles
    bool raw_valid = false, raw_is_diag = false;
    int has_valid_keff = 0; double last_keff = 1.0;
    // stats
    int last_outer = 0, coarse_outer = 0; long last_cg_total = 0;
    std::vector<double> hist_k, hist_dk, hist_dphi; std::vector<int> hist_cg;
    std::vector<int> last_its;
    // profiling
    bool profile = false;
    std::map<std::string, ProfSlot> prof;
    struct Ev { hipEvent_t a, b; int slot; };
    std::vector<Ev> ev_pending; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
    int cg_batch = 0;
    int opt_s_tx = 0, opt_s_seg = 0, opt_s_pair = 0;      // tuning overrides (nf_set_option)
};

const char *nf_last_error(void) { return g_err.c_str(); }

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

static const char *SLOT_NAMES[4] = { "schur_x", "schur_y", "schur_z", "schur_apply" };

static Geom make_geom(const nf_solver *S)
{
    Geom G; G.dim = S->dim; G.nx = S->nx; G.ny = S->ny; G.nz = S->nz; G.hx = S->d_hx; G.hy = S->d_hy; G.hz = S->d_hz;
    const double p2 = (double)(1 << (S->dim - 1));
    G.cLL = p2 * 2.0 / 3.0; G.cLR = p2 / 3.0; G.beta = p2;
    // GetBoundaryAttribute, src/NeutFEM.cpp:2338-2347
    for (int d = 0; d < 3; ++d) {
        int lo, hi;
        if (S->dim == 1) { lo = 1; hi = 2; }
        else if (S->dim == 2) { if (d == 0) { lo = 1; hi = 2; } else { lo = 4; hi = 3; } }
        else { if (d == 0) { lo = 3; hi = 4; } else if (d == 1) { lo = 6; hi = 5; } else { lo = 1; hi = 2; } }
        G.dir_lo[d] = S->bc_set[lo] && S->bc_type[lo] == NF_BC_DIRICHLET;
        G.dir_hi[d] = S->bc_set[hi] && S->bc_type[hi] == NF_BC_DIRICHLET;
    }
    return G;
}

int nf_create(int rt_order, int p_order, int ng, int nxb, const double *xb, int nyb, const double *yb, int nzb,
              const double *zb, int device, nf_handle *out)
{
    if (!out || !xb || nxb < 2 || ng < 1 || ng > 64) return fail(NF_ERR_ARG, "nf_create: bad arguments");
    int k = std::min(rt_order, 2), m = std::min(p_order, 2);
    if (k < m) m = k;                                            // src/NeutFEM.cpp:149-169
    if (k != 0 || m != 0)
        return fail(NF_ERR_UNSUPPORTED, "nf_create: RT%d-P%d is not implemented on the HIP path yet (RT0-P0 only)", k, m);
    int ndev = nf_device_count();
    if (ndev <= 0) return fail(NF_ERR_NO_DEVICE, "nf_create: no HIP device visible (the gfx950 path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(NF_ERR_ARG, "nf_create: device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    nf_solver *S = new nf_solver();
    S->device = device; S->k = k; S->m = m; S->ng = ng;
    S->xb.assign(xb, xb + nxb);
    if (nyb > 1) S->yb.assign(yb, yb + nyb); else S->yb.assign(1, nyb == 1 && yb ? yb[0] : 0.0);
    if (nzb > 1) S->zb.assign(zb, zb + nzb); else S->zb.assign(1, nzb == 1 && zb ? zb[0] : 0.0);
    S->nx = nxb - 1; S->ny = nyb > 1 ? nyb - 1 : 1; S->nz = nzb > 1 ? nzb - 1 : 1;
    S->dim = S->nz > 1 ? 3 : (S->ny > 1 ? 2 : 1);               // src/FEM.cpp:33-35
    S->N = (long)S->nx * S->ny * S->nz;
    S->hx.resize(S->nx); S->hy.assign(S->ny, 1.0); S->hz.assign(S->nz, 1.0);
    for (int i = 0; i < S->nx; ++i) S->hx[i] = xb[i + 1] - xb[i];
    if (S->dim >= 2) for (int i = 0; i < S->ny; ++i) S->hy[i] = yb[i + 1] - yb[i];
    if (S->dim == 3) for (int i = 0; i < S->nz; ++i) S->hz[i] = zb[i + 1] - zb[i];
    S->nJx = (long)(S->nx + 1) * S->ny * S->nz;
    S->nJy = S->dim >= 2 ? (long)S->nx * (S->ny + 1) * S->nz : 0;
    S->nJz = S->dim == 3 ? (long)S->nx * S->ny * (S->nz + 1) : 0;
    S->nJ = S->nJx + S->nJy + S->nJz;
    S->nlines[0] = (long)S->ny * S->nz; S->nlines[1] = (long)S->nx * S->nz; S->nlines[2] = (long)S->nx * S->ny;
    const char *cb = getenv("NEUTFEM_CG_BATCH"); S->cg_batch = cb ? atoi(cb) : 0;
    *out = S;
    int rc = NF_OK;
    auto up = [&](double **d, const std::vector<double> &h) {
        if (rc != NF_OK) return;
        rc = dalloc(d, h.size());
        if (rc == NF_OK && hipMemcpy(*d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(NF_ERR_HIP, "nf_create: upload failed");
    };
    if (hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(NF_ERR_HIP, "hipStreamCreate failed");
    up(&S->d_hx, S->hx); up(&S->d_hy, S->hy); up(&S->d_hz, S->hz); up(&S->d_xb, S->xb); up(&S->d_yb, S->yb); up(&S->d_zb, S->zb);
    const size_t NN = (size_t)S->N * ng;
    if (rc == NF_OK) rc = dalloc(&S->d_phi, NN);
    if (rc == NF_OK) rc = dalloc(&S->d_raw, NN);
    if (rc == NF_OK) rc = dalloc(&S->d_tf, S->N);
    if (rc == NF_OK) rc = dalloc(&S->d_rhs, S->N);
    if (rc == NF_OK) rc = dalloc(&S->d_r, S->N);
    if (rc == NF_OK) rc = dalloc(&S->d_p, S->N);
    if (rc == NF_OK) rc = dalloc(&S->d_q, S->N);
    // partial sums: the Schur passes write one partial per block
    long maxblocks = std::max<long>(RED_GRID, std::max(S->nlines[0], (long)(((S->nx + 7) / 8)) * std::max(S->ny, S->nz)) + 16);
    S->partial_stride = maxblocks;
    if (rc == NF_OK) rc = dalloc(&S->d_partials, (size_t)maxblocks * 3);
    if (rc == NF_OK) rc = dalloc(&S->d_cg, 1);
    if (rc == NF_OK) rc = dalloc(&S->d_out, 4);
    if (rc != NF_OK) { nf_destroy(S); *out = nullptr; return rc; }
    S->d_SigS.assign((size_t)ng * ng, nullptr); S->d_Ms.assign((size_t)ng * ng, nullptr);
    S->last_its.assign(ng, 0);
    return nf_reset_flux(S);
}

int nf_destroy(nf_handle S)
{
    if (!S) return NF_OK;
    (void)hipSetDevice(S->device);
    if (S->stream) (void)hipStreamSynchronize(S->stream);
    dfree(S->d_hx); dfree(S->d_hy); dfree(S->d_hz); dfree(S->d_xb); dfree(S->d_yb); dfree(S->d_zb);
    dfree(S->d_D); dfree(S->d_SigR); dfree(S->d_NSF); dfree(S->d_Chi);
    for (auto &p : S->d_SigS) dfree(p);
    for (auto &p : S->d_Ms) dfree(p);
    dfree(S->d_Cd); dfree(S->d_Mf); dfree(S->d_Sinv);
    for (int d = 0; d < 3; ++d) { dfree(S->d_L[d]); dfree(S->d_DR[d]); dfree(S->d_D0[d]); }
    dfree(S->d_phi); dfree(S->d_raw); dfree(S->d_p0); dfree(S->d_p1);
    dfree(S->d_tf); dfree(S->d_rhs); dfree(S->d_r); dfree(S->d_p); dfree(S->d_q);
    dfree(S->d_partials); dfree(S->d_cg); dfree(S->d_out);
    for (auto &e : S->ev_pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : S->ev_free) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (S->stream) (void)hipStreamDestroy(S->stream);
    delete S;
    return NF_OK;
}

long nf_info(nf_handle S, const char *key)
{
    if (!S || !key) return -1;
#define K(s, v) if (!strcmp(key, s)) return (long)(v)
    K("dim", S->dim); K("nx", S->nx); K("ny", S->ny); K("nz", S->nz); K("ne", S->N); K("ng", S->ng);
    K("n_phi", S->N); K("n_J", S->nJ); K("n_loc", 1); K("last_outer", S->last_outer);
    K("last_cg_total", S->last_cg_total); K("coarse_outer", S->coarse_outer); K("device", S->device);
#undef K
    return -1;
}

int nf_set_bc(nf_handle S, int attr, int bc_type)
{
    if (!S || attr < 0 || attr >= 8) return fail(NF_ERR_ARG, "nf_set_bc: bad attribute %d", attr);
    S->bc_set[attr] = 1; S->bc_type[attr] = bc_type;
    return NF_OK;
}

int nf_upload_xs(nf_handle S, const double *D, const double *SigR, const double *NSF, const double *Chi, const double *SigS)
{
    if (!S || !D || !SigR || !NSF || !Chi || !SigS) return fail(NF_ERR_ARG, "nf_upload_xs: null pointer");
    HIPCHK(hipSetDevice(S->device));
    const size_t NN = (size_t)S->N * S->ng, B = NN * sizeof(double);
    NFCHK(dalloc(&S->d_D, NN)); NFCHK(dalloc(&S->d_SigR, NN)); NFCHK(dalloc(&S->d_NSF, NN)); NFCHK(dalloc(&S->d_Chi, NN));
    HIPCHK(hipMemcpyAsync(S->d_D, D, B, hipMemcpyHostToDevice, S->stream));
    HIPCHK(hipMemcpyAsync(S->d_SigR, SigR, B, hipMemcpyHostToDevice, S->stream));
    HIPCHK(hipMemcpyAsync(S->d_NSF, NSF, B, hipMemcpyHostToDevice, S->stream));
    HIPCHK(hipMemcpyAsync(S->d_Chi, Chi, B, hipMemcpyHostToDevice, S->stream));
    const int ng = S->ng;
    for (int i = 0; i < ng * ng; ++i) {
        const double *blk = SigS + (size_t)i * S->N;
        bool nz = false;                                          // src/NeutFEM.cpp:1265 : |sigs| > 1e-14
        for (long e = 0; e < S->N; ++e) if (std::fabs(blk[e]) > 1e-14) { nz = true; break; }
        if (!nz) { dfree(S->d_SigS[i]); continue; }
        NFCHK(dalloc(&S->d_SigS[i], S->N));
        HIPCHK(hipMemcpyAsync(S->d_SigS[i], blk, S->N * sizeof(double), hipMemcpyHostToDevice, S->stream));
    }
    HIPCHK(hipStreamSynchronize(S->stream));
    S->xs_uploaded = true; S->built = false;
    return NF_OK;
}

static int grid_for(long n, int block = 256, int cap = RED_GRID)
{
    long g = (n + block - 1) / block; if (g < 1) g = 1; if (g > cap) g = cap; return (int)g;
}

int nf_build(nf_handle S)
{
    if (!S) return fail(NF_ERR_ARG, "nf_build: null handle");
    if (!S->xs_uploaded) return fail(NF_ERR_STATE, "nf_build: call nf_upload_xs first");
    HIPCHK(hipSetDevice(S->device));
    const int ng = S->ng; const long N = S->N; const size_t NN = (size_t)N * ng;
    NFCHK(dalloc(&S->d_Cd, NN)); NFCHK(dalloc(&S->d_Mf, NN));
    for (int d = 0; d < S->dim; ++d) {
        NFCHK(dalloc(&S->d_L[d], NN)); NFCHK(dalloc(&S->d_DR[d], NN)); NFCHK(dalloc(&S->d_D0[d], (size_t)S->nlines[d] * ng));
    }
    Geom G = make_geom(S);
    const int gN = grid_for(N, 256, 65535);
    for (int g = 0; g < ng; ++g) {
        hipLaunchKernelGGL(k_cell_coef, dim3(gN), dim3(256), 0, S->stream, S->d_SigR + g * N, S->d_Cd + g * N, S->d_hx, S->d_hy, S->d_hz, S->nx, S->ny, N, 0);
        hipLaunchKernelGGL(k_cell_coef, dim3(gN), dim3(256), 0, S->stream, S->d_NSF + g * N, S->d_Mf + g * N, S->d_hx, S->d_hy, S->d_hz, S->nx, S->ny, N, 1);
        for (int gp = 0; gp < ng; ++gp) {
            const int i = g * ng + gp;
            if (!S->d_SigS[i]) { dfree(S->d_Ms[i]); continue; }
            NFCHK(dalloc(&S->d_Ms[i], N));
            hipLaunchKernelGGL(k_cell_coef, dim3(gN), dim3(256), 0, S->stream, S->d_SigS[i], S->d_Ms[i], S->d_hx, S->d_hy, S->d_hz, S->nx, S->ny, N, 1);
        }
        for (int d = 0; d < S->dim; ++d) {
            const long nl = S->nlines[d];
            hipLaunchKernelGGL(k_factor_lines, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, S->stream, G, d, S->d_D + g * N,
                               S->d_L[d] + g * N, S->d_DR[d] + g * N, S->d_D0[d] + g * nl, nl);
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(S->stream));
    S->built = true; S->diag_valid = false;                       // src/NeutFEM.cpp:454-456
    return NF_OK;
}

// ---- profiling helpers -----------------------------------------------------------------------
static void prof_begin(nf_solver *S, int slot, hipEvent_t *a, hipEvent_t *b)
{
    if (S->ev_free.empty()) {
        hipEvent_t x, y; (void)hipEventCreate(&x); (void)hipEventCreate(&y); S->ev_free.push_back({x, y});
    }
    auto pr = S->ev_free.back(); S->ev_free.pop_back();
    *a = pr.first; *b = pr.second;
    (void)hipEventRecord(*a, S->stream);
    S->ev_pending.push_back({*a, *b, slot});
}
static void prof_collect(nf_solver *S)
{
    for (auto &e : S->ev_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            auto &sl = S->prof[SLOT_NAMES[e.slot]]; sl.count += 1; sl.ms += ms;
        }
        S->ev_free.push_back({e.a, e.b});
    }
    S->ev_pending.clear();
}

// ---- Schur apply -----------------------------------------------------------------------------
template <int K, int NCH>
static void launch_x_t(nf_solver *S, int g, const double *x, double *y, int lpl_log2, int first, int last,
                       double *partials, const CgScalars *cg, unsigned grid)
{
    const long N = S->N; const double beta = (double)(1 << (S->dim - 1));
    const bool vec = (S->nx % 2 == 0);
    if (vec)
        hipLaunchKernelGGL((k_schur_x<K, NCH, true>), dim3(grid), dim3(256), 0, S->stream, x, y, S->d_L[0] + g * N, S->d_DR[0] + g * N,
                           S->d_D0[0] + g * S->nlines[0], S->d_Cd + g * N, S->nx, S->nlines[0], lpl_log2, beta, first, last, partials, cg);
    else
        hipLaunchKernelGGL((k_schur_x<K, NCH, false>), dim3(grid), dim3(256), 0, S->stream, x, y, S->d_L[0] + g * N, S->d_DR[0] + g * N,
                           S->d_D0[0] + g * S->nlines[0], S->d_Cd + g * N, S->nx, S->nlines[0], lpl_log2, beta, first, last, partials, cg);
}

static int launch_x(nf_solver *S, int g, const double *x, double *y, int last, double *partials, const CgScalars *cg, int *nparts)
{
    const int K = 2;
    int lanes = (S->nx + K - 1) / K, lpl_log2 = 0;
    while ((1 << lpl_log2) < lanes && lpl_log2 < 6) ++lpl_log2;
    const int LPL = 1 << lpl_log2, LPW = 64 / LPL;
    const int nch = (S->nx + LPL * K - 1) / (LPL * K);
    const unsigned grid = (unsigned)((S->nlines[0] + 4 * LPW - 1) / (4 * LPW));
    if (nparts) *nparts = (int)grid;
    if (nch <= 1) launch_x_t<2, 1>(S, g, x, y, lpl_log2, 1, last, partials, cg, grid);
    else if (nch <= 2) launch_x_t<2, 2>(S, g, x, y, lpl_log2, 1, last, partials, cg, grid);
    else if (nch <= 4) launch_x_t<2, 4>(S, g, x, y, lpl_log2, 1, last, partials, cg, grid);
    else if (nch <= 8) launch_x_t<2, 8>(S, g, x, y, lpl_log2, 1, last, partials, cg, grid);
    else return fail(NF_ERR_UNSUPPORTED, "nx = %d exceeds the x-line kernel limit (1024 cells)", S->nx);
    return NF_OK;
}

static int launch_s(nf_solver *S, int d, int g, const double *x, double *y, int last, double *partials, const CgScalars *cg, int *nparts)
{
    const long N = S->N; const double beta = (double)(1 << (S->dim - 1));
    const int n = d == 1 ? S->ny : S->nz;
    const long nxy = (long)S->nx * S->ny;
    const long sl = d == 1 ? S->nx : nxy, ostride = d == 1 ? nxy : S->nx;
    const int nouter = d == 1 ? S->nz : S->ny;
    const bool pair = S->opt_s_pair && (S->nx % 2 == 0);          // two columns per thread, double2 accesses
    int SEG = S->opt_s_seg ? S->opt_s_seg : (pair ? (n <= 256 ? 4 : 8) : (n <= 512 ? 8 : (n <= 1024 ? 16 : 32)));
    int NSEG = (n + SEG - 1) / SEG;
    if (NSEG > 128) return fail(NF_ERR_UNSUPPORTED, "line length %d exceeds the segmented kernel limit", n);
    const int cols = pair ? (S->nx + 1) / 2 : S->nx;              // thread columns needed
    int TX = S->opt_s_tx ? S->opt_s_tx : 64;
    while (TX > 8 && TX * NSEG > 1024) TX >>= 1;
    if (TX * NSEG > 1024) return fail(NF_ERR_UNSUPPORTED, "line length %d needs more than 1024 threads per block", n);
    while (TX > 8 && TX / 2 >= cols) TX >>= 1;                    // narrow meshes
    dim3 grid((unsigned)((cols + TX - 1) / TX), (unsigned)nouter), block((unsigned)(TX * NSEG));
    if (nparts) *nparts = (int)(grid.x * grid.y);
    const double *L = S->d_L[d] + g * N, *DR = S->d_DR[d] + g * N, *D0 = S->d_D0[d] + g * S->nlines[d];
    if (pair) {
        const size_t lds = (size_t)(4 * TX * NSEG + TX) * sizeof(double2) + 16 * sizeof(double);
#define NF_LAUNCH_S2(SEGV, DIRV) hipLaunchKernelGGL((k_schur_s2<SEGV, DIRV>), grid, block, lds, S->stream, x, y, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, beta, last, partials, cg)
        if (d == 1) { if (SEG == 4) NF_LAUNCH_S2(4, 1); else if (SEG == 8) NF_LAUNCH_S2(8, 1); else if (SEG == 16) NF_LAUNCH_S2(16, 1); else return fail(NF_ERR_ARG, "bad s_seg"); }
        else        { if (SEG == 4) NF_LAUNCH_S2(4, 2); else if (SEG == 8) NF_LAUNCH_S2(8, 2); else if (SEG == 16) NF_LAUNCH_S2(16, 2); else return fail(NF_ERR_ARG, "bad s_seg"); }
#undef NF_LAUNCH_S2
    } else {
        const size_t lds = (size_t)(4 * TX * NSEG + TX + 16) * sizeof(double);
#define NF_LAUNCH_S(SEGV, DIRV) hipLaunchKernelGGL((k_schur_s<SEGV, DIRV>), grid, block, lds, S->stream, x, y, L, DR, D0, n, sl, ostride, S->nx, TX, NSEG, beta, last, partials, cg)
        if (d == 1) { if (SEG == 4) NF_LAUNCH_S(4, 1); else if (SEG == 8) NF_LAUNCH_S(8, 1); else if (SEG == 16) NF_LAUNCH_S(16, 1); else if (SEG == 32) NF_LAUNCH_S(32, 1); else return fail(NF_ERR_ARG, "bad s_seg"); }
        else        { if (SEG == 4) NF_LAUNCH_S(4, 2); else if (SEG == 8) NF_LAUNCH_S(8, 2); else if (SEG == 16) NF_LAUNCH_S(16, 2); else if (SEG == 32) NF_LAUNCH_S(32, 2); else return fail(NF_ERR_ARG, "bad s_seg"); }
#undef NF_LAUNCH_S
    }
    return NF_OK;
}

// y = S_g x ; if partials != NULL the last pass leaves *nparts block partials of x.y there
static int schur_apply(nf_solver *S, int g, const double *x, double *y, double *partials, const CgScalars *cg, int *nparts)
{
    hipEvent_t a, b, ta = nullptr, tb = nullptr;
    if (S->profile) prof_begin(S, 3, &ta, &tb);
    for (int d = 0; d < S->dim; ++d) {
        const int last = d == S->dim - 1;
        if (S->profile) prof_begin(S, d, &a, &b);
        if (d == 0) NFCHK(launch_x(S, g, x, y, last, partials, cg, last ? nparts : nullptr));
        else NFCHK(launch_s(S, d, g, x, y, last, partials, cg, last ? nparts : nullptr));
        if (S->profile) (void)hipEventRecord(b, S->stream);
    }
    if (S->profile) (void)hipEventRecord(tb, S->stream);
    return NF_OK;
}

int nf_schur_apply(nf_handle S, int g, const double *x_dev, double *y_dev)
{
    if (!S || !x_dev || !y_dev || g < 0 || g >= S->ng) return fail(NF_ERR_ARG, "nf_schur_apply: bad arguments");
    if (!S->built) return fail(NF_ERR_STATE, "nf_schur_apply: call nf_build first");
    HIPCHK(hipSetDevice(S->device));
    NFCHK(schur_apply(S, g, x_dev, y_dev, nullptr, nullptr, nullptr));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(S->stream));
    if (S->profile) prof_collect(S);
    return NF_OK;
}

// ---- CG (SchurSolver::SolveSchurImplicit, src/solvers.cpp:577-636) -----------------------------
static int cg_solve(nf_solver *S, int g, const double *rhs, double *x, double tol, int maxit, int *its_out, double *res_out)
{
    const long N = S->N; const int G = grid_for(N);
    hipLaunchKernelGGL(k_cg_init, dim3(G), dim3(256), 0, S->stream, rhs, x, S->d_r, S->d_p, N, S->d_partials);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, S->stream, (int)FIN_RHS, S->d_partials, G, S->partial_stride, 1, S->d_cg, S->d_out, tol, maxit);
    CgScalars sc; memset(&sc, 0, sizeof sc);
    int launched = 0;
    int batch = S->cg_batch > 0 ? S->cg_batch : std::max(1, S->last_its[g] - 1);
    while (launched < maxit) {
        int nb = std::min(batch, maxit - launched);
        for (int i = 0; i < nb; ++i) {
            int nparts = 0;
            NFCHK(schur_apply(S, g, S->d_p, S->d_q, S->d_partials, S->d_cg, &nparts));
            hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, S->stream, (int)FIN_PAP, S->d_partials, nparts, S->partial_stride, 1, S->d_cg, S->d_out, 0.0, 0);
            hipLaunchKernelGGL(k_cg_update, dim3(G), dim3(256), 0, S->stream, x, S->d_r, S->d_p, S->d_q, N, S->d_cg, S->d_partials);
            hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, S->stream, (int)FIN_RR, S->d_partials, G, S->partial_stride, 1, S->d_cg, S->d_out, 0.0, 0);
            hipLaunchKernelGGL(k_cg_pupdate, dim3(G), dim3(256), 0, S->stream, S->d_p, S->d_r, N, S->d_cg);
        }
        launched += nb;
        HIPCHK(hipMemcpyAsync(&sc, S->d_cg, sizeof sc, hipMemcpyDeviceToHost, S->stream));
        HIPCHK(hipStreamSynchronize(S->stream));
        if (sc.done) break;
        batch = S->cg_batch > 0 ? S->cg_batch : (launched < 8 ? 1 : 2);
    }
    if (launched == 0) {
        HIPCHK(hipMemcpyAsync(&sc, S->d_cg, sizeof sc, hipMemcpyDeviceToHost, S->stream));
        HIPCHK(hipStreamSynchronize(S->stream));
    }
    HIPCHK(hipGetLastError());
    if (S->profile) prof_collect(S);
    if (!std::isfinite(sc.rr)) return fail(NF_ERR_NUMERIC, "CG produced a non-finite residual (group %d)", g);
    S->last_its[g] = sc.its;
    if (its_out) *its_out = sc.its;
    if (res_out) *res_out = sc.rhs_norm > 0 ? std::sqrt(sc.rr) / sc.rhs_norm : 0.0;
    return NF_OK;
}

int nf_solve_group(nf_handle S, int g, const double *rhs_dev, double *phi_dev, double tol, int maxit, int *its, double *res)
{
    if (!S || !rhs_dev || !phi_dev || g < 0 || g >= S->ng) return fail(NF_ERR_ARG, "nf_solve_group: bad arguments");
    if (!S->built) return fail(NF_ERR_STATE, "nf_solve_group: call nf_build first");
    HIPCHK(hipSetDevice(S->device));
    return cg_solve(S, g, rhs_dev, phi_dev, tol, maxit, its, res);
}

// ---- diagonal cache ----------------------------------------------------------------------------
int nf_build_diagonal_cache(nf_handle S)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    if (!S->built) return fail(NF_ERR_STATE, "nf_build_diagonal_cache: call nf_build first");
    if (S->diag_valid) return NF_OK;
    HIPCHK(hipSetDevice(S->device));
    const long N = S->N;
    NFCHK(dalloc(&S->d_Sinv, (size_t)N * S->ng));
    Geom G = make_geom(S);
    for (int g = 0; g < S->ng; ++g)
        hipLaunchKernelGGL(k_diag_cache, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, S->stream, G, S->d_D + g * N, S->d_Cd + g * N, S->d_Sinv + g * N, N);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(S->stream));
    S->diag_valid = true;
    return NF_OK;
}
int nf_get_diagonal_cache(nf_handle S, int g, double *sinv_host)
{
    if (!S || g < 0 || g >= S->ng || !sinv_host) return fail(NF_ERR_ARG, "nf_get_diagonal_cache: bad arguments");
    NFCHK(nf_build_diagonal_cache(S));
    HIPCHK(hipMemcpy(sinv_host, S->d_Sinv + g * S->N, S->N * sizeof(double), hipMemcpyDeviceToHost));
    return NF_OK;
}

// ---- state -------------------------------------------------------------------------------------
int nf_set_phi(nf_handle S, const double *phi)
{
    if (!S || !phi) return fail(NF_ERR_ARG, "nf_set_phi: bad arguments");
    HIPCHK(hipSetDevice(S->device));
    HIPCHK(hipMemcpy(S->d_phi, phi, (size_t)S->N * S->ng * sizeof(double), hipMemcpyHostToDevice));
    return NF_OK;
}
int nf_get_phi(nf_handle S, double *phi)
{
    if (!S || !phi) return fail(NF_ERR_ARG, "nf_get_phi: bad arguments");
    HIPCHK(hipSetDevice(S->device));
    HIPCHK(hipStreamSynchronize(S->stream));
    HIPCHK(hipMemcpy(phi, S->d_phi, (size_t)S->N * S->ng * sizeof(double), hipMemcpyDeviceToHost));
    return NF_OK;
}
int nf_reset_flux(nf_handle S)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(S->device));
    std::vector<double> ones((size_t)S->N * S->ng, 1.0);        // Sol_Phi_ = 1, src/NeutFEM.cpp:347-354
    HIPCHK(hipMemcpy(S->d_phi, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
    S->has_valid_keff = 0; S->raw_valid = false;
    return NF_OK;
}
int nf_set_warm_state(nf_handle S, int v, double k) { if (!S) return fail(NF_ERR_ARG, "null handle"); S->has_valid_keff = v; S->last_keff = k; return NF_OK; }
int nf_get_warm_state(nf_handle S, int *v, double *k) { if (!S) return fail(NF_ERR_ARG, "null handle"); if (v) *v = S->has_valid_keff; if (k) *k = S->last_keff; return NF_OK; }

int nf_get_J(nf_handle S, double *J_host)
{
    if (!S || !J_host) return fail(NF_ERR_ARG, "nf_get_J: bad arguments");
    HIPCHK(hipSetDevice(S->device));
    const long N = S->N, nJ = S->nJ;
    if (!S->raw_valid) { memset(J_host, 0, sizeof(double) * nJ * S->ng); return NF_OK; }   // Sol_J_ = 0 before any solve
    double *dJ = nullptr; NFCHK(dalloc(&dJ, (size_t)nJ));
    Geom G = make_geom(S);
    const long off[3] = { 0, S->nJx, S->nJx + S->nJy };
    for (int g = 0; g < S->ng; ++g) {
        for (int d = 0; d < S->dim; ++d)
            hipLaunchKernelGGL(k_flux_to_J, dim3((unsigned)((S->nlines[d] + 63) / 64)), dim3(64), 0, S->stream, G, d, S->d_D + g * N,
                               S->d_raw + g * N, S->d_L[d] + g * N, S->d_DR[d] + g * N, S->d_D0[d] + g * S->nlines[d], dJ + off[d],
                               S->nlines[d], S->raw_is_diag ? 1 : 0);
        HIPCHK(hipMemcpyAsync(J_host + (size_t)g * nJ, dJ, nJ * sizeof(double), hipMemcpyDeviceToHost, S->stream));
        HIPCHK(hipStreamSynchronize(S->stream));
    }
    dfree(dJ);
    HIPCHK(hipGetLastError());
    return NF_OK;
}

// ---- SolveCoarse (src/NeutFEM.cpp:2380-2611) ---------------------------------------------------
static int solve_keff_impl(nf_solver *S, const nf_keff_opts *o, double *keff, int *n_outer);

// builds + solves the coarse problem on the device; the prolonged flux is written to d_dst (ng*N)
static int coarse_init(nf_solver *S, const nf_keff_opts *o, double *k_coarse, double *d_dst, bool *done)
{
    *done = false;
    const int dim = S->dim, ng = S->ng;
    const int nf = o->n_coarse_factors;
    int rx = nf > 0 ? std::max(o->coarse_factors[0], 1) : 1;
    int ry = (nf > 1 && dim >= 2) ? std::max(o->coarse_factors[1], 1) : 1;
    int rz = (nf > 2 && dim >= 3) ? std::max(o->coarse_factors[2], 1) : 1;
    if (S->nx % rx || S->ny % ry || S->nz % rz) return NF_OK;   // :2402-2407 -> (1.0, Sol_Phi_)
    const int nxc = S->nx / rx, nyc = S->ny / ry, nzc = S->nz / rz;
    std::vector<double> xc(nxc + 1), yc(dim >= 2 ? nyc + 1 : 1), zc(dim >= 3 ? nzc + 1 : 1);
    for (int i = 0; i <= nxc; ++i) xc[i] = S->xb[i * rx];
    if (dim >= 2) for (int j = 0; j <= nyc; ++j) yc[j] = S->yb[j * ry]; else yc[0] = 0.0;
    if (dim >= 3) for (int kk = 0; kk <= nzc; ++kk) zc[kk] = S->zb[kk * rz]; else zc[0] = 0.0;
    nf_handle C = nullptr;
    NFCHK(nf_create(0, 0, ng, nxc + 1, xc.data(), (int)yc.size(), yc.data(), (int)zc.size(), zc.data(), S->device, &C));
    for (int a = 0; a < 8; ++a) if (S->bc_set[a]) nf_set_bc(C, a, S->bc_type[a]);
    const long Nc = C->N; const size_t NNc = (size_t)Nc * ng;
    int rc = NF_OK;
    auto coarsen = [&](const double *fine, double **coarse, int nfields) -> int {
        NFCHK(dalloc(coarse, (size_t)Nc * nfields));
        hipLaunchKernelGGL(k_coarsen, dim3((unsigned)((Nc + 127) / 128)), dim3(128), 0, S->stream, fine, *coarse, S->d_xb, S->d_yb, S->d_zb,
                           dim, S->nx, S->ny, S->nz, rx, ry, rz, nfields);
        return NF_OK;
    };
    (void)NNc;
    rc = coarsen(S->d_D, &C->d_D, ng);
    if (rc == NF_OK) rc = coarsen(S->d_SigR, &C->d_SigR, ng);
    if (rc == NF_OK) rc = coarsen(S->d_NSF, &C->d_NSF, ng);
    if (rc == NF_OK) rc = coarsen(S->d_Chi, &C->d_Chi, ng);
    for (int i = 0; i < ng * ng && rc == NF_OK; ++i)
        if (S->d_SigS[i]) rc = coarsen(S->d_SigS[i], &C->d_SigS[i], 1);    // mean of an all-zero block is zero
    if (rc == NF_OK && hipStreamSynchronize(S->stream) != hipSuccess) rc = fail(NF_ERR_HIP, "coarsen failed");
    if (rc == NF_OK) { C->xs_uploaded = true; rc = nf_build(C); }
    double kc = 1.0; int nout = 0;
    if (rc == NF_OK) {
        nf_keff_opts co = *o;                                    // :2460-2467
        co.tol_keff = o->tol_keff * 10.0; co.tol_flux = o->tol_flux * 10.0; co.max_outer = o->max_outer / 2;
        co.use_coarse_init = 0; co.n_coarse_factors = 0; co.use_diagonal_solver = 0; co.solver_type_pushed = 1; co.profile = 0;
        rc = solve_keff_impl(C, &co, &kc, &nout);
    }
    if (rc == NF_OK) {
        S->coarse_outer = nout;
        hipLaunchKernelGGL(k_prolong, dim3((unsigned)((S->N + 255) / 256)), dim3(256), 0, C->stream, C->d_phi, d_dst, S->nx, S->ny, S->nz, rx, ry, rz, ng);
        if (hipStreamSynchronize(C->stream) != hipSuccess) rc = fail(NF_ERR_HIP, "prolong failed");
    }
    std::string keep = g_err;
    nf_destroy(C);
    if (rc != NF_OK) { g_err = keep; return rc; }
    *k_coarse = kc; *done = true;
    return NF_OK;
}

int nf_solve_coarse(nf_handle S, const nf_keff_opts *o, double *k_coarse, double *phi_host)
{
    if (!S || !o || !k_coarse || !phi_host) return fail(NF_ERR_ARG, "nf_solve_coarse: bad arguments");
    if (!S->built) return fail(NF_ERR_STATE, "nf_solve_coarse: call nf_build first");
    HIPCHK(hipSetDevice(S->device));
    const size_t NN = (size_t)S->N * S->ng;
    bool done = false; double kc = 1.0;
    if (o->n_coarse_factors > 0) NFCHK(coarse_init(S, o, &kc, S->d_raw, &done));
    S->raw_valid = false;
    HIPCHK(hipMemcpy(phi_host, done ? S->d_raw : S->d_phi, NN * sizeof(double), hipMemcpyDeviceToHost));
    *k_coarse = done ? kc : 1.0;
    return NF_OK;
}

// ---- SolveKeff (src/NeutFEM.cpp:1627-1815) -----------------------------------------------------
static int solve_keff_impl(nf_solver *S, const nf_keff_opts *o, double *keff_out, int *n_outer)
{
    const int ng = S->ng; const long N = S->N; const long NT = N * ng;
    const int G = grid_for(N), GT = grid_for(NT);
    int use_diag = o->use_diagonal_solver ? 1 : 0;                // RT0-P0 only exists here
    if (use_diag) NFCHK(nf_build_diagonal_cache(S));
    double keff = S->has_valid_keff ? S->last_keff : 1.0;        // :1662
    S->coarse_outer = 0;
    if (o->use_coarse_init && o->n_coarse_factors > 0) {          // :1665-1670
        bool done = false; double kc = 1.0;
        NFCHK(coarse_init(S, o, &kc, S->d_phi, &done));
        keff = done ? kc : 1.0;
    }
    // SchurSolver type: DIRECT_* or n_phi < 200 -> "exact" solve (CG to 1e-14 stands in)
    const bool direct = !o->solver_type_pushed || o->solver_type <= 2 || N < 200;
    const double cg_tol = direct ? 1e-14 : o->tol_flux;           // SetTolerance forwards tol_flux (:334)
    const int cg_max = direct ? 100000 : o->max_inner;
    // ChebyshevAccel(15, 0.98), src/solvers.cpp:664-700
    const int nmax = 15; const double sigma = 0.98;
    double ca[16], cbv[16];
    { const double Gm = std::acosh(2. / sigma - 1.); ca[0] = cbv[0] = 0.; ca[1] = 2. / (2. - sigma); cbv[1] = 0.;
      for (int i = 2; i < nmax; ++i) { ca[i] = std::cosh((i - 1) * Gm) / std::cosh(i * Gm); cbv[i] = std::cosh((i - 2) * Gm) / std::cosh(i * Gm); } }
    int cheb_it = 0;
    S->hist_k.clear(); S->hist_dk.clear(); S->hist_dphi.clear(); S->hist_cg.clear();
    S->last_outer = 0; S->last_cg_total = 0;
    S->profile = o->profile != 0;
    ScatterArgs sa; sa.ng = ng;
    double hout[4];
    for (int it = 0; it < o->max_outer; ++it) {
        // total_fiss and prod_old (:1700-1707)
        hipLaunchKernelGGL(k_fission, dim3(G), dim3(256), 0, S->stream, S->d_Mf, S->d_phi, ng, N, S->d_tf, S->d_partials);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, S->stream, (int)FIN_SUM, S->d_partials, G, S->partial_stride, 1, S->d_cg, S->d_out, 0.0, 0);
        for (int g = 0; g < ng; ++g) {
            for (int gp = 0; gp < 64; ++gp) sa.M[gp] = gp < ng ? S->d_Ms[g * ng + gp] : nullptr;
            double *dst = use_diag ? S->d_raw + g * N : S->d_rhs;
            hipLaunchKernelGGL(k_group_rhs, dim3(G), dim3(256), 0, S->stream, sa, g, S->d_Chi + g * N, S->d_tf, 1.0 / keff, S->d_raw, S->d_phi,
                               use_diag ? S->d_Sinv + g * N : (const double *)nullptr, dst, N);
            int its = 0;
            if (!use_diag) NFCHK(cg_solve(S, g, S->d_rhs, S->d_raw + g * N, cg_tol, cg_max, &its, nullptr));
            S->hist_cg.push_back(its); S->last_cg_total += its;
        }
        // prod_new, norms (:1766-1779)
        hipLaunchKernelGGL(k_outer_reduce, dim3(GT), dim3(256), 0, S->stream, S->d_Mf, S->d_raw, S->d_phi, NT, S->d_partials, S->partial_stride);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, S->stream, (int)FIN_SUM, S->d_partials, GT, S->partial_stride, 3, S->d_cg, S->d_out + 1, 0.0, 0);
        HIPCHK(hipMemcpyAsync(hout, S->d_out, 4 * sizeof(double), hipMemcpyDeviceToHost, S->stream));
        HIPCHK(hipStreamSynchronize(S->stream));
        const double prod_old = hout[0], prod_new = hout[1], nsq = hout[2], dsq = hout[3];
        const double keff_new = keff * (prod_new / prod_old);
        const double dk = std::fabs(keff_new - keff);
        if (it >= 1) keff = keff_new;                             // :1774
        const double dphi = std::sqrt(dsq / nsq), norm = std::sqrt(nsq);
        if (!std::isfinite(keff_new) || !std::isfinite(dphi))
            return fail(NF_ERR_NUMERIC, "power iteration diverged (outer %d: k=%g dphi=%g)", it, keff_new, dphi);
        // normalise + Chebyshev (:1780-1788, src/solvers.cpp:720-756)
        int mode = 0; double a = 0.0, b = 0.0;
        if (it >= 2) {
            if (cheb_it == nmax) cheb_it = 0;
            if (cheb_it == 0) { mode = 1; }
            else if (cheb_it == 1) { mode = 2; a = ca[1]; }
            else { mode = 3; a = (4. / sigma) * ca[cheb_it]; b = cbv[cheb_it]; }
            if (!S->d_p0) { NFCHK(dalloc(&S->d_p0, (size_t)NT)); NFCHK(dalloc(&S->d_p1, (size_t)NT)); }
        }
        hipLaunchKernelGGL(k_normalize_cheb, dim3(GT), dim3(256), 0, S->stream, S->d_raw, S->d_phi, S->d_p0, S->d_p1, NT, norm,
                           norm > 1e-14 ? 1 : 0, mode, a, b);
        if (mode == 3) std::swap(S->d_p0, S->d_p1);
        if (it >= 2) ++cheb_it;
        S->hist_k.push_back(keff); S->hist_dk.push_back(dk); S->hist_dphi.push_back(dphi);
        S->last_outer = it + 1;
        if (dk < o->tol_keff && dphi < o->tol_flux) break;        // :1799-1802
    }
    HIPCHK(hipStreamSynchronize(S->stream));
    HIPCHK(hipGetLastError());
    if (S->profile) prof_collect(S);
    S->profile = false;
    S->raw_valid = S->last_outer > 0; S->raw_is_diag = use_diag != 0;
    S->has_valid_keff = 1; S->last_keff = keff;                   // :1808-1809
    if (keff_out) *keff_out = keff;
    if (n_outer) *n_outer = S->last_outer;
    return NF_OK;
}

int nf_solve_keff(nf_handle S, const nf_keff_opts *o, double *keff, int *n_outer)
{
    if (!S || !o) return fail(NF_ERR_ARG, "nf_solve_keff: bad arguments");
    if (!S->built) return fail(NF_ERR_STATE, "nf_solve_keff: call nf_build first");
    HIPCHK(hipSetDevice(S->device));
    return solve_keff_impl(S, o, keff, n_outer);
}

int nf_get_history(nf_handle S, double *k, double *dk, double *dphi, int *cg, int cap)
{
    if (!S) return fail(NF_ERR_ARG, "null handle");
    const int n = std::min<int>(cap, S->last_outer);
    for (int i = 0; i < n; ++i) {
        if (k) k[i] = S->hist_k[i];
        if (dk) dk[i] = S->hist_dk[i];
        if (dphi) dphi[i] = S->hist_dphi[i];
        if (cg) for (int g = 0; g < S->ng; ++g) cg[i * S->ng + g] = S->hist_cg[i * S->ng + g];
    }
    return NF_OK;
}

int nf_profile_get(nf_handle S, const char *name, long *count, double *total_ms)
{
    if (!S || !name) return fail(NF_ERR_ARG, "nf_profile_get: bad arguments");
    auto it = S->prof.find(name);
    if (count) *count = it == S->prof.end() ? 0 : it->second.count;
    if (total_ms) *total_ms = it == S->prof.end() ? 0.0 : it->second.ms;
    return NF_OK;
}
int nf_profile_reset(nf_handle S) { if (!S) return fail(NF_ERR_ARG, "null handle"); S->prof.clear(); return NF_OK; }

int nf_time_schur_apply(nf_handle S, int g, int reps, double *avg_ms)
{
    if (!S || g < 0 || g >= S->ng || reps < 1 || !avg_ms) return fail(NF_ERR_ARG, "nf_time_schur_apply: bad arguments");
    if (!S->built) return fail(NF_ERR_STATE, "nf_time_schur_apply: call nf_build first");
    HIPCHK(hipSetDevice(S->device));
    hipLaunchKernelGGL(k_fill_pattern, dim3(grid_for(S->N)), dim3(256), 0, S->stream, S->d_p, S->N);
    NFCHK(schur_apply(S, g, S->d_p, S->d_q, nullptr, nullptr, nullptr));   // warm-up
    S->profile = true;
    for (int i = 0; i < std::min(reps, 8); ++i) NFCHK(schur_apply(S, g, S->d_p, S->d_q, nullptr, nullptr, nullptr));
    HIPCHK(hipStreamSynchronize(S->stream));
    prof_collect(S); S->profile = false;
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, S->stream));
    for (int i = 0; i < reps; ++i) NFCHK(schur_apply(S, g, S->d_p, S->d_q, nullptr, nullptr, nullptr));
    HIPCHK(hipEventRecord(b, S->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *avg_ms = ms / reps;
    return NF_OK;
}

int nf_set_option(nf_handle S, const char *key, long value)
{
    if (!S || !key) return fail(NF_ERR_ARG, "nf_set_option: bad arguments");
    if (!strcmp(key, "s_tx")) S->opt_s_tx = (int)value;
    else if (!strcmp(key, "s_seg")) S->opt_s_seg = (int)value;
    else if (!strcmp(key, "s_pair")) S->opt_s_pair = (int)value;
    else if (!strcmp(key, "cg_batch")) S->cg_batch = (int)value;
    else return fail(NF_ERR_ARG, "nf_set_option: unknown key %s", key);
    return NF_OK;
}

// ---- raw memory helpers ------------------------------------------------------------------------
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
    HIPCHK(hipSetDevice(S->device)); HIPCHK(hipStreamSynchronize(S->stream)); HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return NF_OK;
}
int nf_synchronize(nf_handle S) { if (!S) return fail(NF_ERR_ARG, "null handle"); HIPCHK(hipSetDevice(S->device)); HIPCHK(hipStreamSynchronize(S->stream)); return NF_OK; }
void *nf_stream(nf_handle S) { return S ? (void *)S->stream : nullptr; }
