// host_module.cpp -- pybind11 module `neutfem._neutfem_eigen`: the reference's Python surface
// (src/wrapper.cpp:20-1066) re-exported name for name over the MI355X C ABI (include/neutfem_hip.h).
//
// The class keeps what the reference keeps on the host -- cross-section arrays, flux, tolerances,
// BC map, warm-start flags -- and hands the hot path (BuildMatrices, SolveKeff, SolveCoarse,
// build_diagonal_cache) to the HIP library.  There is NO CPU fallback: without a HIP device those
// methods raise RuntimeError.  Methods the reference binds but that lie outside the accelerated
// path (never-defined projections, reflectors) raise RuntimeError with that explanation.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/neutfem_hip.h"

namespace py = pybind11;

// include/NeutFEM.hpp:51-91, include/solvers.hpp:176-190
enum class BCType { DIRICHLET, NEUMANN, MIRROR, ROBIN, PERIODIC };
enum class VerbosityLevel { SILENT = 0, LIGHT = 1, NORMAL = 2, VERBOSE = 3, DEBUG = 4 };
enum class BoundaryID { LEFT_1D = 1, RIGHT_1D = 2, LEFT_2D = 1, RIGHT_2D = 2, TOP_2D = 3, BOTTOM_2D = 4,
                        BACK_3D = 1, FRONT_3D = 2, LEFT_3D = 3, RIGHT_3D = 4, TOP_3D = 5, BOTTOM_3D = 6 };
enum class LinearSolverType { DIRECT_LU, DIRECT_LDLT, DIRECT_LLT, CG, CG_DIAG, CG_ICHOL, BICGSTAB, BICGSTAB_DIAG, BICGSTAB_ILU, LCG };

using arr_t = py::array_t<double, py::array::c_style | py::array::forcecast>;

class NeutFEM {
public:
    NeutFEM(int order, int ng, arr_t xb, arr_t yb, arr_t zb) : NeutFEM(order, order, ng, xb, yb, zb) {}
    NeutFEM(int rt_order, int p_order, int ng, arr_t xb, arr_t yb, arr_t zb)
    {
        if (xb.ndim() != 1 || yb.ndim() != 1 || zb.ndim() != 1) throw std::runtime_error("breaks must be 1-D arrays");
        xb_.assign(xb.data(), xb.data() + xb.size()); yb_.assign(yb.data(), yb.data() + yb.size()); zb_.assign(zb.data(), zb.data() + zb.size());
        if (xb_.size() < 2) throw std::runtime_error("x_breaks needs at least 2 entries");
        nx_ = (int)xb_.size() - 1; ny_ = yb_.size() > 1 ? (int)yb_.size() - 1 : 1; nz_ = zb_.size() > 1 ? (int)zb_.size() - 1 : 1;
        dim_ = nz_ > 1 ? 3 : (ny_ > 1 ? 2 : 1);
        ng_ = ng; rt_ = std::min(rt_order, 2); p_ = std::min(p_order, 2);
        if (rt_ < p_) {                                           // src/NeutFEM.cpp:149-169
            Log(VerbosityLevel::NORMAL, "");
            Log(VerbosityLevel::NORMAL, "!!! ERREUR: RT" + std::to_string(rt_) + "-P" + std::to_string(p_) + " est instable !!!");
            Log(VerbosityLevel::NORMAL, "    Pour la stabilite inf-sup, il faut k_RT >= k_P");
            Log(VerbosityLevel::NORMAL, "    Combinaisons valides: RT0-P0, RT1-P0, RT1-P1, RT2-P0, RT2-P1, RT2-P2");
            Log(VerbosityLevel::NORMAL, "    Forçage à RT" + std::to_string(rt_) + "-P" + std::to_string(rt_));
            Log(VerbosityLevel::NORMAL, "");
            p_ = rt_;
        }
        const int k = rt_, m = p_;                                // src/FEM.cpp:177-259
        nloc_ = dim_ == 1 ? m + 1 : dim_ == 2 ? (m + 1) * (m + 1) : (m + 1) * (m + 1) * (m + 1);
        const int nf = dim_ == 1 ? 1 : dim_ == 2 ? k + 1 : (k + 1) * (k + 1);
        const int ni = dim_ == 1 ? k : dim_ == 2 ? k * (k + 1) : k * (k + 1) * (k + 1);
        ne_ = (long)nx_ * ny_ * nz_; nphi_ = ne_ * nloc_;
        long njx = (long)(nx_ + 1) * ny_ * nz_ * nf, njy = dim_ >= 2 ? (long)nx_ * (ny_ + 1) * nz_ * nf : 0,
             njz = dim_ == 3 ? (long)nx_ * ny_ * (nz_ + 1) * nf : 0;
        nJface_ = njx + njy + njz; nJ_ = nJface_ + ne_ * dim_ * ni;
        D_.assign(ng * ne_, 1.0); SRC_.assign(ng * ne_, 0.0); SigR_.assign(ng * ne_, 0.01); NSF_.assign(ng * ne_, 0.0);
        KSF_.assign(ng * ne_, 0.0); Chi_.assign(ng * ne_, 0.0); SigS_.assign((size_t)ng * ng * ne_, 0.0);
        if (ng > 0) std::fill(Chi_.begin(), Chi_.begin() + ne_, 1.0);
        Phi_.assign(ng * nphi_, 1.0); PhiAdj_.assign(ng * nphi_, 1.0);
        const std::string o = "RT" + std::to_string(rt_) + "-P" + std::to_string(p_);
        Log(VerbosityLevel::NORMAL, "========================================");
        Log(VerbosityLevel::NORMAL, "  NeutFEM - Solveur " + o + " (MI355X / HIP)");
        Log(VerbosityLevel::NORMAL, "========================================");
        Log(VerbosityLevel::NORMAL, "  Dimension     : " + std::to_string(dim_) + "D");
        Log(VerbosityLevel::NORMAL, "  Maillage      : " + std::to_string(nx_) + " x " + std::to_string(ny_) + " x " + std::to_string(nz_));
        Log(VerbosityLevel::NORMAL, "  Elements      : " + std::to_string(ne_));
        Log(VerbosityLevel::NORMAL, "  Groupes       : " + std::to_string(ng));
        Log(VerbosityLevel::NORMAL, "  DOFs flux     : " + std::to_string(nphi_) + " par groupe");
        Log(VerbosityLevel::NORMAL, "  DOFs courant  : " + std::to_string(nJ_) + " par groupe");
        Log(VerbosityLevel::NORMAL, "    - Face DOFs   : " + std::to_string(nJface_));
        Log(VerbosityLevel::NORMAL, "    - Interior DOFs: " + std::to_string(nJ_ - nJface_));
        Log(VerbosityLevel::NORMAL, "========================================\n");
    }
    ~NeutFEM() { if (h_) nf_destroy(h_); }
    NeutFEM(const NeutFEM &) = delete;

    // ---- configuration (src/NeutFEM.cpp:306-362) ------------------------------------------------
    void SetBC(int attr, BCType t, double v) { bc_types_[attr] = t; bc_values_[attr] = v; if (h_ && attr >= 0 && attr < 8) nf_set_bc(h_, attr, (int)t); }
    void SetRobin(int attr, double a, double b) { robin_[attr] = {a, b}; }
    void SetLinearSolver(LinearSolverType t) { solver_ = t; solver_pushed_ = true; }
    void SetTolerance(double tk, double tf, double tl, int mo, int mi) { tol_keff_ = tk; tol_flux_ = tf; tol_L2_ = tl; max_outer_ = mo; max_inner_ = mi; }
    void SetVerbosity(VerbosityLevel v) { verb_ = v; }
    void SetCMFDRelaxation(double w) { cmfd_omega_ = w; if (h_) nf_set_cmfd_relaxation(h_, w); }   // include/NeutFEM.hpp:232-235
    void InitializeCMFD() { need_built("initialize_cmfd"); chk(nf_initialize_cmfd(h_)); }                 // src/NeutFEM.cpp:662-704
    void ApplyQuarterSymmetry(int, int) { SetBC(1, BCType::MIRROR, 0.0); SetBC(4, BCType::MIRROR, 0.0); }   // :356-362
    int AddReflector(py::array_t<double>, py::array_t<double>, py::array_t<double>) { return 0; }           // :2614-2620
    void SetReflector(int, int, bool) {}
    void ClearReflectors() {}
    void ResetFlux()
    {
        std::fill(Phi_.begin(), Phi_.end(), 1.0); std::fill(PhiAdj_.begin(), PhiAdj_.end(), 1.0);
        has_valid_keff_ = false; has_valid_adjoint_ = false;     // src/NeutFEM.cpp:347-354
        if (h_) nf_reset_flux(h_);
    }
    std::string GetSolverName() const
    {
        static const char *n[] = { "SparseLU", "SimplicialLDLT", "SimplicialLLT", "CG", "CG + Diag", "CG + IChol", "BiCGSTAB", "BiCGSTAB + Diag", "BiCGSTAB + ILU", "LSCG" };
        int i = (int)solver_; return i >= 0 && i < 10 ? n[i] : "Unknown";
    }

    // ---- hot path ---------------------------------------------------------------------------------
    void BuildMatrices()
    {
        Log(VerbosityLevel::NORMAL, "Assemblage des matrices...");
        ensure_handle();
        for (auto &kv : bc_types_) if (kv.first >= 0 && kv.first < 8) chk(nf_set_bc(h_, kv.first, (int)kv.second));
        chk(nf_upload_xs(h_, D_.data(), SigR_.data(), NSF_.data(), Chi_.data(), SigS_.data()));
        chk(nf_build(h_));
        Log(VerbosityLevel::NORMAL, "  Assemblage termine (coefficients + factorisation par lignes sur GPU)");
    }
    nf_keff_opts make_opts(bool use_coarse, const std::vector<int> &f, bool use_diag) const
    {
        nf_keff_opts o{}; o.tol_keff = tol_keff_; o.tol_flux = tol_flux_; o.max_outer = max_outer_; o.max_inner = max_inner_;
        o.use_coarse_init = use_coarse && !f.empty();
        o.n_coarse_factors = (int)std::min<size_t>(f.size(), 3);
        for (int i = 0; i < o.n_coarse_factors; ++i) o.coarse_factors[i] = f[i];
        o.use_diagonal_solver = use_diag; o.solver_type = (int)solver_; o.solver_type_pushed = solver_pushed_; o.profile = 0;
        return o;
    }
    int printed_upto_ = 0;                                        // first outer whose progress line has not been printed yet (multiple of 5)
    static void progress_line(void *self, int it, double keff, double dk, double dphi)
    {
        NeutFEM *me = static_cast<NeutFEM *>(self);
        if (it % 5 != 0) return;
        std::cout << "  It " << std::setw(4) << it << " : k = " << std::fixed << std::setprecision(8) << keff << "  dk = " << std::scientific
                  << std::setprecision(2) << dk << "  dphi = " << dphi << std::defaultfloat << std::endl;
        me->printed_upto_ = it + 5;
    }
    double SolveKeff(bool use_coarse_init, const std::vector<int> &coarse_factors, bool use_diagonal_solver, bool use_cmfd)
    {
        need_built("SolveKeff");
        Log(VerbosityLevel::NORMAL, "\n=== CALCUL DE K-EFFECTIF (DIRECT) ===");
        if (use_diagonal_solver && !(rt_ == 0 && p_ == 0)) { Log(VerbosityLevel::NORMAL, "  Note: Solveur diagonal non disponible (ordre > 0)"); use_diagonal_solver = false; }
        if (use_diagonal_solver) Log(VerbosityLevel::NORMAL, "  Mode: Solveur diagonal RT0-P0 (faible RAM)");
        Log(VerbosityLevel::NORMAL, use_cmfd ? "  Acceleration: CMFD active" : "  Acceleration: Chebyshev");
        chk(nf_set_phi(h_, Phi_.data()));
        chk(nf_set_warm_state(h_, has_valid_keff_ ? 1 : 0, last_keff_));
        nf_keff_opts o = make_opts(use_coarse_init, coarse_factors, use_diagonal_solver);
        o.use_cmfd = use_cmfd ? 1 : 0;
        chk(nf_set_cmfd_relaxation(h_, cmfd_omega_));
        double k = 1.0; int nout = 0;
        // the reference prints its progress line every 5th outer DURING the solve (src/NeutFEM.cpp:1791-1796): the host-driven loop calls back after
        // every outer; the in-kernel paths (milliseconds) have no host in between, their lines follow from the recorded history
        printed_upto_ = 0;
        chk(nf_set_progress_callback(h_, verb_ >= VerbosityLevel::NORMAL ? &NeutFEM::progress_line : nullptr, this));
        const int rc_solve = nf_solve_keff(h_, &o, &k, &nout);
        (void)nf_set_progress_callback(h_, nullptr, nullptr);
        chk(rc_solve);
        chk(nf_get_phi(h_, Phi_.data()));
        if (verb_ >= VerbosityLevel::NORMAL) {
            std::vector<double> hk(nout), hdk(nout), hdp(nout);
            nf_get_history(h_, hk.data(), hdk.data(), hdp.data(), nullptr, nout);
            for (int it = printed_upto_; it < nout; it += 5)
                std::cout << "  It " << std::setw(4) << it << " : k = " << std::fixed << std::setprecision(8) << hk[it] << "  dk = " << std::scientific
                          << std::setprecision(2) << hdk[it] << "  dphi = " << hdp[it] << std::defaultfloat << std::endl;
            if (nout < max_outer_) std::cout << "  Convergence en " << nout << " iterations" << std::endl;
            std::cout << "  k-eff direct = " << std::fixed << std::setprecision(8) << k << std::defaultfloat << std::endl;
        }
        has_valid_keff_ = true; last_keff_ = k;
        return k;
    }
    std::pair<double, py::array_t<double>> SolveCoarse(const std::vector<int> &refine)
    {
        need_built("SolveCoarse");
        chk(nf_set_phi(h_, Phi_.data()));
        nf_keff_opts o = make_opts(true, refine, false);
        py::array_t<double> out((py::ssize_t)(ng_ * nphi_));
        double k = 1.0;
        chk(nf_solve_coarse(h_, &o, &k, out.mutable_data()));
        return {k, out};
    }
    double SolveAdjoint(bool normalize_to_direct, bool use_direct_keff)
    {
        need_built("SolveAdjoint");
        Log(VerbosityLevel::NORMAL, "\n=== CALCUL DE K-EFFECTIF (ADJOINT) ===");
        chk(nf_set_phi(h_, Phi_.data()));                         // the bi-orthonormalisation uses the direct flux
        chk(nf_set_warm_state(h_, has_valid_keff_ ? 1 : 0, last_keff_));
        nf_keff_opts o = make_opts(false, {}, false);
        double k = 1.0; int nout = 0;
        chk(nf_solve_adjoint(h_, &o, normalize_to_direct ? 1 : 0, use_direct_keff ? 1 : 0, &k, &nout));
        chk(nf_get_phi_adj(h_, PhiAdj_.data()));
        if (verb_ >= VerbosityLevel::NORMAL) std::cout << "  k-eff adjoint = " << std::fixed << std::setprecision(8) << k << std::defaultfloat << std::endl;
        last_keff_adj_ = k; has_valid_adjoint_ = true;            // src/NeutFEM.cpp:2075
        return k;
    }
    void BuildDiagonalCache() { need_built("build_diagonal_cache"); chk(nf_build_diagonal_cache(h_)); }
    py::array_t<double> GetCurrent()
    {
        need_built("get_current");
        py::array_t<double> out((py::ssize_t)(ng_ * nJ_));
        chk(nf_get_J(h_, out.mutable_data()));
        return out;
    }
    std::map<int, int> GetBCMap() const { std::map<int, int> r; for (auto &kv : bc_types_) r[kv.first] = (int)kv.second; return r; }

    // ExportVTK (src/NeutFEM.cpp:2137-2332): legacy ASCII STRUCTURED_GRID, cell data.  Same dataset layout and field
    // names as the reference so existing ParaView states keep working; written on the host from the host mirrors
    // (flux, XS) and, for the cell-centred currents, from Sol_J_ fetched through nf_get_J.
    void ExportVTK(const std::string &filename, bool export_flux, bool export_current, bool export_xs, bool export_adjoint)
    {
        const std::string full = filename + ".vtk";
        std::ofstream f(full);
        if (!f.is_open()) throw std::runtime_error("Cannot open file: " + full);
        Log(VerbosityLevel::NORMAL, "Export VTK vers " + full);
        const long nc = ne_;
        f << "# vtk DataFile Version 3.0\n" << "NeutFEM Output - k-eff=" << std::fixed << std::setprecision(6) << last_keff_ << "\n"
          << "ASCII\nDATASET STRUCTURED_GRID\n" << "DIMENSIONS " << nx_ + 1 << " " << ny_ + 1 << " " << nz_ + 1 << "\n"
          << "POINTS " << (long)(nx_ + 1) * (ny_ + 1) * (nz_ + 1) << " double\n";
        for (int k = 0; k <= nz_; ++k) for (int j = 0; j <= ny_; ++j) for (int i = 0; i <= nx_; ++i)
            f << xb_[i] << " " << (dim_ >= 2 ? yb_[j] : 0.0) << " " << (dim_ == 3 ? zb_[k] : 0.0) << "\n";
        f << "\nCELL_DATA " << nc << "\n";
        auto scalars = [&](const std::string &name, const std::function<double(long)> &val) {
            f << "SCALARS " << name << " double 1\nLOOKUP_TABLE default\n";
            for (long e = 0; e < nc; ++e) f << val(e) << "\n";
        };
        if (export_flux) {
            for (int g = 0; g < ng_; ++g) scalars("Flux_g" + std::to_string(g), [&](long e) { return Phi_[g * nphi_ + e * nloc_]; });
            scalars("Flux_total", [&](long e) { double t = 0; for (int g = 0; g < ng_; ++g) t += Phi_[g * nphi_ + e * nloc_]; return t; });
        }
        if (export_adjoint && has_valid_adjoint_)                 // src/NeutFEM.cpp:2206-2214: only after a SolveAdjoint
            for (int g = 0; g < ng_; ++g) scalars("Flux_adj_g" + std::to_string(g), [&](long e) { return PhiAdj_[g * nphi_ + e * nloc_]; });
        if (export_current) {
            need_built("ExportVTK(export_current=True)");
            std::vector<double> J((size_t)ng_ * nJ_);
            chk(nf_get_J(h_, J.data()));
            int nf = 1; for (int t = 1; t < dim_; ++t) nf *= rt_ + 1;          // first DOF of every face (JxFaceIndex(..., local_dof = 0))
            const long njx = (long)(nx_ + 1) * ny_ * nz_ * nf, njy = dim_ >= 2 ? (long)nx_ * (ny_ + 1) * nz_ * nf : 0;
            for (int g = 0; g < ng_; ++g) {
                const double *Jg = J.data() + (size_t)g * nJ_;
                f << "VECTORS Current_g" << g << " double\n";
                for (int k = 0; k < nz_; ++k) for (int j = 0; j < ny_; ++j) for (int i = 0; i < nx_; ++i) {
                    const long fx = ((long)k * ny_ + j) * (nx_ + 1) + i;             // src/FEM.cpp:264-275
                    double jx = 0.5 * (Jg[fx * nf] + Jg[(fx + 1) * nf]), jy = 0.0, jz = 0.0;
                    if (dim_ >= 2) { const long fy = ((long)k * (ny_ + 1) + j) * nx_ + i; jy = 0.5 * (Jg[njx + fy * nf] + Jg[njx + (fy + nx_) * nf]); }
                    if (dim_ == 3) { const long fz = ((long)k * ny_ + j) * nx_ + i; jz = 0.5 * (Jg[njx + njy + fz * nf] + Jg[njx + njy + (fz + (long)nx_ * ny_) * nf]); }
                    f << jx << " " << jy << " " << jz << "\n";
                }
            }
        }
        if (export_xs) {
            const std::pair<const char *, const std::vector<double> *> fields[] = {
                {"D_g", &D_}, {"SigmaR_g", &SigR_}, {"NuSigF_g", &NSF_}, {"Chi_g", &Chi_}, {"KappaSigF_g", &KSF_}, {"Source_g", &SRC_}};
            for (auto &fd : fields)
                for (int g = 0; g < ng_; ++g) scalars(fd.first + std::to_string(g), [&](long e) { return (*fd.second)[g * nc + e]; });
            for (int gf = 0; gf < ng_; ++gf) for (int gt = 0; gt < ng_; ++gt)
                scalars("SigS_" + std::to_string(gf) + "_to_" + std::to_string(gt), [&](long e) { return SigS_[((size_t)gt * ng_ + gf) * nc + e]; });
        }
        f.close();
        Log(VerbosityLevel::NORMAL, "  Export termine: " + std::to_string(nc) + " cellules");
    }
    [[noreturn]] void oos(const char *what, const char *ref) const
    {
        throw std::runtime_error(std::string(what) + " is outside the accelerated hot path of neutfem_amd (reference: " + ref + ")");
    }

    // ---- numpy views (src/NeutFEM.cpp:2626-2730) ---------------------------------------------------
    py::array_t<double> view(std::vector<double> &v, bool sigs = false)
    {
        std::vector<py::ssize_t> shape; shape.push_back(ng_); if (sigs) shape.push_back(ng_);
        if (dim_ >= 3) shape.push_back(nz_);
        if (dim_ >= 2) shape.push_back(ny_);
        shape.push_back(nx_);
        std::vector<py::ssize_t> strides(shape.size()); py::ssize_t s = sizeof(double);
        for (int i = (int)shape.size() - 1; i >= 0; --i) { strides[i] = s; s *= shape[i]; }
        return py::array_t<double>(shape, strides, v.data(), py::cast(this));
    }
    py::array_t<double> flux(std::vector<double> &src, std::vector<double> &p0)
    {
        if (nloc_ == 1) return view(src);
        p0.resize(ng_ * ne_);
        for (int g = 0; g < ng_; ++g) for (long e = 0; e < ne_; ++e) p0[g * ne_ + e] = src[g * nphi_ + e * nloc_];
        return view(p0);
    }

    int dim_, nx_, ny_, nz_, ng_, rt_, p_, nloc_;
    long ne_, nphi_, nJ_, nJface_;
    std::vector<double> xb_, yb_, zb_, D_, SRC_, SigR_, NSF_, KSF_, Chi_, SigS_, Phi_, PhiAdj_, fluxP0_, fluxAdjP0_;
    double last_keff_ = 1.0, last_keff_adj_ = 1.0;

private:
    void Log(VerbosityLevel lvl, const std::string &s) const { if (verb_ >= lvl) std::cout << s << std::endl; }
    void chk(int rc) const { if (rc != NF_OK) throw std::runtime_error(std::string("neutfem_amd: ") + nf_last_error()); }
    void ensure_handle()
    {
        if (h_) return;
        int dev = 0; if (const char *e = std::getenv("NEUTFEM_DEVICE")) dev = std::atoi(e);
        else if (const char *lr = std::getenv("LOCAL_RANK")) dev = std::atoi(lr) % std::max(1, nf_device_count());
        chk(nf_create(rt_, p_, ng_, (int)xb_.size(), xb_.data(), (int)yb_.size(), yb_.data(), (int)zb_.size(), zb_.data(), dev, &h_));
    }
    void need_built(const char *who) const { if (!h_) throw std::runtime_error(std::string(who) + ": call BuildMatrices() first"); }

    nf_handle h_ = nullptr;
    std::map<int, BCType> bc_types_; std::map<int, double> bc_values_; std::map<int, std::pair<double, double>> robin_;
    LinearSolverType solver_ = LinearSolverType::BICGSTAB; bool solver_pushed_ = false;   // src/NeutFEM.cpp:126 vs solvers.cpp:68
    double tol_keff_ = 1e-5, tol_flux_ = 1e-5, tol_L2_ = 1e-5; int max_outer_ = 200, max_inner_ = 1000;
    VerbosityLevel verb_ = VerbosityLevel::NORMAL; double cmfd_omega_ = 1.0;
    bool has_valid_keff_ = false, has_valid_adjoint_ = false;
};

PYBIND11_MODULE(_neutfem_eigen, m)
{
    m.doc() = "neutfem_amd: MI355X-native drop-in for jujuC31/NeutFEM's neutfem._neutfem_eigen (src/wrapper.cpp)";
    m.attr("__backend__") = "hip-gfx950";
    m.def("device_count", &nf_device_count, "number of visible HIP devices");

    py::enum_<VerbosityLevel>(m, "VerbosityLevel")
        .value("SILENT", VerbosityLevel::SILENT).value("NORMAL", VerbosityLevel::NORMAL)
        .value("VERBOSE", VerbosityLevel::VERBOSE).value("DEBUG", VerbosityLevel::DEBUG);
    py::enum_<BCType>(m, "BCType")
        .value("DIRICHLET", BCType::DIRICHLET).value("NEUMANN", BCType::NEUMANN).value("ROBIN", BCType::ROBIN)
        .value("MIRROR", BCType::MIRROR).value("PERIODIC", BCType::PERIODIC);
    py::enum_<BoundaryID>(m, "BoundaryID")
        .value("LEFT_1D", BoundaryID::LEFT_1D).value("RIGHT_1D", BoundaryID::RIGHT_1D)
        .value("LEFT_2D", BoundaryID::LEFT_2D).value("RIGHT_2D", BoundaryID::RIGHT_2D)
        .value("TOP_2D", BoundaryID::TOP_2D).value("BOTTOM_2D", BoundaryID::BOTTOM_2D)
        .value("FRONT_3D", BoundaryID::FRONT_3D).value("BACK_3D", BoundaryID::BACK_3D)
        .value("LEFT_3D", BoundaryID::LEFT_3D).value("RIGHT_3D", BoundaryID::RIGHT_3D)
        .value("TOP_3D", BoundaryID::TOP_3D).value("BOTTOM_3D", BoundaryID::BOTTOM_3D);
    py::enum_<LinearSolverType>(m, "LinearSolverType")
        .value("DIRECT_LU", LinearSolverType::DIRECT_LU).value("DIRECT_LLT", LinearSolverType::DIRECT_LLT)
        .value("DIRECT_LDLT", LinearSolverType::DIRECT_LDLT).value("CG", LinearSolverType::CG)
        .value("CG_DIAG", LinearSolverType::CG_DIAG).value("CG_ICHOL", LinearSolverType::CG_ICHOL)
        .value("BICGSTAB", LinearSolverType::BICGSTAB).value("BICGSTAB_DIAG", LinearSolverType::BICGSTAB_DIAG)
        .value("BICGSTAB_ILU", LinearSolverType::BICGSTAB_ILU).value("LCG", LinearSolverType::LCG);

    py::class_<NeutFEM>(m, "NeutFEM")
        .def(py::init<int, int, arr_t, arr_t, arr_t>(), py::arg("order"), py::arg("ng"), py::arg("x_breaks"), py::arg("y_breaks"), py::arg("z_breaks"))
        .def(py::init<int, int, int, arr_t, arr_t, arr_t>(), py::arg("rt_order"), py::arg("p_order"), py::arg("ng"), py::arg("x_breaks"),
             py::arg("y_breaks"), py::arg("z_breaks"))
        .def("set_bc", &NeutFEM::SetBC, py::arg("attr"), py::arg("type"), py::arg("value") = 0.0)
        .def("set_robin_coefficients", &NeutFEM::SetRobin, py::arg("attr"), py::arg("alpha"), py::arg("beta"))
        .def("set_linear_solver", &NeutFEM::SetLinearSolver, py::arg("solver_type"))
        .def("set_tol", &NeutFEM::SetTolerance, py::arg("tol_keff"), py::arg("tol_flux"), py::arg("tol_L2"), py::arg("max_outer"), py::arg("max_inner"))
        .def("set_verbosity", &NeutFEM::SetVerbosity, py::arg("level"))
        .def("set_cmfd_relaxation", &NeutFEM::SetCMFDRelaxation, py::arg("omega"))
        .def("apply_quarter_symmetry", &NeutFEM::ApplyQuarterSymmetry, py::arg("axis1") = 0, py::arg("axis2") = 1)
        .def("add_refl", &NeutFEM::AddReflector, py::arg("D"), py::arg("SigR"), py::arg("SigS"))
        .def("set_refl", &NeutFEM::SetReflector, py::arg("refl_id"), py::arg("dimension"), py::arg("is_upper"))
        .def("clean_refl", &NeutFEM::ClearReflectors)
        .def("BuildMatrices", &NeutFEM::BuildMatrices)
        .def("SolveKeff", &NeutFEM::SolveKeff, py::arg("use_coarse_init") = false, py::arg("coarse_factors") = std::vector<int>{},
             py::arg("use_diagonal_solver") = false, py::arg("use_cmfd") = false)
        .def("SolveAdjoint", &NeutFEM::SolveAdjoint, py::arg("normalize_to_direct") = true, py::arg("use_direct_keff") = true)
        .def("SolveSubcritical", [](NeutFEM &s) { s.oos("SolveSubcritical", "declared at include/NeutFEM.hpp:279, never defined"); })
        .def("SolveCoarse", &NeutFEM::SolveCoarse, py::arg("refine"))
        .def("build_diagonal_cache", &NeutFEM::BuildDiagonalCache)
        .def("initialize_cmfd", &NeutFEM::InitializeCMFD)
        .def("ExportVTK", &NeutFEM::ExportVTK, py::arg("filename"), py::arg("export_flux") = true, py::arg("export_current") = true,
             py::arg("export_xs") = false, py::arg("export_adjoint") = false)
        .def("ExportFluxVTK", [](NeutFEM &s, const std::string &fn, bool adj) { s.ExportVTK(fn, true, false, false, adj); }, py::arg("filename"), py::arg("adjoint") = false)
        .def("ExportXSVTK", [](NeutFEM &s, const std::string &fn) { s.ExportVTK(fn, false, false, true, false); }, py::arg("filename"))
        .def("get_D", [](NeutFEM &s) { return s.view(s.D_); })
        .def("get_SRC", [](NeutFEM &s) { return s.view(s.SRC_); })
        .def("get_SigR", [](NeutFEM &s) { return s.view(s.SigR_); })
        .def("get_NSF", [](NeutFEM &s) { return s.view(s.NSF_); })
        .def("get_KSF", [](NeutFEM &s) { return s.view(s.KSF_); })
        .def("get_Chi", [](NeutFEM &s) { return s.view(s.Chi_); })
        .def("get_SigS", [](NeutFEM &s) { return s.view(s.SigS_, true); })
        .def("get_flux", [](NeutFEM &s) { return s.flux(s.Phi_, s.fluxP0_); })
        .def("get_flux_adj", [](NeutFEM &s) { return s.flux(s.PhiAdj_, s.fluxAdjP0_); })
        .def("get_current", &NeutFEM::GetCurrent, "extension: Sol_J_ as a flat (ng*n_J) array in the reference DOF order")
        .def("get_bc_map", &NeutFEM::GetBCMap, "extension: {attr: BCType value} as set through set_bc")
        .def("reset_flux", &NeutFEM::ResetFlux)
        .def("GetNumElements", [](const NeutFEM &s) { return s.ne_; })
        .def("GetNumGroups", [](const NeutFEM &s) { return s.nphi_ / s.ne_; })      // reference bug kept, src/wrapper.cpp:953-955
        .def("GetDimension", [](const NeutFEM &s) { return s.dim_; })
        .def("GetLastKeff", [](const NeutFEM &s) { return s.last_keff_; })
        .def("GetLastKeffAdjoint", [](const NeutFEM &s) { return s.last_keff_adj_; })
        .def("GetSolverName", &NeutFEM::GetSolverName)
        .def("project_flux", [](NeutFEM &s, const std::vector<int> &, bool) { s.oos("project_flux", "bound at src/wrapper.cpp:1003, never defined"); }, py::arg("refine"), py::arg("adjoint") = false)
        .def("project_power", [](NeutFEM &s, const std::vector<int> &, bool) { s.oos("project_power", "bound at src/wrapper.cpp:1024, never defined"); }, py::arg("refine"), py::arg("adjoint") = false)
        .def("zoom_resolved", [](NeutFEM &s, const std::vector<int> &, bool) { s.oos("zoom_resolved", "bound at src/wrapper.cpp:1045, never defined"); }, py::arg("refine"), py::arg("adjoint") = false);
}
