"""CMFD acceleration (src/NeutFEM.cpp:662-1017, :1750-1761) on the device against the oracle.

What can be asserted.  The reference's CMFD step is not a contraction: D-hat exists for x faces only, its right-hand
side has no scattering source, and with the full Schur solver's sign of Sol_J_ the effective coefficient D~ + D^ goes
negative, so Eigen's CG runs its 100 iterations on an indefinite matrix.  On such inputs the step amplifies a 1e-12
perturbation of phi to O(1) (tests/test_oracle.py::test_cmfd_full_path_is_rounding_chaotic shows it on the oracle
alone), so no two implementations -- not even two builds of the reference -- can agree there.  Parity is therefore
asserted (a) to rounding on whole trajectories with the diagonal solver, whose J sign keeps the operator definite,
(b) to rounding on the D~ coefficients everywhere, and (c) on the first CMFD steps of small full-path problems.
"""
import numpy as np
import pytest

from helpers import load_inputs, make_hip, make_oracle, rel_l2, synthetic_inputs

pytestmark = pytest.mark.gpu


def _pair(inp, rt=0, p=0, tol=1e-9, max_outer=300):
    o = make_oracle(inp, rt, p); o.set_linear_solver(6)
    s = make_hip(inp, rt, p)
    t = (tol, tol, tol, max_outer, 3000)
    o.set_tol(*t); s.set_tol(*t)
    return o, s


@pytest.mark.parametrize("shape,rt", [((9, 1, 1), 0), ((8, 7, 1), 1), ((7, 6, 5), 0), ((5, 4, 3), 2)])
def test_dtilde_matches_on_every_direction(shape, rt):
    """ComputeDtildeCoefficients (:723-821): harmonic means inside, 2D/h on the boundary, non-uniform meshes"""
    inp = synthetic_inputs(*shape, ng=2, seed=5)
    o, s = _pair(inp, rt, min(rt, 1))
    s.initialize_cmfd()
    for g in range(2):
        for d in range(o.dim):
            dto, dho = o.cmfd_coefficients(g, d); dts, dhs = s.cmfd_coefficients(g, d)
            assert np.abs(dts - dto).max() <= 4e-16 * np.abs(dto).max()
            assert not dhs.any() and not dho.any()                # D-hat = 0 until the first correction
    s.close()


@pytest.mark.parametrize("name", ["iaea2d", "iaea3d_1x1", "koeberg2d"])
def test_diagonal_solver_with_cmfd_on_benchmarks(name):
    """SolveKeff(use_diagonal_solver=True, use_cmfd=True) (src/wrapper.cpp:662): every outer to rounding"""
    inp = load_inputs(name)
    o, s = _pair(inp, tol=1e-9, max_outer=12)
    ko = o.SolveKeff(False, [], True, True); ks, n = s.solve_keff(use_diag=True, use_cmfd=True)
    ho, hs = o.history(), s.history()
    assert n == ho["n_outer"] == 12
    assert np.abs(hs["k"] - ho["k"]).max() < 1e-12 and np.abs(hs["dphi"] - ho["dphi"]).max() < 1e-11
    assert abs(ks - ko) / ko < 1e-12
    assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-11
    dto, dho = o.cmfd_coefficients(1, 0); dts, dhs = s.cmfd_coefficients(1, 0)
    assert np.abs(dhs - dho).max() < 1e-11 * np.abs(dho).max()
    s.close()


@pytest.mark.parametrize("shape,ng,omega", [((12, 12, 6), 2, 1.0), ((9, 1, 1), 2, 1.0), ((14, 9, 1), 1, 0.7), ((6, 5, 4), 1, 0.55)])
def test_diagonal_solver_with_cmfd_to_convergence(shape, ng, omega):
    """whole trajectory (up to 300 outers, clamp [0.5, 2] and relaxation included): same outer count, k and flux"""
    inp = synthetic_inputs(*shape, ng=ng, seed=3, dirichlet=(1, 2, 3, 5))
    o, s = _pair(inp, tol=1e-7)
    o.set_cmfd_relaxation(omega); s.set_cmfd_relaxation(omega)
    ko = o.SolveKeff(False, [], True, True); ks, n = s.solve_keff(use_diag=True, use_cmfd=True)
    assert n == o.info("last_outer")
    assert abs(ks - ko) / ko < 1e-11
    assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-10
    # CMFD changes the fixed point (x-only D-hat), which the port must reproduce rather than "fix"
    o2 = make_oracle(inp); o2.set_tol(1e-7, 1e-7, 1e-7, 300, 3000)
    assert abs(o2.SolveKeff(False, [], True, False) - ko) / ko > 1e-3
    s.close()


@pytest.mark.parametrize("shape,rt,p,tol", [((9, 1, 1), 0, 0, 1e-9), ((8, 7, 1), 0, 0, 2e-6), ((8, 7, 1), 1, 1, 2e-6), ((6, 5, 4), 1, 1, 1e-7),
                                            ((8, 7, 1), 2, 2, 5e-5)])
def test_full_solver_first_cmfd_step(shape, rt, p, tol):
    """full Schur solver + CMFD: the first correction (outer 2) on small meshes; Sol_J_ mode 0 of every RT order feeds D-hat.
    tol reflects the conditioning of the reference's (indefinite) CMFD matrix, see the module docstring."""
    inp = synthetic_inputs(*shape, ng=2 if rt < 2 and shape != (6, 5, 4) else 1, seed=3, dirichlet=(1, 2, 3, 5))
    o, s = _pair(inp, rt, p, tol=1e-10, max_outer=3)
    ko = o.SolveKeff(False, [], False, True); ks, n = s.solve_keff(use_cmfd=True)
    assert n == 3 and abs(ks - ko) / ko < tol
    assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < tol
    dto, dho = o.cmfd_coefficients(0, 0); dts, dhs = s.cmfd_coefficients(0, 0)
    assert np.abs(dhs - dho).max() < 1e-6 * np.abs(dho).max() and np.abs(dho).max() > 0.1
    for d in range(1, o.dim):
        assert not s.cmfd_coefficients(0, d)[1].any()             # D-hat y/z never updated (:866-867)
    s.close()


def test_dhat_persists_across_solves_and_resets_on_build():
    """cmfd_data_ outlives SolveKeff: a second call starts with the D-hat of the previous one; BuildMatrices clears
    is_initialized (:456), so the next use_cmfd solve recomputes D-tilde and restarts from D-hat = 0"""
    inp = synthetic_inputs(10, 9, 4, 2, seed=8, dirichlet=(1, 2, 3, 5))
    o, s = _pair(inp, tol=1e-9, max_outer=6)
    for _ in range(2):                                            # second call: warm flux, warm k, D-hat carried over
        ko = o.SolveKeff(False, [], True, True); ks, _n = s.solve_keff(use_diag=True, use_cmfd=True)
        assert abs(ks - ko) / ko < 1e-12
        assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-11
    assert s.cmfd_coefficients(0, 0)[1].any()
    s.build(); o.BuildMatrices()
    with pytest.raises(RuntimeError, match="not initialised"):
        s.cmfd_coefficients(0, 0)
    ko = o.SolveKeff(False, [], True, True); ks, _n = s.solve_keff(use_diag=True, use_cmfd=True)
    assert abs(ks - ko) / ko < 1e-12 and rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-11
    s.close()


def test_cmfd_replaces_chebyshev_not_adds_to_it():
    """:1786 -- with use_cmfd the Chebyshev step is skipped: relaxation 0 makes the correction the identity, so the run
    must equal plain (unaccelerated) power iteration, which differs from the default accelerated run"""
    inp = synthetic_inputs(9, 8, 7, 2, seed=21)
    o, s = _pair(inp, tol=1e-8, max_outer=40)
    o.set_cmfd_relaxation(0.0); s.set_cmfd_relaxation(0.0)
    ko = o.SolveKeff(False, [], True, True); ks, n = s.solve_keff(use_diag=True, use_cmfd=True)
    assert n == o.info("last_outer") and abs(ks - ko) / ko < 1e-12
    s.reset_flux(); s.set_warm_state(0, 1.0)
    kc, nc = s.solve_keff(use_diag=True)
    assert abs(kc - ks) / ks > 1e-9 or nc != n
    s.close()


def test_pybind_solvekeff_use_cmfd():
    """reference call shape: solver.SolveKeff(use_diagonal_solver=True, use_cmfd=True) + set_cmfd_relaxation (wrapper.cpp:501,662)"""
    import neutfem_amd
    neutfem_amd.install_compat()
    import neutfem._neutfem_eigen as ns
    inp = load_inputs("iaea2d")
    m = ns.NeutFEM(0, 2, inp["x_breaks"], inp["y_breaks"], inp["z_breaks"])
    m.set_verbosity(ns.VerbosityLevel.SILENT); m.set_linear_solver(ns.LinearSolverType.BICGSTAB)
    for b in (ns.BoundaryID.LEFT_2D, ns.BoundaryID.RIGHT_2D, ns.BoundaryID.TOP_2D, ns.BoundaryID.BOTTOM_2D):
        m.set_bc(int(b), ns.BCType.DIRICHLET, 0.0)
    m.get_D()[...] = inp["D"]; m.get_SigR()[...] = inp["SigR"]; m.get_NSF()[...] = inp["NSF"]
    m.get_Chi()[...] = inp["Chi"]; m.get_SigS()[...] = inp["SigS"]
    with pytest.raises(RuntimeError):
        m.initialize_cmfd()                                       # before BuildMatrices
    m.BuildMatrices()
    m.initialize_cmfd()
    m.set_cmfd_relaxation(0.8)
    m.set_tol(1e-9, 1e-9, 1e-9, 10, 1000)
    k = m.SolveKeff(use_diagonal_solver=True, use_cmfd=True)
    o = make_oracle(inp); o.set_tol(1e-9, 1e-9, 1e-9, 10, 1000); o.set_cmfd_relaxation(0.8)
    ko = o.SolveKeff(False, [], True, True)
    assert abs(k - ko) / ko < 1e-12
    assert rel_l2(m.get_flux().ravel(), o.get_flux().ravel()) < 1e-11


def _team(inp, planes, rt=0, p=0):
    from neutfem_amd.capi import HipTeam
    t = HipTeam(rt, p, int(inp["ng"]), inp["x_breaks"], inp["y_breaks"], inp["z_breaks"], planes)
    t.set_linear_solver(6)
    for a, b in zip(inp["bc_attr"], inp["bc_type"]):
        t.set_bc(int(a), int(b))
    t.upload_xs_global(inp["D"], inp["SigR"], inp["NSF"], inp["Chi"], inp["SigS"]); t.build()
    return t


@pytest.mark.parametrize("planes,omega", [([(0, 20), (20, 40)], 1.0), ([(0, 9), (9, 20), (20, 31), (31, 40)], 0.7)])
def test_cmfd_on_slab_teams_with_the_diagonal_solver(planes, omega):
    """CMFD on a decomposed mesh (src/NeutFEM.cpp:662-1017): D-tilde of the interface z faces couples the edge cells of two slabs
    (harmonic mean with both cell heights, non-uniform mesh), the 7-point operator of the PCG reads one plane of p across every cut
    and its dot products are team-wide; x-only D-hat stays slab-local.  Whole trajectories against the undivided oracle."""
    nz = planes[-1][1]
    inp = synthetic_inputs(9, 7, nz, 2, seed=5, dirichlet=(1, 2, 3, 5))
    o = make_oracle(inp); t = _team(inp, planes)
    tol = (1e-12, 1e-12, 1e-12, 14, 1000); o.set_tol(*tol); t.set_tol(*tol)
    o.set_cmfd_relaxation(omega); t.set_cmfd_relaxation(omega)
    ko = o.SolveKeff(False, [], True, True); kt, n = t.solve_keff(use_diag=True, use_cmfd=True)
    assert n == 14 == o.info("last_outer")
    np.testing.assert_allclose(t.history()["k"], o.history()["k"][:14], rtol=1e-11)
    assert abs(kt - ko) / ko < 1e-11
    assert rel_l2(t.get_phi_local().ravel(), o.phi_dofs().reshape(2, nz, 7, 9).ravel()) < 1e-10
    # D-tilde of every z face, interface faces included (both neighbours hold the same value)
    for g in range(2):
        dto, _ = o.cmfd_coefficients(g, 2)
        dto = dto.reshape(nz + 1, 7 * 9)
        for s, (k0, k1) in zip(t.slabs, planes):
            dts, dhs = s.cmfd_coefficients(g, 2)
            assert np.abs(dts.reshape(k1 - k0 + 1, -1) - dto[k0:k1 + 1]).max() <= 4e-16 * np.abs(dto).max()
            assert not dhs.any()
    t.close()


@pytest.mark.parametrize("rt,p", [(0, 0), (1, 1)])
def test_cmfd_on_slab_teams_full_solver_first_step(rt, p):
    """full Schur solver + CMFD on slabs.  With the Schur solver's sign of Sol_J_ the reference's CMFD matrix is indefinite and its
    100-iteration CG amplifies a 1e-12 perturbation of the flux to O(0.1) (module docstring, tests/test_oracle.py::
    test_cmfd_full_path_is_rounding_chaotic): the partition method changes the rounding of the z-line solves, so nothing after that
    solve can be compared.  What is well defined is everything that enters it: D-tilde, and the D-hat of the first update (outer 2),
    built from the x currents and fluxes of a sweep no correction has touched yet."""
    ng = 2 if rt == 0 else 1
    inp = synthetic_inputs(6, 5, 24, ng, seed=3, dirichlet=(1, 2, 3, 5))
    planes = [(0, 11), (11, 24)]
    o = make_oracle(inp, rt, p); t = _team(inp, planes, rt, p)
    tl = (1e-10, 1e-10, 1e-10, 3, 3000); o.set_tol(*tl); t.set_tol(*tl)
    o.SolveKeff(False, [], False, True); kt, n = t.solve_keff(use_cmfd=True)
    assert n == 3 and np.isfinite(kt)
    for g in range(ng):
        dto, dho = o.cmfd_coefficients(g, 0)
        dto = dto.reshape(24, -1); dho = dho.reshape(24, -1)
        assert np.abs(dho).max() > 0.1
        for s, (k0, k1) in zip(t.slabs, planes):
            dts, dhs = s.cmfd_coefficients(g, 0)
            assert np.abs(dts.reshape(k1 - k0, -1) - dto[k0:k1]).max() <= 4e-16 * np.abs(dto).max()
            assert np.abs(dhs.reshape(k1 - k0, -1) - dho[k0:k1]).max() < 1e-6 * np.abs(dho).max()
    t.close()


def test_diagonal_cmfd_with_coarse_init():
    """the fastest advertised combination, src/wrapper.cpp:663: SolveKeff(True, factors, use_diagonal_solver=True, use_cmfd=True)
    (the coarse solve itself runs the full Schur path without CMFD, src/NeutFEM.cpp:2460-2467).  On a problem where the CMFD
    map contracts, so that the CG-tolerance-level difference of the two coarse starts is not amplified."""
    inp = synthetic_inputs(12, 12, 6, 2, seed=3, dirichlet=(1, 2, 3, 5))
    o, s = _pair(inp, tol=1e-10, max_outer=300)
    ko = o.SolveKeff(True, [2, 2, 2], True, True); ks, n = s.solve_keff(True, [2, 2, 2], use_diag=True, use_cmfd=True)
    ho, hs = o.history(), s.history()
    assert abs(n - ho["n_outer"]) <= 1 and hs["coarse_outer"] == ho["coarse_outer"] > 0
    assert abs(ks - ko) / ko < 1e-8
    assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-7
    s.close()
