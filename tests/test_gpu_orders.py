"""GPU parity for higher orders (BASELINE config 3 and SURVEY 8f-2): RT1/RT2 with P0..P2 in 1D/2D/3D against the oracle."""
import numpy as np
import pytest

from helpers import TEST_TOL, load_golden, load_inputs, make_hip, make_oracle, rel_l2, synthetic_inputs

pytestmark = pytest.mark.gpu

ORDERS = [(1, 1), (1, 0), (2, 2), (2, 1), (2, 0)]


@pytest.mark.parametrize("rt,p", ORDERS)
@pytest.mark.parametrize("shape", [(9, 1, 1), (66, 1, 1), (12, 7, 1), (37, 20, 1), (130, 5, 1), (5, 130, 1), (8, 6, 5), (17, 9, 12), (4, 3, 70)])
def test_schur_apply_orders(rt, p, shape):
    nx, ny, nz = shape
    inp = synthetic_inputs(nx, ny, nz, 2, seed=nx + 3 * ny + 7 * nz + rt, dirichlet=(1, 2, 3, 4, 5, 6) if nx % 2 else (1, 4, 5))
    o, s = make_oracle(inp, rt, p), make_hip(inp, rt, p)
    assert s.n_phi == o.n_phi and s.n_J == o.n_J
    rng = np.random.default_rng(2)
    for g in range(2):
        x = rng.standard_normal(o.n_phi)
        ya, yb = s.schur_apply(g, x), o.schur_apply(g, x)
        assert np.abs(ya - yb).max() <= 1e-12 * np.abs(yb).max(), np.abs(ya - yb).max() / np.abs(yb).max()
    s.close()


@pytest.mark.parametrize("name,rt,p", [("iaea2d", 1, 1), ("iaea2d", 1, 0), ("koeberg2d", 1, 1)])
def test_solve_keff_golden_higher_order(name, rt, p):
    """the reference drivers' `--order 1` runs: KOEBERG-2D 4-group RT1-P1 = BASELINE config 3"""
    inp = load_inputs(name)
    run = [r for r in load_golden(name)["runs"] if r["rt"] == rt and r["p"] == p][0]
    s = make_hip(inp, rt, p)
    s.set_tol(*run["tol"])
    k, n = s.solve_keff(run["coarse"], [int(v) for v in inp["coarse_factors"]], run["diag"])
    h = s.history()
    assert abs(k - run["keff"]) / run["keff"] < 1e-5
    assert n == run["n_outer"] and h["coarse_outer"] == run["coarse_outer"]
    assert np.array_equal(h["cg"], np.array(run["cg"]).reshape(h["cg"].shape))
    phi = s.get_phi().ravel()                      # reference DOF order [e*n_loc + p]
    assert rel_l2(phi[::run["phi_stride"]], run["phi_samples"]) < 1e-8
    s.close()


ORDER_CASES = [(1, 1, (8, 7, 6)), (2, 2, (5, 4, 4)), (2, 1, (20, 14, 1)), (1, 1, (40, 1, 1))]
ORDER_TOL = (1e-11, 1e-11, 1e-11, 1500, 3000)


@pytest.mark.parametrize("rt,p,shape", ORDER_CASES)
def test_solve_keff_orders_vs_oracle(rt, p, shape):
    from helpers import solved_oracle
    inp = synthetic_inputs(*shape, ng=2, seed=5 + rt + p)
    tol = ORDER_TOL
    o, s = solved_oracle(inp, rt, p, tol, want_J=False), make_hip(inp, rt, p)       # the oracle's converged run: committed (helpers.solved_oracle)
    s.set_tol(*tol)
    ko = o.k; ks, n = s.solve_keff()
    assert abs(ks - ko) / ko < 1e-9
    assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-8
    s.close()


def test_pybind_module_rt1_p1_flux_is_dof0():
    import neutfem_amd
    neutfem_amd.install_compat()
    import neutfem._neutfem_eigen as ns
    inp = load_inputs("iaea2d")
    m = ns.NeutFEM(1, 2, inp["x_breaks"], inp["y_breaks"], inp["z_breaks"])
    m.set_verbosity(ns.VerbosityLevel.SILENT)
    m.set_linear_solver(ns.LinearSolverType.BICGSTAB)
    for b in (1, 2, 3, 4):
        m.set_bc(b, ns.BCType.DIRICHLET, 0.0)
    m.get_D()[...] = inp["D"]; m.get_SigR()[...] = inp["SigR"]; m.get_NSF()[...] = inp["NSF"]
    m.get_Chi()[...] = inp["Chi"]; m.get_SigS()[...] = inp["SigS"]
    m.BuildMatrices()
    m.set_tol(*TEST_TOL)
    k = m.SolveKeff(use_coarse_init=True, coarse_factors=[2, 2, 1])
    run = [r for r in load_golden("iaea2d")["runs"] if r["rt"] == 1 and r["p"] == 1][0]
    assert abs(k - run["keff"]) / run["keff"] < 1e-5
    f = m.get_flux()                                # P1: copy of DOF 0 of every cell (src/NeutFEM.cpp:2696-2713)
    assert f.shape == (2, 38, 38) and m.GetNumGroups() == 4
    full = np.array(run["phi_samples"])
    assert np.isfinite(f).all() and abs(1e5 * (1 / 1.029585 - 1 / k)) < 10.0    # literature k within 10 pcm


@pytest.mark.parametrize("rt,p,shape", [(1, 1, (9, 8, 1)), (1, 0, (9, 8, 1)), (2, 2, (6, 5, 1)), (2, 1, (7, 1, 1)), (1, 1, (6, 5, 4)), (2, 2, (4, 3, 3))])
def test_current_reconstruction_orders(rt, p, shape):
    """Sol_J_ = -A^-1 B^T phi for every order (src/solvers.cpp:227-228): face DOFs of all transverse modes + bubbles,
    in the reference's DOF numbering (src/FEM.cpp:264-334)"""
    inp = synthetic_inputs(*shape, ng=2, seed=30 + rt + p)
    o, s = make_oracle(inp, rt, p), make_hip(inp, rt, p)
    tol = (1e-11, 1e-11, 1e-11, 1500, 3000)
    o.set_tol(*tol); s.set_tol(*tol)
    o.SolveKeff(); s.solve_keff()
    Jo, Js = o.J_dofs(), s.get_J()
    assert Js.shape == Jo.shape
    assert rel_l2(Js, Jo) < 1e-8
    s.close()


def test_iaea3d_rt1p1_reaches_the_literature_k():
    """tests/iaea3d/iaea3d.py --order 1 on its default 38x38x19 mesh: the driver's k_ref = 1.029096 (:41).  The CPU oracle needs
    320 s for this case and gives 1.029138 (+4.0 pcm, driver tolerances tightened to 1e-6 / 1e-5); the device has to land there."""
    inp = load_inputs("iaea3d")
    s = make_hip(inp, 1, 1)
    s.set_tol(1e-6, 1e-5, 1e-5, 300, 2000)
    k, n = s.solve_keff(True, [int(v) for v in inp["coarse_factors"]])
    assert abs(1e5 * (1 / 1.029096 - 1 / k)) < 6.0
    assert abs(k - 1.029138) / k < 2e-5                           # oracle value (scratch run, 6 digits); tolerance = the solve's own
    s.close()
