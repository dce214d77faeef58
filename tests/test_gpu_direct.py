"""Explicit-S branch of the Schur solver on the device (src/solvers.cpp:114-124, 259-509): n_phi < 200 with any solver type, the
DIRECT_* types and a never-pushed solver type at any size.  The oracle forms S column by column with its banded
solver and LU-factors it (tests/test_oracle.py pins that against scipy); the HIP path forms S with the matrix-free apply, inverts
it on the device and solves a group with one matrix-vector product.  Both are exact solves: agreement to rounding x cond(S)."""
import numpy as np
import pytest

from helpers import make_oracle, rel_l2, synthetic_inputs

pytestmark = pytest.mark.gpu


def _pair(inp, rt, p, solver):
    from neutfem_amd.capi import HipSolver
    from oracle.oracle import OracleNeutFEM
    ng = int(inp["ng"])
    o = OracleNeutFEM(rt, p, ng, inp["x_breaks"], inp["y_breaks"], inp["z_breaks"])
    s = HipSolver(rt, p, ng, inp["x_breaks"], inp["y_breaks"], inp["z_breaks"])
    if solver is not None:
        o.set_linear_solver(solver); s.set_linear_solver(solver)
    for a, t in zip(inp["bc_attr"], inp["bc_type"]):
        o.set_bc(int(a), int(t), 0.0); s.set_bc(int(a), int(t))
    o.get_D()[...] = inp["D"]; o.get_SigR()[...] = inp["SigR"]; o.get_NSF()[...] = inp["NSF"]; o.get_Chi()[...] = inp["Chi"]; o.get_SigS()[...] = inp["SigS"]
    o.BuildMatrices()
    s.upload_xs(inp["D"], inp["SigR"], inp["NSF"], inp["Chi"], inp["SigS"]); s.build()
    return o, s


@pytest.mark.parametrize("shape,rt,p,solver", [((9, 1, 1), 0, 0, 6), ((13, 12, 1), 0, 0, 6), ((8, 7, 1), 1, 0, 3), ((5, 4, 3), 0, 0, 6), ((4, 3, 2), 1, 1, 6),
                                               ((6, 5, 1), 2, 2, None), ((20, 18, 1), 0, 0, 0), ((9, 8, 7), 0, 0, 1), ((12, 10, 1), 1, 1, 2),
                                               ((30, 24, 1), 0, 0, None), ((7, 6, 5), 1, 1, 0)])
def test_explicit_schur_branch_matches_the_exact_oracle(shape, rt, p, solver):
    inp = synthetic_inputs(*shape, ng=2, seed=sum(shape) + 5 * rt, dirichlet=(1, 2, 3, 5))
    o, s = _pair(inp, rt, p, solver)
    assert solver in (None, 0, 1, 2) or o.n_phi < 200
    tol = (1e-11, 1e-11, 1e-11, 600, 2000)
    o.set_tol(*tol); s.set_tol(*tol)
    ko = o.SolveKeff(); ks, n = s.solve_keff()
    assert s.info("last_direct") == 1 and s.info("last_path") == 0
    assert n == o.info("last_outer")
    assert abs(ks - ko) / ko < 1e-11
    assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-10
    assert (s.history()["cg"] == 1).all() and (o.history()["cg"] == 1).all()      # last_iterations_ = 1 (src/solvers.cpp:447)
    assert rel_l2(s.get_J().ravel(), o.J_dofs().ravel()) < 1e-9
    # the adjoint uses the same (symmetric) S
    ka_o = o.SolveAdjoint(True, True); ka_s, na = s.solve_adjoint(True, True)
    assert ka_s == ks and na == o.info("last_outer")
    assert rel_l2(s.get_phi_adj().ravel(), o.phi_adj_dofs().ravel()) < 1e-9
    s.close()


def test_explicit_schur_beyond_2048_unknowns():
    """round 3: the dense S^-1 of the direct branch now goes up to the oracle's own limit (6000 unknowns per group; was 2048, beyond
    which CG to 1e-14 stood in).  55 x 55 RT0-P0 = 3025 unknowns per group, DIRECT_LDLT: the same exact solve on both sides."""
    inp = synthetic_inputs(55, 55, 1, ng=2, seed=12)
    o, s = _pair(inp, 0, 0, 1)
    assert o.n_phi == 3025
    tol = (1e-10, 1e-10, 1e-10, 40, 2000); o.set_tol(*tol); s.set_tol(*tol)
    ko = o.SolveKeff(); ks, n = s.solve_keff()
    assert s.info("last_direct") == 1 and n == o.info("last_outer")
    assert abs(ks - ko) / ko < 1e-10
    assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-9
    assert (s.history()["cg"] == 1).all()
    s.close()


def test_direct_branch_follows_rebuilds_and_size_limit():
    inp = synthetic_inputs(16, 14, 1, ng=2, seed=4)
    o, s = _pair(inp, 0, 0, 0)                                       # DIRECT_LU, 224 unknowns per group
    tol = (1e-11, 1e-11, 1e-11, 600, 2000); o.set_tol(*tol); s.set_tol(*tol)
    k1 = s.solve_keff()[0]; assert abs(k1 - o.SolveKeff()) / k1 < 1e-11 and s.info("last_direct") == 1
    # new cross sections + BuildMatrices: S is formed again
    inp["SigR"] = inp["SigR"] * 1.07
    o.get_SigR()[...] = inp["SigR"]; o.BuildMatrices(); o.reset_flux()
    s.upload_xs(inp["D"], inp["SigR"], inp["NSF"], inp["Chi"], inp["SigS"]); s.build(); s.reset_flux()
    k2 = s.solve_keff()[0]; ko2 = o.SolveKeff()
    assert abs(k2 - ko2) / ko2 < 1e-11 and abs(k2 - k1) / k1 > 1e-3
    # beyond direct_max_dofs the documented stand-in (CG to 1e-14) takes over and still agrees with the exact oracle
    s.set_option("direct_max_dofs", 100); s.reset_flux(); o.reset_flux()
    k3, n3 = s.solve_keff(); ko3 = o.SolveKeff()
    assert s.info("last_direct") == 2 and s.info("direct_standin_unconverged") == 0
    assert abs(k3 - ko3) / ko3 < 1e-10 and n3 == o.info("last_outer")
    assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-9
    s.close()


def test_pybind_default_solver_is_the_direct_one():
    """a NeutFEM object on which set_linear_solver was never called reports "BiCGSTAB" but solves with SchurSolver's own default,
    DIRECT_LU (src/solvers.cpp:68 vs src/NeutFEM.cpp:126)"""
    import neutfem_amd
    neutfem_amd.install_compat()
    import neutfem._neutfem_eigen as ns
    inp = synthetic_inputs(18, 15, 1, ng=2, seed=9)
    m = ns.NeutFEM(0, 2, inp["x_breaks"], inp["y_breaks"], inp["z_breaks"]); m.set_verbosity(ns.VerbosityLevel.SILENT)
    from oracle.oracle import OracleNeutFEM
    o = OracleNeutFEM(0, 0, 2, inp["x_breaks"], inp["y_breaks"], inp["z_breaks"])
    for a in inp["bc_attr"]:
        m.set_bc(int(a), ns.BCType.DIRICHLET, 0.0); o.set_bc(int(a), 0, 0.0)
    for name in ("D", "SigR", "NSF", "Chi", "SigS"):
        getattr(m, "get_" + name)()[...] = inp[name]; getattr(o, "get_" + name)()[...] = inp[name]
    m.BuildMatrices(); o.BuildMatrices()
    assert m.GetSolverName() == "BiCGSTAB"
    m.set_tol(1e-10, 1e-10, 1e-10, 500, 2000); o.set_tol(1e-10, 1e-10, 1e-10, 500, 2000)
    k = m.SolveKeff(); ko = o.SolveKeff()
    assert abs(k - ko) / ko < 1e-11
    assert rel_l2(m.get_flux().ravel(), o.get_flux().ravel()) < 1e-10
