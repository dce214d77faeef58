"""k_cg_xcd / k_keff_xcd: the whole CG solve (src/solvers.cpp:577-636) of a mid-size mesh in ONE launch on the workgroups of one
XCD.  It is the default between 2 000 and 28 000 unknowns per group (every order); these tests run it there, with default options, against the
oracle -- tests/test_gpu_paths.py forces it onto the small shapes of the path matrix as well."""
import numpy as np
import pytest

from helpers import make_hip, make_oracle, rel_l2, synthetic_inputs

pytestmark = pytest.mark.gpu

# inside the default window: 3D with all three roles in one round, 3D that needs two chunks per lane on its x lines (33..64 cells),
# 2D, 1D, RT1-P0 (P0 flux, RT1 currents), an odd x length (scalar loads instead of pairs)
SHAPES = [((30, 28, 9), 0, 0, 2), ((38, 30, 12), 0, 0, 2), ((64, 48, 1), 0, 0, 2), ((100, 90, 1), 0, 0, 3), ((23, 21, 17), 0, 0, 2), ((40, 30, 4), 1, 0, 2),
          ((37, 35, 19), 0, 0, 2),                                 # x lines of 37 cells in two chunks per lane, odd length: k_keff_xcd<2, false>
          # orders with bubble moments (the window counts unknowns per group): RT1-P1 2D and 3D, RT2-P2, RT2-P1, x lines in two chunks
          ((40, 30, 1), 1, 1, 2), ((14, 12, 6), 1, 1, 2), ((30, 26, 1), 2, 2, 2), ((48, 30, 1), 2, 1, 1), ((70, 20, 1), 1, 1, 2)]      # (beyond the 5120 unknowns the one-workgroup resident kernel takes)


@pytest.mark.parametrize("shape,rt,p,ng", SHAPES)
def test_default_path_in_the_window_matches_the_oracle(shape, rt, p, ng):
    """fixed work (6 outers, CG to 1e-11): k-history 1e-9, flux 1e-8, currents 1e-7 against the oracle; the kernel did run; a second solver
    gives the same bits"""
    inp = synthetic_inputs(*shape, ng=ng, seed=23)
    tol = (0.0, 1e-11, 1e-11, 6, 5000)
    o = make_oracle(inp, rt, p); o.set_tol(*tol); o.SolveKeff(); ho = o.history()
    runs = []
    for _ in range(2):
        s = make_hip(inp, rt, p); s.set_tol(*tol)
        k, n = s.solve_keff()
        assert s.info("last_path") == 3 and s.info("xcd_solves") == 6 * ng and s.info("xcd_refused") == 0     # k_keff_xcd: the whole power iteration
        runs.append((k, s.get_phi().copy(), s.history()["k"].copy(), s.get_J().copy(), s.history()["cg"].copy()))
        s.close()
    k, phi, hk, J, cg = runs[0]
    np.testing.assert_allclose(hk, ho["k"], rtol=1e-9)
    assert rel_l2(phi.ravel(), o.phi_dofs().ravel()) < 1e-8
    assert rel_l2(J.ravel(), o.J_dofs().ravel()) < 1e-7
    assert np.all(np.abs(cg.astype(int) - ho["cg"].astype(int)) <= np.maximum(2, 0.02 * ho["cg"]))     # counts: a stop test at 1e-11 may move by an iteration
    assert runs[1][0] == k and np.array_equal(runs[1][1], phi) and np.array_equal(runs[1][4], cg)       # fixed summation order: reproducible to the bit


def test_iteration_cap_and_launch_path_agree():
    """the cap of the inner iteration (maxit) ends a solve at the same count on both routes; the launch path with the same options
    otherwise gives the same k to rounding"""
    inp = synthetic_inputs(30, 28, 9, 2, seed=5)
    tol = (0.0, 1e-13, 1e-13, 4, 9)                                # 9 CG iterations per group solve, never converged
    o = make_oracle(inp); o.set_tol(*tol); o.SolveKeff(); ho = o.history()
    res = {}
    for xcd, whole in ((1, 1), (1, 0), (0, 0)):
        s = make_hip(inp); s.set_tol(*tol); s.set_option("cg_xcd", xcd); s.set_option("keff_xcd", whole)
        k, n = s.solve_keff()
        assert (s.info("xcd_solves") > 0) == bool(xcd) and s.info("last_path") == (3 if whole else 0)
        assert np.all(s.history()["cg"] == 9) and np.all(ho["cg"] == 9)
        np.testing.assert_allclose(s.history()["k"], ho["k"], rtol=1e-9)
        res[xcd + whole] = (k, s.get_phi().copy())
        s.close()
    for a in (1, 2):                                                # k_cg_xcd under the host's outer loop, k_keff_xcd: both against the launches
        assert abs(res[a][0] - res[0][0]) < 1e-12 and rel_l2(res[a][1], res[0][1]) < 1e-11


def test_refused_start_falls_back_within_the_solve():
    """no workgroup on the chosen XCD (cg_xcd_id 9 names none): the kernel says so before touching a vector, the solver finishes the same
    solve through the launches -- bitwise what cg_xcd = 0 gives -- and stays there"""
    inp = synthetic_inputs(30, 28, 9, 2, seed=5)
    tol = (0.0, 1e-10, 1e-10, 3, 3000)
    out = []
    for opts in (dict(cg_xcd=0), dict(cg_xcd=1, cg_xcd_id=9)):
        s = make_hip(inp); s.set_tol(*tol)
        for key, v in opts.items():
            s.set_option(key, v)
        k, n = s.solve_keff()
        out.append((k, s.get_phi().copy(), s.info("xcd_solves"), s.info("xcd_refused")))
        s.close()
    assert out[1][2] == 0 and out[1][3] == 1 and out[0][3] == 0
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


def test_two_rounds_beyond_the_window():
    """a mesh past the default window (forced): the three roles of a workgroup no longer fit its wavefronts at once and phase A takes two
    rounds -- slower than the launches there, but it must still be right"""
    inp = synthetic_inputs(40, 36, 30, 2, seed=29)
    tol = (0.0, 1e-11, 1e-11, 3, 5000)
    o = make_oracle(inp); o.set_tol(*tol); o.SolveKeff(); ho = o.history()
    s = make_hip(inp); s.set_tol(*tol); s.set_option("cg_xcd_max_cells", 1 << 30)
    k, n = s.solve_keff()
    assert s.info("last_path") == 3 and s.info("xcd_solves") == 6 and s.info("xcd_refused") == 0
    np.testing.assert_allclose(s.history()["k"], ho["k"], rtol=1e-9)
    assert rel_l2(s.get_phi().ravel(), o.phi_dofs().ravel()) < 1e-8
    s.close()

