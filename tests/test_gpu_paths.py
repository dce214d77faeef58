"""The execution shapes of the full-Schur SolveKeff -- (0) host-driven outer loop with the classic four-launch CG,
(0') the same with the fused-direction two-launch CG (k_apply3), (0'') the same with every CG solve as one launch on the workgroups of
one XCD (k_cg_xcd), (2) the resident one-workgroup kernel (k_resident_keff) --
run the same per-cell arithmetic and differ only in the summation order of the dot products.  They must agree with each
other and with the oracle: tightly at tight tolerances, and with the same iteration counts on well-conditioned problems."""
import numpy as np
import pytest

from helpers import TEST_TOL, degenerate_inputs, load_inputs, make_hip, make_oracle, rel_l2, solved_oracle, synthetic_inputs

pytestmark = pytest.mark.gpu

PATHS = [("classic", dict(resident=0, cg_fuse3=0), 0), ("fuse3", dict(resident=0, cg_fuse3=1, cg_fuse3_max_cells=4 << 20, cg_xcd=0), 0), ("resident", dict(resident=1, resident_max_dofs=100000), 2),
         # whole CG solve in one launch on one XCD, forced onto every RT0-P0 mesh it can take; and the same aimed at an XCD that does not
         # exist: nobody registers, the kernel reports it before touching a vector and the solver carries on through the launches
         ("xcd", dict(resident=0, cg_fuse3=1, cg_fuse3_max_cells=4 << 20, cg_xcd=1, keff_xcd=0, cg_xcd_min_cells=0, cg_xcd_max_cells=1 << 30), 0),
         # the whole power iteration in the same kind of launch (k_keff_xcd; nf_info last_path 3 where it can run)
         ("xcd-keff", dict(resident=0, cg_fuse3=1, cg_fuse3_max_cells=4 << 20, cg_xcd=1, keff_xcd=1, cg_xcd_min_cells=0, cg_xcd_max_cells=1 << 30), 3),
         ("xcd-refused", dict(resident=0, cg_fuse3=1, cg_fuse3_max_cells=4 << 20, cg_xcd=1, cg_xcd_min_cells=0, cg_xcd_max_cells=1 << 30, cg_xcd_id=9), 0),
         ("resident-scans", dict(resident=1, resident_max_dofs=100000, resident_serial=0), 2),
         ("resident-one-sided", dict(resident=1, resident_max_dofs=100000, resident_two_sided=0), 2),   # one lane per line instead of a pair meeting in the middle
         ("classic-streaming", dict(resident=0, cg_fuse3=0, nt_min_cells=0), 0),   # the big-mesh instantiations (non-temporal loads) forced onto small meshes   # RT0-P0 small enough for LDS: "resident" is the line-per-lane variant
         # what meshes beyond 4 M cells run (no lean CG: k_finalize sums the partials), forced onto small ones: per-pass shares of p.q with the
         # z.w form in the y / z passes (split_dot), the same with the whole dot in the last pass, the chunked long-line kernel for every y / z
         # line, and the scalar readback through a D2H copy instead of the mapped host page
         ("big-split-dot", dict(resident=0, cg_fuse3=0, cg_lean=0, nt_min_cells=0, split_dot=2), 0),
         ("big-whole-dot", dict(resident=0, cg_fuse3=0, cg_lean=0, nt_min_cells=0, split_dot=0), 0),
         ("big-chunked-lines", dict(resident=0, cg_fuse3=0, cg_lean=0, split_dot=1, s_long=1), 0),
         ("classic-no-host-page", dict(resident=0, cg_fuse3=0, host_pub=0), 0)]


TIGHT = (1e-11, 1e-11, 1e-11, 1500, 3000)
TIGHT_SHAPES = [((24, 20, 6), 0, 0, 2), ((7, 6, 5), 0, 0, 3), ((19, 19, 1), 0, 0, 2), ((110, 1, 1), 1, 1, 2),
                ((12, 10, 1), 1, 1, 2), ((9, 8, 7), 1, 1, 2), ((10, 9, 1), 2, 2, 1), ((8, 6, 5), 2, 1, 2),
                ((16, 14, 1), 1, 0, 2), ((40, 33, 3), 0, 0, 2),
                # line-per-lane resident variant: lines of 4 cells (two cells per half), directions too short for
                # the two-sided sweep next to long ones, more lane slots than threads, 1D, the wide-pitch instance
                ((4, 8, 8), 0, 0, 2), ((9, 8, 3), 0, 0, 2), ((12, 11, 10), 0, 0, 2), ((250, 1, 1), 0, 0, 2),
                ((47, 45, 1), 0, 0, 2), ((6, 3, 2), 1, 1, 2), ((21, 5, 1), 2, 2, 1)]

# paths that run the SAME arithmetic as another one (other load instructions, another readback route, a refused XCD launch falling back)
BITWISE_TWINS = {"classic-streaming": "classic", "classic-no-host-page": "classic", "xcd-refused": "fuse3"}


def _run(inp, rt, p, tol, opts, coarse=False, factors=()):
    s = make_hip(inp, rt, p); s.set_tol(*tol)
    for k, v in opts.items():
        s.set_option(k, v)
    k, n = s.solve_keff(coarse, factors)
    out = dict(k=k, n=n, cg=s.history()["cg"].copy(), hk=s.history()["k"].copy(), phi=s.get_phi().copy(), path=s.info("last_path"), J=s.get_J().copy(),
               xcd=s.info("xcd_solves"), refused=s.info("xcd_refused"))
    s.close()
    return out


def _check_xcd(name, r, shape, p):
    """k_cg_xcd / k_keff_xcd ran where they can (every order; x lines of at most 128 cells) and only where they were asked to"""
    can = shape[0] <= 128
    if name in ("xcd", "xcd-keff"):
        assert (r["xcd"] > 0) == can and r["refused"] == 0, (name, r["xcd"], r["refused"])
    elif name == "xcd-refused":
        assert r["xcd"] == 0 and r["refused"] == (1 if can else 0), (name, r["xcd"], r["refused"])
    else:
        assert r["xcd"] == 0 and r["refused"] == 0, (name, r["xcd"], r["refused"])


@pytest.mark.parametrize("shape,rt,p,ng", TIGHT_SHAPES)
def test_paths_agree_at_tight_tolerance(shape, rt, p, ng):
    inp = synthetic_inputs(*shape, ng=ng, seed=7)
    tol = TIGHT
    o = solved_oracle(inp, rt, p, tol); ko = o.k                    # the oracle's converged run of this input (committed; computed live when the cache key does not match)
    res = {}
    for name, opts, path in PATHS:
        if name.startswith("big-") and (rt > 0 or shape[1] == 1):
            continue                                                # split dot / chunked lines exist for RT0-P0 y / z passes only: elsewhere these options change nothing
        if shape[0] * shape[1] * shape[2] > 2500 and name in ("resident-one-sided", "big-whole-dot", "big-split-dot"):
            continue                                                # the two largest shapes (a minute of solves): these three differ from a neighbour by one switch that 15 smaller shapes cover
        if shape[0] * shape[1] * shape[2] > 3000 and name in ("resident-scans", "big-chunked-lines"):
            continue                                                # the largest shape keeps one instance per kernel family (launches, fused, one-XCD x 2, resident); test_gpu_longlines.py has the chunked lines
        if shape[0] * shape[1] * shape[2] > 1000 and name in BITWISE_TWINS:
            continue                                                # asserted bit-identical to their twin below on the eleven smaller shapes: converging them again on the big ones adds run time, not coverage
        r = res[name] = _run(inp, rt, p, tol, opts)
        if name == "resident-scans" and shape[0] > 128:
            path = 0                                                # the scan variant takes x lines of at most 128 cells (one chunk per line)
        if name == "xcd-keff" and shape[0] > 128:
            path = 0
        assert r["path"] == path, (name, r["path"])
        _check_xcd(name, r, shape, p)
        assert abs(r["k"] - ko) / ko < 1e-9, (name, r["k"], ko)
        assert rel_l2(r["phi"].ravel(), o.phi_dofs().ravel()) < 1e-8, name
        assert rel_l2(r["J"].ravel(), o.J_dofs().ravel()) < 1e-7, name
    for twin, base in BITWISE_TWINS.items():
        if twin in res:
            assert res[twin]["k"] == res[base]["k"] and np.array_equal(res[twin]["phi"], res[base]["phi"]), twin
    for name in ("fuse3", "xcd", "xcd-keff", "resident", "resident-scans", "resident-one-sided", "big-split-dot", "big-whole-dot", "big-chunked-lines"):
        if name not in res:
            continue
        assert abs(res[name]["k"] - res["classic"]["k"]) / ko < 1e-10
        assert rel_l2(res[name]["phi"], res["classic"]["phi"]) < 1e-9


@pytest.mark.parametrize("shape,rt,ng", [((24, 20, 6), 0, 2), ((40, 33, 3), 0, 2), ((9, 8, 7), 1, 2), ((47, 45, 1), 0, 2)])
def test_paths_fixed_work_histories(shape, rt, ng):
    """pin every path independently of knife-edge stop tests -- a FIXED number of outer iterations (no stop test can move
    the count), inner CG converged to 1e-11, then the whole k-history at 1e-9 and the flux at 1e-8 against the oracle, per path, with
    nf_info last_path asserted"""
    inp = synthetic_inputs(*shape, ng=ng, seed=17)
    tol = (0.0, 1e-11, 1e-11, 6, 3000)                            # tol_keff = 0: exactly 6 outers
    o = make_oracle(inp, rt, rt); o.set_tol(*tol); o.SolveKeff(); ho = o.history()
    assert ho["n_outer"] == 6
    for name, opts, path in PATHS:
        if name == "resident-scans" and shape[0] > 128:
            path = 0
        if name.startswith("big-") and (rt > 0 or shape[1] == 1):
            continue
        if name == "xcd-keff" and shape[0] > 128:
            path = 0
        r = _run(inp, rt, rt, tol, opts)
        assert r["path"] == path and r["n"] == 6, (name, r["path"], r["n"])
        _check_xcd(name, r, shape, rt)
        np.testing.assert_allclose(r["hk"], ho["k"], rtol=1e-9, err_msg=name)
        assert rel_l2(r["phi"].ravel(), o.phi_dofs().ravel()) < 1e-8, name


@pytest.mark.parametrize("name,rt", [("iaea2d", 0), ("koeberg2d", 0), ("koeberg2d", 1), ("iaea2d", 1), ("biblis2d", 0), ("zion2d", 0)])
def test_paths_on_benchmarks_with_driver_settings(name, rt):
    """the reference drivers' own settings (loose tolerances, coarse-mesh start): same outer and CG counts as the oracle on
    every path, k-history to 1e-9 -- these problems are well conditioned, so the summation order does not move a stop test"""
    inp = load_inputs(name); f = [int(v) for v in inp["coarse_factors"]]
    o = make_oracle(inp, rt, rt); o.set_tol(*TEST_TOL); ko = o.SolveKeff(True, f); ho = o.history()
    for pname, opts, path in PATHS:
        r = _run(inp, rt, rt, TEST_TOL, opts, True, f)
        assert r["path"] == path
        assert r["n"] == ho["n_outer"], (pname, r["n"], ho["n_outer"])
        assert np.array_equal(r["cg"], ho["cg"]), (pname, r["cg"].ravel(), ho["cg"].ravel())
        np.testing.assert_allclose(r["hk"], ho["k"][:r["n"]], rtol=1e-9)
        assert abs(r["k"] - ko) / ko < 1e-9
        assert rel_l2(r["phi"].ravel(), o.phi_dofs().ravel()) < 1e-8, pname


def test_resident_path_limits_and_warm_start():
    """the resident kernel takes over only below resident_max_dofs and only for the iterative full-Schur path; warm start, reset
    and history behave like the host-driven loop"""
    inp = load_inputs("iaea2d")
    o = make_oracle(inp); s = make_hip(inp)
    o.set_tol(*TEST_TOL); s.set_tol(*TEST_TOL)
    k1, n1 = s.solve_keff(); assert s.info("last_path") == 2 and s.info("last_resident_serial") == 1
    ko1 = o.SolveKeff(); assert abs(k1 - ko1) / ko1 < 1e-9 and n1 == o.info("last_outer")
    k2, n2 = s.solve_keff(); ko2 = o.SolveKeff()                    # warm start from the last k and flux (src/NeutFEM.cpp:1662)
    assert abs(k2 - ko2) / ko2 < 1e-9 and n2 == o.info("last_outer") and n2 < n1
    s.set_option("resident_max_dofs", 1000)
    s.reset_flux(); k3, n3 = s.solve_keff(); assert s.info("last_path") == 0 and n3 == n1 and abs(k3 - k1) / k1 < 1e-9
    s.set_option("resident_max_dofs", 5000)
    s.reset_flux(); s.solve_keff(False, (), True); assert s.info("last_path") == 1      # diagonal path keeps its own device loop
    s.set_tol(0.0, 1e-4, 1e-4, 3, 1000); s.reset_flux()
    k4, n4 = s.solve_keff(); assert n4 == 3 and s.info("last_path") == 2 and len(s.history()["k"]) == 3
    s.set_option("resident_serial", 0); s.set_tol(*TEST_TOL); s.reset_flux()
    k5, n5 = s.solve_keff(); assert s.info("last_resident_serial") == 0 and n5 == n1 and abs(k5 - k1) / k1 < 1e-9
    s.close()


def test_resident_reports_divergence():
    inp = synthetic_inputs(8, 6, 5, 2, seed=1)
    from neutfem_amd.capi import HipSolver
    s = HipSolver(0, 0, 2, inp["x_breaks"], inp["y_breaks"], inp["z_breaks"]); s.set_linear_solver(6)
    s.upload_xs(inp["D"], inp["SigR"], 0.0 * inp["NSF"], inp["Chi"], inp["SigS"]); s.build()
    with pytest.raises(RuntimeError, match="diverged"):            # no fission: prod_old = 0 -> k is NaN
        s.solve_keff()
    assert s.info("last_path") == 2
    s.close()
