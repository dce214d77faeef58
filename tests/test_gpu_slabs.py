"""GPU tests of the z-slab decomposition in loopback (all slabs of the team on the one GPU of the test box):
the partition-method z-line solve, the team CG and the team power iteration against the undivided HIP solve and
the oracle.  RCCL refuses two ranks on one device: here the real library is driven with one rank (all-reduces in the solve,
ncclSend/ncclRecv to the own rank in nf_comm_selftest); several real processes run in tests/test_gpu_multiproc.py over a
host-staged stand-in transport; RCCL between GPUs is exercised by the driver's multi-GPU bench."""
import numpy as np
import pytest

from helpers import make_hip, make_oracle, rel_l2, solved_oracle, synthetic_inputs
from neutfem_amd.capi import HipTeam

pytestmark = pytest.mark.gpu


def make_team(inp, planes):
    t = HipTeam(0, 0, int(inp["ng"]), inp["x_breaks"], inp["y_breaks"], inp["z_breaks"], planes)
    t.set_linear_solver(6)
    for a, ty in zip(inp["bc_attr"], inp["bc_type"]):
        t.set_bc(int(a), int(ty))
    t.upload_xs_global(inp["D"], inp["SigR"], inp["NSF"], inp["Chi"], inp["SigS"])
    t.build()
    return t


@pytest.mark.parametrize("shape,planes", [((12, 10, 9), [(0, 4), (4, 9)]),               # 2 slabs: exact for any thickness
                                          ((16, 8, 70), [(0, 35), (35, 70)]),
                                          ((8, 6, 100), [(0, 33), (33, 67), (67, 100)]),   # middle slab of 34 planes
                                          ((20, 12, 128), [(0, 32), (32, 64), (64, 96), (96, 128)]),
                                          ((37, 9, 66), [(0, 3), (3, 66)])])               # thinnest legal edge slab, odd nx
def test_team_schur_apply_matches_undivided(shape, planes):
    nx, ny, nz = shape
    inp = synthetic_inputs(nx, ny, nz, 2, seed=nz, dirichlet=(1, 2, 3, 4, 5, 6) if nz % 2 == 0 else (1, 3, 6))
    o, t = make_oracle(inp), make_team(inp, planes)
    rng = np.random.default_rng(4)
    for g in range(2):
        x = rng.standard_normal((nz, ny, nx))
        y = t.schur_apply(g, x)
        assert rel_l2(y.ravel(), o.schur_apply(g, x.ravel())) < 1e-12
    t.close()


@pytest.mark.parametrize("planes", [[(0, 10), (10, 20), (20, 30)], [(0, 6), (6, 11), (11, 16), (16, 21), (21, 30)], [(0, 4), (4, 8), (8, 12), (12, 30)]])
def test_thin_slabs_use_separator_sweeps(planes):
    """slabs too thin for the separator system to be diagonal (coupling 0.27^planes): Jacobi sweeps on it, each one more
    neighbour exchange, bring the partition-method apply back to rounding level; the CG / power iteration follow"""
    inp = synthetic_inputs(9, 8, 30, 2, seed=2)
    o, t = make_oracle(inp), make_team(inp, planes)
    rng = np.random.default_rng(4)
    for g in range(2):
        x = rng.standard_normal((30, 8, 9))
        assert rel_l2(t.schur_apply(g, x).ravel(), o.schur_apply(g, x.ravel())) < 1e-12
    tol = (1e-12, 1e-10, 1e-10, 10, 2000)                          # fixed work: 10 outers with tight inner solves
    o.set_tol(*tol); t.set_tol(*tol)
    ko = o.SolveKeff(); kt, n = t.solve_keff()
    assert n == o.info("last_outer") == 10 and abs(kt - ko) / ko < 1e-9
    assert rel_l2(t.get_phi_local().ravel(), o.phi_dofs().reshape(2, 30, 8, 9).ravel()) < 1e-8
    t.close()


def test_slabs_of_three_planes_are_refused():
    inp = synthetic_inputs(8, 8, 30, 1, seed=2)
    t = make_team(inp, [(0, 3), (3, 6), (6, 30)])
    with pytest.raises(RuntimeError, match="too thin"):
        t.schur_apply(0, np.ones((30, 8, 8)))
    t.close()


TEAM_TOL = (1e-10, 1e-10, 1e-10, 1000, 2000)


@pytest.mark.parametrize("planes", [[(0, 48), (48, 96)], [(0, 32), (32, 64), (64, 96)]])
def test_team_solve_keff_matches_undivided_and_oracle(planes):
    inp = synthetic_inputs(8, 6, 96, 2, seed=9)
    tol = TEAM_TOL
    o, s, t = solved_oracle(inp, 0, 0, tol, want_J=False), make_hip(inp), make_team(inp, planes)
    s.set_tol(*tol); t.set_tol(*tol)
    ko = o.k; ks, ns = s.solve_keff(); kt, nt = t.solve_keff()
    phi_t = t.get_phi_local().ravel()
    assert abs(kt - ks) / ks < 1e-10 and abs(kt - ko) / ko < 1e-9
    assert rel_l2(phi_t, s.get_phi().ravel()) < 1e-8
    assert rel_l2(phi_t, o.phi_dofs().ravel()) < 1e-8
    assert abs(nt - ns) <= 1
    # same algorithm, only the summation order of the dot products differs: CG counts within 2 %
    hs, ht = s.history()["cg"][:min(ns, nt)], t.history()["cg"][:min(ns, nt)]
    assert np.abs(ht - hs).max() <= 0.02 * hs.max()
    # the big-mesh instantiations of the passes (non-temporal loads; on by themselves beyond 8 M cells per slab): same arithmetic, same bits
    fixed = (0.0, 1e-10, 1e-10, 6, 2000)                           # bit-identity shows after any number of outers: six, not the whole convergence again
    t.set_tol(*fixed); t.reset_flux(); k1, n1 = t.solve_keff(); phi_1 = t.get_phi_local().ravel()
    for x in t.slabs:
        x.set_option("nt_min_cells", 0)
    t.reset_flux(); k2, n2 = t.solve_keff()
    assert k2 == k1 and n2 == n1 == 6 and np.array_equal(t.get_phi_local().ravel(), phi_1)
    s.close(); t.close()


@pytest.mark.parametrize("planes", [[(0, 32), (32, 64), (64, 96)], [(0, 10), (10, 22), (22, 40), (40, 96)], [(0, 48), (48, 96)]])   # the last: beyond 40 planes from its interface a weight is not read
def test_endpoint_pass_as_weighted_sums_matches_the_chain_solve(planes):
    """The single-reduction CG's endpoint pass without a line solve (k_endpoint_w): c_lo / c_hi of every z line as weighted sums of the
    line's cells, the weights measured once per BuildMatrices by sending every plane's unit vector through the chain-solve endpoint pass.
    Same fixed work with the weights on and off (thick slabs, and thin ones with separator sweeps): same k-history to 1e-11, flux to 1e-9,
    and both against the oracle."""
    inp = synthetic_inputs(8, 6, 96, 2, seed=9)
    tol = (0.0, 1e-10, 1e-10, 12, 2000)                            # 12 outers, inner solves converged
    o = make_oracle(inp); o.set_tol(*tol); ko = o.SolveKeff()
    res = []
    for w in (1, 0):
        t = make_team(inp, planes); t.set_tol(*tol)
        t.head.set_option("cg_single_reduce", 1); t.head.set_option("endpoint_weights", w)
        k, n = t.solve_keff()
        assert n == 12 and t.head.info("cg_reductions") == 1 and t.head.info("endpoint_weights") == w
        res.append((k, t.history()["k"].copy(), t.get_phi_local().ravel().copy(), t.history()["cg"].sum()))
        t.close()
    (k1, h1, p1, c1), (k0, h0, p0, c0) = res
    np.testing.assert_allclose(h1, h0, rtol=1e-11)
    assert rel_l2(p1, p0) < 1e-9 and abs(c1 - c0) <= 0.02 * c0
    np.testing.assert_allclose(h1, o.history()["k"], rtol=1e-9)
    assert rel_l2(p1, o.phi_dofs().ravel()) < 1e-8


def test_team_driver_tolerances_iaea3d_like():
    """reference-driver tolerances on a resampled IAEA-3D core cut into 2 slabs"""
    from neutfem_amd import cases
    c = cases.iaea3d_resampled(38, 76)
    inp = dict(x_breaks=c["x_breaks"], y_breaks=c["y_breaks"], z_breaks=c["z_breaks"], D=c["D"], SigR=c["SigR"], NSF=c["NSF"],
               Chi=c["Chi"], SigS=c["SigS"], bc_attr=np.array([a for a, _ in c["bc"]]), bc_type=np.array([t for _, t in c["bc"]]), ng=2)
    s, t = make_hip(inp), make_team(inp, [(0, 38), (38, 76)])
    tol = (1e-5, 1e-4, 1e-4, 200, 1000)
    s.set_tol(*tol); t.set_tol(*tol)
    ks, ns = s.solve_keff(); kt, nt = t.solve_keff()
    assert abs(kt - ks) / ks < 1e-5 and abs(nt - ns) <= 1
    assert rel_l2(t.get_phi_local().ravel(), s.get_phi().ravel()) < 2e-4
    s.close(); t.close()


def test_rccl_allreduce_path_single_rank(monkeypatch):
    """NEUTFEM_FORCE_RCCL=1 builds a real 1-rank RCCL communicator and routes every scalar reduction of the solve
    through ncclAllReduce on the solver's stream; nf_comm_selftest drives ncclSend/ncclRecv of the same library (to the own
    rank -- between ranks the call pattern is covered by tests/test_gpu_multiproc.py with the stand-in transport)."""
    from neutfem_amd.capi import HipTeam
    monkeypatch.setenv("NEUTFEM_FORCE_RCCL", "1")
    inp = synthetic_inputs(12, 8, 80, 2, seed=3)
    s, t = make_hip(inp), make_team(inp, [(0, 40), (40, 80)])
    t.comm_init(HipTeam.unique_id(), 1, 0)
    t.comm_selftest()            # real RCCL: grouped ncclSend/ncclRecv (to self) on the comm stream with the apply's event pattern, all-reduce(max)
    tol = (1e-9, 1e-9, 1e-9, 500, 2000)
    s.set_tol(*tol); t.set_tol(*tol)
    ks, ns = s.solve_keff(); kt, nt = t.solve_keff()
    assert abs(kt - ks) / ks < 1e-10 and abs(nt - ns) <= 1
    assert rel_l2(t.get_phi_local().ravel(), s.get_phi().ravel()) < 1e-8
    s.close(); t.close()


def test_team_coarse_init_matches_undivided():
    """SolveKeff(use_coarse_init=True) on a slab team: every slab coarsens its own planes, the coarse slabs solve as a
    team, the prolonged flux seeds the fine solve (src/NeutFEM.cpp:2380-2611) -- same k, outer counts and flux as the
    undivided mesh."""
    inp = synthetic_inputs(12, 8, 96, 2, seed=21)
    s, t = make_hip(inp), make_team(inp, [(0, 48), (48, 96)])
    tol = (1e-7, 1e-7, 1e-7, 400, 2000)
    s.set_tol(*tol); t.set_tol(*tol)
    ks, ns = s.solve_keff(True, [2, 2, 2]); kt, nt = t.solve_keff(True, [2, 2, 2])
    assert s.history()["coarse_outer"] > 0 and abs(t.history()["coarse_outer"] - s.history()["coarse_outer"]) <= 1
    assert abs(kt - ks) / ks < 1e-8 and abs(nt - ns) <= 1
    assert rel_l2(t.get_phi_local().ravel(), s.get_phi().ravel()) < 1e-6
    with pytest.raises(RuntimeError, match="do not divide"):
        t.solve_keff(True, [2, 2, 5])
    s.close(); t.close()


@pytest.mark.parametrize("planes", [[(0, 5), (5, 12)], [(0, 32), (32, 64), (64, 96)]])
def test_team_diagonal_path_matches_oracle(planes):
    """diagonal-Schur path on slabs: one edge plane of a2 per interface at cache-build time, then only the
    per-outer scalars; S_inv per slab and the whole power iteration against the undivided oracle"""
    nz = planes[-1][1]
    inp = synthetic_inputs(11, 9, nz, 2, seed=17, dirichlet=(1, 2, 4, 5, 6))
    o, t = make_oracle(inp), make_team(inp, planes)
    tol = (1e-10, 1e-10, 1e-10, 500, 1000)
    o.set_tol(*tol); t.set_tol(*tol)
    ko = o.SolveKeff(False, [], True); kt, n = t.solve_keff(use_diag=True)
    assert n == o.info("last_outer") and abs(kt - ko) / ko < 1e-12
    assert rel_l2(t.get_phi_local().ravel(), o.phi_dofs().reshape(2, nz, 9, 11).ravel()) < 1e-11
    for g in range(2):
        sinv = np.concatenate([s.diagonal_cache(g) for s in t.slabs])
        assert np.abs(sinv / o.diag_cache(g) - 1).max() < 1e-14
    # Sol_J_ after a diagonal solve (J_f = +(B^T phi)_f / A_ff, src/NeutFEM.cpp:620-633): the interface z faces need the neighbour's
    # edge cells (one phi plane and one a2 plane per interface and group)
    J = t.get_J_local()
    assert J.shape == (2, o.n_J) and rel_l2(J.ravel(), o.J_dofs().ravel()) < 1e-9
    lo = o.info("n_Jx") + o.info("n_Jy")
    assert np.abs(J[:, lo:] - o.J_dofs()[:, lo:]).max() <= 1e-12 * np.abs(o.J_dofs()[:, lo:]).max()
    t.close()


@pytest.mark.parametrize("planes", [[(0, 40), (40, 80)], [(0, 9), (9, 20), (20, 31), (31, 40)]])
def test_team_currents_match_oracle(planes):
    """Sol_J_ on a decomposed mesh: x / y faces slab-locally, z faces through the partition method in emit mode (k_schur_s
    mode 3), separators included; thick and thin slabs, against the oracle's J = -A^-1 B^T phi of the same solve"""
    nz = planes[-1][1]
    inp = synthetic_inputs(8, 6, nz, 2, seed=13, dirichlet=(1, 2, 4, 5, 6))
    o, t = make_oracle(inp), make_team(inp, planes)
    tol = (1e-12, 1e-11, 1e-11, 12, 3000)                          # fixed work: 12 outers, tight inner solves
    o.set_tol(*tol); t.set_tol(*tol)
    ko = o.SolveKeff(); kt, _ = t.solve_keff()
    assert abs(kt - ko) / ko < 1e-9
    J = t.get_J_local()
    assert J.shape == (2, o.n_J)
    assert rel_l2(J.ravel(), o.J_dofs().ravel()) < 1e-7
    for lo, hi in ((0, o.info("n_Jx")), (o.info("n_Jx") + o.info("n_Jy"), o.n_J)):      # x faces and z faces separately
        assert rel_l2(J[:, lo:hi].ravel(), o.J_dofs()[:, lo:hi].ravel()) < 1e-7
    t.close()


HO_ORDERS = [(1, 1), (1, 0), (2, 2), (2, 1)]
HO_SHAPES = [((5, 4, 20), [(0, 9), (9, 20)]), ((4, 3, 30), [(0, 10), (10, 20), (20, 30)])]
HO_CURRENT_TOL = (1e-12, 1e-11, 1e-11, 5, 3000)      # five outers: the oracle's RT2-P2 solve is what this test waits for (12 outers: 12 s per case), and the
                                                      # currents of a fixed amount of work are as comparable after 5 outers as after 12
HO_SOLVE_TOL = (1e-12, 1e-10, 1e-10, 10, 3000)


@pytest.mark.parametrize("rt,p", HO_ORDERS)
@pytest.mark.parametrize("shape,planes", HO_SHAPES)
def test_team_currents_higher_orders(shape, planes, rt, p):
    """Sol_J_ of RT1 / RT2 on slabs: the partition-method solve in emit mode stores, per transverse mode, the face DOFs of every local
    z face (separators included) and the z bubbles of every local cell (edge cells included); x / y DOFs are slab-local"""
    inp = synthetic_inputs(*shape, ng=2, seed=21 + rt, dirichlet=(1, 2, 3, 5, 6))
    o, t = make_oracle(inp, rt, p), make_team_order(inp, planes, rt, p)
    tol = HO_CURRENT_TOL                                          # fixed work: 5 outers, tight inner solves
    o.set_tol(*tol); t.set_tol(*tol)
    ko = o.SolveKeff(); kt, _ = t.solve_keff()
    assert abs(kt - ko) / ko < 1e-9
    J, Jo = t.get_J_local(), o.J_dofs()
    assert J.shape == Jo.shape
    nface = o.info("n_Jx") + o.info("n_Jy") + o.info("n_Jz")
    for lo, hi in ((0, o.info("n_Jx") + o.info("n_Jy")), (o.info("n_Jx") + o.info("n_Jy"), nface), (nface, o.n_J)):   # x/y faces, z faces, bubbles
        assert rel_l2(J[:, lo:hi].ravel(), Jo[:, lo:hi].ravel()) < 1e-7, (lo, hi)
    t.close()


def make_team_order(inp, planes, rt, p):
    t = HipTeam(rt, p, int(inp["ng"]), inp["x_breaks"], inp["y_breaks"], inp["z_breaks"], planes)
    t.set_linear_solver(6)
    for a, ty in zip(inp["bc_attr"], inp["bc_type"]):
        t.set_bc(int(a), int(ty))
    t.upload_xs_global(inp["D"], inp["SigR"], inp["NSF"], inp["Chi"], inp["SigS"])
    t.build()
    return t


def _team_vector(t, nz, ny, nx, nloc, v):
    """host DOF order [e * nloc + p] of the whole mesh -> per-slab pieces and back (cells are z-major: slabs are contiguous)"""
    return v.reshape(nz, ny * nx * nloc)


@pytest.mark.parametrize("rt,p", HO_ORDERS)
@pytest.mark.parametrize("shape,planes", HO_SHAPES)
def test_team_higher_orders_match_oracle(shape, planes, rt, p):
    """RT1 / RT2 on slab teams: every transverse mode has its own separator planes (same unit-scaled factors), the edge cells
    contribute xL / xR and receive their bubble moments; apply and power iteration against the undivided oracle"""
    nx, ny, nz = shape
    inp = synthetic_inputs(nx, ny, nz, 2, seed=rt + 7 * p + nz, dirichlet=(1, 2, 3, 5, 6))
    o, t = make_oracle(inp, rt, p), make_team_order(inp, planes, rt, p)
    nloc = o.n_phi // o.ne
    rng = np.random.default_rng(3)
    x = rng.standard_normal(o.n_phi)
    xs = _team_vector(t, nz, ny, nx, nloc, x)
    # apply through the team API: per-slab device vectors in the slabs' own DOF order
    xd, yd = [], []
    for s, (k0, k1) in zip(t.slabs, planes):
        xd.append(s.vector().upload(s._to_dev(np.ascontiguousarray(xs[k0:k1]).ravel()))); yd.append(s.vector())
    import ctypes as C
    arr = (C.c_void_p * len(t.slabs))(*[v.ptr for v in xd]); out = (C.c_void_p * len(t.slabs))(*[v.ptr for v in yd])
    t.head._chk(t.L.nf_team_schur_apply(t.head.h, 1, arr, out))
    y = np.concatenate([s._from_dev(v.download()) for s, v in zip(t.slabs, yd)])
    assert rel_l2(y, o.schur_apply(1, x)) < 1e-12
    for v in xd + yd: v.free()
    tol = HO_SOLVE_TOL                                             # fixed work
    so = solved_oracle(inp, rt, p, tol, want_J=False); t.set_tol(*tol)   # the oracle's ten outers: committed (helpers.solved_oracle)
    ko = so.k; kt, n = t.solve_keff()
    assert n == 10 == so.n_outer and abs(kt - ko) / ko < 1e-9
    phi = np.concatenate([s.get_phi().reshape(2, -1) for s in t.slabs], axis=1)
    assert rel_l2(phi.ravel(), so.phi_dofs().ravel()) < 1e-8
    t.close()


@pytest.mark.parametrize("planes", [[(0, 20), (20, 40)], [(0, 10), (10, 22), (22, 40)]])
def test_team_adjoint_matches_undivided_and_oracle(planes):
    """SolveAdjoint (src/NeutFEM.cpp:1877-2082) on a slab team: fixed-k mode to convergence, bi-orthonormalised against the
    team's direct flux; and the first free-k outers (before the reference's Chebyshev step destabilises them)"""
    inp = synthetic_inputs(8, 7, 40, 2, seed=23, dirichlet=(1, 2, 4, 5, 6))
    o, s, t = make_oracle(inp), make_hip(inp), make_team(inp, planes)
    tol = (1e-10, 1e-9, 1e-9, 400, 3000)
    o.set_tol(*tol); s.set_tol(*tol); t.set_tol(*tol)
    ko = o.SolveKeff(); ks, _ = s.solve_keff(); kt, _ = t.solve_keff()
    for x in (o, s, t): x.set_tol(1e-10, 1e-9, 1e-9, 30, 3000)        # fixed work: 30 adjoint outers
    ka_o = o.SolveAdjoint(True, True); ka_s, na_s = s.solve_adjoint(True, True); ka_t, na_t = t.solve_adjoint(True, True)
    assert ka_t == kt and abs(ka_t - ka_o) / ka_o < 1e-9 and na_t == na_s == 30
    pa = t.get_phi_adj_local().ravel()
    assert rel_l2(pa, s.get_phi_adj().ravel()) < 1e-6 and rel_l2(pa, o.phi_adj_dofs().ravel()) < 1e-6
    o.set_tol(1e-10, 1e-9, 1e-9, 5, 3000); t.set_tol(1e-10, 1e-9, 1e-9, 5, 3000)
    kf_o = o.SolveAdjoint(False, False); kf_t, n = t.solve_adjoint(False, False)
    assert n == 5 and abs(kf_t - kf_o) / kf_o < 1e-8
    assert rel_l2(t.get_phi_adj_local().ravel(), o.phi_adj_dofs().ravel()) < 1e-6
    s.close(); t.close()
