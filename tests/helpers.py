"""Shared set-up for the parity tests: build the oracle and the HIP solver from the same input dict."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TEST_TOL = (1e-5, 1e-4, 1e-4, 200, 1000)      # reference drivers: set_tol(1e-5,1e-4,1e-4,200,1000)


def load_inputs(name):
    z = np.load(os.path.join(GOLDEN, f"inputs_{name}.npz"))
    return {k: z[k] for k in z.files}


def load_golden(name):
    with open(os.path.join(GOLDEN, f"golden_{name}.json")) as f:
        return json.load(f)


def make_oracle(inp, rt=0, p=0):
    from oracle.oracle import OracleNeutFEM
    o = OracleNeutFEM(rt, p, int(inp["ng"]), inp["x_breaks"], inp["y_breaks"], inp["z_breaks"])
    o.set_linear_solver(6)
    for a, t in zip(inp["bc_attr"], inp["bc_type"]):
        o.set_bc(int(a), int(t), 0.0)
    o.get_D()[...] = inp["D"]; o.get_SigR()[...] = inp["SigR"]; o.get_NSF()[...] = inp["NSF"]
    o.get_Chi()[...] = inp["Chi"]; o.get_SigS()[...] = inp["SigS"]
    o.BuildMatrices()
    return o


def make_hip(inp, rt=0, p=0, device=0):
    from neutfem_amd.capi import HipSolver
    s = HipSolver(rt, p, int(inp["ng"]), inp["x_breaks"], inp["y_breaks"], inp["z_breaks"], device)
    s.set_linear_solver(6)
    for a, t in zip(inp["bc_attr"], inp["bc_type"]):
        s.set_bc(int(a), int(t))
    s.upload_xs(inp["D"], inp["SigR"], inp["NSF"], inp["Chi"], inp["SigS"])
    s.build()
    return s


def synthetic_inputs(nx, ny, nz, ng, seed=0, dirichlet=(1, 2, 3, 4, 5, 6), nonuniform=True, void_frac=0.05):
    """random heterogeneous problem with a few strong absorbers.  (Not Sigma = 1e15 like IAEA-3D's F6: scattered
    cells of that kind give cond(S) ~ 1e20, on which the reference's unpreconditioned CG stagnates on any
    hardware -- such a case tests nothing.)"""
    rng = np.random.default_rng(seed)
    def brk(n):
        if n <= 1: return np.array([0.0])
        h = rng.uniform(0.5, 2.5, n) if nonuniform else np.full(n, 1.25)
        return np.concatenate([[0.0], np.cumsum(h)])
    shape = tuple(n for n in (nz, ny, nx) if n > 1) or (nx,)
    if len(shape) == 1: shape = (nx,)
    D = rng.uniform(0.2, 2.0, (ng,) + shape); SigR = rng.uniform(0.01, 0.2, (ng,) + shape)
    NSF = rng.uniform(0.0, 0.15, (ng,) + shape) * (rng.random((ng,) + shape) > 0.3)
    Chi = np.zeros((ng,) + shape); Chi[0] = 1.0
    if ng > 1: Chi[0] = 0.8; Chi[1] = 0.2
    SigS = np.zeros((ng, ng) + shape)
    for g in range(1, ng): SigS[g, g - 1] = rng.uniform(0.005, 0.05, shape)
    if ng > 2: SigS[ng - 2, ng - 1] = rng.uniform(0.0, 0.003, shape)      # one up-scatter block
    void = rng.random(shape) < void_frac
    D[:, void] = 0.05; SigR[:, void] = 50.0; NSF[:, void] = 0.0; Chi[:, void] = 0.0; SigS[:, :, void] = 0.0
    dim = 3 if nz > 1 else (2 if ny > 1 else 1)
    valid = {1: (1, 2), 2: (1, 2, 3, 4), 3: (1, 2, 3, 4, 5, 6)}[dim]
    attrs = [a for a in dirichlet if a in valid]
    return dict(x_breaks=brk(nx), y_breaks=brk(ny), z_breaks=brk(nz), D=D, SigR=SigR, NSF=NSF, Chi=Chi, SigS=SigS,
                bc_attr=np.array(attrs, int), bc_type=np.zeros(len(attrs), int), coarse_factors=np.array([1, 1, 1]), kref=1.0, ng=ng)


def rel_l2(a, b):
    a = np.asarray(a).ravel(); b = np.asarray(b).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def degenerate_inputs(nx, ny, nz, ng=2, seed=3):
    """meshes with a single cell along some axis: every axis that exists keeps its breaks (an axis with one cell has two);
    the dimension follows the reference's rule dim = 3 if nz > 1 else 2 if ny > 1 else 1 (src/FEM.cpp:23-60)"""
    rng = np.random.default_rng(seed)
    brk = lambda n: np.concatenate([[0.0], np.cumsum(rng.uniform(0.5, 2.5, n))])
    xb = brk(nx); yb = brk(ny) if (ny > 1 or nz > 1) else np.array([0.0]); zb = brk(nz) if nz > 1 else np.array([0.0])
    dim = 3 if nz > 1 else (2 if ny > 1 else 1)
    shp = (nz, ny, nx)[3 - dim:]
    D = rng.uniform(0.2, 2.0, (ng,) + shp); SigR = rng.uniform(0.01, 0.2, (ng,) + shp); NSF = rng.uniform(0.0, 0.15, (ng,) + shp)
    Chi = np.zeros((ng,) + shp); Chi[0] = 0.8; Chi[1] = 0.2
    SigS = np.zeros((ng, ng) + shp); SigS[1, 0] = 0.02
    attrs = {1: (1, 2), 2: (1, 2, 3, 4), 3: (1, 2, 3, 4, 5, 6)}[dim]
    return dict(x_breaks=xb, y_breaks=yb, z_breaks=zb, D=D, SigR=SigR, NSF=NSF, Chi=Chi, SigS=SigS, bc_attr=np.array(attrs),
                bc_type=np.zeros(len(attrs), dtype=int), ng=ng, coarse_factors=np.array([1, 1, 1]), kref=1.0)


# ---- converged oracle runs, cached -----------------------------------------------------------------------------------------------------
# The tight-tolerance parity tests compare the GPU with a CONVERGED oracle solve of the same input; on one host core those solves are most
# of the GPU suite's wall time (the 40 x 33 x 3 shape: 30 s of oracle for 10 s of GPU paths).  The oracle is deterministic, so its outputs
# for the heavy cases are committed under tests/golden/oracle_cache/ -- keyed by a hash of the inputs, the settings AND oracle/nf_oracle.c:
# edit the oracle (or an input generator) and the key no longer matches, the test computes live again.  tests/golden/make_oracle_cache.py
# regenerates the files; tests/test_oracle.py::test_oracle_cache_is_current recomputes entries and compares them bit for bit.
ORACLE_CACHE = os.path.join(GOLDEN, "oracle_cache")


def _oracle_key(inp, rt, p, tol, coarse):
    import hashlib
    h = hashlib.sha256()
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "nf_oracle.c"), "rb") as f:
        h.update(f.read())
    for k in ("x_breaks", "y_breaks", "z_breaks", "D", "SigR", "NSF", "Chi", "SigS", "bc_attr", "bc_type"):
        a = np.ascontiguousarray(inp[k]); h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    h.update(repr((int(inp["ng"]), int(rt), int(p), tuple(float(t) for t in tol), None if coarse is None else tuple(int(c) for c in coarse))).encode())
    return h.hexdigest()[:24]


class SolvedOracle:
    """what a parity test reads from a converged oracle run"""

    def __init__(self, k, n_outer, phi, J, hist_k, hist_cg, cached):
        self.k, self.n_outer, self.phi, self.J, self.hist_k, self.hist_cg, self.cached = float(k), int(n_outer), phi, J, hist_k, hist_cg, cached

    def phi_dofs(self): return self.phi
    def J_dofs(self): return self.J


def solved_oracle(inp, rt=0, p=0, tol=TEST_TOL, coarse=None, want_J=True, write=False):
    key = _oracle_key(inp, rt, p, tol, coarse)
    path = os.path.join(ORACLE_CACHE, key + ".npz")
    if os.path.exists(path) and not write:
        z = np.load(path)
        return SolvedOracle(z["k"], z["n_outer"], z["phi"], z["J"] if "J" in z.files else None, z["hist_k"], z["hist_cg"], True)
    o = make_oracle(inp, rt, p); o.set_tol(*tol)
    k = o.SolveKeff(True, [int(c) for c in coarse]) if coarse is not None else o.SolveKeff()
    h = o.history()
    r = SolvedOracle(k, h["n_outer"], o.phi_dofs().copy(), o.J_dofs().copy() if want_J else None, h["k"], h["cg"], False)
    if write:
        os.makedirs(ORACLE_CACHE, exist_ok=True)
        extra = dict(J=r.J) if want_J else {}
        np.savez_compressed(path, k=np.float64(r.k), n_outer=np.int64(r.n_outer), phi=r.phi, hist_k=r.hist_k, hist_cg=r.hist_cg, **extra)
    return r
