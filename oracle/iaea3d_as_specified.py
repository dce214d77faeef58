"""IAEA-3D AS SPECIFIED (vacuum condition J.n = 0.4692 phi on the stepped outline of the core and on its top and bottom) next to the driver's
variant of it (blank assemblies filled with F6 = (D 1e-3, Sigma 1e15), the reference's boundary term on the 380 cm box; tests/iaea3d/iaea3d.py:231-258),
on the CPU oracle with its nfo_set_void probe -- test infrastructure; what pins the oracle on BASELINE config 1 against the literature k (1.029096).
m cells per assembly and axis from the drivers' own input (tests/golden/inputs_iaea3d.npz).  Results: tests/golden/iaea3d_as_specified.json
(tests/test_oracle.py::test_iaea3d_as_specified_approaches_the_literature_k recomputes the cheap entries).
usage: python oracle/iaea3d_as_specified.py <rt = p> <m> <spec|driver>      (RT1-P1 m=4 and RT2-P2 m=2: 1.5 - 2.5 hours of one core each)"""
import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.dirname(os.path.abspath(__file__))]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_inputs
from oracle.oracle import OracleNeutFEM
KREF = 1.029096


def run(rt, m, mode, tol=(1e-8, 1e-7, 1e-7, 1000, 5000)):
    base = load_inputs("iaea3d")                          # 2 x 2 x 1 cells per assembly
    per_assembly = lambda a: a[..., :, ::2, ::2]
    refine = lambda a: np.repeat(np.repeat(np.repeat(a, m, axis=-1), m, axis=-2), m, axis=-3)
    inp = dict(base)
    for k in ("D", "SigR", "NSF", "Chi", "SigS"):
        inp[k] = np.ascontiguousarray(refine(per_assembly(base[k])))
    xb = np.linspace(0.0, 380.0, 19 * m + 1); zb = np.linspace(0.0, 380.0, 19 * m + 1)
    blank = inp["D"][0] == 1e-3                               # what the driver filled with F6
    if mode == "spec":
        # one void plane below and above: the true bottom / top faces become kept | void faces and carry the vacuum term too
        pad = lambda a: np.concatenate([a[..., :1, :, :], a, a[..., -1:, :, :]], axis=-3)
        for k in ("D", "SigR", "NSF", "Chi", "SigS"):
            inp[k] = np.ascontiguousarray(pad(inp[k]))
        blank = np.concatenate([np.ones_like(blank[:1]), blank, np.ones_like(blank[:1])], axis=0)
        zb = np.concatenate([[-1.0], zb, [381.0]])
        inp["D"][:, blank] = 1.0; inp["SigR"][:, blank] = 1.0; inp["NSF"][:, blank] = 0.0; inp["Chi"][:, blank] = 0.0; inp["SigS"][:, :, blank] = 0.0
    o = OracleNeutFEM(rt, rt, 2, xb, xb.copy(), zb); o.set_linear_solver(6)
    for a, t in zip(base["bc_attr"], base["bc_type"]):
        o.set_bc(int(a), int(t), 0.0)
    o.get_D()[...] = inp["D"]; o.get_SigR()[...] = inp["SigR"]; o.get_NSF()[...] = inp["NSF"]; o.get_Chi()[...] = inp["Chi"]; o.get_SigS()[...] = inp["SigS"]
    if mode == "spec":
        o.set_void(blank, 1.0 / 0.4692)
    o.BuildMatrices(); o.set_tol(*tol)
    k = o.SolveKeff()
    return k, int(o.info("last_outer")), int(o.history()["cg"].sum())


if __name__ == "__main__":
    rt, m, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    t0 = time.time(); k, n, cg = run(rt, m, mode)
    print(f"IAEA-3D {mode:6s} RT{rt}-P{rt} m={m}: k={k:.7f} pcm vs k_ref={1e5 * (1 / KREF - 1 / k):+.2f} ({n} outers, {cg} CG, {time.time() - t0:.0f}s)", flush=True)
