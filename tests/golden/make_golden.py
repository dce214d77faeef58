#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the build container only).

inputs_<case>.npz  -- exactly what the reference's benchmark drivers feed the solver.  The driver
                      modules under /root/reference/tests are imported IN PLACE (never copied), with
                      `neutfem._neutfem_eigen` resolving to neutfem_amd's drop-in module; their own
                      load_*_mat / mesh_initialisation / init_solver code fills the cross sections through
                      the get_D()/get_SigS() views.  BuildMatrices raises here (no GPU), which is caught
                      after the fill is complete.  What is saved is DATA: break arrays, XS arrays, BC map,
                      coarse factors, literature k_ref.
golden_<case>.json -- expected outputs from the CPU oracle (oracle/nf_oracle.c): k-eff, outer count,
                      per-outer k history, CG iterations, flux checksums and strided samples.  The oracle
                      itself is pinned against oracle/ref_scipy.py by tests/test_oracle.py.

Usage: python tests/golden/make_golden.py
"""
import contextlib
import importlib.util
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neutfem_amd  # noqa: E402

neutfem_amd.install_compat(with_shims=True)
from oracle.oracle import OracleNeutFEM  # noqa: E402

REF = "/root/reference/tests"
OUT = os.path.dirname(os.path.abspath(__file__))
TEST_TOL = (1e-5, 1e-4, 1e-4, 200, 1000)       # tests/iaea3d/iaea3d.py:313


def load_driver(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def capture(obj, loader):
    """run the driver's own set-up; returns the input dict"""
    with contextlib.redirect_stdout(io.StringIO()):
        getattr(obj, loader)()
        obj.mesh_initialisation()
        try:
            obj.init_solver()
        except RuntimeError as e:            # BuildMatrices without a GPU
            assert "no HIP device" in str(e), e
    s = obj.mysolv
    bc = s.get_bc_map()
    return dict(x_breaks=np.array(obj.x_breaks, float), y_breaks=np.array(obj.y_breaks, float), z_breaks=np.array(obj.z_breaks, float),
                D=np.array(s.get_D()), SigR=np.array(s.get_SigR()), NSF=np.array(s.get_NSF()), Chi=np.array(s.get_Chi()),
                SigS=np.array(s.get_SigS()), bc_attr=np.array(sorted(bc), int), bc_type=np.array([bc[a] for a in sorted(bc)], int),
                coarse_factors=np.array(obj.coarse_factors, int), kref=float(obj.kref), ng=int(obj.num_groups))


def oracle_from(inp, rt, p):
    o = OracleNeutFEM(rt, p, int(inp["ng"]), inp["x_breaks"], inp["y_breaks"], inp["z_breaks"])
    o.set_linear_solver(6)                   # every driver: set_linear_solver(BICGSTAB)
    for a, t in zip(inp["bc_attr"], inp["bc_type"]):
        o.set_bc(int(a), int(t), 0.0)
    o.get_D()[...] = inp["D"]; o.get_SigR()[...] = inp["SigR"]; o.get_NSF()[...] = inp["NSF"]
    o.get_Chi()[...] = inp["Chi"]; o.get_SigS()[...] = inp["SigS"]
    o.BuildMatrices()
    return o


def run_case(inp, rt, p, tol, coarse, diag):
    o = oracle_from(inp, rt, p)
    o.set_tol(*tol)
    k = o.SolveKeff(coarse, [int(v) for v in inp["coarse_factors"]] if coarse else [], diag)
    h = o.history()
    phi = o.phi_dofs().ravel()
    stride = max(1, phi.size // 2000)
    return dict(rt=rt, p=p, tol=list(tol), coarse=bool(coarse), diag=bool(diag), keff=k, n_outer=int(h["n_outer"]),
                coarse_outer=int(h["coarse_outer"]), k_hist=h["k"].tolist(), cg=h["cg"].astype(int).tolist(),
                phi_sum=float(phi.sum()), phi_abs_sum=float(np.abs(phi).sum()), phi_sq=float(phi @ phi),
                phi_stride=int(stride), phi_samples=phi[::stride].tolist(),
                pcm_vs_kref=1e5 * (1.0 / float(inp["kref"]) - 1.0 / k))


def main():
    cases = {}
    m = load_driver("iaea2d/iaea2d.py", "ref_iaea2d")
    cases["iaea2d"] = (capture(m.Iaea2D(meshtype="2x2"), "load_iaea2d_mat"),
                       [(0, 0, TEST_TOL, True, False), (0, 0, TEST_TOL, False, False), (0, 0, TEST_TOL, False, True),
                        (0, 0, (1e-10, 1e-10, 1e-10, 1000, 1000), False, False), (1, 1, TEST_TOL, True, False), (1, 0, TEST_TOL, False, False)])
    m = load_driver("iaea3d/iaea3d.py", "ref_iaea3d")
    cases["iaea3d"] = (capture(m.Iaea3D(meshtype="2x2", nmeshes_z=1), "load_iaea3d_mat"),
                       [(0, 0, TEST_TOL, True, False), (0, 0, TEST_TOL, False, False), (0, 0, TEST_TOL, False, True)])
    cases["iaea3d_1x1"] = (capture(m.Iaea3D(meshtype="1x1", nmeshes_z=1), "load_iaea3d_mat"),
                           [(0, 0, TEST_TOL, False, False), (0, 0, (1e-9, 1e-9, 1e-9, 1000, 1000), False, False)])
    m = load_driver("koeberg2d/koeberg2d.py", "ref_koeberg2d")
    cases["koeberg2d"] = (capture(m.Koeberg2D(meshtype="2x2"), "load_koeberg2d_mat"),
                          [(0, 0, TEST_TOL, True, False), (1, 1, TEST_TOL, True, False)])
    m = load_driver("biblis2d/biblis2D.py", "ref_biblis2d")
    cases["biblis2d"] = (capture(m.Biblis2D(meshtype="2x2"), "load_biblis2d_mat"), [(0, 0, TEST_TOL, True, False)])
    m = load_driver("zion2d/zion2d.py", "ref_zion2d")
    cases["zion2d"] = (capture(m.Zion2D(), "load_zion2d_mat"), [(0, 0, TEST_TOL, True, False)])
    for name, (inp, runs) in cases.items():
        np.savez_compressed(os.path.join(OUT, f"inputs_{name}.npz"), **inp)
        res = [run_case(inp, *r) for r in runs]
        with open(os.path.join(OUT, f"golden_{name}.json"), "w") as f:
            json.dump(dict(case=name, kref=inp["kref"], runs=res), f)
        for r in res:
            print(f"{name:12s} RT{r['rt']}-P{r['p']} coarse={r['coarse']!s:5} diag={r['diag']!s:5} tol={r['tol'][0]:.0e} "
                  f"k={r['keff']:.10f} outers={r['n_outer']:3d} pcm_vs_kref={r['pcm_vs_kref']:+.1f}")


if __name__ == "__main__":
    main()
