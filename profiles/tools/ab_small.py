"""Wall time of the small BASELINE configs (whole SolveKeff as the drivers run it) with the library NEUTFEM_HIP_LIB names (default: this tree's):
min and median of `reps` solves after one warm-up.  Used to compare two builds on the same box:
    NEUTFEM_HIP_LIB=profiles/tools/_ab/libneutfem_hip_prev.so python profiles/tools/ab_small.py; python profiles/tools/ab_small.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from neutfem_amd.capi import HipSolver
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
print("library:", os.environ.get("NEUTFEM_HIP_LIB", "(tree)"))
for label, name, rt, coarse, diag in [("IAEA-2D 38x38 2g RT0", "iaea2d", 0, True, False), ("IAEA-3D 38x38x19 2g RT0", "iaea3d", 0, True, False),
                                      ("KOEBERG-2D 34x34 4g RT1", "koeberg2d", 1, True, False), ("KOEBERG-2D 34x34 4g RT2", "koeberg2d", 2, True, False)]:
    z = np.load(os.path.join(ROOT, "tests", "golden", f"inputs_{name}.npz"))
    ng = int(z["ng"]); f = [int(v) for v in z["coarse_factors"]]
    s = HipSolver(rt, rt, ng, z["x_breaks"], z["y_breaks"], z["z_breaks"], 0)
    s.set_linear_solver(6)
    for at, ty in zip(z["bc_attr"], z["bc_type"]):
        s.set_bc(int(at), int(ty))
    s.upload_xs(z["D"], z["SigR"], z["NSF"], z["Chi"], z["SigS"]); s.build()
    s.set_tol(1e-5, 1e-4, 1e-4, 200, 1000)
    s.solve_keff(coarse, f, diag)
    ts = []
    for _ in range(reps):
        s.reset_flux(); t0 = time.perf_counter(); k, n = s.solve_keff(coarse, f, diag); ts.append((time.perf_counter() - t0) * 1e3)
    h = s.history()
    print(f"{label:28s} k = {k:.13f}  outers {n:3d} (+{int(h['coarse_outer'])} coarse)  CG {int(h['cg'].sum()):5d}   min {min(ts):7.3f} ms  median {np.median(ts):7.3f} ms", flush=True)
    s.close()
