"""k_cg_xcd (whole CG solve on one XCD) against the launch path on the small BASELINE configurations: k, counts, flux difference, wall time.
usage: xcd_dev.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from neutfem_amd.capi import HipSolver
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
for label, name, opts in [("IAEA-3D 38x38x19 (+19^3 twin)", "iaea3d", {}), ("IAEA-2D 38x38, resident kernels off", "iaea2d", {"resident": 0}),
                          ("BIBLIS-2D, resident kernels off", "biblis2d", {"resident": 0})]:
    path = os.path.join(ROOT, "tests", "golden", f"inputs_{name}.npz")
    if not os.path.exists(path):
        continue
    z = np.load(path)
    ng = int(z["ng"]); f = [int(v) for v in z["coarse_factors"]]
    res = {}
    for xcd in (0, 1):
        s = HipSolver(0, 0, ng, z["x_breaks"], z["y_breaks"], z["z_breaks"], 0)
        s.set_linear_solver(6)
        for at, ty in zip(z["bc_attr"], z["bc_type"]):
            s.set_bc(int(at), int(ty))
        for k_, v_ in opts.items():
            s.set_option(k_, v_)
        s.set_option("cg_xcd", xcd)
        s.upload_xs(z["D"], z["SigR"], z["NSF"], z["Chi"], z["SigS"]); s.build()
        s.set_tol(1e-5, 1e-4, 1e-4, 200, 1000)
        s.solve_keff(True, f, False)
        ts = []
        for _ in range(reps):
            s.reset_flux(); t0 = time.perf_counter(); k, n = s.solve_keff(True, f, False); ts.append((time.perf_counter() - t0) * 1e3)
        h = s.history()
        res[xcd] = (k, n, int(h["cg"].sum()), s.get_phi().ravel().copy(), min(ts), float(np.median(ts)), s.info("xcd_solves"), s.info("last_path"))
        print(f"{label:38s} cg_xcd={xcd}: k = {k:.13f} outers {n} CG {res[xcd][2]} min {min(ts):7.3f} ms median {np.median(ts):7.3f} ms  xcd_solves {res[xcd][6]} last_path {res[xcd][7]}", flush=True)
        s.close()
    a, b = res[0], res[1]
    print(f"    dk = {abs(a[0] - b[0]):.2e}, flux rel l2 = {np.linalg.norm(a[3] - b[3]) / np.linalg.norm(a[3]):.2e}, outers {a[1]} / {b[1]}, CG {a[2]} / {b[2]}, time x{a[4] / b[4]:.2f}", flush=True)
