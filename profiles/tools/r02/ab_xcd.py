import sys, os, time; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R)
import numpy as np
from neutfem_amd import capi, cases
for n in (96, 128, 192, 256):
    c = cases.iaea3d_resampled(n)
    s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
    s.set_linear_solver(6)
    for a, t in c["bc"]: s.set_bc(a, t)
    s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
    s.set_tol(0.0, 1e-4, 1e-4, 1, 1000); s.solve_keff()
    for rep in range(2):
        for x in (0, 1):
            s.set_option("xcd", x)
            best = 1e9
            for r in range(3):
                s.set_tol(0.0, 1e-4, 1e-4, 2, 1000)
                t0 = time.perf_counter(); k, no = s.solve_keff(); dt = time.perf_counter() - t0
                best = min(best, dt / s.history()["cg"].sum() * 1e6)
            print(f"{n}^3 xcd={x}: {best:.2f} us per CG it", flush=True)
    s.close()
