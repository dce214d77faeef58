import sys, os, time; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R)
import numpy as np
from neutfem_amd import capi, cases
def run(c, label, variants):
    s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
    s.set_linear_solver(6)
    for a, t in c["bc"]: s.set_bc(a, t)
    s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
    s.set_tol(0.0, 1e-4, 1e-4, 1, 1000); s.solve_keff()
    for rep in range(2):
        for name, opts in variants:
            for k_, v in opts.items(): s.set_option(k_, v)
            best = 1e9
            for r in range(3):
                s.set_tol(0.0, 1e-4, 1e-4, 2, 1000)
                t0 = time.perf_counter(); k, no = s.solve_keff(); dt = time.perf_counter() - t0
                best = min(best, dt / s.history()["cg"].sum() * 1e6)
            print(f"{label} {name:28s}: {best:.2f} us per CG it", flush=True)
    s.close()
FC = [("fused", dict(cg_fuse3=1)), ("classic lean", dict(cg_fuse3=0))]
for n in (48, 64, 80, 96, 112):
    run(cases.iaea3d_resampled(n), f"{n}^3", FC)
run(cases.iaea3d_resampled(38, 19), "38x38x19", FC)
for n in (224,):
    run(cases.iaea3d_resampled(n), f"{n}^3", [("classic", dict(nt_loads=0)), ("classic + streaming", dict(nt_loads=1, nt_min_cells=0))])
