import sys, os, time; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R)
import numpy as np
from neutfem_amd import capi, cases
n = 256
c = cases.iaea3d_resampled(n)
s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
s.set_linear_solver(6)
for a, t in c["bc"]: s.set_bc(a, t)
s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
s.set_tol(0.0, 1e-4, 1e-4, 1, 1000); s.solve_keff()
s.set_option("prof_every", 1)
for rep in range(2):
    for name, opts in (("default", dict(xcd=-1)),):
        for k_, v in opts.items(): s.set_option(k_, v)
        s.set_tol(0.0, 1e-4, 1e-4, 1, 1000); s.profile_reset()
        t0 = time.perf_counter(); k, no = s.solve_keff(profile=True); dt = time.perf_counter() - t0
        cg = s.history()["cg"].sum()
        p = {nm: s.profile(nm) for nm in ("schur_x", "schur_y", "schur_z")}
        print(f"{name:14s}: {dt/cg*1e6:.1f} us per CG it  " + "  ".join(f"{k_}={v[1]/max(v[0],1)*1e3:.1f}us" for k_, v in p.items()), flush=True)
s.close()
