import sys, os, json, time; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R)
import numpy as np
from neutfem_amd import capi, cases
from bench import split_planes
n = 256; slabs = int(sys.argv[1])
allp = split_planes(n, slabs)
case = cases.iaea3d_resampled(n)
s = capi.HipTeam(0, 0, 2, case["x_breaks"], case["y_breaks"], np.linspace(0.0, 380.0, n + 1), allp, device=0)
s.set_linear_solver(6)
for at, ty in case["bc"]: s.set_bc(at, ty)
s.upload_xs_global(case["D"], case["SigR"], case["NSF"], case["Chi"], case["SigS"], k_offset=0); s.build()
s.set_tol(0.0, 1e-4, 1e-4, 1, 1000); s.solve_keff()
for rep in range(2):
    for nt in (0, 1):
        for x in s.slabs: x.set_option("nt_loads", nt)
        s.set_tol(0.0, 1e-4, 1e-4, 2, 1000); s.synchronize(); t0 = time.perf_counter(); k, no = s.solve_keff(); s.synchronize(); dt = time.perf_counter() - t0
        cg = s.history()["cg"].sum()
        print(f"slabs {slabs} nt_loads={nt}: {2/dt:.3f} outer/s  {dt/cg*1e6:.1f} us per CG it  k={k:.10f}", flush=True)
s.close()
