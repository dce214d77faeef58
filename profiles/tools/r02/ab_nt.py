import sys, os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R)
import numpy as np
from neutfem_amd import capi, cases
for n in (256, 384):
    c = cases.iaea3d_resampled(n)
    s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
    s.set_linear_solver(6)
    for a, t in c["bc"]: s.set_bc(a, t)
    s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
    for rep in range(2):
        for nt in (0, 1):
            s.set_option("nt_loads", nt); s.profile_reset()
            ms = s.time_schur_apply(0, 30)
            p = {k: s.profile(k) for k in ("schur_x","schur_y","schur_z")}
            print(f"{n}^3 nt_loads={nt}: apply {ms:.4f} ms  " + "  ".join(f"{k}={v[1]/max(v[0],1)*1e3:.1f}us" for k, v in p.items()), flush=True)
    s.close()
