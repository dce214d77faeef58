"""apply-only workload for PMC passes: usage pmc_apply.py nx ny nz reps"""
import sys, os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R)
import numpy as np
from neutfem_amd import capi, cases
nx, ny, nz, reps = [int(v) for v in sys.argv[1:5]]
c = cases.iaea3d_resampled(nx, nz)
if ny != nx:
    raise SystemExit("ny must equal nx")
s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
s.set_linear_solver(6)
for a, t in c["bc"]: s.set_bc(a, t)
s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
ms = s.time_schur_apply(0, reps)
print(f"{nx}x{ny}x{nz}: apply {ms:.4f} ms", [ (n, s.profile(n)) for n in ("schur_x","schur_y","schur_z")])
s.close()
