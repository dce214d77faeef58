import sys, os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R)
import numpy as np
from neutfem_amd import capi, cases
nx, ny, nz = [int(v) for v in sys.argv[1:4]]
c = cases.iaea3d_resampled(nx, nz)
s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
s.set_linear_solver(6)
for a, t in c["bc"]: s.set_bc(a, t)
s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
for rep in range(2):
    for ws in (9, 100000):
        for tx in (0, 16):
            s.set_option("s_wsmin", ws); s.set_option("s_tx", tx); s.profile_reset()
            ms = s.time_schur_apply(0, 30)
            p = {n: s.profile(n) for n in ("schur_x","schur_y","schur_z")}
            print(f"{nx}x{ny}x{nz} wsmin={ws:6d} tx={tx:2d}: apply {ms:.4f} ms  " + "  ".join(f"{n}={v[1]/max(v[0],1)*1e3:.1f}us" for n, v in p.items()), flush=True)
s.close()
