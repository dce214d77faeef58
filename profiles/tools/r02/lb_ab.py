import sys, os, json, time; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R)
import numpy as np
from neutfem_amd import capi, cases
if sys.argv[1] == "old": capi.lib_path = lambda: os.path.join(R, "scratch", "libneutfem_old.so")
sys.path.insert(0, R)
from bench import split_planes
n = 256; slabs = int(sys.argv[2])
allp = split_planes(n, slabs)
case = cases.iaea3d_resampled(n)
s = capi.HipTeam(0, 0, 2, case["x_breaks"], case["y_breaks"], np.linspace(0.0, 380.0, n + 1), allp, device=0)
s.set_linear_solver(6)
for at, ty in case["bc"]: s.set_bc(at, ty)
s.upload_xs_global(case["D"], case["SigR"], case["NSF"], case["Chi"], case["SigS"], k_offset=0); s.build()
s.set_tol(0.0, 1e-4, 1e-4, 1, 1000); s.solve_keff()
s.set_tol(0.0, 1e-4, 1e-4, 2, 1000); s.profile_reset(); s.synchronize(); t0 = time.perf_counter(); k, no = s.solve_keff(profile=True); s.synchronize(); dt = time.perf_counter() - t0
cg = s.history()["cg"].sum()
print(sys.argv[1], "slabs", slabs, "outer/s", 2 / dt, "us per CG it", dt / cg * 1e6, {nm: round(s.profile(nm)[1] / max(s.profile(nm)[0], 1) * 1e3, 1) for nm in ("schur_x", "schur_y", "schur_z", "schur_z1", "schur_apply")})
s.close()
