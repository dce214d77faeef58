"""Input for DESIGN section 10's schedule: what does carrying the CG vector update (+32 B per cell: read r, x_sol; write p, x_sol) cost the x pass at
SLAB size?  An undivided 256 x 256 x 32 mesh (2 M cells, the size of one rank's slab of the 256^3 run), the same fixed work with cg_fuse = 1 (the x
pass carries x_sol += alpha p, p = r + beta p of the previous iteration) and cg_fuse = 0 (separate update kernel), per-pass times from the library's
HIP-event profile (every 8th apply).  usage: python profiles/tools/r04_x_pass_fused_update.py   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neutfem_amd import cases
from neutfem_amd.capi import HipSolver
c = cases.iaea3d_resampled(256, nz=32)
for fuse in (1, 0, 1, 0):
    s = HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
    s.set_linear_solver(6)
    for a, t in c["bc"]:
        s.set_bc(a, t)
    s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
    s.set_option("cg_fuse", fuse); s.set_option("cg_fuse3", 0); s.set_option("cg_xcd", 0); s.set_option("resident", 0)
    s.set_tol(0.0, 0.0, 1e-4, 1, 300); s.solve_keff()
    s.set_tol(0.0, 0.0, 1e-4, 3, 300); s.profile_reset()
    import time
    t0 = time.perf_counter(); k, n = s.solve_keff(profile=True); dt = time.perf_counter() - t0
    its = int(s.history()["cg"].sum())
    per = {nm: (lambda ct: round(1e3 * ct[1] / ct[0], 2) if ct[0] else None)(s.profile(nm)) for nm in ("schur_x", "schur_y", "schur_z", "schur_apply")}
    print(f"cg_fuse={fuse}: {its} CG iterations, {dt / its * 1e6:6.1f} us per CG iteration; us per launch {per}", flush=True)
    s.close()
