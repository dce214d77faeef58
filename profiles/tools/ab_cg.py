#!/usr/bin/env python3
"""A/B of solver options INSIDE the CG (per-pass HIP events around every apply, one outer iteration per variant, same process, same box).
usage: ab_cg.py <checker|iaea3d> n groups key=v[,key=v...] [key=v,...] ...     (each argument = one variant; "default" = no options)"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from neutfem_amd import capi, cases  # noqa: E402

kind, n, ng = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
variants = sys.argv[4:] or ["default"]
c = cases.synthetic_checkerboard(n, ng) if kind == "checker" else cases.iaea3d_resampled(n)
s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
s.set_linear_solver(6)
for a, t in c["bc"]:
    s.set_bc(a, t)
s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
tol = (0.0, 0.0, 1e-4, 1, 50) if kind == "checker" else (0.0, 1e-4, 1e-4, 1, 1000)
s.set_tol(*tol); s.solve_keff()                                    # warm-up
keys = sorted({kv.split("=")[0] for v in variants if v != "default" for kv in v.split(",")})
defaults = dict(x_two_phase=0, split_dot=1, s_long=-1, s_long_dirs=3, s_long_min=256, nt_loads=1, xcd=-1, s_tx=0, cg_lean=1, vec_reduce=1)
for rep in range(2):
    for v in variants:
        for k in keys:
            s.set_option(k, defaults[k])
        if v != "default":
            for kv in v.split(","):
                k, val = kv.split("="); s.set_option(k, int(val))
        for every in (1, 1000000):                                 # with events around every pass, then without (wall time per CG iteration)
            s.set_option("prof_every", every)
            s.reset_flux(); s.set_tol(*tol); s.profile_reset()
            t0 = time.perf_counter(); k, no = s.solve_keff(profile=True); dt = time.perf_counter() - t0
            cg = s.history()["cg"].sum()
            if every == 1:
                p = {nm: s.profile(nm) for nm in ("schur_x", "schur_y", "schur_z")}
                line = "  ".join(f"{nm[-1]}={m / max(cnt, 1) * 1e3:7.1f}us" for nm, (cnt, m) in p.items())
            else:
                print(f"{v:40s} {line}   {dt / cg * 1e6:8.1f} us per CG iteration without events ({int(cg)} its, k={k:.9f})", flush=True)
s.close()
