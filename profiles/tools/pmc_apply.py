#!/usr/bin/env python3
"""Schur-apply-only workload for rocprofv3 passes (kernel trace or one --pmc counter group per run).
usage: pmc_apply.py <checker|iaea3d|uniform> nx ny nz groups reps [key=value nf_set_option pairs]
checker / iaea3d need nx == ny == nz resp. nx == ny; `uniform` = tests/helpers.synthetic_inputs on a uniform mesh (any shape)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from neutfem_amd import capi, cases  # noqa: E402

kind = sys.argv[1]
nx, ny, nz, ng, reps = [int(v) for v in sys.argv[2:7]]
if kind == "checker":
    c = cases.synthetic_checkerboard(nx, ng)
elif kind == "iaea3d":
    c = cases.iaea3d_resampled(nx, nz)
else:
    from helpers import synthetic_inputs
    c = synthetic_inputs(nx, ny, nz, ng, seed=1, nonuniform=False)
    c["bc"] = list(zip(c["bc_attr"].tolist(), c["bc_type"].tolist()))
s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
s.set_linear_solver(6)
for a, t in c["bc"]:
    s.set_bc(int(a), int(t))
s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
for kv in sys.argv[7:]:
    k, v = kv.split("="); s.set_option(k, int(v))
s.set_option("prof_every", 1)
s.profile_reset()
ms = s.time_schur_apply(0, reps)
p = {n: s.profile(n) for n in ("schur_x", "schur_y", "schur_z")}
N = s.ne
print(f"{kind} {nx}x{ny}x{nz} g{ng}: apply {ms:.4f} ms  " + "  ".join(f"{n} {m / max(k, 1) * 1e3:.1f} us ({48.2 * N / (m / max(k, 1) * 1e-3) / 1e9:.0f} GB/s alg.)" for n, (k, m) in p.items()), flush=True)
s.close()
