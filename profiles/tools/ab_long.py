#!/usr/bin/env python3
"""A/B of the y / z line kernels on one box, same process: classic one-chunk k_schur_s (s_long=0) against the chunked k_schur_c
(s_long=1) at several tile widths.  usage: ab_long.py <checker|iaea3d> n groups [reps]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from neutfem_amd import capi, cases  # noqa: E402

kind, n, ng = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
c = cases.synthetic_checkerboard(n, ng) if kind == "checker" else cases.iaea3d_resampled(n)
s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
s.set_linear_solver(6)
for a, t in c["bc"]:
    s.set_bc(a, t)
s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
s.set_option("prof_every", 1)
N = s.ne


def run(label, **opt):
    for k in ("s_tx", "s_seg", "s_wsmin"):
        s.set_option(k, opt.get(k, 0))
    s.set_option("s_long", opt.get("s_long", -1)); s.set_option("xcd", opt.get("xcd", -1))
    try:
        s.time_schur_apply(0, 2)
        s.profile_reset()
        ms = s.time_schur_apply(0, reps)
    except RuntimeError as e:
        print(f"{label:44s} ERR {str(e)[:90]}", flush=True); return
    p = {nm: (lambda cnt, m: m / max(cnt, 1))(*s.profile(nm)) for nm in ("schur_x", "schur_y", "schur_z")}
    print(f"{label:44s} apply {ms * 1e3:8.1f} us   x {p['schur_x'] * 1e3:7.1f}  y {p['schur_y'] * 1e3:7.1f} ({48.2 * N / p['schur_y'] / 1e6:5.0f} GB/s alg.)"
          f"  z {p['schur_z'] * 1e3:7.1f} ({48.2 * N / p['schur_z'] / 1e6:5.0f})", flush=True)


for rep in range(2):
    run("classic (s_long=0)", s_long=0)
    run("classic, serial summaries (s_wsmin=100000)", s_long=0, s_wsmin=100000)
    run("chunked default (s_long=-1)")
    run("chunked forced (s_long=1)", s_long=1)
    for tx in (16, 32, 64):
        run(f"chunked s_tx={tx}", s_long=1, s_tx=tx)
    run("chunked, xcd order y+z", s_long=1, xcd=3)
    run("chunked, xcd order off", s_long=1, xcd=0)
s.close()
