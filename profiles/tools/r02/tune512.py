import sys; import os; sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from neutfem_amd import cases
from bench import make_solver
n = int(sys.argv[1]) if len(sys.argv)>1 else 512
case = cases.iaea3d_resampled(n)
s = make_solver(case, 0)
def run(**opt):
    for k in ("s_tx","s_seg"): s.set_option(k, opt.get(k, 0))
    s.profile_reset()
    try:
        ms = s.time_schur_apply(0, 10)
    except RuntimeError as e:
        print(opt, "ERR", str(e)[:80], flush=True); return
    p = {nm: (lambda c,m: m/max(c,1))(*s.profile(nm)) for nm in ("schur_x","schur_y","schur_z")}
    print(f"{opt!s:35s} apply {ms*1e3:7.1f} us  x {p['schur_x']*1e3:6.1f} y {p['schur_y']*1e3:6.1f} z {p['schur_z']*1e3:6.1f}", flush=True)
run()
for seg in (8, 16, 32):
    for tx in (8, 16, 32, 64):
        if tx*((n+seg-1)//seg) <= 1024: run(s_tx=tx, s_seg=seg)
