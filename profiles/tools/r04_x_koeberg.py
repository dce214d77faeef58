"""round 4, call x: KOEBERG-2D RT1-P1 (BASELINE config 2) on its three possible shapes: resident one-workgroup kernel (default), one-XCD kernel, launch path"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import load_inputs, make_hip
for name, rt in (("koeberg2d", 1), ("iaea2d", 0), ("iaea2d", 1)):
    inp = load_inputs(name); f = [int(v) for v in inp["coarse_factors"]]
    for label, opts in (("resident (default)", {}), ("one XCD", dict(resident=0)), ("launch path", dict(resident=0, keff_xcd=0, cg_xcd=0))):
        s = make_hip(inp, rt, rt); s.set_tol(1e-5, 1e-4, 1e-4, 200, 1000)
        for k_, v in opts.items(): s.set_option(k_, v)
        s.solve_keff(True, f)
        best = 1e9
        for _ in range(5):
            s.reset_flux(); t0 = time.perf_counter(); k, n = s.solve_keff(True, f); best = min(best, time.perf_counter() - t0)
        print(f"{name} RT{rt}-P{rt} {label:20s}: {best * 1e3:7.2f} ms  k {k:.10f} outers {n} CG {int(s.history()['cg'].sum())} path {s.info('last_path')}", flush=True)
        s.close()
