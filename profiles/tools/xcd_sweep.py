"""Per-CG-iteration time of the launch path (cg_xcd 0) and of k_cg_xcd (1) on synthetic RT0-P0 cubes: where does one XCD stop paying?
fixed work: 3 outers x 2 groups x 60 CG iterations.  usage: xcd_sweep.py [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import synthetic_inputs, make_hip
ns = [int(v) for v in sys.argv[1:]] or [10, 14, 18, 22, 26, 30, 34, 38, 44, 50, 56, 64]
for n in ns:
    inp = synthetic_inputs(n, n, n, 2, seed=3)
    row = []
    for xcd in (0, 1):
        s = make_hip(inp)
        s.set_option("resident", 0); s.set_option("cg_xcd", xcd); s.set_option("cg_xcd_max_cells", 10**9)
        s.set_tol(0.0, 0.0, 1e-4, 3, 60)
        s.solve_keff()
        ts = []
        for _ in range(5):
            s.reset_flux(); t0 = time.perf_counter(); k, no = s.solve_keff(); ts.append(time.perf_counter() - t0)
        its = int(s.history()["cg"].sum())
        row.append((min(ts) * 1e6 / its, k, s.info("xcd_solves")))
        s.close()
    print(f"{n:3d}^3 = {n**3:7d} cells: launches {row[0][0]:7.2f} us per CG iteration (all-in), one XCD {row[1][0]:7.2f}  ratio {row[0][0] / row[1][0]:5.2f}  dk {abs(row[0][1] - row[1][1]):.1e}  xcd solves {row[1][2]}", flush=True)
