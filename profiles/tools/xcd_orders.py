"""One-XCD kernels against the launch path for orders with bubble moments: the drivers' settings (coarse start) on the 2D benchmarks at RT1-P1 /
RT2-P2 with the resident kernels off where they would take the mesh.  usage: xcd_orders.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import TEST_TOL, load_inputs, make_hip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, rt in [("iaea2d", 1), ("koeberg2d", 1), ("biblis2d", 1), ("iaea2d", 2), ("zion2d", 1)]:
    inp = load_inputs(name); f = [int(v) for v in inp["coarse_factors"]]
    res = {}
    for xcd in (0, 1):
        s = make_hip(inp, rt, rt); s.set_tol(*TEST_TOL); s.set_option("resident", 0); s.set_option("cg_xcd", xcd); s.set_option("cg_xcd_max_cells", 1 << 30)
        s.solve_keff(True, f)
        ts = []
        for _ in range(reps):
            s.reset_flux(); t0 = time.perf_counter(); k, n = s.solve_keff(True, f); ts.append((time.perf_counter() - t0) * 1e3)
        res[xcd] = (k, n, int(s.history()["cg"].sum()), min(ts), s.info("last_path"), s.get_phi().copy(), s.ne)
        s.close()
    a, b = res[0], res[1]
    unk = a[5].size // int(inp["ng"])
    print(f"{name:10s} RT{rt}-P{rt} {a[6]:5d} cells {unk:6d} unknowns/group: launches {a[3]:8.2f} ms (CG {a[2]}), one XCD {b[3]:8.2f} ms (CG {b[2]}, path {b[4]})  x{a[3] / b[3]:.2f}  "
          f"dk {abs(a[0] - b[0]):.1e} dphi {np.linalg.norm(a[5] - b[5]) / np.linalg.norm(a[5]):.1e}", flush=True)
