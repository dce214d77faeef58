"""k_keff_xcd / k_cg_xcd against the launch path on random shapes (forced onto every mesh they can take): the planner's integer edge cases --
wavefront split among the roles, sub-tile packing, rounds, chunks per lane, odd lengths, 1D / 2D / 3D, RT0-P0 and RT1-P0.
fixed work: 4 outers, CG to 1e-11; k to 1e-10, flux to 1e-9 between the routes.  usage: xcd_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_hip, rel_l2, synthetic_inputs
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; ran = 0
for c in range(cases):
    dim = int(rng.choice([1, 2, 2, 3, 3, 3]))
    nx = int(rng.integers(2, 129)); ny = int(rng.integers(2, 70)) if dim >= 2 else 1; nz = int(rng.integers(2, 48)) if dim == 3 else 1
    while nx * ny * nz > 60000:
        if nz > 2: nz = max(2, nz // 2)
        elif ny > 2: ny = max(2, ny // 2)
        else: nx = max(2, nx // 2)
    if nx * ny * nz < 220:                                          # below 200 unknowns the explicit-S branch takes over (no CG at all): 1D meshes
        ny = max(ny, 3)                                             # of at most 128 cells never reach the CG, so a tiny case becomes 2D
        if nx * ny * nz < 220: nx = max(nx, 80)
    ng = int(rng.integers(1, 4)); rt = int(rng.choice([0, 0, 1, 1, 2])); pm = int(rng.choice([0, rt])) if rt else 0
    if pm > 0:                                                      # (m + 1)^dim unknowns per cell: keep the case small
        while nx * ny * nz * (pm + 1) ** dim > 60000:
            if nz > 2: nz = max(2, nz // 2)
            elif ny > 2: ny = max(2, ny // 2)
            else: nx = max(2, nx // 2)
    inp = synthetic_inputs(nx, ny, nz, ng, seed=100 + c)
    tol = (0.0, 1e-11, 1e-11, 4, 4000)
    res = {}
    for name, opts in (("launches", dict(cg_xcd=0)), ("cg", dict(cg_xcd=1, keff_xcd=0)), ("keff", dict(cg_xcd=1, keff_xcd=1))):
        s = make_hip(inp, rt, pm); s.set_tol(*tol); s.set_option("resident", 0); s.set_option("cg_xcd_min_cells", 0); s.set_option("cg_xcd_max_cells", 1 << 30)
        for k_, v_ in opts.items():
            s.set_option(k_, v_)
        k, n = s.solve_keff()
        res[name] = (k, s.get_phi().copy(), s.info("xcd_solves"), s.info("last_path"), s.info("xcd_refused"), int(s.history()["cg"].sum()))
        s.close()
    ref = res["launches"]
    line = f"case {c:3d}: {nx:3d} x {ny:2d} x {nz:2d} RT{rt}-P{pm} {ng}g  CG {ref[5]:5d}"
    for name in ("cg", "keff"):
        r = res[name]
        dk = abs(r[0] - ref[0]) / abs(ref[0]); dphi = rel_l2(r[1], ref[1])
        ok = dk < 1e-10 and dphi < 1e-9 and r[4] == 0 and r[2] > 0 and (r[3] == 3) == (name == "keff")
        line += f" | {name}: dk {dk:.1e} dphi {dphi:.1e} solves {r[2]} path {r[3]}{'' if ok else '  <-- BAD'}"
        bad += 0 if ok else 1
    ran += 1
    print(line, flush=True)
print(f"{ran} cases, {bad} bad", flush=True)
sys.exit(1 if bad else 0)
