"""round 4, call g: higher orders at sizes where the chip is busy.  (i) bench.py's extra: IAEA-3D 128^3 RT1-P1 Schur apply; (ii) RT1 / RT2
y and z passes on lines beyond 256 cells (the 8-cell-segment instantiations that still spill) with 4- and 8-cell segments."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bench import higher_order_throughput
from helpers import make_hip, make_oracle, rel_l2, synthetic_inputs
print(json.dumps(higher_order_throughput(0)))
print(json.dumps(higher_order_throughput(0, 96, 2)))
for shape in ((48, 512, 48), (48, 48, 512), (32, 400, 64)):
    for rt in (1, 2):
        inp = synthetic_inputs(*shape, 1, seed=3)
        for seg in (0, 4, 8):
            s = make_hip(inp, rt, rt)
            s.set_option("s_seg", seg)
            ms = s.time_schur_apply(0, 10)
            ps = {nm: (lambda c, t: round(t / c, 4) if c else None)(*s.profile(nm)) for nm in ("schur_x", "schur_y", "schur_z")}
            alg = 24.0 * s.n_phi + 40.0 * s.n_J
            print(f"shape {shape} RT{rt}-P{rt} s_seg={seg}: apply {ms:.3f} ms = {alg / ms / 1e6:.0f} GB/s (8d bytes), passes {ps}", flush=True)
            s.close()
