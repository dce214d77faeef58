#!/usr/bin/env python3
"""Mesh-refinement sweep of the CPU oracle against the literature k_ref scalars the reference's drivers hold
(tests/iaea2d/iaea2d.py:39, tests/biblis2d/biblis2D.py:39, tests/koeberg2d/koeberg2d.py:40) -- test infrastructure.

For every case and element order the oracle is run on the drivers' own input (tests/golden/inputs_*.npz: 2 cells per assembly)
coarsened / refined to m = 1, 2, 4, 8(, 16) cells per assembly at tight tolerances, and the h -> 0 limit of each order is
Richardson-extrapolated from its three finest meshes: p = log2((k_a - k_b) / (k_b - k_c)), k_inf = k_c + (k_c - k_b) / (2^p - 1).
What this pins: the three orders, which share nothing but the reference's formulas for their local matrices, must extrapolate to
the SAME limit (to ~1 pcm), and that limit's distance from the literature value is the benchmark's own uncertainty, stated per case.
Writes tests/golden/kref_richardson.json and prints the table of DESIGN.md section 2.  About 40 minutes on 6 cores.
Usage: python oracle/sweep_kref.py [--quick]"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.dirname(os.path.abspath(__file__))]   # `oracle` must be the package, not oracle/oracle.py
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {"iaea2d": 1.029585, "biblis2d": 1.02511, "koeberg2d": 1.007954}
LADDER = {0: (2, 4, 8, 16), 1: (1, 2, 4, 8), 2: (1, 2, 4)}
TOL = (1e-10, 1e-9, 1e-9, 2000, 5000)


def resample(inp, m):
    """m cells per assembly from the drivers' 2-cells-per-assembly input (piecewise-constant cross sections)"""
    out = dict(inp)
    for k in ("D", "SigR", "NSF", "Chi", "SigS"):
        a = inp[k][..., ::2, ::2]                                 # one value per assembly
        out[k] = np.ascontiguousarray(np.repeat(np.repeat(a, m, axis=-1), m, axis=-2))
    for k in ("x_breaks", "y_breaks"):
        b = inp[k][::2]; out[k] = np.interp(np.arange((len(b) - 1) * m + 1) / m, np.arange(len(b)), b)
    return out


def run(job):
    name, rt, m = job
    from helpers import load_inputs, make_oracle
    inp = resample(load_inputs(name), m)
    o = make_oracle(inp, rt, rt); o.set_tol(*TOL)
    t0 = time.time(); k = o.SolveKeff(); dt = time.time() - t0
    return name, rt, m, k, o.info("last_outer"), dt


def richardson(ks):
    a, b, c = ks[-3:]
    d1, d2 = a - b, b - c
    if d1 == 0 or d2 == 0 or d1 * d2 < 0:
        return None, None                                         # not monotone: not in the asymptotic range
    p = np.log2(d1 / d2)
    return float(p), float(c + (c - b) / (2.0 ** p - 1.0))


def main():
    quick = "--quick" in sys.argv
    jobs = [(n, rt, m) for n in CASES for rt, ms in LADDER.items() for m in (ms[:3] if quick and rt == 0 else ms[:2] if quick else ms)]
    jobs.sort(key=lambda j: -(j[2] ** 2) * (j[1] + 1) ** 2)         # big ones first
    with mp.Pool(int(os.environ.get("NEUTFEM_SWEEP_PROCS", "6"))) as pool:
        res = []
        for r in pool.imap_unordered(run, jobs):
            res.append(r); print(f"  {r[0]:10s} RT{r[1]}-P{r[1]} m={r[2]:2d}: k={r[3]:.9f} ({r[4]} outers, {r[5]:.0f} s)", flush=True)
    doc = {}
    for name, kref in CASES.items():
        doc[name] = dict(kref=kref, orders={})
        for rt in LADDER:
            rows = sorted((m, k) for n, r, m, k, _, _ in res if n == name and r == rt)
            ms = [m for m, _ in rows]; ks = [k for _, k in rows]
            p, kinf = richardson(ks) if len(ks) >= 3 else (None, None)
            doc[name]["orders"][f"RT{rt}-P{rt}"] = dict(cells_per_assembly=ms, keff=ks, pcm_vs_kref=[1e5 * (1 / kref - 1 / k) for k in ks],
                                                       observed_order=p, k_limit=kinf, limit_pcm_vs_kref=None if kinf is None else 1e5 * (1 / kref - 1 / kinf))
    if not quick:
        with open(os.path.join(ROOT, "tests", "golden", "kref_richardson.json"), "w") as f:
            json.dump(dict(tolerances=list(TOL), cases=doc), f, indent=1)
    print("\n| case | order | cells per assembly: pcm vs k_ref | observed order | Richardson limit | limit - k_ref (pcm) |\n|---|---|---|---|---|---|")
    for name, d in doc.items():
        for od, v in d["orders"].items():
            seq = ", ".join(f"{m}: {x:+.2f}" for m, x in zip(v["cells_per_assembly"], v["pcm_vs_kref"]))
            print(f"| {name} | {od} | {seq} | {'-' if v['observed_order'] is None else format(v['observed_order'], '.2f')} | "
                  f"{'-' if v['k_limit'] is None else format(v['k_limit'], '.7f')} | {'-' if v['k_limit'] is None else format(v['limit_pcm_vs_kref'], '+.2f')} |")


if __name__ == "__main__":
    main()
