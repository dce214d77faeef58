#!/usr/bin/env python3
"""Converged oracle runs of the heavy tight-tolerance parity cases -> tests/golden/oracle_cache/<key>.npz (see tests/helpers.py: solved_oracle).
Outputs of oracle/nf_oracle.c only: k, outer count, flux DOFs, current DOFs, k history, CG counts.  The key hashes the inputs, the settings and
nf_oracle.c itself, so a stale file is never read -- after editing the oracle or an input generator run this again (about 6 minutes of 6 cores).
Usage: python tests/golden/make_oracle_cache.py [--missing]"""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def cases():
    """(label, kwargs of helpers.solved_oracle without the input) + a function building the input"""
    import test_gpu_more, test_gpu_orders, test_gpu_paths, test_gpu_slabs
    from helpers import synthetic_inputs
    out = []
    for shape, rt, p, ng in test_gpu_paths.TIGHT_SHAPES:
        out.append((f"paths tight {shape} RT{rt}-P{p} {ng}g", ("synthetic", shape, ng, 7), dict(rt=rt, p=p, tol=test_gpu_paths.TIGHT)))
    for rt, p, shape in test_gpu_orders.ORDER_CASES:
        out.append((f"orders {shape} RT{rt}-P{p}", ("synthetic", shape, 2, 5 + rt + p), dict(rt=rt, p=p, tol=test_gpu_orders.ORDER_TOL, want_J=False)))
    for shape, _planes in test_gpu_slabs.HO_SHAPES:
        for rt, p in test_gpu_slabs.HO_ORDERS:
            out.append((f"team solve {shape} RT{rt}-P{p}", ("synthetic5", shape, 2, rt + 7 * p + shape[2]), dict(rt=rt, p=p, tol=test_gpu_slabs.HO_SOLVE_TOL, want_J=False)))
    out.append(("checkerboard 24^3 x 8 groups", ("checker", 24, 8), dict(rt=0, p=0, tol=test_gpu_more.CHECKER_TOL, coarse=[2, 2, 2], want_J=False)))
    out.append(("team solve 8 x 6 x 96", ("synthetic", (8, 6, 96), 2, 9), dict(rt=0, p=0, tol=test_gpu_slabs.TEAM_TOL, want_J=False)))
    return out


def build_input(spec):
    from helpers import synthetic_inputs
    if spec[0] == "synthetic":
        return synthetic_inputs(*spec[1], ng=spec[2], seed=spec[3])
    if spec[0] == "synthetic5":                                  # five Dirichlet sides (the slab tests of the higher orders)
        return synthetic_inputs(*spec[1], ng=spec[2], seed=spec[3], dirichlet=(1, 2, 3, 5, 6))
    import test_gpu_more
    from neutfem_amd import cases as gen
    return test_gpu_more._from_case(gen.synthetic_checkerboard(spec[1], spec[2]))


def run(job):
    import time
    from helpers import solved_oracle
    label, spec, kw = job
    t0 = time.time()
    r = solved_oracle(build_input(spec), write=True, **kw)
    return f"{label}: k = {r.k:.12f}, {r.n_outer} outers, {int(r.hist_cg.sum())} CG iterations, {time.time() - t0:.0f} s"


if __name__ == "__main__":
    from helpers import ORACLE_CACHE
    from helpers import _oracle_key
    jobs = cases()
    if "--missing" in sys.argv:                                   # only the cases that have no file under today's key
        jobs = [j for j in jobs if not os.path.exists(os.path.join(ORACLE_CACHE, _oracle_key(build_input(j[1]), j[2]["rt"], j[2]["p"], j[2]["tol"], j[2].get("coarse")) + ".npz"))]
    elif os.path.isdir(ORACLE_CACHE):
        for f in os.listdir(ORACLE_CACHE):
            os.remove(os.path.join(ORACLE_CACHE, f))
    with mp.Pool(int(os.environ.get("NEUTFEM_SWEEP_PROCS", "6"))) as pool:
        for line in pool.imap_unordered(run, jobs):
            print(line, flush=True)
