#!/usr/bin/env python3
"""does MFMA beat plain FMA where the work is a dense contraction?  Times nf_local_matrices (LocalMatrices::Compute on the
device, one element per workgroup) in both variants for every order in 3D and prints elements/s and the fp64 rate of the contractions.
usage: assembly_bench.py [n_elems]"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import make_hip, synthetic_inputs  # noqa: E402

n_el = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
print("# nf_local_matrices: A_d = Psi_d^T W Psi_d (3 directions), B = Phi^T W divPsi, C = Phi^T W Phi per element, one element per 256-thread workgroup")
print("# order     dim  nq^dim  nJ   nP   flops/element   FMA: ms  Melem/s  GFLOP/s    MFMA f64: ms  Melem/s  GFLOP/s   (GFLOP/s counts the padded MFMA tiles' useful part only)")
for dim, shape in ((3, (32, 32, 16)), (2, (128, 128, 1))):
    for rt, p in ((0, 0), (1, 1), (2, 2)):
        inp = synthetic_inputs(*shape, ng=1, seed=1)
        s = make_hip(inp, rt, p)
        ne = s.ne
        elems = (np.arange(n_el) * 7919) % ne
        k = rt; nf, ni = (k + 1) ** (dim - 1), k * (k + 1) ** (dim - 1)
        nper, nP = 2 * nf + ni, (p + 1) ** dim
        nq = (3 if rt == 0 else 5) ** dim
        flops = 2.0 * nq * (dim * nper * nper + nP * dim * nper + nP * nP) + nq * (dim * nper + nP)
        row = f"RT{rt}-P{p}   {dim}    {nq:4d}   {dim * nper:3d}  {nP:3d}   {flops:12.0f}  "
        for variant in (0, 1):
            _, _, _, ms = s.local_matrices(0, elems, variant, reps=5)
            row += f"   {ms:8.3f}  {n_el / ms / 1e3:7.2f}  {flops * n_el / ms / 1e6:8.1f}      "
        print(row, flush=True)
        s.close()
