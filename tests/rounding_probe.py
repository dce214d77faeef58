#!/usr/bin/env python3
"""Helper of tests/test_rounding_sensitivity.py (run as a subprocess with NF_ORACLE_LIB pointing at one build of oracle/nf_oracle.c):
solves the named benchmark inputs with the reference drivers' settings (set_tol(1e-5, 1e-4, 1e-4, 200, 1000), coarse start;
tests/iaea3d/iaea3d.py:313,321) and writes k, the histories and the flux.   usage: rounding_probe.py <out.npz> <name> [<name> ...]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from helpers import TEST_TOL, load_inputs, make_oracle  # noqa: E402

out = {}
for name in sys.argv[2:]:
    inp = load_inputs(name)
    o = make_oracle(inp)
    o.set_tol(*TEST_TOL)
    k = o.SolveKeff(True, [int(v) for v in inp["coarse_factors"]])
    h = o.history()
    out[name + "_k"] = k; out[name + "_khist"] = h["k"]; out[name + "_cg"] = h["cg"]; out[name + "_phi"] = o.phi_dofs().ravel().copy()
    out[name + "_coarse_outer"] = h["coarse_outer"]
np.savez(sys.argv[1], **out)
