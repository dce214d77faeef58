#!/usr/bin/env python3
"""Oracle-vs-oracle spread at 128^3 with the drivers' settings: the committed golden run (tests/golden/golden_iaea3d_128cube.json, made
by the -ffp-contract=off build of oracle/nf_oracle.c) against the same run of the -ffp-contract=fast build of the same source.
Adds the entry "iaea3d_128cube_driver" to tests/golden/rounding_spread.json (the bar test_iaea3d_128cube_golden derives from).
~25 minutes of one core.  usage: python tests/golden/make_rounding_spread_128cube.py"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if "NF_ORACLE_LIB" not in os.environ:
    tmp = tempfile.mkdtemp()
    lib = os.path.join(tmp, "libnf_oracle_fast.so")
    subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fPIC", "-std=c99", "-fno-fast-math", "-shared", "-ffp-contract=fast", "-o", lib,
                           os.path.join(ROOT, "oracle", "nf_oracle.c"), "-lm"])
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, NF_ORACLE_LIB=lib)))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
from make_golden_128cube import oracle  # noqa: E402
from neutfem_amd import cases  # noqa: E402

with open(os.path.join(HERE, "golden_iaea3d_128cube.json")) as f:
    r = json.load(f)["runs"]["driver"]
o = oracle(cases.iaea3d_resampled(128))
o.set_tol(*r["tol"])
k = o.SolveKeff(True, r["factors"])
h = o.history(); phi = o.phi_dofs().ravel()[::r["phi_stride"]]; ref = np.array(r["phi_samples"])
rec = dict(flux_rel_l2=float(np.linalg.norm(phi - ref) / np.linalg.norm(ref)), k_pcm=float(1e5 * abs(k - r["keff"]) / r["keff"]),
           outers=[int(r["n_outer"]), int(h["n_outer"])], cg=[int(np.sum(r["cg"])), int(h["cg"].sum())])
print(rec, flush=True)
path = os.path.join(HERE, "rounding_spread.json")
with open(path) as f:
    all_ = json.load(f)
all_["iaea3d_128cube_driver"] = rec
with open(path, "w") as f:
    json.dump(all_, f, indent=1)
