"""run small configs once warm + once timed (for rocprofv3 --kernel-trace); usage: small_trace.py name rt coarse"""
import sys, time, os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from helpers import *
name, rt, coarse = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
inp = load_inputs(name); f=[int(v) for v in inp["coarse_factors"]]
s = make_hip(inp, rt, rt); s.set_tol(*TEST_TOL)
s.solve_keff(bool(coarse), f, False)
s.reset_flux(); t=time.perf_counter(); k,n = s.solve_keff(bool(coarse), f, False); dt=time.perf_counter()-t
h = s.history(); cg=int(h['cg'].sum())
print(f"{name} RT{rt} cells={s.ne} outers={n} cg={cg} {dt*1e3:.2f} ms {dt/max(cg,1)*1e6:.1f} us/it k={k:.9f}")
s.close()
