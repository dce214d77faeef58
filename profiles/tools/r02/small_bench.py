import sys, time, os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from helpers import *
def run(name, rt, p, coarse, diag, reps=3):
    inp = load_inputs(name); f=[int(v) for v in inp["coarse_factors"]]
    s = make_hip(inp, rt, p); s.set_tol(*TEST_TOL)
    s.solve_keff(coarse, f, diag)            # warm
    ts=[]
    for _ in range(reps):
        s.reset_flux(); t=time.perf_counter(); k,n = s.solve_keff(coarse, f, diag); ts.append(time.perf_counter()-t)
    h = s.history(); cg = int(h['cg'].sum()); co = h['coarse_outer']
    o = make_oracle(inp, rt, p); o.set_tol(*TEST_TOL); t=time.perf_counter(); ko = o.SolveKeff(coarse, f if coarse else [], diag); to=time.perf_counter()-t
    tg=min(ts)
    print(f"{name:10s} RT{rt}P{p} coarse={coarse!s:5} diag={diag!s:5} cells={s.ne:6d} outers={n:3d}(+{co}) cg={cg:5d}  GPU {tg*1e3:8.1f} ms = {n/tg:7.1f} outer/s ({tg/max(cg,1)*1e6:6.1f} us/CG-it) | CPU oracle {to*1e3:8.1f} ms = {o.info('last_outer')/to:7.1f} outer/s  k={k:.8f}", flush=True)
    s.close()
run("iaea3d",0,0,True,False); run("iaea3d",0,0,False,False); run("iaea3d",0,0,False,True)
run("iaea2d",0,0,True,False); run("koeberg2d",1,1,True,False); run("iaea2d",1,1,True,False)
