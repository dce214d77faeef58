import sys, time, os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from neutfem_amd import capi, cases
if sys.argv[1] == "old": capi.lib_path = lambda: os.path.join(R, "scratch", "libneutfem_old.so")
from helpers import *
def run(name, rt, p, coarse, diag, reps=5):
    inp = load_inputs(name); f=[int(v) for v in inp["coarse_factors"]]
    s = make_hip(inp, rt, p); s.set_tol(*TEST_TOL)
    s.solve_keff(coarse, f, diag)
    ts=[]
    for _ in range(reps):
        s.reset_flux(); t=time.perf_counter(); k,n = s.solve_keff(coarse, f, diag); ts.append(time.perf_counter()-t)
    h = s.history(); cg = int(h['cg'].sum())
    print(f"{sys.argv[1]} {name:10s} RT{rt}P{p} coarse={coarse!s:5} outers={n:3d} cg={cg:5d} GPU {min(ts)*1e3:8.2f} ms ({min(ts)/max(cg,1)*1e6:6.2f} us/CG-it) path={s.info('last_path') if hasattr(s,'info') else ''}", flush=True)
    s.close()
run("iaea3d",0,0,True,False); run("iaea3d",0,0,False,False); run("iaea2d",0,0,True,False); run("koeberg2d",1,1,True,False)
for n in (64,):
    c = cases.iaea3d_resampled(n)
    s = capi.HipSolver(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], 0)
    s.set_linear_solver(6)
    for a, t in c["bc"]: s.set_bc(a, t)
    s.upload_xs(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); s.build()
    s.set_tol(0.0, 1e-4, 1e-4, 1, 1000); s.solve_keff()
    best = 1e9
    for rep in range(3):
        s.set_tol(0.0, 1e-4, 1e-4, 3, 1000); t0=time.perf_counter(); k,no = s.solve_keff(); dt=time.perf_counter()-t0
        cg = s.history()["cg"].sum(); best = min(best, dt/cg*1e6)
    print(sys.argv[1], n, "us per CG it", round(best,2), flush=True)
    s.close()
