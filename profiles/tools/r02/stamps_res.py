import sys, os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
from neutfem_amd import capi
capi.lib_path = lambda: os.path.join(R, "scratch", "libneutfem_stamps.so")
from helpers import *
import time
for name, rt in (("iaea2d", 0), ("koeberg2d", 1), ("iaea2d", 1), ("koeberg2d", 0)):
    inp = load_inputs(name)
    s = make_hip(inp, rt, rt); s.set_tol(*TEST_TOL)
    s.solve_keff(False, (), False); s.reset_flux()
    t = time.perf_counter(); k, n = s.solve_keff(False, (), False); dt = time.perf_counter() - t
    print(name, rt, "path", s.info("last_path"), "serial", s.info("last_resident_serial"), "outers", n, "cg", s.history()["cg"].sum(), f"{dt*1e3:.2f} ms", f"{dt/s.history()['cg'].sum()*1e6:.2f} us/it", f"k={k:.9f}", flush=True)
    s.close()
