"""in-kernel cycle stamps of k_apply3 (diagnostic library scratch/libneutfem_stamps.so)"""
import sys, os, ctypes as C; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from neutfem_amd import capi
capi.lib_path = lambda: os.path.join(R, "scratch", "libneutfem_stamps.so")
from helpers import *
name, rt = sys.argv[1], int(sys.argv[2])
inp = load_inputs(name)
s = make_hip(inp, rt, rt); s.set_tol(*TEST_TOL); s.set_option("resident", 0)
s.solve_keff(False, (), False)
L = capi.load(); L.nf_debug_stamps.argtypes = [C.POINTER(C.c_longlong)]
acc = []
for rep in range(6):
    s.reset_flux(); s.set_tol(1e-5, 1e-4, 1e-4, 2 + rep, 1000); s.solve_keff(False, (), False)
    buf = (C.c_longlong * 48)(); assert L.nf_debug_stamps(buf) == 0
    acc.append(np.array(buf[:], dtype=np.int64).reshape(3, 16))
lab = ["entry", "done-flag", "prologue(rr)", "-", "s:pre-bar1", "s:post-bar1", "s:pre-bar2", "s:post-bar2", "body-end", "blocksum-end"]
for role, rn in enumerate(["x block", "y block", "z block"]):
    print(rn)
    for a in acc[-3:]:
        t = a[role]; t0 = t[0]
        real = (t[13] - t[12]) * 10.0      # ns (100 MHz)
        cyc = t[9] - t[0]
        print("   ", " ".join(f"{lab[i]}:{t[i]-t0}" for i in (1, 2, 4, 5, 6, 7, 8, 9) if t[i] != 0), f"| total {cyc} cyc in {real:.0f} ns -> {cyc/real if real else 0:.2f} GHz")
s.close()
