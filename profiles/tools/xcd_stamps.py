"""Phase times inside k_cg_xcd (library built with -DNF_XSTAMPS, NEUTFEM_HIP_LIB): s_memrealtime (100 MHz) sums of workgroup 0, thread 0.
usage: NEUTFEM_HIP_LIB=profiles/tools/_ab/libneutfem_hip_xstamps.so python profiles/tools/xcd_stamps.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from neutfem_amd.capi import HipSolver
names = ["phase A (roles)", "block sum + partial store", "barrier 1", "sum of p.q partials", "phase B (r update) + block sum", "barrier 2", "sum of |r|^2 partials"]
for label, name, coarse in [("IAEA-3D 38x38x19, no coarse start", "iaea3d", False), ("IAEA-2D 38x38, resident off, no coarse start", "iaea2d", False)]:
    z = np.load(os.path.join(ROOT, "tests", "golden", f"inputs_{name}.npz"))
    ng = int(z["ng"])
    s = HipSolver(0, 0, ng, z["x_breaks"], z["y_breaks"], z["z_breaks"], 0)
    s.set_linear_solver(6)
    for at, ty in zip(z["bc_attr"], z["bc_type"]):
        s.set_bc(int(at), int(ty))
    s.set_option("resident", 0); s.set_option("keff_xcd", 0); s.set_option("cg_xcd_min_cells", 0)   # the stamps sit in k_cg_xcd (the CG under the host's outer loop)
    s.upload_xs(z["D"], z["SigR"], z["NSF"], z["Chi"], z["SigS"]); s.build()
    s.set_tol(1e-5, 1e-4, 1e-4, 200, 1000)
    k, n = s.solve_keff(False, [], False)
    buf = (C.c_double * 512)()
    s.L.nf_debug_xcd_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]
    s.L.nf_debug_xcd_buffer(s.h, buf, 512, 0)
    b = np.array(buf[384:396])
    its, solves = b[7], b[8]
    print(f"{label}: k = {k:.10f}, {n} outers, {int(its)} CG iterations in {int(solves)} solves; P = {int(b[9])}, rounds {int(b[10])}, x wavefronts per round {int(b[11])}")
    for i, nm in enumerate(names):
        print(f"    {nm:34s} {b[i] / 100.0 / max(its, 1):7.3f} us per iteration")
    print(f"    {'total':34s} {b[:7].sum() / 100.0 / max(its, 1):7.3f} us per iteration", flush=True)
    s.close()
