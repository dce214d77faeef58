"""What the one all-reduce per CG iteration costs through the REAL librccl on one GPU: IAEA-3D 256^3 as 8 slabs in one process (loopback), the
same fixed work (2 outers, CG to 1e-4) with the five doubles reduced by the library's own kernels (default) and through ncclAllReduce on a 1-rank
communicator (NEUTFEM_FORCE_RCCL=1: host-side enqueue + RCCL's kernel, no peer) -- a LOWER bound of what a rank pays on 8 GPUs.
usage: python profiles/tools/r04_real_rccl_one_rank.py   (on the GPU box; one process per variant, the switch is read at comm_init)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from neutfem_amd import capi, cases
    from bench import split_planes
    c = cases.iaea3d_resampled(256)
    t = capi.HipTeam(0, 0, c["ng"], c["x_breaks"], c["y_breaks"], c["z_breaks"], split_planes(256, 8))
    t.set_linear_solver(6)
    for a, ty in c["bc"]:
        t.set_bc(a, ty)
    t.upload_xs_global(c["D"], c["SigR"], c["NSF"], c["Chi"], c["SigS"]); t.build()
    if sys.argv[1] == "rccl":
        t.comm_init(capi.HipTeam.unique_id(), 1, 0)
    t.set_tol(0.0, 0.0, 1e-4, 1, 1000); t.solve_keff()           # warm-up outer
    t.set_tol(0.0, 0.0, 1e-4, 2, 1000)
    t0 = time.perf_counter(); k, n = t.solve_keff(); dt = time.perf_counter() - t0
    its = int(t.history()["cg"].sum())
    print(f"{sys.argv[1]:6s}: {n} outers, {its} CG iterations, {dt * 1e3:8.1f} ms -> {dt / its * 1e6:7.1f} us per CG iteration; reductions per iteration {t.head.info('cg_reductions')}, k = {k:.10f}", flush=True)
    t.close()
else:
    for v in ("own", "rccl", "own", "rccl"):
        e = dict(os.environ)
        if v == "rccl": e["NEUTFEM_FORCE_RCCL"] = "1"
        else: e.pop("NEUTFEM_FORCE_RCCL", None)
        subprocess.run([sys.executable, os.path.abspath(__file__), v], env=e, check=True)
