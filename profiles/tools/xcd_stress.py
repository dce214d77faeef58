"""k_keff_xcd under uneven load: repeated IAEA-3D 38x38x19 solves (coarse start, drivers' settings) while ANOTHER process streams the 256^3
benchmark through every compute unit of the same GPU.  Every solve must give the bits of the first one (fixed summation order), or --
when the workgroups did not assemble in time and the launch path took over (xcd_refused) -- the launch path's bits.
usage: xcd_stress.py [solves] [whole = 1] [case = iaea3d] [rt = 0]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import TEST_TOL, load_inputs, make_hip
n_solves = int(sys.argv[1]) if len(sys.argv) > 1 else 40
whole = int(sys.argv[2]) if len(sys.argv) > 2 else 1                # 1: k_keff_xcd (whole SolveKeff), 0: k_cg_xcd under the host's outer loop
case = sys.argv[3] if len(sys.argv) > 3 else "iaea3d"; rt = int(sys.argv[4]) if len(sys.argv) > 4 else 0
inp = load_inputs(case); f = [int(v) for v in inp["coarse_factors"]]


def solve(xcd):
    s = make_hip(inp, rt, rt); s.set_tol(*TEST_TOL); s.set_option("cg_xcd", xcd); s.set_option("keff_xcd", whole)
    k, n = s.solve_keff(True, f)
    out = (k, n, s.get_phi().copy(), s.info("xcd_refused"), s.info("last_path"))
    s.close()
    return out


ref_x, ref_l = solve(1), solve(0)                                  # quiet GPU: the two reference results
print(f"quiet: one XCD k = {ref_x[0]:.13f} ({ref_x[1]} outers, path {ref_x[4]}), launches k = {ref_l[0]:.13f} ({ref_l[1]} outers)", flush=True)
load = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "80", "--warmup", "0", "--no-converge", "--no-parity", "--no-small", "--no-c5",
                         "--cpu-sample-iters", "0"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
time.sleep(8.0)                                                    # its set-up (case generation, upload) is over, the timed steps run
same_x = same_l = other = refused = 0; ts = []
for i in range(n_solves):
    t0 = time.perf_counter(); r = solve(1); ts.append(time.perf_counter() - t0)
    busy = load.poll() is None
    if r[3]:
        refused += 1
    if r[0] == ref_x[0] and np.array_equal(r[2], ref_x[2]): same_x += 1
    elif r[0] == ref_l[0] and np.array_equal(r[2], ref_l[2]): same_l += 1
    else:
        other += 1; print(f"solve {i}: k = {r[0]:.13f} differs from both references (refused {r[3]}, path {r[4]}, load running {busy})", flush=True)
    if not busy:
        print(f"the load ended after {i + 1} solves", flush=True); break
load.kill(); load.wait()
print(f"{same_x + same_l + other} solves under load: {same_x} with the one-XCD bits, {same_l} with the launch path's bits ({refused} refused starts), {other} with other bits; "
      f"solve time median {np.median(ts) * 1e3:.1f} ms (quiet: ~22)", flush=True)
sys.exit(1 if other else 0)
