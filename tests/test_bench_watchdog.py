"""bench.py's per-rank watchdog (multi-rank runs): a solve whose outer-iteration count stops moving must end the process with a
non-zero exit code of its own accord -- one dead peer must not pin the other ranks of an 8-GPU node until the driver's limit
(VERDICT r2 item 9).  CPU only: the solver is a stand-in whose progress counter can be frozen."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = """
import sys, time
sys.path.insert(0, {root!r})
import bench
class Stuck:
    def __init__(self, moving): self.n, self.moving = 0, moving
    def progress(self):
        if self.moving: self.n += 1
        return self.n
dog = bench.Watchdog(Stuck({moving}), 0.5, 3)
dog.arm(True)
time.sleep(2.5)
dog.arm(False); dog.stop = True
print("survived")
"""


def _run(moving):
    return subprocess.run([sys.executable, "-c", PROG.format(root=ROOT, moving=moving)], capture_output=True, text=True, timeout=60)


def test_watchdog_ends_a_rank_whose_solve_makes_no_progress():
    r = _run(False)
    assert r.returncode == 3 and "survived" not in r.stdout
    assert "rank 3: no outer iteration completed" in r.stderr


def test_watchdog_leaves_a_progressing_solve_alone():
    r = _run(True)
    assert r.returncode == 0 and "survived" in r.stdout
