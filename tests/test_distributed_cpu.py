"""world_size-2 (and 3) CPU tests of the multi-GPU algorithm over torch.distributed/gloo: the z-slab partition
method + distributed CG (tests/slab_numpy.py, the pattern neutfem_hip.hip drives through RCCL) against the
single-domain oracle, and the plumbing bench.py uses for N > 1 (plane split, rendezvous, max-over-ranks)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


class GlooComm:
    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def exchange(self, lo, hi):
        """send lo down / hi up, return (from below, from above) -- one plane per interface"""
        r_lo = r_hi = None
        reqs = []
        if lo is not None:
            buf_lo = torch.empty(lo.shape, dtype=torch.float64)
            reqs += [dist.isend(torch.from_numpy(np.ascontiguousarray(lo)), self.rank - 1), dist.irecv(buf_lo, self.rank - 1)]
        if hi is not None:
            buf_hi = torch.empty(hi.shape, dtype=torch.float64)
            reqs += [dist.isend(torch.from_numpy(np.ascontiguousarray(hi)), self.rank + 1), dist.irecv(buf_hi, self.rank + 1)]
        for q in reqs: q.wait()
        if lo is not None: r_lo = buf_lo.numpy()
        if hi is not None: r_hi = buf_hi.numpy()
        return r_lo, r_hi

    def allreduce(self, v):
        t = torch.tensor([v], dtype=torch.float64); dist.all_reduce(t); self.n_allreduce += 1; return float(t.item())

    def allreduce4(self, v):
        t = torch.tensor(v, dtype=torch.float64); dist.all_reduce(t); self.n_allreduce += 1; return [float(a) for a in t]

    n_allreduce = 0


def _worker(rank, world, port, shape, seed, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bench import split_planes
    from helpers import make_oracle, synthetic_inputs
    from slab_numpy import SlabOperator, distributed_cg, distributed_cg_single_reduction
    nx, ny, nz = shape
    inp = synthetic_inputs(nx, ny, nz, 1, seed=seed, dirichlet=(1, 2, 4, 5, 6))
    k0, k1 = split_planes(nz, world)[rank]
    hx, hy, hz = (np.diff(inp[k]) for k in ("x_breaks", "y_breaks", "z_breaks"))
    dirichlet = {int(a): True for a in inp["bc_attr"]}
    op = SlabOperator(hx, hy, hz[k0:k1], inp["D"][0, k0:k1], inp["SigR"][0, k0:k1], dirichlet, rank > 0, rank < world - 1, GlooComm(rank, world))
    rng = np.random.default_rng(5)
    xg = rng.standard_normal((nz, ny, nx)); bg = np.abs(rng.standard_normal((nz, ny, nx)))
    y = op.apply(xg[k0:k1])
    sinv = op.diag_sinv()
    c2 = GlooComm(rank, world); x, its = distributed_cg(op, bg[k0:k1], 1e-10, 2000, c2)
    c1 = GlooComm(rank, world); x1, its1 = distributed_cg_single_reduction(op, bg[k0:k1], 1e-10, 2000, c1)
    # gather on rank 0 and compare with the undivided oracle
    ys = [None] * world; xs = [None] * world; ss = [None] * world; x1s = [None] * world
    dist.gather_object(y, ys if rank == 0 else None); dist.gather_object(x, xs if rank == 0 else None); dist.gather_object(sinv, ss if rank == 0 else None)
    dist.gather_object(x1, x1s if rank == 0 else None)
    tmax = torch.tensor([float(rank + 1)], dtype=torch.float64); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        o = make_oracle(inp)
        o.set_tol(1e-5, 1e-10, 1e-5, 10, 2000)
        yo = o.schur_apply(0, xg.ravel()); xo, _, its_o = o.solve_group(0, bg.ravel())
        ya = np.concatenate(ys).ravel(); xa = np.concatenate(xs).ravel()
        so = o.diag_cache(0); sa = np.concatenate(ss).ravel()
        x1a = np.concatenate(x1s).ravel()
        np.save(out, np.array([np.linalg.norm(ya - yo) / np.linalg.norm(yo), np.linalg.norm(xa - xo) / np.linalg.norm(xo), its, its_o, tmax.item(),
                               np.abs(sa / so - 1).max(), np.linalg.norm(x1a - xo) / np.linalg.norm(xo), its1, c2.n_allreduce, c1.n_allreduce]))
    dist.barrier(); dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,shape", [(2, (7, 6, 12)), (3, (4, 3, 99)), (2, (5, 6, 7))])
def test_slab_partition_method_gloo(world, shape, tmp_path):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), shape, 11, out), nprocs=world, join=True)
    err_apply, err_solve, its, its_o, tmax, err_sinv, err_solve1, its1, nred2, nred1 = np.load(out)
    # 2 slabs: exact for any thickness.  3 slabs: the middle slab (33 planes) is thick enough that its two separators
    # decouple to rounding (0.268^32); thinner middle slabs are refused by the HIP path (test_gpu_slabs.py)
    assert err_apply < 1e-12, err_apply
    assert err_solve < 1e-8 and abs(its - its_o) <= max(2, 0.05 * its_o)
    assert tmax == world                       # max-over-ranks reduction used for the bench timing
    assert err_sinv < 1e-13                    # diagonal-Schur cache on slabs (one edge plane per interface)
    # single-reduction CG (what slab teams run by default): the same answer, the reference's iteration count to within the spread two
    # correct summation orders show on these 170-iteration solves, half the all-reduces
    assert err_solve1 < 1e-8 and abs(its1 - its) <= max(2, 0.03 * its), (err_solve1, its1, its)
    assert nred2 == 2 * its + 1 and nred1 == its1 + 2, (nred2, nred1, its, its1)    # + |b|^2; the single-reduction form pays one apply to see the measured stop


def test_split_planes():
    from bench import split_planes
    assert split_planes(256, 8) == [(32 * i, 32 * i + 32) for i in range(8)]
    p = split_planes(19, 3)
    assert p[0][0] == 0 and p[-1][1] == 19 and all(a[1] == b[0] for a, b in zip(p, p[1:]))
