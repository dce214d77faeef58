"""Worker of tests/test_gpu_multiproc.py: one rank of a slab-decomposed solve, all ranks on GPU 0, transport =
tests/fake_rccl (NEUTFEM_RCCL_LIB).  Rendezvous and unique-id broadcast over torch.distributed/gloo, exactly as bench.py.
usage: multiproc_worker.py <rank> <world> <port> <out.npz> <slabs_per_rank> <use_diag> <planes_per_slab> [<nz or 0> [<rt> [<outers>]]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out, per, use_diag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    planes = int(sys.argv[7])
    from neutfem_amd import capi                                # HIP library first, then torch (as bench.py)
    import numpy as np
    import torch
    import torch.distributed as dist
    from bench import split_planes
    from helpers import synthetic_inputs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nx, ny, nz = 10, 8, planes * world * per
    if len(sys.argv) > 8 and int(sys.argv[8]) > 0:
        nz = int(sys.argv[8])                                     # uneven split (e.g. 100 planes on 3 ranks = 33 / 34 / 33)
    rt = int(sys.argv[9]) if len(sys.argv) > 9 else 0            # RT order (P = RT): higher orders exchange one plane per transverse mode
    outers = int(sys.argv[10]) if len(sys.argv) > 10 else 16      # fixed work of the power iteration (half of it on the coarse twin)
    inp = synthetic_inputs(nx, ny, nz, 2, seed=9, dirichlet=(1, 2, 4, 5, 6))
    allp = split_planes(nz, world * per)
    mine = allp[rank * per:(rank + 1) * per]
    k0, k1 = mine[0][0], mine[-1][1]
    dev = rank if os.environ.get("NEUTFEM_WORKER_RANK_IS_DEVICE") else 0      # real RCCL: one GPU per rank; stand-in transport: all ranks on GPU 0
    t = capi.HipTeam(rt, rt, 2, inp["x_breaks"], inp["y_breaks"], inp["z_breaks"], mine, device=dev, below=rank > 0, above=rank < world - 1)
    t.set_linear_solver(6)
    for a, ty in zip(inp["bc_attr"], inp["bc_type"]):
        t.set_bc(int(a), int(ty))
    t.upload_xs_global(inp["D"][:, k0:k1], inp["SigR"][:, k0:k1], inp["NSF"][:, k0:k1], inp["Chi"][:, k0:k1], inp["SigS"][:, :, k0:k1], k_offset=k0)
    t.build()
    idt = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        idt = torch.frombuffer(bytearray(capi.HipTeam.unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(idt, 0)
    t.comm_init(bytes(idt.numpy().tobytes()), world, rank)
    assert t.head.info("n_ranks") == world and t.head.info("rank") == rank
    if os.environ.get("NEUTFEM_TEST_VEC_REDUCE") is not None:     # A/B of the vector all-reduce of block partials (nf_set_option "vec_reduce")
        t.head.set_option("vec_reduce", int(os.environ["NEUTFEM_TEST_VEC_REDUCE"]))
    if os.environ.get("NEUTFEM_TEST_CG1") is not None:            # single-reduction CG on / off (nf_set_option "cg_single_reduce"; default on)
        t.head.set_option("cg_single_reduce", int(os.environ["NEUTFEM_TEST_CG1"]))
    if os.environ.get("NEUTFEM_TEST_XCHG_COMM") is not None:      # interface planes on a communicator of their own
        t.head.set_option("xchg_comm", int(os.environ["NEUTFEM_TEST_XCHG_COMM"]))
    # 1. distributed Schur apply
    xg = np.random.default_rng(4).standard_normal((nz, ny, nx))
    y = t.schur_apply(1, xg[k0:k1]) if rt == 0 else np.zeros((k1 - k0, ny, nx))
    # 2. distributed power iteration (coarse-mesh start on the team, borrowed communicator)
    import time
    if os.environ.get("NEUTFEM_WORKER_LOSE_RANK") is not None:    # a peer that is simply gone (test_a_lost_peer_...): this rank's collectives must time out
        if rank == int(os.environ["NEUTFEM_WORKER_LOSE_RANK"]):
            os._exit(7)
        t.set_tol(1e-12, 1e-9, 1e-9, 16, 2000)
        t0 = time.time()
        try:
            t.solve_keff()
            print(f"rank {rank}: solve returned although a peer is gone", flush=True); os._exit(1)
        except RuntimeError as e:
            print(f"rank {rank}: solve ended after {time.time() - t0:.1f} s: {e}", flush=True)
        try:
            t.solve_keff()
        except RuntimeError as e:
            print(f"rank {rank}: second solve refused: {e}", flush=True)
        t1 = time.time(); t.close()
        print(f"rank {rank}: closed in {time.time() - t1:.2f} s", flush=True)
        os._exit(5)                                               # a fresh, non-zero exit: what include/neutfem_hip.h asks of the caller after NF_ERR_COMM
    t.set_tol(1e-12, 1e-9, 1e-9, outers, 2000)                    # fixed work: `outers` fine outers (half as many coarse ones) with tight inner solves
    t0 = time.time()
    if use_diag == 2:                                             # diagonal solver + CMFD: interface D-tilde, halo planes and dots of the PCG across ranks
        k, n = t.solve_keff(use_diag=True, use_cmfd=True)
    else:
        k, n = t.solve_keff(True, [2, 1, 2], use_diag=bool(use_diag)) if not use_diag else t.solve_keff(use_diag=True)
    print(f"rank {rank}: k = {k:.12f} after {n} outers, {t.history()['cg'].sum()} CG iterations, {time.time() - t0:.1f} s", flush=True)
    phi = t.get_phi_local() if rt == 0 else np.concatenate([s.get_phi().reshape(2, -1) for s in t.slabs], axis=1)
    J = t.get_J_local() if (not use_diag and rt == 0) else None                 # collective: the z currents cross slabs
    ys = [None] * world; ps = [None] * world; ks = [None] * world; js = [None] * world
    vec = t.head.info("vec_reduce")                               # 1: the last CG solve all-reduced the partial vectors themselves (no k_finalize)
    red = t.head.info("cg_reductions")                            # cross-rank reductions per CG iteration of the last CG solve (1: single-reduction CG)
    xc = t.head.info("xchg_comm")
    dist.gather_object(y, ys if rank == 0 else None); dist.gather_object(phi, ps if rank == 0 else None); dist.gather_object((k, n, vec, red, xc), ks if rank == 0 else None)
    dist.gather_object((J, k1 - k0), js if rank == 0 else None)
    if rank == 0:
        extra = {}
        if not use_diag and rt == 0:                              # global Sol_J_ from the ranks' pieces (x | y | z faces, planes stacked)
            xs, ysf, zs = [], [], []
            for r, (Jr, nzr) in enumerate(js):
                nxf, nyf = (nx + 1) * ny * nzr, nx * (ny + 1) * nzr
                xs.append(Jr[:, :nxf]); ysf.append(Jr[:, nxf:nxf + nyf])
                z = Jr[:, nxf + nyf:].reshape(2, nzr + 1, ny * nx)
                zs.append(z if r == world - 1 else z[:, :-1])
            extra["J"] = np.concatenate([np.concatenate(xs, axis=1), np.concatenate(ysf, axis=1), np.concatenate(zs, axis=1).reshape(2, -1)], axis=1)
        np.savez(out, y=np.concatenate(ys, axis=0), phi=np.concatenate(ps, axis=1), k=np.array([v[0] for v in ks]), n=np.array([v[1] for v in ks]),
                 vec=np.array([v[2] for v in ks]), red=np.array([v[3] for v in ks]), xc=np.array([v[4] for v in ks]), x=xg, **extra)
    dist.barrier()
    t.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
