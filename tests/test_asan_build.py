"""Checks that only mean something under `make test-asan` (ASan + UBSan builds of the CPU-side native code):
the sanitized builds are really the ones loaded, the fake RCCL transport exports what libneutfem_hip.so binds, and the pybind11
host module survives a set-up / tear-down cycle without a GPU.  Skipped in the plain CPU suite."""
import ctypes
import os

import numpy as np
import pytest

ASAN = "NF_ORACLE_LIB" in os.environ and "asan" in os.environ.get("NF_ORACLE_LIB", "")
pytestmark = pytest.mark.skipif(not ASAN, reason="sanitizer builds only (make test-asan)")


def _maps():
    with open("/proc/self/maps") as f:
        return f.read()


def test_sanitized_oracle_is_the_one_loaded():
    from oracle.oracle import OracleNeutFEM
    o = OracleNeutFEM(1, 1, 2, np.linspace(0, 4, 5), np.linspace(0, 3, 4), np.array([0.0]))
    o.set_linear_solver(6)
    for a in (1, 2, 3, 4):
        o.set_bc(a, 0, 0.0)
    o.get_NSF()[...] = 0.02
    o.BuildMatrices(); o.set_tol(1e-8, 1e-8, 1e-8, 50, 200)
    assert np.isfinite(o.SolveKeff())
    m = _maps()
    assert "build/asan/libnf_oracle.so" in m and "libasan" in m
    del o


def test_fake_rccl_exports_the_bound_entry_points():
    lib = ctypes.CDLL(os.environ["NEUTFEM_ASAN_FAKE_RCCL"])
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd",
                 "ncclGetErrorString"):
        assert hasattr(lib, name), name


def test_host_module_setup_teardown_without_gpu():
    import neutfem_amd
    neutfem_amd.install_compat()
    import neutfem._neutfem_eigen as ns
    assert "build/asan/neutfem/_neutfem_eigen" in _maps()
    for _ in range(3):
        m = ns.NeutFEM(1, 2, np.linspace(0, 4, 9), np.linspace(0, 3, 7), np.array([0.0]))
        m.set_verbosity(ns.VerbosityLevel.SILENT)
        m.set_bc(1, ns.BCType.DIRICHLET, 0.0)
        d = m.get_D(); d[...] = 1.3
        assert m.get_D()[1, 2, 3] == 1.3 and m.get_SigS().shape == (2, 2, 6, 8) and m.get_flux().shape == (2, 6, 8)
        v = m.get_NSF()
        del m
        v[...] = 0.5                                   # the view keeps the solver alive (src/NeutFEM.cpp:2643)
        assert v.sum() == 0.5 * v.size
