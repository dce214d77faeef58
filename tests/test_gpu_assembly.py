"""LocalMatrices::Compute on the device (nf_local_matrices, neutfem_amd/csrc/nf_assembly.h: one element per workgroup, quadrature points
and basis values staged in LDS) against the oracle's literal quadrature (nfo_local_matrices, src/FEM.cpp:748-953) for every RTk-Pm
order and dimension, in both variants (plain fp64 FMA and v_mfma_f64_16x16x4_f64) -- and against the closed forms the hot path relies on
(SURVEY 8a "known answers"): the device now checks them against the quadrature itself."""
import numpy as np
import pytest

from helpers import make_hip, make_oracle, synthetic_inputs

pytestmark = pytest.mark.gpu

ORDERS = [(0, 0), (1, 1), (1, 0), (2, 2), (2, 1), (2, 0)]


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("rt,p", ORDERS)
@pytest.mark.parametrize("shape", [(7, 1, 1), (6, 5, 1), (5, 4, 3)])
def test_local_matrices_match_the_oracle_quadrature(shape, rt, p, variant):
    inp = synthetic_inputs(*shape, ng=2, seed=3 + rt + p)          # non-uniform mesh: every geometric factor is exercised
    o, s = make_oracle(inp, rt, p), make_hip(inp, rt, p)
    ne = o.ne
    elems = sorted({0, ne - 1, ne // 2, ne // 3, (2 * ne) // 3})
    for g in range(2):
        A, B, C, _ = s.local_matrices(g, elems, variant)
        D, Sig = np.asarray(inp["D"][g]).ravel(), np.asarray(inp["SigR"][g]).ravel()
        for i, e in enumerate(elems):
            Ao, Bo, Co = o.local_matrices(e, float(D[e]), float(Sig[e]))
            assert A[i].shape == Ao.shape and B[i].shape == Bo.shape and C[i].shape == Co.shape
            for got, ref in ((A[i], Ao), (B[i], Bo), (C[i], Co)):
                assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), (rt, p, e, np.abs(got - ref).max() / np.abs(ref).max())
    s.close()


def test_closed_forms_of_the_hot_path_against_the_device_quadrature():
    """RT0 3D  A_LL = 16 hx / (3 D hy hz), A_LR = 8 hx / (3 D hy hz), B = -+4, C = Sigma V;  RT1-P1 2D x chain
    [[4/3, 2/3, 4/3], [2/3, 4/3, 4/3], [4/3, 4/3, 32/15]] x factor / D per transverse mode (mode 1: x 1/3), C-hat = diag(4, 4/3, 4/3, 4/9)"""
    inp = synthetic_inputs(5, 4, 3, ng=1, seed=8)
    s = make_hip(inp, 0, 0)
    hx, hy, hz = (np.diff(inp[k]) for k in ("x_breaks", "y_breaks", "z_breaks"))
    e = 1 + 5 * (2 + 4 * 1); ix, iy, iz = 1, 2, 1
    D, Sig = np.asarray(inp["D"][0]).ravel()[e], np.asarray(inp["SigR"][0]).ravel()[e]
    for variant in (0, 1):
        A, B, C, _ = s.local_matrices(0, [e], variant)
        A, B, C = A[0], B[0], C[0]
        f = hx[ix] / (D * hy[iy] * hz[iz])
        np.testing.assert_allclose([A[0, 0], A[0, 1], A[1, 1]], [16 * f / 3, 8 * f / 3, 16 * f / 3], rtol=1e-13)
        np.testing.assert_allclose(B[0], [-4, 4, -4, 4, -4, 4], rtol=1e-13)
        np.testing.assert_allclose(C[0, 0], Sig * hx[ix] * hy[iy] * hz[iz], rtol=1e-13)
        assert A[0, 2] == 0.0 and A[2, 4] == 0.0                       # no coupling across directions
    s.close()
    inp = synthetic_inputs(6, 5, 1, ng=1, seed=9)
    s = make_hip(inp, 1, 1)
    hx, hy = np.diff(inp["x_breaks"]), np.diff(inp["y_breaks"])
    e = 2 + 6 * 3
    D, Sig = np.asarray(inp["D"][0]).ravel()[e], np.asarray(inp["SigR"][0]).ravel()[e]
    A, B, C, _ = s.local_matrices(0, [e], 1)
    A, B, C = A[0], B[0], C[0]
    fx = (hy[3] / hx[2]) / D                                           # the reference's 2D factor_x = hy / hx (src/FEM.cpp:803-804)
    chain = np.array([[4 / 3, 2 / 3, 4 / 3], [2 / 3, 4 / 3, 4 / 3], [4 / 3, 4 / 3, 32 / 15]])
    idx0, idx1 = [0, 2, 4], [1, 3, 5]                                  # local x DOFs [L0, L1, R0, R1, b0, b1]: mode 0 / mode 1 chains
    np.testing.assert_allclose(A[np.ix_(idx0, idx0)], fx * chain, rtol=1e-12)
    np.testing.assert_allclose(A[np.ix_(idx1, idx1)], fx * chain / 3, rtol=1e-12)
    assert np.abs(A[np.ix_(idx0, idx1)]).max() < 1e-15 * fx
    np.testing.assert_allclose(np.diag(C), Sig * (hx[2] / 2) * (hy[3] / 2) * np.array([4, 4 / 3, 4 / 3, 4 / 9]), rtol=1e-12)
    np.testing.assert_allclose(B[0, :6], [-2, 0, 2, 0, 0, 0], atol=1e-14)
    np.testing.assert_allclose(B[1, 4], -8 / 3, rtol=1e-12)
    s.close()
