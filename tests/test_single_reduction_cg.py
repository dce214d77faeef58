"""Which single-reduction CG can stand in for the reference's recurrence (src/solvers.cpp:577-636) on slab teams -- decided on the CPU
oracle's Schur operator, before any GPU is involved (VERDICT r3 item 1 proposed Chronopoulos-Gear).

IAEA-3D fills its blank assemblies with Sigma = 1e15, D = 1e-3 (tests/iaea3d/iaea3d.py:254): cond(S) ~ 1e17 and |r|^2 swings by ten
orders of magnitude between consecutive iterations of the reference's unpreconditioned CG.  On that operator
  * Chronopoulos-Gear (p.Sp from r.Sr and the previous scalars), its pipelined form (Ghysels-Vanroose: the same scalars plus recurrences
    for S p and S S p, which is what would let the reduction run under the next apply -- item 1b) and the two-term prediction
    |r_new|^2 = alpha^2 q.q - |r|^2 (which assumes r_new . r = 0) NEVER converge;
  * the form the device runs (Cg1 in nf_kernels.h: one reduction carrying p.q, q.q, r.q and the MEASURED |r|^2; the predicted
    |r_new|^2 = |r|^2 - 2 alpha r.q + alpha^2 q.q feeds beta only; stop test on measured values, one apply late) converges with the
    reference's iteration counts to within a few per cent and to the same solution within the spread that two builds of the
    reference's own recurrence show (tests/golden/rounding_spread.json);
  * on the well-conditioned benchmarks all variants reproduce the reference's counts exactly and its iterates to rounding."""
import numpy as np
import pytest

from helpers import load_inputs, make_oracle


def std_cg(S, b, tol, maxit):
    x = np.zeros_like(b); r = b.copy(); p = b.copy(); rr = r @ r; t2 = tol * tol * rr
    for it in range(maxit):
        q = S(p); pq = p @ q
        if abs(pq) < 1e-30: return x, it
        a = rr / pq; x += a * p; r -= a * q; rn = r @ r
        if rn < t2: return x, it + 1
        p = r + (rn / rr) * p; rr = rn
    return x, maxit


def chronopoulos_gear(S, b, tol, maxit):
    x = np.zeros_like(b); r = b.copy(); t2 = tol * tol * (b @ b)
    s = S(r); g = r @ r; d = r @ s; beta = 0.0; a = g / d; p = np.zeros_like(b); q = np.zeros_like(b)
    for it in range(maxit):
        p = r + beta * p; q = s + beta * q; x += a * p; r -= a * q
        s = S(r); gn = r @ r; d = r @ s
        if gn < t2: return x, it + 1
        beta = gn / g; a = gn / (d - beta * gn / a); g = gn
    return x, maxit


def pipelined(S, b, tol, maxit):
    """Ghysels-Vanroose pipelined CG: Chronopoulos-Gear's scalars plus recurrences for s = S p and z = S s, so that the one reduction of an
    iteration can run under the next apply"""
    x = np.zeros_like(b); r = b.copy(); w = S(r); t2 = tol * tol * (b @ b)
    p = np.zeros_like(b); s = np.zeros_like(b); z = np.zeros_like(b); g0 = a = 1.0
    for it in range(maxit):
        g = r @ r; d = w @ r; q = S(w)                           # the reduction of (g, d) is what the apply S(w) would hide
        if g < t2: return x, it
        beta = g / g0 if it else 0.0
        a = g / (d - beta * g / a) if it else g / d
        z = q + beta * z; s = w + beta * s; p = r + beta * p
        x += a * p; r -= a * s; w -= a * z; g0 = g
    return x, maxit


def two_term(S, b, tol, maxit):
    x = np.zeros_like(b); r = b.copy(); p = b.copy(); t2 = tol * tol * (b @ b)
    for it in range(maxit):
        q = S(p); pq, qq, rr = p @ q, q @ q, r @ r
        a = rr / pq; rn = a * a * qq - rr; x += a * p; r -= a * q
        if rn < t2: return x, it + 1
        p = r + (rn / rr) * p
    return x, maxit


def device_form(S, b, tol, maxit):
    """Cg1 (neutfem_amd/csrc/nf_kernels.h), statement for statement"""
    x = np.zeros_like(b); r = b.copy(); p = b.copy(); t2 = tol * tol * (b @ b)
    for j in range(maxit + 1):
        q = S(p); pq, qq, rq, rr = p @ q, q @ q, r @ q, r @ r
        if j >= 1 and rr < t2: return x, j
        if abs(pq) < 1e-30: return x, j
        a = rr / pq; rn = max(rr - 2 * a * rq + a * a * qq, 0.0); x += a * p
        if j + 1 >= maxit: return x, j + 1
        r -= a * q; p = r + (rn / rr) * p
    return x, maxit


def _problem(name, g, seed=1):
    inp = load_inputs(name); o = make_oracle(inp); ng = int(inp["ng"])
    fuel = np.asarray(inp["NSF"]).reshape(ng, -1).sum(0) > 0
    b = np.abs(np.random.default_rng(seed).standard_normal(o.n_phi)) * fuel          # a fission-like source: nothing in the blank cells
    return o, (lambda v: o.schur_apply(g, v)), b


@pytest.mark.parametrize("g", [0, 1])
def test_recurrence_variants_fail_on_iaea3d_and_the_device_form_does_not(g):
    o, S, b = _problem("iaea3d", g)
    res = lambda x: np.linalg.norm(b - S(x)) / np.linalg.norm(b)
    x0, n0 = std_cg(S, b, 1e-4, 1000)
    assert n0 < 60 and res(x0) < 1.01e-4
    # the proposed single-reduction recurrences: 400 iterations (ten times the reference's count) and nowhere near
    for f in (chronopoulos_gear, two_term, pipelined):
        x, n = f(S, b, 1e-4, 400)
        assert n == 400 or res(x) > 1e-2, (f.__name__, n, res(x))
    # the device's form: the drivers' tolerance ...
    x1, n1 = device_form(S, b, 1e-4, 1000)
    assert abs(n1 - n0) <= 2 and res(x1) < 1.3e-4, (n0, n1, res(x1))
    assert np.linalg.norm(x1 - x0) / np.linalg.norm(x0) < 2e-4        # the spread of two builds of the reference recurrence on this case is 6.6e-5
    # ... and a tight one: same solution, iteration count within the chaos of this operator (two correct builds differ by 4 % here)
    xt0, nt0 = std_cg(S, b, 1e-10, 5000); xt1, nt1 = device_form(S, b, 1e-10, 5000)
    assert res(xt1) < 1.3e-10 and np.linalg.norm(xt1 - xt0) / np.linalg.norm(xt0) < 2e-9, (nt0, nt1)
    assert nt1 <= 1.5 * nt0, (nt0, nt1)


@pytest.mark.parametrize("name,ng", [("iaea2d", 2), ("koeberg2d", 4), ("biblis2d", 2)])
def test_all_variants_agree_on_the_well_conditioned_benchmarks(name, ng):
    for g in range(ng):
        o, S, b = _problem(name, g)
        for tol in (1e-4, 1e-10):
            x0, n0 = std_cg(S, b, tol, 2000)
            for f in (device_form, chronopoulos_gear, two_term) + ((pipelined,) if name != "koeberg2d" else ()):
                x, n = f(S, b, tol, 2000)
                assert n == n0, (name, g, tol, f.__name__, n, n0)
                bar = (1e-9 if tol > 1e-6 else 1e-12) if f is device_form else 1e-6       # the recurrence variants drift first (KOEBERG: 2e-9 at 28 iterations)
                assert np.linalg.norm(x - x0) / np.linalg.norm(x0) < bar, (name, g, tol, f.__name__)


def test_pipelined_form_loses_koeberg_too():
    """VERDICT r3 item 1b (hide the all-reduce under the next apply) needs S p and S S p by recurrence.  KOEBERG's fast group already breaks that: |r|^2
    swings by seven orders of magnitude between iterations there, the recurred w = S r drifts to 1e-3 relative within 35 iterations and the true residual
    stalls where the reference's recurrence is done after 28 iterations."""
    o, S, b = _problem("koeberg2d", 0)
    x0, n0 = std_cg(S, b, 1e-4, 2000)
    x, n = pipelined(S, b, 1e-4, 20 * n0)
    assert n0 < 40 and (n == 20 * n0 or np.linalg.norm(b - S(x)) / np.linalg.norm(b) > 1e-3), (n0, n)
    x1, n1 = device_form(S, b, 1e-4, 2000)
    assert n1 == n0


def test_device_form_honours_maxit_and_zero_rhs():
    o, S, b = _problem("iaea2d", 0)
    x0, n0 = std_cg(S, b, 0.0, 7); x1, n1 = device_form(S, b, 0.0, 7)          # fixed work (bench.py's C5 leg): exactly maxit iterations, same x
    assert n0 == n1 == 7 and np.linalg.norm(x1 - x0) / np.linalg.norm(x0) < 1e-13
    xz, nz = device_form(S, np.zeros_like(b), 1e-4, 10)
    assert nz == 0 and not xz.any()
