"""numpy restatement of the z-slab partition method used by the multi-GPU HIP path (test infrastructure).

RT0-P0, 3D.  Each rank owns the z-planes [k0,k1) of the mesh.  x/y-line solves are slab-local; a z-line solve
crosses slabs: the interface faces are separators, each slab factors only its interior chain (partition / SPIKE):
  step 1  v = T_II^-1 t_I with the neighbour cells as plain boundary data ->  c_lo = -beta x_0 - a_lo v_first,
          c_hi = +beta x_last - a_hi v_last
  step 2  exchange c with the two neighbours;  u_sep = (c_hi(below) + c_lo(above)) / S_red
  step 3  T_II u_I = t_I - T_I,sep u_sep ;  y += B u
Communication goes through a tiny interface (send_up/send_down/recv..., allreduce) implemented with
torch.distributed (gloo) in the tests -- the same pattern neutfem_hip.hip issues through RCCL."""
import numpy as np


def thomas(diag, off, rhs):
    """batched tridiagonal solve along axis 0: diag (n, ...), off (n-1, ...) symmetric, rhs (n, ...)"""
    n = diag.shape[0]
    d = diag.copy(); b = rhs.copy()
    for i in range(1, n):
        w = off[i - 1] / d[i - 1]
        d[i] = d[i] - w * off[i - 1]
        b[i] = b[i] - w * b[i - 1]
    x = np.empty_like(b)
    x[n - 1] = b[n - 1] / d[n - 1]
    for i in range(n - 2, -1, -1):
        x[i] = (b[i] - off[i] * x[i + 1]) / d[i]
    return x


class SlabOperator:
    """S = C + B A^-1 B^T restricted to one z-slab (src/solvers.cpp:535-547 on a decomposed mesh)"""

    def __init__(self, hx, hy, hz_slab, D, SigR, dirichlet, if_lo, if_hi, comm):
        # D, SigR: (nzl, ny, nx) of one group; dirichlet: dict attr->bool for attrs 1..6 (3D numbering)
        self.hx, self.hy, self.hz = hx, hy, hz_slab
        self.if_lo, self.if_hi, self.comm = if_lo, if_hi, comm
        nzl, ny, nx = D.shape
        HX, HY, HZ = hx[None, None, :], hy[None, :, None], hz_slab[:, None, None]
        self.C = SigR * (HX * HY * HZ)
        fac = [2 * HX / (HY * HZ), 2 * HY / (HX * HZ), 2 * HZ / (HX * HY)]
        self.beta = 4.0
        self.a2 = [(8.0 / 3.0) * f / D for f in fac]           # A_LL = A_RR  (2^(d-1) * 2/3)
        self.a1 = [(4.0 / 3.0) * f / D for f in fac]           # A_LR
        area = [HY * HZ, HX * HZ, HX * HY]
        self.dir = []                                          # Dirichlet terms 32 D / area on the boundary faces
        lo_attr, hi_attr = {0: 3, 1: 6, 2: 1}, {0: 4, 1: 5, 2: 2}
        for d in range(3):
            t = 32.0 * D / area[d]
            ax = 2 - d
            lo = np.take(t, 0, axis=ax) * (1.0 if dirichlet.get(lo_attr[d]) else 0.0)
            hi = np.take(t, -1, axis=ax) * (1.0 if dirichlet.get(hi_attr[d]) else 0.0)
            self.dir.append((lo, hi))
        if if_lo or if_hi:                                     # separator diagonal halves -> S_red (one exchange, build time)
            self._setup_separators()

    # tridiagonal of a full line set along numpy axis `ax` (cells 0..n-1, faces 0..n), Dirichlet ends
    def _line_matrix(self, d):
        ax = 2 - d
        a2 = np.moveaxis(self.a2[d], ax, 0); a1 = np.moveaxis(self.a1[d], ax, 0)
        n = a2.shape[0]
        diag = np.zeros((n + 1,) + a2.shape[1:]); diag[:-1] += a2; diag[1:] += a2
        lo, hi = self.dir[d]
        diag[0] += lo; diag[-1] += hi
        return diag, a1

    def _line_apply(self, x, d):
        ax = 2 - d
        xm = np.moveaxis(x, ax, 0)
        diag, off = self._line_matrix(d)
        n = xm.shape[0]
        t = np.zeros((n + 1,) + xm.shape[1:]); t[1:] += self.beta * xm; t[:-1] -= self.beta * xm
        u = thomas(diag, off, t)
        return np.moveaxis(self.beta * (u[1:] - u[:-1]), 0, ax)

    # ---- z direction on a slab ----------------------------------------------------------------------------------
    def _chain(self):
        """interior chain of the slab-local z lines: faces fs..fe (local face j = lower face of local cell j)"""
        a2, a1 = self.a2[2], self.a1[2]
        m = a2.shape[0]
        fs, fe = (1 if self.if_lo else 0), (m - 1 if self.if_hi else m)
        diag = np.zeros((fe - fs + 1,) + a2.shape[1:])
        for j, f in enumerate(range(fs, fe + 1)):
            if f - 1 >= 0: diag[j] += a2[f - 1]
            if f < m: diag[j] += a2[f]
        if not self.if_lo: diag[0] += self.dir[2][0]
        if not self.if_hi: diag[-1] += self.dir[2][1]
        off = a1[fs:fe]
        return fs, fe, diag, off

    def _setup_separators(self):
        fs, fe, diag, off = self._chain()
        m = self.a2[2].shape[0]
        e0 = np.zeros_like(diag); e0[0] = 1.0
        e1 = np.zeros_like(diag); e1[-1] = 1.0
        g0 = thomas(diag, off, e0)[0]; g1 = thomas(diag, off, e1)[-1]
        half_lo = self.a2[2][0] - self.a1[2][0] ** 2 * g0 if self.if_lo else None
        half_hi = self.a2[2][m - 1] - self.a1[2][m - 1] ** 2 * g1 if self.if_hi else None
        r_lo, r_hi = self.comm.exchange(half_lo, half_hi)      # neighbour's halves
        self.sred_lo = (r_lo + half_lo) if self.if_lo else None
        self.sred_hi = (half_hi + r_hi) if self.if_hi else None

    def _z_apply(self, x):
        a1 = self.a1[2]; m = x.shape[0]; beta = self.beta
        fs, fe, diag, off = self._chain()
        def rhs(x_before, x_after):
            xe = np.concatenate([x_before[None], x[fs:fe], x_after[None]]) if True else None
            return beta * (xe[:-1] - xe[1:])                   # t_f = beta (x_{f-1} - x_f), f = fs..fe
        zero = np.zeros(x.shape[1:])
        xb = x[0] if self.if_lo else zero
        xa = x[m - 1] if self.if_hi else zero
        v = thomas(diag, off, rhs(xb, xa))
        c_lo = -beta * x[0] - a1[0] * v[0] if self.if_lo else None
        c_hi = beta * x[m - 1] - a1[m - 1] * v[-1] if self.if_hi else None
        r_lo, r_hi = self.comm.exchange(c_lo, c_hi)
        u_lo = (r_lo + c_lo) / self.sred_lo if self.if_lo else None
        u_hi = (c_hi + r_hi) / self.sred_hi if self.if_hi else None
        if self.if_lo: xb = x[0] - (a1[0] / beta) * u_lo
        if self.if_hi: xa = x[m - 1] + (a1[m - 1] / beta) * u_hi
        u = thomas(diag, off, rhs(xb, xa))                     # chain faces fs..fe
        uf = np.zeros((m + 1,) + x.shape[1:])
        uf[fs:fe + 1] = u
        if self.if_lo: uf[0] = u_lo
        if self.if_hi: uf[m] = u_hi
        return beta * (uf[1:] - uf[:-1])

    def diag_sinv(self):
        """BuildDiagonalSchurCache (src/NeutFEM.cpp:483-597) on a slab: S_inv = 1/(C + sum_faces B^2/A_ff); the A_ff of an
        interface z face needs the neighbour's edge a2 -- one plane exchanged once (what nf_build_diagonal_cache does)"""
        S = self.C.copy()
        for d in range(3):
            ax = 2 - d
            a2 = np.moveaxis(self.a2[d], ax, 0)
            Aff = np.zeros((a2.shape[0] + 1,) + a2.shape[1:]); Aff[:-1] += a2; Aff[1:] += a2
            lo, hi = self.dir[d]
            if d == 2 and (self.if_lo or self.if_hi):
                r_lo, r_hi = self.comm.exchange(a2[0] if self.if_lo else None, a2[-1] if self.if_hi else None)
                if self.if_lo: Aff[0] += r_lo
                else: Aff[0] += lo
                if self.if_hi: Aff[-1] += r_hi
                else: Aff[-1] += hi
            else:
                Aff[0] += lo; Aff[-1] += hi
            S += np.moveaxis(self.beta ** 2 / Aff[:-1] + self.beta ** 2 / Aff[1:], 0, ax)
        return 1.0 / S

    def apply(self, x):
        y = self.C * x + self._line_apply(x, 0) + self._line_apply(x, 1)
        if self.if_lo or self.if_hi:
            return y + self._z_apply(x)
        return y + self._line_apply(x, 2)


def distributed_cg(op, b, tol, maxit, comm):
    """SolveSchurImplicit (src/solvers.cpp:577-636) with the two dot products all-reduced over the slabs"""
    x = np.zeros_like(b); r = b.copy(); p = b.copy()
    rr = comm.allreduce(float((r * r).sum()))
    tol_sq = tol * tol * rr
    its = 0
    for k in range(maxit):
        q = op.apply(p)
        pq = comm.allreduce(float((p * q).sum()))
        if abs(pq) < 1e-30: break
        al = rr / pq
        x += al * p; r -= al * q
        rrn = comm.allreduce(float((r * r).sum())); its = k + 1
        if rrn < tol_sq: break
        p = r + (rrn / rr) * p; rr = rrn
    return x, its


def distributed_cg_single_reduction(op, b, tol, maxit, comm):
    """The same solve with ONE all-reduce per iteration (Cg1 in neutfem_amd/csrc/nf_kernels.h, what slab teams run by default): the
    reduction of iteration j carries [p.q, q.q, r.q, |r_j|^2 measured]; its consumer forms alpha_j, the predicted
    |r_{j+1}|^2 = |r|^2 - 2 alpha r.q + alpha^2 q.q (for beta_j only) and applies r -= alpha q, x += alpha p, p = r + beta p in one sweep;
    the stop test is taken on the measured |r_j|^2, one apply late, so the returned x is the one the reference recurrence returns."""
    x = np.zeros_like(b); r = b.copy(); p = b.copy()
    tol_sq = tol * tol * comm.allreduce(float((b * b).sum()))
    for j in range(maxit + 1):
        q = op.apply(p)
        pq, qq, rq, rr = comm.allreduce4([float((p * q).sum()), float((q * q).sum()), float((r * q).sum()), float((r * r).sum())])
        if j >= 1 and rr < tol_sq: return x, j
        if abs(pq) < 1e-30: return x, j
        al = rr / pq
        rn = max(rr - 2.0 * al * rq + al * al * qq, 0.0)
        x += al * p
        if j + 1 >= maxit: return x, j + 1
        r -= al * q
        p = r + (rn / rr) * p
    return x, maxit
