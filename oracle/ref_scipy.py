"""Independent numpy/scipy restatement of the reference hot path (TEST INFRASTRUCTURE).

Purpose: pin oracle/nf_oracle.c.  Unlike the C oracle (matrix-free, banded line
solves in chain order) this file follows the reference's *data structures*: it
assembles the explicit global sparse matrices A_g, B, C_g, M_fiss, M_scatter from
per-element quadrature (src/FEM.cpp:748-953, src/NeutFEM.cpp:1036-1302), applies
the Dirichlet diagonal (src/NeutFEM.cpp:1328-1456), factors A with SuperLU
(scipy.sparse.linalg.splu -- the library Eigen::SparseLU is a port of,
src/solvers.cpp:163) and runs the same CG / Chebyshev / power iteration
(src/solvers.cpp:577-636,664-756; src/NeutFEM.cpp:1627-1815,2380-2611).

Only tests/ may import this module.  It is slow (Python loops over elements) and
is used on meshes of at most a few thousand cells.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from numpy.polynomial import legendre as npleg

_GAUSS = {  # include/FEM.hpp:82-123 (tabulated digits kept)
    3: ([-np.sqrt(0.6), 0.0, np.sqrt(0.6)], [5 / 9, 8 / 9, 5 / 9]),
    5: ([-0.906179845938664, -0.538469310105683, 0.0, 0.538469310105683, 0.906179845938664],
        [0.236926885056189, 0.478628670499366, 0.568888888888889, 0.478628670499366, 0.236926885056189]),
}


def _P(n, x):
    c = np.zeros(n + 1); c[n] = 1.0
    return npleg.legval(x, c)


def _dP(n, x):
    c = np.zeros(n + 1); c[n] = 1.0
    return npleg.legval(x, npleg.legder(c)) if n > 0 else np.zeros_like(x)


class RefScipy:
    def __init__(self, rt, p, ng, xb, yb, zb):
        xb, yb, zb = (np.asarray(a, float) for a in (xb, yb, zb))
        self.nx = len(xb) - 1
        self.ny = len(yb) - 1 if len(yb) > 1 else 1
        self.nz = len(zb) - 1 if len(zb) > 1 else 1
        self.dim = 3 if self.nz > 1 else (2 if self.ny > 1 else 1)
        self.xb, self.yb, self.zb = xb, yb, zb
        self.hx = np.diff(xb)
        self.hy = np.diff(yb) if self.dim >= 2 else np.ones(1)
        self.hz = np.diff(zb) if self.dim == 3 else np.ones(1)
        self.k = min(rt, 2); self.m = min(min(p, 2), self.k); self.ng = ng
        k, m, d = self.k, self.m, self.dim
        self.nf = (k + 1) ** (d - 1); self.ni = k * (k + 1) ** (d - 1); self.nloc = (m + 1) ** d
        self.nper = 2 * self.nf + self.ni; self.nJloc = d * self.nper
        self.ne = self.nx * self.ny * self.nz
        nx, ny, nz, nf = self.nx, self.ny, self.nz, self.nf
        self.nJx = (nx + 1) * ny * nz * nf
        self.nJy = nx * (ny + 1) * nz * nf if d >= 2 else 0
        self.nJz = nx * ny * (nz + 1) * nf if d == 3 else 0
        self.nJface = self.nJx + self.nJy + self.nJz
        self.nJ = self.nJface + self.ne * d * self.ni
        self.nPhi = self.ne * self.nloc
        order = 2 * max(k, m) + 3
        self.qp, self.qw = (np.array(a) for a in _GAUSS[order if order in _GAUSS else 5])
        ne = self.ne
        self.D = np.ones((ng, ne)); self.SigR = np.full((ng, ne), 0.01); self.NSF = np.zeros((ng, ne))
        self.Chi = np.zeros((ng, ne)); self.Chi[0] = 1.0; self.SigS = np.zeros((ng, ng, ne))
        self.phi = np.ones(ng * self.nPhi)
        self.bc = {}
        self.tol = (1e-5, 1e-5, 1e-5, 200, 1000); self.cg_tol = 1e-10; self.cg_max = 1000
        self.keff = 1.0; self.valid = False
        self._tables()

    # reference tables by tensor quadrature (numpy, independent of the C loops)
    def _tables(self):
        d, k, m = self.dim, self.k, self.m
        grids = np.meshgrid(*([self.qp] * d), indexing="ij")
        w = np.ones_like(grids[0])
        for a in range(d):
            sh = [1] * d; sh[a] = len(self.qw)
            w = w * self.qw.reshape(sh)
        pts = [g.ravel() for g in grids] + [np.zeros(w.size)] * (3 - d)
        w = w.ravel()
        Jv = np.zeros((self.nJloc, w.size)); dv = np.zeros_like(Jv)
        for a in range(d):
            s = pts[a]
            tr = [pts[b] for b in range(3) if b != a]           # x:(eta,zeta) y:(xi,zeta) z:(xi,eta)
            def Pt(idx):
                i, j = (idx % (k + 1), idx // (k + 1)) if d == 3 else (idx, 0)
                v = np.ones_like(s)
                if d >= 2: v = v * _P(i, tr[0])
                if d == 3: v = v * _P(j, tr[1])
                return v
            o = a * self.nper
            for f in range(self.nf):
                Jv[o + f] = 0.5 * (1 - s) * Pt(f); dv[o + f] = -0.5 * Pt(f)
                Jv[o + self.nf + f] = 0.5 * (1 + s) * Pt(f); dv[o + self.nf + f] = 0.5 * Pt(f)
            for b in range(self.ni):
                l, t = (b % k, b // k) if d >= 2 else (b, 0)
                Jv[o + 2 * self.nf + b] = (1 - s * s) * _P(l, s) * Pt(t)
                dv[o + 2 * self.nf + b] = (-2 * s * _P(l, s) + (1 - s * s) * _dP(l, s)) * Pt(t)
        n = m + 1
        pv = np.zeros((self.nloc, w.size))
        for q in range(self.nloc):
            i, j, kk = q % n, (q // n) % n, q // (n * n)
            v = _P(i, pts[0])
            if d >= 2: v = v * _P(j, pts[1])
            if d == 3: v = v * _P(kk, pts[2])
            pv[q] = v
        self.Ahat = []
        for a in range(d):
            blk = Jv[a * self.nper:(a + 1) * self.nper]
            self.Ahat.append((blk * w) @ blk.T)
        self.Bhat = (pv * w) @ dv.T
        self.Chat = (pv * w) @ pv.T

    def _faces(self, ix, iy, iz):
        """global J indices of one element, src/FEM.cpp:955-999"""
        nx, ny, nf, ni, d = self.nx, self.ny, self.nf, self.ni, self.dim
        e = iz * nx * ny + iy * nx + ix
        out = []
        fx = lambda i: ((iz * ny + iy) * (nx + 1) + i) * nf
        out += [fx(ix) + l for l in range(nf)] + [fx(ix + 1) + l for l in range(nf)]
        out += [self.nJface + e * ni + b for b in range(ni)]
        if d >= 2:
            fy = lambda j: self.nJx + ((iz * (ny + 1) + j) * nx + ix) * nf
            out += [fy(iy) + l for l in range(nf)] + [fy(iy + 1) + l for l in range(nf)]
            out += [self.nJface + self.ne * ni + e * ni + b for b in range(ni)]
        if d == 3:
            fz = lambda kz: self.nJx + self.nJy + ((kz * ny + iy) * nx + ix) * nf
            out += [fz(iz) + l for l in range(nf)] + [fz(iz + 1) + l for l in range(nf)]
            out += [self.nJface + 2 * self.ne * ni + e * ni + b for b in range(ni)]
        return np.array(out)

    def _geom(self, ix, iy, iz):
        hx, hy, hz = self.hx[ix], self.hy[iy], self.hz[iz]
        if self.dim == 1: return [hx / 2], hx / 2
        if self.dim == 2: return [hy / hx, hx / hy], hx * hy / 4
        return [2 * hx / (hy * hz), 2 * hy / (hx * hz), 2 * hz / (hx * hy)], hx * hy * hz / 8

    def _attr(self, direction, upper):  # src/NeutFEM.cpp:2338-2347
        if self.dim == 1: return 2 if upper else 1
        if self.dim == 2: return (2 if upper else 1) if direction == 0 else (3 if upper else 4)
        return [(3, 4), (6, 5), (1, 2)][direction][1 if upper else 0]

    def build(self):
        ng, ne, nper, nP, d = self.ng, self.ne, self.nper, self.nloc, self.dim
        rows, cols, bvals = [], [], []
        arows, acols = [], []
        avals = [[] for _ in range(ng)]
        cdiag = np.zeros((ng, self.nPhi)); self.Mf = np.zeros((ng, self.nPhi)); self.Ms = {}
        Ms = np.zeros((ng, ng, self.nPhi))
        for iz in range(self.nz):
            for iy in range(self.ny):
                for ix in range(self.nx):
                    e = iz * self.nx * self.ny + iy * self.nx + ix
                    J = self._faces(ix, iy, iz); fac, detJ = self._geom(ix, iy, iz)
                    vol = self.hx[ix] * self.hy[iy] * self.hz[iz]
                    for a in range(d):
                        blk = J[a * nper:(a + 1) * nper]
                        arows.append(np.repeat(blk, nper)); acols.append(np.tile(blk, nper))
                        for g in range(ng):
                            avals[g].append((self.Ahat[a] * fac[a] / self.D[g, e]).ravel())
                    rows.append(np.repeat(e * nP + np.arange(nP), self.nJloc)); cols.append(np.tile(J, nP)); bvals.append(self.Bhat.ravel())
                    cd = np.diag(self.Chat) * detJ
                    for g in range(ng):
                        cdiag[g, e * nP:(e + 1) * nP] = self.SigR[g, e] * cd
                        self.Mf[g, e * nP:(e + 1) * nP] = self.NSF[g, e] * (vol if self.m == 0 else cd)
                        for gp in range(ng):
                            Ms[g, gp, e * nP:(e + 1) * nP] = self.SigS[g, gp, e] * (vol if self.m == 0 else cd)
        arows, acols = np.concatenate(arows), np.concatenate(acols)
        B = sp.csr_matrix((np.concatenate(bvals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.nPhi, self.nJ))
        B.data[np.abs(B.data) <= 1e-14] = 0; B.eliminate_zeros()
        self.B = B; self.BT = B.T.tocsr()
        self.A, self.lu, self.C = [], [], cdiag
        self._Sfac = {}
        for g in range(ng):
            v = np.concatenate(avals[g]); keep = np.abs(v) > 1e-13 * np.abs(v).max()
            A = sp.coo_matrix((v[keep], (arows[keep], acols[keep])), shape=(self.nJ, self.nJ)).tolil()
            for a in range(d):                                   # Dirichlet, NeutFEM.cpp:1328-1456
                for upper in (False, True):
                    if self.bc.get(self._attr(a, upper)) != 0: continue
                    rng = [range(self.nx), range(self.ny), range(self.nz)]
                    n_a = [self.nx, self.ny, self.nz][a]
                    rng[a] = [n_a - 1] if upper else [0]
                    for iz in rng[2]:
                        for iy in rng[1]:
                            for ix in rng[0]:
                                e = iz * self.nx * self.ny + iy * self.nx + ix
                                J = self._faces(ix, iy, iz)
                                area = [self.hy[iy] * self.hz[iz], self.hx[ix] * self.hz[iz], self.hx[ix] * self.hy[iy]][a]
                                for f in range(self.nf):
                                    if d == 1: I = 1.0
                                    elif d == 2: I = 2 * (2 / (2 * f + 1)) / area
                                    else: I = 4 * (2 / (2 * (f % (self.k + 1)) + 1)) * (2 / (2 * (f // (self.k + 1)) + 1)) / area
                                    dof = J[a * nper + (self.nf if upper else 0) + f]
                                    A[dof, dof] += I * 2.0 * self.D[g, e]
            A = A.tocsc(); self.A.append(A); self.lu.append(spla.splu(A))
        for g in range(ng):
            for gp in range(ng):
                if np.any(np.abs(Ms[g, gp]) > 1e-14): self.Ms[(g, gp)] = Ms[g, gp]

    def schur_apply(self, g, x):                                 # solvers.cpp:535-547
        return self.C[g] * x + self.B @ self.lu[g].solve(self.BT @ x)

    def cg(self, g, b):                                          # solvers.cpp:577-636
        x = np.zeros_like(b); r = b.copy(); p = b.copy(); rr = r @ r
        tol_sq = self.cg_tol ** 2 * (b @ b); its = 0
        for kk in range(self.cg_max):
            Ap = self.schur_apply(g, p); pAp = p @ Ap
            if abs(pAp) < 1e-30: break
            al = rr / pAp; x += al * p; r -= al * Ap; rrn = r @ r; its = kk + 1
            if rrn < tol_sq: break
            p = r + (rrn / rr) * p; rr = rrn
        return x, its

    def direct_solve(self, g, b):                                # solvers.cpp:259-310 (FormSchurComplement) + :441-452 (S_lu_.solve)
        if g not in self._Sfac:
            X = self.lu[g].solve(self.BT.toarray())              # A x_j = B^T[:, j] for every column j
            X[np.abs(X) <= 1e-14] = 0.0                          # :291
            S = np.diag(self.C[g]) + self.B @ X
            import scipy.linalg as sla
            self._Sfac[g] = sla.lu_factor(S)
        import scipy.linalg as sla
        return sla.lu_solve(self._Sfac[g], b)

    def set_tol(self, tk, tf, tl, mo, mi):
        self.tol = (tk, tf, tl, mo, mi); self.cg_tol = tf; self.cg_max = mi

    # ---- CMFD (NeutFEM.cpp:662-1017) with an explicit scipy matrix and Eigen's PCG written out in numpy -------------
    def cmfd_init(self):
        nx, ny, nz, ng = self.nx, self.ny, self.nz, self.ng
        shp = [(nz, ny, nx + 1), (nz, ny + 1, nx), (nz + 1, ny, nx)]
        self.Dt = [np.zeros((ng,) + s_) for s_ in shp]; self.Dh = [np.zeros((ng,) + s_) for s_ in shp]
        hs = [self.hx, self.hy, self.hz]
        for g in range(ng):
            D = self.D[g].reshape(nz, ny, nx)
            for d in range(self.dim):
                ax = 2 - d                                       # array axis of direction d
                Dm = np.moveaxis(D, ax, 0); h = hs[d].reshape((-1,) + (1,) * 2)
                out = np.moveaxis(self.Dt[d][g], ax, 0)
                out[0] = 2 * Dm[0] / h[0]; out[-1] = 2 * Dm[-1] / h[-1]
                out[1:-1] = 2 * Dm[:-1] * Dm[1:] / (Dm[:-1] * h[1:] + Dm[1:] * h[:-1])

    def cmfd_dhat(self, J):
        nx, ny, nz, nf = self.nx, self.ny, self.nz, self.nf
        for g in range(self.ng):
            phi = self.phi[g * self.nPhi:(g + 1) * self.nPhi].reshape(self.ne, self.nloc)[:, 0].reshape(nz, ny, nx)
            Jx = J[g][:self.nJx].reshape(nz, ny, nx + 1, nf)[..., 0]
            pd = np.zeros((nz, ny, nx + 1)); pd[..., 0] = -phi[..., 0]; pd[..., -1] = phi[..., -1]
            pd[..., 1:-1] = phi[..., :-1] - phi[..., 1:]
            with np.errstate(divide="ignore", invalid="ignore"):
                self.Dh[0][g] = np.where(np.abs(pd) > 1e-14, Jx / pd - self.Dt[0][g], 0.0)

    def cmfd_matrix(self, g):
        nx, ny, nz = self.nx, self.ny, self.nz
        idx = np.arange(self.ne).reshape(nz, ny, nx)
        area = [np.einsum("k,j,i->kji", self.hz, self.hy, np.ones(nx)), np.einsum("k,j,i->kji", self.hz, np.ones(ny), self.hx),
                np.einsum("k,j,i->kji", np.ones(nz), self.hy, self.hx)]
        diag = (self.C[g].reshape(self.ne, self.nloc)[:, 0]).reshape(nz, ny, nx).copy()
        rows, cols, vals = [], [], []
        for d in range(self.dim):
            ax = 2 - d
            De = np.moveaxis(self.Dt[d][g] + self.Dh[d][g], ax, 0); A = np.moveaxis(area[d], ax, 0); I = np.moveaxis(idx, ax, 0)
            dg = np.moveaxis(diag, ax, 0)
            dg += (De[:-1] + De[1:]) * A
            rows += [I[1:].ravel(), I[:-1].ravel()]; cols += [I[:-1].ravel(), I[1:].ravel()]
            vals += [(-De[1:-1] * A[1:]).ravel(), (-De[1:-1] * A[:-1]).ravel()]
        rows.append(idx.ravel()); cols.append(idx.ravel()); vals.append(diag.ravel())
        return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.ne, self.ne))

    def cmfd_correction(self, g, tf, keff, omega=1.0):
        M = self.cmfd_matrix(g)
        b = self.Chi[g] * tf.reshape(self.ne, self.nloc)[:, 0] / keff
        x = np.zeros(self.ne); its = 0
        b2 = b @ b
        if b2 > 0:
            thr = max(1e-16 * b2, np.finfo(float).tiny)
            r = b.copy()
            dg = M.diagonal(); inv = np.where(dg != 0, 1.0 / np.where(dg != 0, dg, 1.0), 1.0)
            p = inv * r; an = r @ p
            for its in range(1, 101):
                t = M @ p; al = an / (p @ t); x += al * p; r -= al * t
                if r @ r < thr: break
                z = inv * r; ao, an = an, r @ z; p = z + (an / ao) * p
        pc = self.phi[g * self.nPhi:(g + 1) * self.nPhi].reshape(self.ne, self.nloc)[:, 0]
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.where(np.abs(pc) > 1e-14, np.clip(x / pc, 0.5, 2.0), 1.0)
        self.cmfd_its = its
        return np.repeat(omega * ratio + (1 - omega), self.nloc)

    def solve_keff(self, use_coarse=False, factors=(), use_diag=False, use_cmfd=False, omega=1.0):   # NeutFEM.cpp:1627-1815
        ng, nP, ne = self.ng, self.nPhi, self.ne
        if use_cmfd: self.cmfd_init()
        k = self.keff if self.valid else 1.0
        if use_coarse and len(factors):
            k, self.phi = self.solve_coarse(factors)
        if use_diag:                                             # NeutFEM.cpp:483-597
            Sinv = []
            for g in range(ng):
                Ad = self.A[g].diagonal(); B2 = self.B.multiply(self.B).tocsr()
                S = self.C[g] + B2 @ np.where(np.abs(Ad) > 1e-14, 1.0 / Ad, 0.0)
                Sinv.append(np.where(np.abs(S) > 1e-14, 1.0 / S, 0.0))
        G = np.arccosh(2 / 0.98 - 1)
        ca = [0, 2 / (2 - 0.98)] + [np.cosh((n - 1) * G) / np.cosh(n * G) for n in range(2, 15)]
        cb = [0, 0] + [np.cosh((n - 2) * G) / np.cosh(n * G) for n in range(2, 15)]
        cit, p0, p1 = 0, None, None
        hist = []
        for it in range(self.tol[3]):
            old = self.phi.copy()
            tf = sum(self.Mf[g] * self.phi[g * nP:(g + 1) * nP] for g in range(ng)); prod_old = tf.sum()
            cgs = []
            for g in range(ng):
                chi = np.repeat(self.Chi[g], self.nloc) / k
                rhs = np.where(np.abs(chi) < 1e-14, 0.0, chi * tf) if self.nloc > 1 else chi * tf
                for gp in range(ng):
                    if gp != g and (g, gp) in self.Ms: rhs = rhs + self.Ms[(g, gp)] * self.phi[gp * nP:(gp + 1) * nP]
                if use_diag: x, its = Sinv[g] * rhs, 0
                elif getattr(self, "direct", False): x, its = self.direct_solve(g, rhs), 1
                else: x, its = self.cg(g, rhs)
                self.phi[g * nP:(g + 1) * nP] = x; cgs.append(its)
            if use_cmfd and it >= 2:                             # :1750-1761 (J = -A^-1 B^T phi, solvers.cpp:227-228)
                self.cmfd_dhat([-self.lu[g].solve(self.BT @ self.phi[g * nP:(g + 1) * nP]) for g in range(ng)])
                for g in range(ng):
                    self.phi[g * nP:(g + 1) * nP] *= self.cmfd_correction(g, tf, k, omega)
            prod_new = sum((self.Mf[g] * self.phi[g * nP:(g + 1) * nP]).sum() for g in range(ng))
            kn = k * prod_new / prod_old; dk = abs(kn - k)
            if it >= 1: k = kn
            nsq = self.phi @ self.phi; dphi = np.sqrt(((self.phi - old) ** 2).sum() / nsq)
            self.phi /= np.sqrt(nsq)
            if it >= 2 and not use_cmfd:                         # solvers.cpp:720-756
                if cit == 15: cit, p0, p1 = 0, None, None
                if cit == 0: p0 = self.phi.copy()
                elif cit == 1: p1 = p0 + ca[1] * (self.phi - p0); self.phi = p1.copy()
                else:
                    nw = p1 + (4 / 0.98) * ca[cit] * (self.phi - p1) + cb[cit] * (p1 - p0)
                    p0, p1 = p1, nw; self.phi = nw.copy()
                cit += 1
            hist.append((k, dk, dphi, cgs))
            if dk < self.tol[0] and dphi < self.tol[1]: break
        self.keff, self.valid, self.hist = k, True, hist
        return k

    def solve_coarse(self, factors):                             # NeutFEM.cpp:2380-2611
        d = self.dim
        r = [max(int(factors[i]), 1) if (i < len(factors) and i < d) else 1 for i in range(3)]
        n = [self.nx, self.ny, self.nz]
        if any(n[i] % r[i] for i in range(3)): return 1.0, self.phi
        c = RefScipy(0, 0, self.ng, self.xb[::r[0]], self.yb[::r[1]] if d >= 2 else [0.0], self.zb[::r[2]] if d == 3 else [0.0])
        c.bc = dict(self.bc); c.set_tol(self.tol[0] * 10, self.tol[1] * 10, self.tol[2], self.tol[3] // 2, self.tol[4])
        vol = (self.hz[:, None, None] * self.hy[None, :, None] * self.hx[None, None, :])
        def coarsen(a):
            w = (a.reshape(-1, self.nz, self.ny, self.nx) * vol)
            s = w.reshape(-1, n[2] // r[2], r[2], n[1] // r[1], r[1], n[0] // r[0], r[0]).sum(axis=(2, 4, 6))
            v = vol.reshape(n[2] // r[2], r[2], n[1] // r[1], r[1], n[0] // r[0], r[0]).sum(axis=(1, 3, 5))
            return (s / v).reshape(a.shape[0], -1)
        c.D, c.SigR, c.NSF, c.Chi = coarsen(self.D), coarsen(self.SigR), coarsen(self.NSF), coarsen(self.Chi)
        c.SigS = coarsen(self.SigS.reshape(self.ng * self.ng, -1)).reshape(self.ng, self.ng, -1)
        c.build(); kc = c.solve_keff()
        self.coarse_outer = len(c.hist)
        out = np.zeros(self.ng * self.nPhi).reshape(self.ng, self.nz, self.ny, self.nx, self.nloc)
        cp = c.phi.reshape(self.ng, n[2] // r[2], n[1] // r[1], n[0] // r[0])
        out[..., 0] = np.repeat(np.repeat(np.repeat(cp, r[2], 1), r[1], 2), r[0], 3)
        return kc, out.ravel()
