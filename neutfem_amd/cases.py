"""Benchmark input generators.  Pure numpy; no RNG.

iaea3d_resampled(n)      C4: IAEA-3D core (19x19x19 assemblies of 20 cm, tests/iaea3d/iaea3d.py:60-158,231-258)
                         resampled on a uniform n^3 mesh over [0,380]^3; a cell takes the material of the
                         assembly that contains its centre.  2 groups, 6 Dirichlet sides, RT0-P0.
synthetic_checkerboard   C5: 16-cell checkerboard fuel/moderator, ng groups, closed-form XS.
The assembly-level XS table is data captured once from the reference driver (tests/golden/make_golden.py).
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "iaea3d_assemblies.npz")
DIRICHLET_3D = [(a, 0) for a in (1, 2, 3, 4, 5, 6)]


def iaea3d_resampled(n, nz=None, z_range=None):
    """returns dict(x_breaks,y_breaks,z_breaks,D,SigR,NSF,Chi,SigS,bc,ng).  z_range=(k0,k1) keeps only the
    z-planes [k0,k1) of the nz-plane mesh (slab of a decomposed run)."""
    nz = nz or n
    a = np.load(_DATA)
    brk = np.linspace(0.0, 380.0, n + 1)
    zb = np.linspace(0.0, 380.0, nz + 1)
    ic = np.minimum(((brk[:-1] + brk[1:]) * 0.5 / 20.0).astype(np.int64), 18)
    kc = np.minimum(((zb[:-1] + zb[1:]) * 0.5 / 20.0).astype(np.int64), 18)
    if z_range is not None:
        kc = kc[z_range[0]:z_range[1]]; zb = zb[z_range[0]:z_range[1] + 1]
    def rs(x):      # x: (..., 19, 19, 19) -> (..., nz, n, n)
        return np.ascontiguousarray(x[..., kc[:, None, None], ic[None, :, None], ic[None, None, :]])
    return dict(x_breaks=brk, y_breaks=brk.copy(), z_breaks=zb, D=rs(a["D"]), SigR=rs(a["SigR"]), NSF=rs(a["NSF"]),
                Chi=rs(a["Chi"]), SigS=rs(a["SigS"]), bc=DIRICHLET_3D, ng=2, coarse_factors=[2, 2, 1],
                name=f"IAEA-3D resampled {n}x{n}x{nz} RT0-P0 2g")


def synthetic_checkerboard(n, ng=8, z_range=None, nxy=None):
    """SURVEY.md 8d C5 (h = 1 cm, material id ((ix>>4)+(iy>>4)+(iz>>4))&1, 0 = fuel, 1 = moderator).
    z_range = (k0, k1): only those z-planes of the XS arrays (slab-decomposed runs); z_breaks stay global.
    nxy: cells along x and y (default n) -- a column of the same pattern with full-length z lines (parity tests, bench self-check)."""
    nxy = nxy or n
    brk = np.linspace(0.0, float(n), n + 1)
    i = np.arange(n) >> 4
    k0, k1 = z_range if z_range is not None else (0, n)
    mod = ((i[k0:k1, None, None] + i[None, :nxy, None] + i[None, None, :nxy]) & 1).astype(bool)
    shp = (k1 - k0, nxy, nxy)
    D = np.empty((ng,) + shp); SigR = np.empty((ng,) + shp); NSF = np.zeros((ng,) + shp); Chi = np.zeros((ng,) + shp)
    SigS = np.zeros((ng, ng) + shp)
    chi = [0.60, 0.30, 0.08, 0.02] + [0.0] * 60
    for g in range(ng):
        D[g] = np.where(mod, 1.9 * 0.75 ** g, 1.6 * 0.8 ** g)
        siga = np.where(mod, 0.0005 * 1.9 ** g, 0.004 * 1.6 ** g)
        out = np.zeros(shp)
        if g < ng - 1:
            s = np.where(mod, 0.08 * 0.9 ** g, 0.06 * 0.9 ** g); SigS[g + 1, g] = s; out = out + s
        if g == ng - 1 and ng >= 2:
            SigS[g - 1, g] = 0.002; out = out + 0.002
        SigR[g] = siga + out
        NSF[g] = np.where(mod, 0.0, 0.004 * 1.7 ** g)
        Chi[g] = np.where(mod, 0.0, chi[g])
    return dict(x_breaks=brk[:nxy + 1].copy(), y_breaks=brk[:nxy + 1].copy(), z_breaks=brk.copy(), D=D, SigR=SigR, NSF=NSF, Chi=Chi, SigS=SigS,
                bc=DIRICHLET_3D, ng=ng, coarse_factors=[2, 2, 2],
                name=f"synthetic checkerboard {n}^3 RT0-P0 {ng}g" if nxy == n else f"synthetic checkerboard {nxy}x{nxy}x{n} RT0-P0 {ng}g")
