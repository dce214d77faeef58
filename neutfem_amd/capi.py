"""ctypes binding of the C ABI declared in include/neutfem_hip.h (libneutfem_hip.so).

Thin by design: every call goes straight to the shared library; errors become RuntimeError with
nf_last_error().  Loading fails loudly if the library has not been built.
"""
import ctypes as C
import os

import numpy as np

from . import lib_path

SYMBOLS = [
    "nf_last_error", "nf_device_count", "nf_create", "nf_destroy", "nf_info", "nf_set_bc", "nf_upload_xs", "nf_build",
    "nf_schur_apply", "nf_solve_group", "nf_build_diagonal_cache", "nf_get_diagonal_cache", "nf_solve_keff",
    "nf_solve_coarse", "nf_set_phi", "nf_get_phi", "nf_get_J", "nf_reset_flux", "nf_set_warm_state",
    "nf_get_warm_state", "nf_get_history", "nf_profile_get", "nf_profile_reset", "nf_time_schur_apply",
    "nf_set_option", "nf_dev_alloc", "nf_dev_free", "nf_memcpy_h2d", "nf_memcpy_d2h", "nf_synchronize", "nf_stream",
]


class KeffOpts(C.Structure):
    _fields_ = [("tol_keff", C.c_double), ("tol_flux", C.c_double), ("max_outer", C.c_int), ("max_inner", C.c_int),
                ("use_coarse_init", C.c_int), ("coarse_factors", C.c_int * 3), ("n_coarse_factors", C.c_int),
                ("use_diagonal_solver", C.c_int), ("solver_type", C.c_int), ("solver_type_pushed", C.c_int),
                ("profile", C.c_int)]


_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(neutfem_amd has no CPU fallback)")
    L = C.CDLL(path)
    dp, vp, ip = C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int)
    L.nf_last_error.restype = C.c_char_p
    L.nf_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int, C.POINTER(vp)]
    L.nf_destroy.argtypes = [vp]
    L.nf_info.restype = C.c_long
    L.nf_info.argtypes = [vp, C.c_char_p]
    L.nf_set_bc.argtypes = [vp, C.c_int, C.c_int]
    L.nf_upload_xs.argtypes = [vp, dp, dp, dp, dp, dp]
    L.nf_build.argtypes = [vp]
    L.nf_schur_apply.argtypes = [vp, C.c_int, vp, vp]
    L.nf_solve_group.argtypes = [vp, C.c_int, vp, vp, C.c_double, C.c_int, ip, dp]
    L.nf_build_diagonal_cache.argtypes = [vp]
    L.nf_get_diagonal_cache.argtypes = [vp, C.c_int, dp]
    L.nf_solve_keff.argtypes = [vp, C.POINTER(KeffOpts), dp, ip]
    L.nf_solve_coarse.argtypes = [vp, C.POINTER(KeffOpts), dp, dp]
    L.nf_set_phi.argtypes = [vp, dp]
    L.nf_get_phi.argtypes = [vp, dp]
    L.nf_get_J.argtypes = [vp, dp]
    L.nf_reset_flux.argtypes = [vp]
    L.nf_set_warm_state.argtypes = [vp, C.c_int, C.c_double]
    L.nf_get_warm_state.argtypes = [vp, ip, dp]
    L.nf_get_history.argtypes = [vp, dp, dp, dp, ip, C.c_int]
    L.nf_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_long), dp]
    L.nf_profile_reset.argtypes = [vp]
    L.nf_time_schur_apply.argtypes = [vp, C.c_int, C.c_int, dp]
    L.nf_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.nf_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.nf_dev_free.argtypes = [vp, vp]
    L.nf_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.nf_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.nf_synchronize.argtypes = [vp]
    L.nf_stream.restype = vp
    L.nf_stream.argtypes = [vp]
    _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class DeviceVector:
    """n doubles of device memory owned through nf_dev_alloc."""

    def __init__(self, solver, n):
        self.s, self.n = solver, n
        p = C.c_void_p()
        solver._chk(solver.L.nf_dev_alloc(solver.h, n * 8, C.byref(p)))
        self.ptr = p

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        self.s._chk(self.s.L.nf_memcpy_h2d(self.s.h, self.ptr, a.ctypes.data_as(C.c_void_p), self.n * 8))
        return self

    def download(self):
        out = np.empty(self.n)
        self.s._chk(self.s.L.nf_memcpy_d2h(self.s.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.n * 8))
        return out

    def free(self):
        if self.ptr:
            self.s.L.nf_dev_free(self.s.h, self.ptr)
            self.ptr = None


class HipSolver:
    """One nf_handle.  Method names follow the C ABI; see include/neutfem_hip.h for reference citations."""

    def __init__(self, rt_order, p_order, ng, x_breaks, y_breaks, z_breaks, device=0):
        self.L = load()
        xb, yb, zb = (np.ascontiguousarray(a, dtype=np.float64) for a in (x_breaks, y_breaks, z_breaks))
        h = C.c_void_p()
        self.h = None
        self._chk(self.L.nf_create(rt_order, p_order, ng, len(xb), _dp(xb), len(yb), _dp(yb), len(zb), _dp(zb), device, C.byref(h)))
        self.h = h
        for key in ("dim", "nx", "ny", "nz", "ne", "ng", "n_phi", "n_J"):
            setattr(self, key, self.L.nf_info(h, key.encode()))
        self.tol = (1e-5, 1e-5, 1e-5, 200, 1000)
        self.solver_type, self.solver_pushed = 6, 0

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(f"neutfem_hip error {rc}: {self.L.nf_last_error().decode()}")

    def close(self):
        if self.h:
            self.L.nf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self, key): return self.L.nf_info(self.h, key.encode())
    def set_bc(self, attr, bctype): self._chk(self.L.nf_set_bc(self.h, int(attr), int(bctype)))
    def set_tol(self, tk, tf, tl, mo, mi): self.tol = (tk, tf, tl, mo, mi)
    def set_linear_solver(self, t): self.solver_type, self.solver_pushed = int(t), 1

    def upload_xs(self, D, SigR, NSF, Chi, SigS):
        arrs = [np.ascontiguousarray(a, dtype=np.float64).ravel() for a in (D, SigR, NSF, Chi, SigS)]
        assert all(a.size == self.ng * self.ne for a in arrs[:4]) and arrs[4].size == self.ng * self.ng * self.ne
        self._chk(self.L.nf_upload_xs(self.h, *[_dp(a) for a in arrs]))

    def build(self): self._chk(self.L.nf_build(self.h))
    def vector(self, n=None): return DeviceVector(self, n or self.n_phi)

    def schur_apply(self, g, x):
        xd, yd = self.vector().upload(x), self.vector()
        self._chk(self.L.nf_schur_apply(self.h, g, xd.ptr, yd.ptr))
        y = yd.download(); xd.free(); yd.free()
        return y

    def solve_group(self, g, rhs, tol, maxit):
        bd, xd = self.vector().upload(rhs), self.vector()
        its, res = C.c_int(), C.c_double()
        self._chk(self.L.nf_solve_group(self.h, g, bd.ptr, xd.ptr, tol, maxit, C.byref(its), C.byref(res)))
        x = xd.download(); bd.free(); xd.free()
        return x, its.value, res.value

    def opts(self, use_coarse=False, factors=(), use_diag=False, profile=False):
        o = KeffOpts()
        o.tol_keff, o.tol_flux, _, o.max_outer, o.max_inner = self.tol
        f = list(factors)[:3]
        o.use_coarse_init = int(bool(use_coarse and f))
        o.n_coarse_factors = len(f)
        for i, v in enumerate(f):
            o.coarse_factors[i] = int(v)
        o.use_diagonal_solver = int(use_diag)
        o.solver_type, o.solver_type_pushed, o.profile = self.solver_type, self.solver_pushed, int(profile)
        return o

    def solve_keff(self, use_coarse=False, factors=(), use_diag=False, profile=False):
        o = self.opts(use_coarse, factors, use_diag, profile)
        k, n = C.c_double(), C.c_int()
        self._chk(self.L.nf_solve_keff(self.h, C.byref(o), C.byref(k), C.byref(n)))
        return k.value, n.value

    def solve_coarse(self, factors):
        o = self.opts(True, factors)
        k = C.c_double(); out = np.empty(self.ng * self.n_phi)
        self._chk(self.L.nf_solve_coarse(self.h, C.byref(o), C.byref(k), _dp(out)))
        return k.value, out

    def diagonal_cache(self, g):
        out = np.empty(self.ne)
        self._chk(self.L.nf_get_diagonal_cache(self.h, g, _dp(out)))
        return out

    def set_phi(self, phi):
        a = np.ascontiguousarray(phi, dtype=np.float64).ravel(); assert a.size == self.ng * self.n_phi
        self._chk(self.L.nf_set_phi(self.h, _dp(a)))

    def get_phi(self):
        out = np.empty(self.ng * self.n_phi); self._chk(self.L.nf_get_phi(self.h, _dp(out))); return out.reshape(self.ng, self.n_phi)

    def get_J(self):
        out = np.empty(self.ng * self.n_J); self._chk(self.L.nf_get_J(self.h, _dp(out))); return out.reshape(self.ng, self.n_J)

    def reset_flux(self): self._chk(self.L.nf_reset_flux(self.h))
    def set_warm_state(self, valid, k): self._chk(self.L.nf_set_warm_state(self.h, int(valid), float(k)))

    def history(self):
        n = self.info("last_outer")
        k, dk, dp = np.empty(n), np.empty(n), np.empty(n); cg = np.empty(n * self.ng, dtype=np.int32)
        self._chk(self.L.nf_get_history(self.h, _dp(k), _dp(dk), _dp(dp), cg.ctypes.data_as(C.POINTER(C.c_int)), n))
        return dict(n_outer=n, k=k, dk=dk, dphi=dp, cg=cg.reshape(n, self.ng), coarse_outer=self.info("coarse_outer"))

    def profile(self, name):
        c, ms = C.c_long(), C.c_double()
        self._chk(self.L.nf_profile_get(self.h, name.encode(), C.byref(c), C.byref(ms)))
        return c.value, ms.value

    def set_option(self, key, value): self._chk(self.L.nf_set_option(self.h, key.encode(), int(value)))

    def profile_reset(self): self._chk(self.L.nf_profile_reset(self.h))

    def time_schur_apply(self, g, reps):
        ms = C.c_double(); self._chk(self.L.nf_time_schur_apply(self.h, g, reps, C.byref(ms))); return ms.value


def device_count():
    return load().nf_device_count()
