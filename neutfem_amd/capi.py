"""ctypes binding of the C ABI declared in include/neutfem_hip.h (libneutfem_hip.so).

Thin by design: every call goes straight to the shared library; errors become RuntimeError with
nf_last_error().  Loading fails loudly if the library has not been built.
"""
import ctypes as C
import os

import numpy as np

from . import lib_path

SYMBOLS = [
    "nf_last_error", "nf_device_count", "nf_create", "nf_destroy", "nf_create_slab", "nf_link_slabs", "nf_comm_unique_id",
    "nf_comm_init", "nf_comm_info", "nf_comm_selftest", "nf_team_schur_apply", "nf_info", "nf_set_bc", "nf_upload_xs", "nf_build",
    "nf_schur_apply", "nf_solve_group", "nf_build_diagonal_cache", "nf_get_diagonal_cache", "nf_solve_keff",
    "nf_solve_coarse", "nf_coarsen", "nf_prolong", "nf_timers", "nf_initialize_cmfd", "nf_set_cmfd_relaxation", "nf_get_cmfd_coefficients", "nf_solve_adjoint", "nf_get_phi_adj", "nf_set_phi", "nf_get_phi", "nf_get_J", "nf_reset_flux", "nf_set_warm_state",
    "nf_get_warm_state", "nf_get_history", "nf_profile_get", "nf_profile_reset", "nf_time_schur_apply", "nf_time_device_copy", "nf_progress", "nf_set_progress_callback", "nf_local_matrices",
    "nf_set_option", "nf_mem_info", "nf_dev_alloc", "nf_dev_free", "nf_memcpy_h2d", "nf_memcpy_d2h", "nf_synchronize", "nf_stream",
]


class KeffOpts(C.Structure):
    _fields_ = [("tol_keff", C.c_double), ("tol_flux", C.c_double), ("max_outer", C.c_int), ("max_inner", C.c_int),
                ("use_coarse_init", C.c_int), ("coarse_factors", C.c_int * 3), ("n_coarse_factors", C.c_int),
                ("use_diagonal_solver", C.c_int), ("solver_type", C.c_int), ("solver_type_pushed", C.c_int),
                ("profile", C.c_int), ("use_cmfd", C.c_int)]


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
    L.nf_create_slab.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.nf_link_slabs.argtypes = [C.POINTER(vp), C.c_int]
    L.nf_comm_unique_id.argtypes = [vp]
    L.nf_comm_selftest.argtypes = [vp]
    L.nf_comm_info.argtypes = [vp, ip, C.c_char_p, C.c_size_t]
    L.nf_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.nf_team_schur_apply.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]
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
    L.nf_solve_adjoint.argtypes = [vp, C.POINTER(KeffOpts), C.c_int, C.c_int, dp, ip]
    L.nf_get_phi_adj.argtypes = [vp, dp]
    L.nf_set_phi.argtypes = [vp, dp]
    L.nf_get_phi.argtypes = [vp, dp]
    L.nf_get_J.argtypes = [vp, dp]
    L.nf_coarsen.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.nf_prolong.argtypes = [vp, vp]
    L.nf_timers.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.nf_initialize_cmfd.argtypes = [vp]
    L.nf_set_cmfd_relaxation.argtypes = [vp, C.c_double]
    L.nf_get_cmfd_coefficients.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.nf_reset_flux.argtypes = [vp]
    L.nf_set_warm_state.argtypes = [vp, C.c_int, C.c_double]
    L.nf_get_warm_state.argtypes = [vp, ip, dp]
    L.nf_get_history.argtypes = [vp, dp, dp, dp, ip, C.c_int]
    L.nf_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_long), dp]
    L.nf_profile_reset.argtypes = [vp]
    L.nf_time_schur_apply.argtypes = [vp, C.c_int, C.c_int, dp]
    L.nf_time_device_copy.argtypes = [vp, C.c_size_t, C.c_int, dp]
    L.nf_progress.argtypes = [vp, C.POINTER(C.c_long)]
    L.nf_set_progress_callback.argtypes = [vp, vp, vp]
    L.nf_local_matrices.argtypes = [vp, C.c_int, C.c_int, ip, dp, dp, dp, C.c_int, C.c_int, dp]
    L.nf_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.nf_mem_info.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
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

    def __init__(self, rt_order, p_order, ng, x_breaks, y_breaks, z_breaks, device=0, interface_below=False, interface_above=False):
        self.L = load()
        xb, yb, zb = (np.ascontiguousarray(a, dtype=np.float64) for a in (x_breaks, y_breaks, z_breaks))
        h = C.c_void_p()
        self.h = None
        if interface_below or interface_above:      # z_breaks are the slab's own breaks
            self._chk(self.L.nf_create_slab(rt_order, p_order, ng, len(xb), _dp(xb), len(yb), _dp(yb), len(zb), _dp(zb),
                                            int(interface_below), int(interface_above), device, C.byref(h)))
        else:
            self._chk(self.L.nf_create(rt_order, p_order, ng, len(xb), _dp(xb), len(yb), _dp(yb), len(zb), _dp(zb), device, C.byref(h)))
        self.h = h
        for key in ("dim", "nx", "ny", "nz", "ne", "ng", "n_phi", "n_J", "n_loc"):
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

    # *_dev vectors of the C ABI use the device DOF order [p*N + e]; these helpers take / return the reference order [e*n_loc + p]
    def _to_dev(self, v): return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(self.ne, self.n_loc).T).ravel()
    def _from_dev(self, v): return np.ascontiguousarray(v.reshape(self.n_loc, self.ne).T).ravel()

    def schur_apply(self, g, x):
        xd, yd = self.vector().upload(self._to_dev(x)), self.vector()
        self._chk(self.L.nf_schur_apply(self.h, g, xd.ptr, yd.ptr))
        y = self._from_dev(yd.download()); xd.free(); yd.free()
        return y

    def solve_group(self, g, rhs, tol, maxit):
        bd, xd = self.vector().upload(self._to_dev(rhs)), self.vector()
        its, res = C.c_int(), C.c_double()
        self._chk(self.L.nf_solve_group(self.h, g, bd.ptr, xd.ptr, tol, maxit, C.byref(its), C.byref(res)))
        x = self._from_dev(xd.download()); bd.free(); xd.free()
        return x, its.value, res.value

    def opts(self, use_coarse=False, factors=(), use_diag=False, profile=False, use_cmfd=False):
        o = KeffOpts()
        o.tol_keff, o.tol_flux, _, o.max_outer, o.max_inner = self.tol
        f = list(factors)[:3]
        o.use_coarse_init = int(bool(use_coarse and f))
        o.n_coarse_factors = len(f)
        for i, v in enumerate(f):
            o.coarse_factors[i] = int(v)
        o.use_diagonal_solver = int(use_diag)
        o.solver_type, o.solver_type_pushed, o.profile = self.solver_type, self.solver_pushed, int(profile)
        o.use_cmfd = int(use_cmfd)
        return o

    def solve_keff(self, use_coarse=False, factors=(), use_diag=False, profile=False, use_cmfd=False):
        o = self.opts(use_coarse, factors, use_diag, profile, use_cmfd)
        k, n = C.c_double(), C.c_int()
        self._chk(self.L.nf_solve_keff(self.h, C.byref(o), C.byref(k), C.byref(n)))
        return k.value, n.value

    def coarsen(self, rx, ry=1, rz=1):
        """built coarse twin (HipSolver over the merged mesh, block-mean XS); close() it when done"""
        h = C.c_void_p()
        self._chk(self.L.nf_coarsen(self.h, rx, ry, rz, C.byref(h)))
        c = HipSolver.__new__(HipSolver)
        c.L, c.h = self.L, h
        c.tol, c.solver_type, c.solver_pushed = self.tol, self.solver_type, self.solver_pushed
        for key in ("dim", "nx", "ny", "nz", "ne", "ng", "n_phi", "n_J"):
            setattr(c, key, c.L.nf_info(c.h, key.encode()))
        c.n_loc = c.L.nf_info(c.h, b"n_loc")
        return c

    def prolong_from(self, coarse): self._chk(self.L.nf_prolong(coarse.h, self.h))

    def timers(self):
        import json
        buf = C.create_string_buffer(2048)
        self._chk(self.L.nf_timers(self.h, buf, 2048))
        return json.loads(buf.value.decode())

    def initialize_cmfd(self): self._chk(self.L.nf_initialize_cmfd(self.h))
    def set_cmfd_relaxation(self, omega): self._chk(self.L.nf_set_cmfd_relaxation(self.h, float(omega)))

    def cmfd_coefficients(self, g, direction):
        """(D-tilde, D-hat) of group g on the faces of `direction`, reference face numbering"""
        nx, ny, nz = self.nx, self.ny, self.nz
        n = [(nx + 1) * ny * nz, nx * (ny + 1) * nz, nx * ny * (nz + 1)][direction]
        dt, dh = np.empty(n), np.empty(n)
        self._chk(self.L.nf_get_cmfd_coefficients(self.h, g, direction, _dp(dt), _dp(dh)))
        return dt, dh

    def solve_coarse(self, factors):
        o = self.opts(True, factors)
        k = C.c_double(); out = np.empty(self.ng * self.n_phi)
        self._chk(self.L.nf_solve_coarse(self.h, C.byref(o), C.byref(k), _dp(out)))
        return k.value, out

    def solve_adjoint(self, normalize_to_direct=True, use_direct_keff=True):
        o = self.opts()
        k, n = C.c_double(), C.c_int()
        self._chk(self.L.nf_solve_adjoint(self.h, C.byref(o), int(normalize_to_direct), int(use_direct_keff), C.byref(k), C.byref(n)))
        return k.value, n.value

    def get_phi_adj(self):
        out = np.empty(self.ng * self.n_phi); self._chk(self.L.nf_get_phi_adj(self.h, _dp(out))); return out.reshape(self.ng, self.n_phi)

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

    def local_matrices(self, g, elems, variant=0, reps=1):
        """LocalMatrices::Compute on the device for the listed elements: (A, B, C, avg_ms) with A (n, nJ, nJ), B (n, nP, nJ), C (n, nP, nP)"""
        el = np.ascontiguousarray(elems, dtype=np.int32); n = el.size
        k, m, dim = self.info("rt_order"), self.info("p_order"), self.dim
        nf, ni = (k + 1) ** (dim - 1), k * (k + 1) ** (dim - 1)
        nJ, nP = dim * (2 * nf + ni), (m + 1) ** dim
        A, B, Cm = np.zeros((n, nJ, nJ)), np.zeros((n, nP, nJ)), np.zeros((n, nP, nP))
        ms = C.c_double()
        self._chk(self.L.nf_local_matrices(self.h, g, n, el.ctypes.data_as(C.POINTER(C.c_int)), A.ctypes.data_as(C.POINTER(C.c_double)),
                                           B.ctypes.data_as(C.POINTER(C.c_double)), Cm.ctypes.data_as(C.POINTER(C.c_double)), variant, reps, C.byref(ms)))
        return A, B, Cm, ms.value

    def progress(self):
        """completed outer iterations of the running / last solve_keff (callable from another thread)"""
        v = C.c_long(); self._chk(self.L.nf_progress(self.h, C.byref(v))); return v.value

    def time_device_copy(self, nbytes, reps=20):
        v = C.c_double(); self._chk(self.L.nf_time_device_copy(self.h, int(nbytes), reps, C.byref(v))); return v.value


def mem_info(device=0):
    """(free, total) bytes of HBM on `device`"""
    L = load(); f, t = C.c_size_t(), C.c_size_t()
    if L.nf_mem_info(device, C.byref(f), C.byref(t)) != 0:
        raise RuntimeError(L.nf_last_error().decode())
    return f.value, t.value


def device_count():
    return load().nf_device_count()


class HipTeam:
    """The z-slabs of one global mesh held by THIS process (include/neutfem_hip.h, multi-GPU section).

    planes = [(k0, k1), ...] consecutive z-plane ranges of the local slabs (bottom to top);
    below / above = True if another process holds the slab below the first / above the last local slab.
    With one process and several ranges this is the single-GPU loopback used by the tests."""

    def __init__(self, rt_order, p_order, ng, x_breaks, y_breaks, z_breaks, planes, device=0, below=False, above=False):
        zb = np.asarray(z_breaks, dtype=np.float64)
        self.planes = [tuple(p) for p in planes]
        self.slabs = []
        for i, (k0, k1) in enumerate(self.planes):
            lo = below if i == 0 else True
            hi = above if i == len(self.planes) - 1 else True
            self.slabs.append(HipSolver(rt_order, p_order, ng, x_breaks, y_breaks, zb[k0:k1 + 1], device, lo, hi))
        self.L = self.slabs[0].L
        self.ng = ng
        self.rt = min(int(rt_order), 2)
        if len(self.slabs) > 1:
            arr = (C.c_void_p * len(self.slabs))(*[s.h for s in self.slabs])
            self.slabs[0]._chk(self.L.nf_link_slabs(arr, len(self.slabs)))
        self.head = self.slabs[0]

    def close(self):
        for s in self.slabs:
            s.close()

    def comm_selftest(self): self.head._chk(self.L.nf_comm_selftest(self.head.h))

    def comm_info(self):
        """(ranks of the live communicator as the transport library counts them, path of the library that carries the data)"""
        n = C.c_int(); buf = C.create_string_buffer(512)
        self.head._chk(self.L.nf_comm_info(self.head.h, C.byref(n), buf, 512))
        return n.value, buf.value.decode()

    def comm_init(self, id_bytes, nranks, rank):
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        self.head._chk(self.L.nf_comm_init(self.head.h, buf, nranks, rank))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        L = load()
        if L.nf_comm_unique_id(buf) != 0:
            raise RuntimeError(L.nf_last_error().decode())
        return buf.raw

    def set_bc(self, attr, t):
        for s in self.slabs: s.set_bc(attr, t)

    def set_tol(self, *a):
        for s in self.slabs: s.set_tol(*a)

    def set_linear_solver(self, t):
        for s in self.slabs: s.set_linear_solver(t)

    def upload_xs_local(self, arrays_per_slab):
        for s, a in zip(self.slabs, arrays_per_slab): s.upload_xs(*a)

    def upload_xs_global(self, D, SigR, NSF, Chi, SigS, k_offset=0):
        """global (.., nz, ny, nx) arrays; k_offset = global plane index of array plane 0"""
        for s, (k0, k1) in zip(self.slabs, self.planes):
            sl = slice(k0 - k_offset, k1 - k_offset)
            s.upload_xs(D[:, sl], SigR[:, sl], NSF[:, sl], Chi[:, sl], SigS[:, :, sl])

    def build(self):
        for s in self.slabs: s.build()

    def solve_keff(self, use_coarse=False, factors=(), profile=False, use_diag=False, use_cmfd=False):
        return self.head.solve_keff(use_coarse, factors, use_diag, profile, use_cmfd)

    def set_cmfd_relaxation(self, omega):
        for s in self.slabs:
            s.set_cmfd_relaxation(omega)

    def reset_flux(self):
        for s in self.slabs: s.reset_flux()

    def solve_adjoint(self, normalize_to_direct=True, use_direct_keff=True):
        return self.head.solve_adjoint(normalize_to_direct, use_direct_keff)

    def get_phi_adj_local(self):
        return np.concatenate([s.get_phi_adj().reshape(self.ng, s.nz, s.ny, s.nx) for s in self.slabs], axis=1)

    def history(self): return self.head.history()
    def profile(self, name): return self.head.profile(name)
    def profile_reset(self): self.head.profile_reset()
    def time_schur_apply(self, g, reps): return self.head.time_schur_apply(g, reps)
    def synchronize(self): self.head._chk(self.L.nf_synchronize(self.head.h))

    def get_J_local(self):
        """Sol_J_ of the local slabs assembled over their planes: (ng, n_J of the stacked local slabs) in the reference's DOF
        numbering (x, y, z face DOFs, then x, y, z bubbles; src/FEM.cpp:264-334).  Collective on a multi-rank run (the z currents
        cross slabs)."""
        nx, ny, ng, k = self.head.nx, self.head.ny, self.ng, self.rt
        nf, ni = (k + 1) ** 2, k * (k + 1) ** 2                   # slabs are 3D: face / interior DOFs per face / cell and direction
        xs, ys, zs, bub = [], [], [], [[], [], []]
        for i, s in enumerate(self.slabs):
            J = s.get_J()
            nxf, nyf, nzf = (nx + 1) * ny * s.nz * nf, nx * (ny + 1) * s.nz * nf, nx * ny * (s.nz + 1) * nf
            xs.append(J[:, :nxf]); ys.append(J[:, nxf:nxf + nyf])
            z = J[:, nxf + nyf:nxf + nyf + nzf].reshape(ng, s.nz + 1, ny * nx * nf)
            zs.append(z if i == len(self.slabs) - 1 else z[:, :-1])      # the shared interface plane is reported by both neighbours
            nb = s.ne * ni
            for d in range(3):
                bub[d].append(J[:, nxf + nyf + nzf + d * nb:nxf + nyf + nzf + (d + 1) * nb])
        return np.concatenate([np.concatenate(xs, axis=1), np.concatenate(ys, axis=1), np.concatenate(zs, axis=1).reshape(ng, -1)] +
                              [np.concatenate(b, axis=1) for b in bub], axis=1)

    def get_phi_local(self):
        """(ng, local planes, ny, nx)"""
        return np.concatenate([s.get_phi().reshape(self.ng, s.nz, s.ny, s.nx) for s in self.slabs], axis=1)

    def set_phi_local(self, phi):
        k = 0
        for s in self.slabs:
            s.set_phi(np.ascontiguousarray(phi[:, k:k + s.nz])); k += s.nz

    def schur_apply(self, g, x_local):
        """x_local: (local planes, ny, nx) -> S x on the local planes (loopback: the whole mesh)"""
        xs, ys, k = [], [], 0
        for s in self.slabs:
            xs.append(s.vector().upload(np.ascontiguousarray(x_local[k:k + s.nz]).ravel())); ys.append(s.vector()); k += s.nz
        xa = (C.c_void_p * len(xs))(*[v.ptr for v in xs]); ya = (C.c_void_p * len(ys))(*[v.ptr for v in ys])
        self.head._chk(self.L.nf_team_schur_apply(self.head.h, g, xa, ya))
        out = np.concatenate([v.download() for v in ys])
        for v in xs + ys: v.free()
        return out.reshape(x_local.shape)
