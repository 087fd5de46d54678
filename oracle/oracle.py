"""ctypes handle on the CPU oracle (oracle/psgsdf_oracle.c).  TEST INFRASTRUCTURE ONLY: imported
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by psgradientsdf_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from psgradientsdf_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libpsgsdf_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "psgsdf_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
    return _lib


class Oracle(capi.Api):
    def __init__(self, sc_or_grid, K, settings, solver_mode=0, threads=1):
        grid = sc_or_grid if isinstance(sc_or_grid, capi.GridDesc) else capi.grid_of(sc_or_grid)
        super().__init__(lib(), "orc_", grid, K, settings, 0)
        self._lib.orc_set_solver_mode(self.ctx, C.c_int(solver_mode))
        self._lib.orc_set_threads(self.ctx, C.c_int(threads))

    def set_faithful(self, on=True):
        """band membership as the reference does it (std::find over surface_points_, O(S) per test) instead of the row_of table: same results"""
        self._lib.orc_set_faithful(self.ctx, C.c_int(1 if on else 0))

    def set_solver_mode(self, mode):
        self._lib.orc_set_solver_mode(self.ctx, C.c_int(mode))

    # ---- oracle-only probes (known-answer tests)
    def probe_residual(self, j, f):
        r = (C.c_float * 3)(); w = (C.c_float * 3)()
        ok = self._lib.orc_probe_residual(self.ctx, C.c_int(j), C.c_int(f), r, w)
        return bool(ok), np.array(r[:], np.float32), np.array(w[:], np.float32)

    def probe_dist_jacobian(self, j, f):
        J = (C.c_float * 12)(); rows = (C.c_int * 4)()
        ok = self._lib.orc_probe_dist_jacobian(self.ctx, C.c_int(j), C.c_int(f), J, rows)
        return bool(ok), np.array(J[:], np.float32).reshape(4, 3), list(rows)

    def probe_pose_jacobian(self, j, f):
        J = (C.c_float * 18)()
        ok = self._lib.orc_probe_pose_jacobian(self.ctx, C.c_int(j), C.c_int(f), J)
        return bool(ok), np.array(J[:], np.float32).reshape(3, 6)

    def probe_eikonal(self, j):
        J = (C.c_float * 4)(); res = C.c_float(); rows = (C.c_int * 4)()
        self._lib.orc_probe_eikonal(self.ctx, C.c_int(j), J, C.byref(res), rows)
        return np.array(J[:], np.float32), float(res.value), list(rows)

    def probe_laplacian(self, j):
        res = C.c_float(); Jd = C.c_float()
        self._lib.orc_probe_laplacian(self.ctx, C.c_int(j), C.byref(res), C.byref(Jd))
        return float(res.value), float(Jd.value)

    def probe_albedo_reg(self, j):
        J = (C.c_float * 12)(); res = (C.c_float * 3)(); nb = (C.c_int64 * 3)()
        self._lib.orc_probe_albedo_reg(self.ctx, C.c_int(j), J, res, nb)
        return np.array(J[:], np.float32).reshape(4, 3), np.array(res[:], np.float32), list(nb)

    def probe_rho_jacobian(self, j, f):
        J = (C.c_float * 3)()
        self._lib.orc_probe_rho_jacobian(self.ctx, C.c_int(j), C.c_int(f), J)
        return np.array(J[:], np.float32)

    def peek_dist(self, lin):
        self._lib.orc_peek_dist.restype = C.c_float
        return float(self._lib.orc_peek_dist(self.ctx, C.c_int(lin)))

    def poke_dist(self, lin, v):
        self._lib.orc_poke_dist(self.ctx, C.c_int(lin), C.c_float(v))

    def peek_grad(self, lin):
        g = (C.c_float * 3)(); self._lib.orc_peek_grad(self.ctx, C.c_int(lin), g); return np.array(g[:], np.float32)

    def poke_grad(self, lin, g):
        self._lib.orc_poke_grad(self.ctx, C.c_int(lin), (C.c_float * 3)(*[float(x) for x in g]))

    def peek_rgb(self, lin):
        g = (C.c_float * 3)(); self._lib.orc_peek_rgb(self.ctx, C.c_int(lin), g); return np.array(g[:], np.float32)

    def poke_rgb(self, lin, g):
        self._lib.orc_poke_rgb(self.ctx, C.c_int(lin), (C.c_float * 3)(*[float(x) for x in g]))

    def poke_pose(self, f, P):
        self._lib.orc_poke_pose(self.ctx, C.c_int(f), (C.c_float * 16)(*[float(x) for x in np.asarray(P).ravel()]))

    def update_grad(self):
        self._lib.orc_update_grad(self.ctx)


def so3_exp(w):
    R = (C.c_float * 9)()
    lib().orc_so3_exp((C.c_float * 3)(*[float(x) for x in w]), R)
    return np.array(R[:], np.float32).reshape(3, 3)


def eigen_cg_dense(A, b):
    n = len(b)
    A = np.ascontiguousarray(A, np.float32); b = np.ascontiguousarray(b, np.float32)
    x = np.zeros(n, np.float32); it = C.c_int(); err = C.c_double()
    ok = lib().orc_eigen_cg_dense(C.c_int(n), A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.byref(it), C.byref(err))
    return x, it.value, err.value, bool(ok)
