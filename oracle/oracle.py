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
LIB_FMA = os.path.join(_HERE, "libpsgsdf_oracle_fma.so")      # the same source with FMA contraction: a second float build (oracle/Makefile), a yardstick for drift
_lib = None
_lib_fma = None


def build(force=False):
    src = os.path.join(_HERE, "psgsdf_oracle.c")
    if force or not os.path.exists(LIB) or not os.path.exists(LIB_FMA) or os.path.getmtime(LIB) < os.path.getmtime(src) or os.path.getmtime(LIB_FMA) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
    return _lib


def lib_fma() -> C.CDLL:
    global _lib_fma
    if _lib_fma is None:
        build()
        _lib_fma = C.CDLL(LIB_FMA)
    return _lib_fma


class Oracle(capi.Api):
    def __init__(self, sc_or_grid, K, settings, solver_mode=0, threads=1, fma=False):
        """fma: the FMA-contracted build of the same source (a second float build: the drift yardstick of the whole-run tests, never the checker)"""
        grid = sc_or_grid if isinstance(sc_or_grid, capi.GridDesc) else capi.grid_of(sc_or_grid)
        super().__init__(lib_fma() if fma else lib(), "orc_", grid, K, settings, 0)
        self._lib.orc_set_solver_mode(self.ctx, C.c_int(solver_mode))
        self._lib.orc_set_threads(self.ctx, C.c_int(threads))

    # ---- the oracle's multi-rank PHASE MIRROR (orc_mg_*): the slab algorithm spelled out phase by phase for the host program tests/_slab_runner.py
    # (gloo, CPU).  The engine has no such API -- its C++ host runs the slab loop itself -- so these bindings live here, not in the product's capi.
    def comm_init(self, rank, n_ranks, unique_id=None):
        """attach this oracle context to a rank (no communicator: the exchanges are done by the host program over the phase API below)"""
        self._check(self._fn("comm_init")(self.ctx, None, C.c_int(rank), C.c_int(n_ranks)), "comm_init")

    def mg_info(self):
        """{S, Spad, row0, row1, halo, F, rank, n_ranks, need_lo, need_hi, -, -} in the mirror's own layout (rows = band rows, not z-planes)"""
        out = (C.c_int32 * 12)()
        self._check(self._fn("mg_info")(self.ctx, out), "mg_info")
        return dict(zip(["S", "Spad", "row0", "row1", "halo", "F", "rank", "n_ranks", "need_lo", "need_hi", "z0", "z1"], list(out)))

    def mg_buffer(self, which):
        ptr = C.c_void_p(); n = C.c_int64()
        self._check(self._fn("mg_buffer")(self.ctx, C.c_int(which), C.byref(ptr), C.byref(n)), "mg_buffer")
        return ptr.value, n.value

    def mg_phase(self, phase, arg=0):
        self._check(self._fn("mg_phase")(self.ctx, C.c_int(phase), C.c_int(arg)), f"mg_phase({phase})")

    def mg_pcg_status(self, k0, n):
        it = C.c_int32(); err = C.c_double()
        self._check(self._fn("mg_pcg_status")(self.ctx, C.c_int(k0), C.c_int(n), C.byref(it), C.byref(err)), "mg_pcg_status")
        return it.value, err.value

    def mg_fold_base(self, base):
        self._check(self._fn("mg_fold_base")(self.ctx, C.c_int(base)), "mg_fold_base")

    def mg_set_reg_sums(self, en_sum, el_sum):
        self._check(self._fn("mg_set_reg_sums")(self.ctx, C.c_double(en_sum), C.c_double(el_sum)), "mg_set_reg_sums")

    def mg_set_weights(self, reg_n, reg_l):
        self._check(self._fn("mg_set_weights")(self.ctx, C.c_float(reg_n), C.c_float(reg_l)), "mg_set_weights")

    def mg_pack_state(self):
        self._check(self._fn("mg_pack_state")(self.ctx), "mg_pack_state")

    def mg_unpack_state(self):
        self._check(self._fn("mg_unpack_state")(self.ctx), "mg_unpack_state")

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
