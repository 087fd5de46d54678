"""Host program of the z-slab multi-GPU path.

One process per GPU.  Every rank's context owns a contiguous range of band rows (= a z-slab of equal band count,
`psgsdf_comm_init`) and exposes the phases of a Gauss-Newton iteration plus the buffers that have to be exchanged
between them (`psgsdf_mg_*`).  This module runs the phases and does the exchanges with torch.distributed:
backend "nccl" (= RCCL over xGMI) on the GPUs, "gloo" in the CPU tests where the same code drives the oracle.

Exchanges per iteration (all <= 1 MiB, latency-bound):
  all-reduce : per-frame light / pose rows (F x 64 doubles), folded scalars (energies, counts), 3 PCG scalars / pass
  halo       : contiguous row ranges [row0-halo,row0) / [row1,row1+halo) of `blk` (14 planes, once), `{z,p}` pairs
               (once per PCG pass) and `dist` (once) with the two z-neighbours only (a chain: <= 2 xGMI links per GPU)
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import capi

BUF_FRAME_ACC, BUF_SCAL, BUF_PCG, BUF_DIST, BUF_BLK, BUF_ZP, BUF_RHO, BUF_GRAD = range(8)
(PH_ENERGY, PH_INIT_ALBEDO, PH_LED_SUMS, PH_LED_SET, PH_SWEEP_ALBEDO, PH_APPLY_ALBEDO, PH_SWEEP_LIGHT, PH_SOLVE_LIGHT,
 PH_SWEEP_POSE, PH_SOLVE_POSE, PH_SWEEP_DIST, PH_ASSEMBLE, PH_PCG_INIT, PH_PCG_MV, PH_PCG_UPD, PH_APPLY_DIST, PH_DERIVE,
 PH_SET_REG_SUMS) = range(18)
FROW = 64


class _DevArray:
    """minimal __cuda_array_interface__ carrier so torch can alias engine-owned device memory"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _alias(ptr, n, dtype, cuda):
    if cuda:
        return torch.as_tensor(_DevArray(ptr, n, "<f8" if dtype == torch.float64 else "<f4"), device="cuda")
    ct = ctypes.c_double if dtype == torch.float64 else ctypes.c_float
    arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(n,))
    return torch.from_numpy(arr)


class SlabRunner:
    """Drives one rank's context (HIP engine or CPU oracle) through the alternation loop."""

    def __init__(self, api: capi.Api, dist=None, cuda=False):
        self.api, self.dist, self.cuda = api, dist, cuda
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.i = api.mg_info()
        assert self.i["n_ranks"] == self.world and self.i["rank"] == self.rank, "psgsdf_comm_init must match the process group"
        self.S, self.Spad, self.r0, self.r1, self.halo = (self.i[k] for k in ("S", "Spad", "row0", "row1", "halo"))
        self.led = api._settings.model == capi.LED
        self.quirks = bool(api._settings.ref_quirks)
        f64, f32 = torch.float64, torch.float32
        self.frame = _alias(*api.mg_buffer(BUF_FRAME_ACC), f64, cuda).view(-1, FROW)
        self.scal = _alias(*api.mg_buffer(BUF_SCAL), f64, cuda)
        self.ext = _alias(*api.mg_buffer(BUF_PCG), f64, cuda)
        self.t_dist = _alias(*api.mg_buffer(BUF_DIST), f32, cuda).view(1, self.Spad)
        self.t_blk = _alias(*api.mg_buffer(BUF_BLK), f32, cuda).view(14, self.Spad)
        self.t_zp = _alias(*api.mg_buffer(BUF_ZP), f32, cuda).view(1, 2 * self.Spad)
        self.t_rho = _alias(*api.mg_buffer(BUF_RHO), f32, cuda).view(3, self.Spad)
        self.t_grad = _alias(*api.mg_buffer(BUF_GRAD), f32, cuda).view(3, self.Spad)
        info = api.info()
        self.reg_n, self.reg_l = float(info.reg_weight_n), float(info.reg_weight_l)
        self.last_cg = 0
        self.n_collectives = 0
        # gloo cannot send/recv device tensors: stage halos through the host (test configuration only)
        self.stage = bool(cuda and dist is not None and dist.get_backend() == "gloo")
        if self.led:                       # computeLightIntensive needs sums over every rank's rows
            self.api.mg_phase(PH_LED_SUMS)
            self._allreduce(self.scal[:6])
            self.api.mg_phase(PH_LED_SET)
        self.api.mg_pack_state()           # oracle: dense grid -> exchange planes (no-op for the engine)
        self._halo(self.t_dist)
        self.e_n, self.e_l = self._derive(0)

    # ---- exchanges
    def _allreduce(self, t):
        if self.world > 1:
            self.dist.all_reduce(t)
            self.n_collectives += 1

    def _halo(self, t, width=1):
        """exchange the halo row ranges of every plane of t ([planes, Spad*width]) with the z-neighbours"""
        if self.world == 1 or self.halo == 0:
            return
        H, r0, r1, S, w = self.halo, self.r0, self.r1, self.S, width
        ops, recvs = [], []

        def send(view, peer):
            ops.append(self.dist.P2POp(self.dist.isend, view.cpu() if self.stage else view, peer))

        def recv(view, peer):
            buf = torch.empty(view.shape, dtype=view.dtype) if self.stage else view
            if self.stage:
                recvs.append((view, buf))
            ops.append(self.dist.P2POp(self.dist.irecv, buf, peer))

        for p in range(t.shape[0]):
            if self.rank > 0:
                send(t[p, r0 * w:min(r0 + H, r1) * w], self.rank - 1)
                recv(t[p, max(r0 - H, 0) * w:r0 * w], self.rank - 1)
            if self.rank < self.world - 1:
                send(t[p, max(r1 - H, r0) * w:r1 * w], self.rank + 1)
                recv(t[p, r1 * w:min(r1 + H, S) * w], self.rank + 1)
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
            for view, buf in recvs:
                view.copy_(buf)
            self.n_collectives += 1

    def _scal(self, n):
        self._allreduce(self.scal[:n])
        return self.scal[:n].tolist()      # host read (synchronises)

    # ---- building blocks
    def _derive(self, update_grad):
        self.api.mg_phase(PH_DERIVE, update_grad)
        en, el = self._scal(2)
        self.api.mg_phase(PH_SET_REG_SUMS)
        return en / self.S, el / self.S

    def energy(self):
        self.api.mg_phase(PH_ENERGY)
        e, n = self._scal(2)
        return e / self.S, int(n)

    def init_albedo(self):
        self.api.mg_phase(PH_INIT_ALBEDO)

    def total(self, E, E_n, E_l):
        f = np.float32
        return float(f(E) + f(self.reg_n) * f(E_n) + f(self.reg_l) * f(E_l))

    def normalize_weights(self):
        """PsOptimizer.cpp:274-285"""
        E = np.float32(self.energy()[0]); E_n = E_l = np.float32(0)
        if self.reg_n != 0.0:
            E_n = np.float32(self.e_n); self.reg_n = float(np.float32(self.reg_n) * (E / E_n))
        if self.reg_l != 0.0:
            E_l = np.float32(self.e_l); self.reg_l = float(np.float32(self.reg_l) * (E / E_l))
        self.api.mg_set_weights(self.reg_n, self.reg_l)
        return self.total(E, E_n, E_l)

    def _pcg(self):
        api = self.api
        api.mg_phase(PH_PCG_INIT)
        self._allreduce(self.ext[:2])
        cap = api._settings.cg_max_it if api._settings.cg_max_it > 0 else min(2 * self.S, 4096)
        k, chunk = 0, max(4, self.last_cg + 1)
        while True:
            n = min(chunk, cap - k, 60)
            for q in range(n):
                self._halo(self.t_zp, 2)
                api.mg_phase(PH_PCG_MV, k + q)
                self._allreduce(self.ext[2:3])
                api.mg_phase(PH_PCG_UPD, k + q)
                self._allreduce(self.ext[:2])
            iters, err = api.mg_pcg_status(k, n)
            if iters >= 0:
                break
            k += n
            if k >= cap:
                iters = cap
                break
            chunk = 4
        self.last_cg = iters
        return iters, err, err <= float(np.finfo(np.float32).eps)

    def step(self, block, laplacian_reg=None):
        api = self.api
        lap = int(self.reg_l != 0.0) if laplacian_reg is None else int(laplacian_reg)
        st = dict(block=block, cg_iters=0, cg_converged=1, applied=1, cg_error=0.0)
        if block == capi.ALBEDO:
            api.mg_phase(PH_SWEEP_ALBEDO)
            e, n = self._scal(2)
            api.mg_phase(PH_APPLY_ALBEDO)
            st["n_accepted"] = int(self._scal(1)[0])
        elif block in (capi.LIGHT, capi.POSE):
            sweep, solve = (PH_SWEEP_LIGHT, PH_SOLVE_LIGHT) if block == capi.LIGHT else (PH_SWEEP_POSE, PH_SOLVE_POSE)
            api.mg_phase(sweep)
            self._allreduce(self.frame)
            if block == capi.POSE:
                col = 27
            else:
                nb = 3 if self.led else (9 if api._settings.model == capi.SH2 else 4)
                col = (3 if self.led else nb * (nb + 1) // 2) + nb
            e, n = self.frame[:, col].sum().item(), self.frame[:, col + 1].sum().item()
            api.mg_phase(solve)
        elif block == capi.DIST:
            api.mg_phase(PH_SWEEP_DIST, lap)
            e, n = self._scal(2)
            self._halo(self.t_blk)
            api.mg_phase(PH_ASSEMBLE)
            iters, err, ok = self._pcg()
            apply = not ((not self.led) and self.quirks and not ok)      # PsOptimizer.cpp:168-170 (B8)
            st.update(cg_iters=iters, cg_error=err, cg_converged=int(ok), applied=int(apply))
            if apply:
                api.mg_phase(PH_APPLY_DIST)
                st["n_accepted"] = int(self._scal(1)[0])
                self._halo(self.t_dist)
                self.e_n, self.e_l = self._derive(1)
        else:
            raise ValueError(block)
        st["e_in"], st["n_obs"] = e / self.S, int(n)
        return st

    def iterate(self, flags, n_iters):
        """n bodies of the alternation loop (PsOptimizer.cpp:303-366) with the engine's energy bookkeeping."""
        f = np.float32
        E = f(self.energy()[0])
        E_n = f(self.e_n) if self.reg_n != 0.0 else f(0)
        E_l = f(self.e_l) if self.reg_l != 0.0 else f(0)
        E_prev = f(self.total(E, E_n, E_l))
        lap = self.reg_l != 0.0
        order = [capi.LIGHT, capi.ALBEDO, capi.DIST, capi.POSE] if self.led else [capi.ALBEDO, capi.LIGHT, capi.DIST, capi.POSE]
        slot = {capi.ALBEDO: 0, capi.LIGHT: 1, capi.DIST: 2, capi.POSE: 3}
        recs = []
        for _ in range(n_iters):
            rec = dict(e_after=[float("nan")] * 4, cg_iters=0)
            pending = None
            for blk in order:
                if not (flags & blk):
                    continue
                st = self.step(blk, lap)
                if pending is not None:
                    E = f(st["e_in"]); rec["e_after"][pending] = float(E)
                if blk == capi.DIST:
                    rec["cg_iters"] = st["cg_iters"]
                    if self.reg_n != 0.0:
                        E_n = f(self.e_n)
                    if lap:
                        E_l = f(self.e_l)
                pending = slot[blk]
            if pending is not None:
                E = f(self.energy()[0]); rec["e_after"][pending] = float(E)
            Et = f(self.total(E, E_n, E_l))
            rec.update(e_n=float(E_n), e_l=float(E_l), e_total=float(Et), rel_diff=float(abs(E_prev - Et) / E_prev))
            E_prev = Et
            recs.append(rec)
        self.gather_state()
        return recs

    def gather_state(self):
        """every rank ends up with the whole refined band (dist, albedo, gradient) for download / writers"""
        if self.world == 1:
            return
        self.api.mg_pack_state()
        C = (self.S + self.world - 1) // self.world
        for t in (self.t_dist, self.t_rho, self.t_grad):
            for p in range(t.shape[0]):
                for r in range(self.world):
                    a, b = min(r * C, self.S), min((r + 1) * C, self.S)
                    if b > a:
                        self.dist.broadcast(t[p, a:b], src=r)
        self.api.mg_unpack_state()
