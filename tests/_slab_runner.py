"""Host program of the z-slab ALGORITHM on the CPU oracle's phase mirror (`orc_mg_*`) -- TEST INFRASTRUCTURE (tests/test_slab_gloo.py).

The HIP engine runs its slab loop natively: its C++ host (psgradientsdf_amd/csrc/comm.hip, loop.hip, engine.hip) issues every exchange on
the engine's stream over its own RCCL communicator, and exports no phase API.  What remains here is the same algorithm spelled out phase
by phase over torch.distributed (gloo), driving one oracle context per rank, so that the partition, the halo ranges, the fused PCG with
globally reduced sums and the energy bookkeeping can be checked on CPU with world sizes 2-4 against the single-rank oracle.

Every rank's context owns a contiguous range of band rows (= a z-slab of equal band count) and exposes the phases of a Gauss-Newton
iteration plus the buffers that have to be exchanged between them.  A phase folds its scalars (energies, counts) into its own slots of
the SCAL buffer, which is all-reduced and read ONCE per iteration; the only other host reads are the PCG stop checks.

Exchanges per iteration (all <= 1 MiB, latency-bound):
  all-reduce : per-frame light / pose rows (F x 64 doubles, twice), the 7 sums of a PCG pass (once per pass), the
               iteration's folded scalars (once)
  halo       : the row ranges the stencils of a slab actually reach into its two z-neighbours (`need_lo`/`need_hi` of
               mg_info): `blk` (14 planes, once), the PCG records (16 B/row, once per pass) and `dist` (once).
               Slabs whose stencils do not cross the cut exchange nothing.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from psgradientsdf_amd import capi

BUF_FRAME_ACC, BUF_SCAL, BUF_PCG, BUF_DIST, BUF_BLK, BUF_REC0, BUF_RHO, BUF_GRAD, BUF_REC1 = range(9)
(PH_ENERGY, PH_INIT_ALBEDO, PH_LED_SUMS, PH_LED_SET, PH_SWEEP_ALBEDO, PH_APPLY_ALBEDO, PH_SWEEP_LIGHT, PH_SOLVE_LIGHT,
 PH_SWEEP_POSE, PH_SOLVE_POSE, PH_SWEEP_DIST, PH_ASSEMBLE, PH_PCG_INIT, PH_PCG_PASS, PH_APPLY_DIST, PH_DERIVE) = range(16)
FROW = 64
# slots of the SCAL buffer within one iteration (local sums until the single all-reduce at its end)
SL_ALB_E, SL_ALB_ACC, SL_DIST_E, SL_DIST_ACC, SL_REG, SL_CLOSE, SL_N = 0, 2, 3, 5, 6, 8, 10


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
        if cuda:   # the collectives below run on torch's current stream: the engine's kernels must be ordered on the same one
            api.set_stream(torch.cuda.current_stream().cuda_stream)
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.i = api.mg_info()
        assert self.i["n_ranks"] == self.world and self.i["rank"] == self.rank, "psgsdf_comm_init must match the process group"
        self.S, self.Spad, self.r0, self.r1 = (self.i[k] for k in ("S", "Spad", "row0", "row1"))
        self.led = api._settings.model == capi.LED
        self.quirks = bool(api._settings.ref_quirks)
        f64, f32 = torch.float64, torch.float32
        self.frame = _alias(*api.mg_buffer(BUF_FRAME_ACC), f64, cuda).view(-1, FROW)
        self.scal = _alias(*api.mg_buffer(BUF_SCAL), f64, cuda)
        self.ext = _alias(*api.mg_buffer(BUF_PCG), f64, cuda)
        self.t_dist = _alias(*api.mg_buffer(BUF_DIST), f32, cuda).view(1, self.Spad)
        self.t_blk = _alias(*api.mg_buffer(BUF_BLK), f32, cuda).view(14, self.Spad)
        self.t_rec = [_alias(*api.mg_buffer(b), f32, cuda).view(1, 4 * self.Spad) for b in (BUF_REC0, BUF_REC1)]
        self.t_rho = _alias(*api.mg_buffer(BUF_RHO), f32, cuda).view(3, self.Spad)
        self.t_grad = _alias(*api.mg_buffer(BUF_GRAD), f32, cuda).view(3, self.Spad)
        info = api.info()
        self.reg_n, self.reg_l = float(info.reg_weight_n), float(info.reg_weight_l)
        self.last_cg = 0
        self.n_collectives = 0
        # gloo cannot send/recv device tensors: stage halos through the host (test configuration only)
        self.stage = bool(cuda and dist is not None and dist.get_backend() == "gloo")
        # what each neighbour needs of this slab (its reach across the cut), what this slab needs of them
        self.need_lo, self.need_hi, self.halo = self.i["need_lo"], self.i["need_hi"], self.i["halo"]
        self.give_lo = self.give_hi = 0
        if self.world > 1:
            needs = torch.tensor([self.need_lo, self.need_hi], dtype=torch.int64, device="cuda" if (cuda and not self.stage) else "cpu")
            allneeds = [torch.zeros_like(needs) for _ in range(self.world)]
            dist.all_gather(allneeds, needs)
            allneeds = [t.tolist() for t in allneeds]
            if self.rank > 0:
                self.give_lo = min(allneeds[self.rank - 1][1], self.r1 - self.r0)     # the lower neighbour reaches up into my first rows
            if self.rank < self.world - 1:
                self.give_hi = min(allneeds[self.rank + 1][0], self.r1 - self.r0)     # the upper neighbour reaches down into my last rows
        self.halo_active = (self.need_lo + self.need_hi + self.give_lo + self.give_hi) > 0
        if self.led:                       # computeLightIntensive needs sums over every rank's rows
            api.mg_fold_base(0)
            api.mg_phase(PH_LED_SUMS)
            self._allreduce(self.scal[:6])
            api.mg_phase(PH_LED_SET)
        api.mg_pack_state()                # oracle: dense grid -> exchange planes (no-op for the engine)
        self._halo(self.t_dist)
        api.mg_fold_base(0)
        api.mg_phase(PH_DERIVE, 0)
        en, el = self._scal_now(0, 2)
        self._set_reg(en, el)

    # ---- exchanges
    def _allreduce(self, t):
        if self.world > 1:
            self.dist.all_reduce(t)
            self.n_collectives += 1

    def _halo(self, t, width=1):
        """exchange the halo row ranges of every plane of t ([planes, Spad*width]) with the z-neighbours"""
        if self.world == 1 or not self.halo_active:
            return
        r0, r1, w = self.r0, self.r1, width
        ops, recvs = [], []

        def send(view, peer):
            ops.append(self.dist.P2POp(self.dist.isend, view.cpu() if self.stage else view, peer))

        def recv(view, peer):
            buf = torch.empty(view.shape, dtype=view.dtype) if self.stage else view
            if self.stage:
                recvs.append((view, buf))
            ops.append(self.dist.P2POp(self.dist.irecv, buf, peer))

        for p in range(t.shape[0]):
            if self.give_lo:
                send(t[p, r0 * w:(r0 + self.give_lo) * w], self.rank - 1)
            if self.need_lo:
                recv(t[p, (r0 - self.need_lo) * w:r0 * w], self.rank - 1)
            if self.give_hi:
                send(t[p, (r1 - self.give_hi) * w:r1 * w], self.rank + 1)
            if self.need_hi:
                recv(t[p, r1 * w:(r1 + self.need_hi) * w], self.rank + 1)
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
            for view, buf in recvs:
                view.copy_(buf)
            self.n_collectives += 1

    def _scal_now(self, base, n):
        """all-reduce and read n folded scalars at once (host sync): set-up paths and the synchronous step()"""
        self._allreduce(self.scal[base:base + n])
        return self.scal[base:base + n].tolist()

    def _set_reg(self, en_sum, el_sum):
        self.e_n, self.e_l = en_sum / self.S, el_sum / self.S
        self.api.mg_set_reg_sums(en_sum, el_sum)

    # ---- building blocks
    def energy(self):
        self.api.mg_fold_base(SL_CLOSE)
        self.api.mg_phase(PH_ENERGY)
        e, n = self._scal_now(SL_CLOSE, 2)
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
        """fused Jacobi-PCG: per pass one halo exchange of the records, one kernel, one all-reduce of 7 doubles; the stop
        test reads back once per chunk (every rank sees the same all-reduced sums, so all ranks stop together)"""
        api = self.api
        api.mg_phase(PH_PCG_INIT)
        self._allreduce(self.ext[:7])
        cap = api._settings.cg_max_it if api._settings.cg_max_it > 0 else min(2 * self.S, 4096)
        k, chunk = 0, max(4, self.last_cg + 2)
        while True:
            n = min(chunk, cap + 1 - k, 64)
            for q in range(n):
                self._halo(self.t_rec[(k + q + 1) & 1], 4)
                api.mg_phase(PH_PCG_PASS, k + q)
                self._allreduce(self.ext[:7])
            iters, err = api.mg_pcg_status(k, n)
            if iters >= 0:
                break
            k += n
            chunk = 4
        self.last_cg = iters
        return iters, err, err <= float(np.finfo(np.float32).eps)

    def _frame_energy(self, block):
        if block == capi.POSE:
            col = 27
        else:
            nb = 3 if self.led else (9 if self.api._settings.model == capi.SH2 else 4)
            col = (3 if self.led else nb * (nb + 1) // 2) + nb
        return self.frame[:, col:col + 2].sum(dim=0)     # device tensor {energy sum, n_obs}: no host sync

    def _step_async(self, block, lap):
        """enqueue one block; returns (stats dict, deferred) where deferred maps stat name -> SCAL slot or device tensor"""
        api = self.api
        st = dict(block=block, cg_iters=0, cg_converged=1, applied=1, cg_error=0.0)
        if block == capi.ALBEDO:
            api.mg_fold_base(SL_ALB_E); api.mg_phase(PH_SWEEP_ALBEDO)
            api.mg_fold_base(SL_ALB_ACC); api.mg_phase(PH_APPLY_ALBEDO)
            return st, dict(e=SL_ALB_E, acc=SL_ALB_ACC)
        if block in (capi.LIGHT, capi.POSE):
            sweep, solve = (PH_SWEEP_LIGHT, PH_SOLVE_LIGHT) if block == capi.LIGHT else (PH_SWEEP_POSE, PH_SOLVE_POSE)
            api.mg_phase(sweep)
            self._allreduce(self.frame)
            en = self._frame_energy(block)
            api.mg_phase(solve)
            return st, dict(e=en)
        if block == capi.DIST:
            api.mg_fold_base(SL_DIST_E); api.mg_phase(PH_SWEEP_DIST, lap)
            self._halo(self.t_blk)
            api.mg_phase(PH_ASSEMBLE)
            iters, err, ok = self._pcg()
            apply = not ((not self.led) and self.quirks and not ok)      # PsOptimizer.cpp:168-170 (B8)
            st.update(cg_iters=iters, cg_error=err, cg_converged=int(ok), applied=int(apply))
            d = dict(e=SL_DIST_E)
            if apply:
                api.mg_fold_base(SL_DIST_ACC); api.mg_phase(PH_APPLY_DIST)
                self._halo(self.t_dist)
                api.mg_fold_base(SL_REG); api.mg_phase(PH_DERIVE, 1)
                d.update(acc=SL_DIST_ACC, reg=SL_REG)
            return st, d
        raise ValueError(block)

    def _resolve_enqueue(self, deferred_list):
        """ONE all-reduce of the iteration's folded scalars, then an asynchronous copy to pinned host memory; nothing is
        read here.  Returns a ticket for _resolve_finish."""
        self._allreduce(self.scal[:SL_N])
        dev = [d["e"] for _, d in deferred_list if torch.is_tensor(d.get("e"))]
        allv = torch.cat([self.scal[:SL_N]] + [t.reshape(-1) for t in dev])
        ev = None
        if self.cuda:
            host = torch.empty(allv.shape, dtype=allv.dtype, pin_memory=True)
            host.copy_(allv, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
        else:
            host = allv.clone()
        return deferred_list, host, ev

    def _resolve_finish(self, ticket):
        """host read of an enqueued resolve (waits only if the copy has not landed yet); fills the stats of its blocks"""
        deferred_list, host, ev = ticket
        if ev is not None:
            ev.synchronize()
        vals = host.tolist()
        di = SL_N
        for st, d in deferred_list:
            if torch.is_tensor(d["e"]):
                e, n = vals[di], vals[di + 1]; di += 2
            else:
                e, n = vals[d["e"]], vals[d["e"] + 1]
            st["e_in"], st["n_obs"] = e / self.S, int(n)
            if "acc" in d:
                st["n_accepted"] = int(vals[d["acc"]])
            if "reg" in d:
                st["reg_sums"] = (vals[d["reg"]], vals[d["reg"] + 1])
        return vals

    def step(self, block, laplacian_reg=None):
        lap = int(self.reg_l != 0.0) if laplacian_reg is None else int(laplacian_reg)
        self.scal[:SL_N].zero_()
        st, d = self._step_async(block, lap)
        self._resolve_finish(self._resolve_enqueue([(st, d)]))
        if "reg_sums" in st:
            self._set_reg(*st.pop("reg_sums"))
        return st

    def iterate(self, flags, n_iters, gather=True):
        """n bodies of the alternation loop (PsOptimizer.cpp:303-366) with the engine's energy bookkeeping.

        The PS energy that closes an iteration is the input energy of the next iteration's first sweep (same state), and
        an iteration's scalars are read late, right after the next PCG stop check has synchronised the stream anyway:
        per iteration the host waits for the device once (twice in the last one)."""
        f = np.float32
        E = f(self.energy()[0])
        state = dict(E=E, E_n=f(self.e_n) if self.reg_n != 0.0 else f(0), E_l=f(self.e_l) if self.reg_l != 0.0 else f(0))
        state["E_prev"] = f(self.total(state["E"], state["E_n"], state["E_l"]))
        lap = self.reg_l != 0.0
        order = [b for b in ([capi.LIGHT, capi.ALBEDO, capi.DIST, capi.POSE] if self.led else [capi.ALBEDO, capi.LIGHT, capi.DIST, capi.POSE]) if flags & b]
        slot = {capi.ALBEDO: 0, capi.LIGHT: 1, capi.DIST: 2, capi.POSE: 3}
        recs, pending = [], []     # pending: iterations whose record is still open, oldest first

        def close(p, close_e):
            """finish the record of a resolved iteration; close_e = PS energy after its last block"""
            rec = dict(e_after=[float("nan")] * 4, cg_iters=0)
            last_slot = None
            for blk, st, _ in p["ran"]:
                if last_slot is not None:      # e_in of a sweep is the energy AFTER the block that ran before it
                    state["E"] = f(st["e_in"]); rec["e_after"][last_slot] = float(state["E"])
                if blk == capi.DIST:
                    rec["cg_iters"] = st["cg_iters"]
                    if "reg_sums" in st:
                        self._set_reg(*st.pop("reg_sums"))
                    if self.reg_n != 0.0:
                        state["E_n"] = f(self.e_n)
                    if lap:
                        state["E_l"] = f(self.e_l)
                last_slot = slot[blk]
            if last_slot is not None:
                state["E"] = f(close_e); rec["e_after"][last_slot] = float(state["E"])
            Et = f(self.total(state["E"], state["E_n"], state["E_l"]))
            rec.update(e_n=float(state["E_n"]), e_l=float(state["E_l"]), e_total=float(Et), rel_diff=float(abs(state["E_prev"] - Et) / state["E_prev"]))
            state["E_prev"] = Et
            recs.append(rec)

        def finish_enqueued():
            for p in pending:
                if p["vals"] is None:
                    p["vals"] = self._resolve_finish(p["ticket"])

        def close_ready():
            while pending and pending[0]["vals"] is not None:
                p = pending[0]
                if p["explicit"]:
                    close(p, p["vals"][SL_CLOSE] / self.S)
                elif len(pending) > 1 and pending[1]["vals"] is not None:
                    close(p, pending[1]["ran"][0][1]["e_in"])      # the next iteration's first sweep saw this iteration's final state
                else:
                    break
                pending.pop(0)

        for it in range(n_iters):
            explicit = it + 1 == n_iters or not order
            self.scal[:SL_N].zero_()       # slots a skipped block leaves untouched must not be re-reduced
            ran = []
            for blk in order:
                st, d = self._step_async(blk, int(lap))
                ran.append((blk, st, d))
                if blk == capi.DIST:
                    finish_enqueued()      # the PCG stop check has just synchronised the stream: earlier tickets have landed
            if explicit:                   # closing PS energy (all other iterations are closed by the next first sweep)
                self.api.mg_fold_base(SL_CLOSE); self.api.mg_phase(PH_ENERGY)
            pending.append(dict(ran=ran, ticket=self._resolve_enqueue([(st, d) for _, st, d in ran]), vals=None, explicit=explicit))
            if explicit or capi.DIST not in order:
                finish_enqueued()
            close_ready()
        assert not pending
        if gather:
            self.gather_state()
        return recs

    def gather_state(self):
        """every rank ends up with the whole refined band (dist, albedo, gradient) for download / writers"""
        if self.world == 1:
            return
        self.api.mg_pack_state()
        C = (self.S + self.world - 1) // self.world        # rows per slab (the last one may be shorter: Spad covers world * C)
        for t in (self.t_dist, self.t_rho, self.t_grad):
            for p in range(t.shape[0]):
                for r in range(self.world):
                    a, b = min(r * C, self.S), min((r + 1) * C, self.S)
                    if b > a:
                        self.dist.broadcast(t[p, a:b], src=r)
        self.api.mg_unpack_state()
