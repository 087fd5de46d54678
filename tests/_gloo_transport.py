"""A caller-supplied transport (psgsdf_comm_ops) built on a torch.distributed gloo process group: TEST INFRASTRUCTURE.  It lets two ranks
share ONE GPU -- which RCCL refuses -- so that the engine's native multi-rank loop (halo exchanges, all-reduces) can be
exercised on a one-GPU box.  Every primitive synchronises the device, stages through the host and blocks: correct, and slow on purpose."""
import ctypes as C

import torch

from psgradientsdf_amd import capi


class _DevBytes:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _alias(ptr, n, typestr):
    return torch.as_tensor(_DevBytes(ptr, n, typestr), device="cuda")


class GlooTransport:
    def __init__(self, dist):
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()
        self.calls = 0
        self._f = (capi.ALLREDUCE_FN(self._allreduce), capi.SENDRECV_FN(self._sendrecv))
        self.ops = capi.CommOps(None, *self._f)

    def _guard(self, fn):
        try:
            fn(); self.calls += 1
            return 0
        except Exception as e:      # never let an exception cross the C boundary
            import traceback; traceback.print_exc()
            return 1

    def _allreduce(self, user, buf, n, stream):
        def run():
            torch.cuda.synchronize()
            t = _alias(buf, n, "<f8"); h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h); torch.cuda.synchronize()
        return self._guard(run)

    def _sendrecv(self, user, sends, ns, recvs, nr, stream):
        def run():
            torch.cuda.synchronize()
            ops, back = [], []
            for i in range(ns):
                x = sends[i]
                ops.append(self.dist.P2POp(self.dist.isend, _alias(x.ptr_dev, x.bytes, "|u1").cpu(), x.peer))
            for i in range(nr):
                x = recvs[i]
                h = torch.empty(x.bytes, dtype=torch.uint8)
                back.append((_alias(x.ptr_dev, x.bytes, "|u1"), h))
                ops.append(self.dist.P2POp(self.dist.irecv, h, x.peer))
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
            for t, h in back:
                t.copy_(h)
            torch.cuda.synchronize()
        return self._guard(run)
