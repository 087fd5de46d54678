"""worker of tests/test_slab_gpu.py: one rank of a world_size-N run of the slab host program on the HIP engine.
All ranks share cuda:0 (the GPU box has one device); backend gloo (RCCL refuses two ranks on one device)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(rank, world, port, model, out, n_iters, N, backend):
    import torch
    import torch.distributed as dist
    from psgradientsdf_amd import capi, synth
    from psgradientsdf_amd.distributed import SlabRunner
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    sc = synth.make_scene(N=N, F=6, W=160, H=120, model=model)
    st = capi.default_settings(sc.model_id)
    eng = capi.load_engine(sc, sc.K, st, 0)
    eng.comm_init(rank, world)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.load_scene(sc)
    run = SlabRunner(eng, dist, cuda=True)
    run.init_albedo()
    e0 = run.normalize_weights()
    recs = run.iterate(capi.ALL, n_iters)
    torch.cuda.synchronize()
    v = eng.download_volume()
    np.savez(out + f".rank{rank}.npz", dist=v["dist"], rgb=v["rgb"], grad=v["grad"], poses=eng.download_poses(), light=eng.download_light(),
             e_total=[r["e_total"] for r in recs], cg=[r["cg_iters"] for r in recs], e0=e0, info=[run.r0, run.r1, run.halo, run.S], ncoll=run.n_collectives)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), sys.argv[8])
