"""worker of tests/test_slab_gloo.py: one rank of a world_size-N gloo run of the slab host program on the oracle"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(rank, world, port, model, out, n_iters, N, tile=False):
    import torch.distributed as dist
    from psgradientsdf_amd import capi, synth
    from _slab_runner import SlabRunner
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(N=N, F=5, W=128, H=96, model=model)
    if tile:          # bench.py's weak-scaling scene: one copy of the scene (and of its keyframes) per rank, stacked along z
        sc = synth.tile_scene(sc, world)
    st = capi.default_settings(sc.model_id, reg_weight_l=2.0 if model == "SH1" else 0.0)
    o = oracle.Oracle(sc, sc.K, st)
    o.comm_init(rank, world)
    o.load_scene(sc)
    run = SlabRunner(o, dist)
    run.init_albedo()
    e0 = run.normalize_weights()
    recs = run.iterate(capi.ALL, n_iters)
    v = o.download_volume()
    np.savez(out + f".rank{rank}.npz", dist=v["dist"], rgb=v["rgb"], grad=v["grad"], poses=o.download_poses(), light=o.download_light(),
             e_total=[r["e_total"] for r in recs], cg=[r["cg_iters"] for r in recs], e0=e0, info=[run.r0, run.r1, run.halo, run.S],
             ncoll=run.n_collectives)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), len(sys.argv) > 8 and sys.argv[8] == "tile")
