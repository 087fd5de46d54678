"""worker of tests/test_slab_gpu.py: one rank of a world_size-N run of the engine's NATIVE z-slab loop (psgsdf_iterate / psgsdf_optimize on a
context attached to a rank).  transport "rccl": the engine's own RCCL communicator (one rank per device: world 1 on the one-GPU box);
transport "gloo": all ranks share cuda:0 and the exchanges go through the caller-supplied transport of tests/_gloo_transport.py."""
import faulthandler
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(rank, world, port, model, out, n_iters, N, transport, mode):
    faulthandler.dump_traceback_later(int(os.environ.get("SLAB_WORKER_TIMEOUT", "110")), exit=True)
    import torch
    from psgradientsdf_amd import capi, synth
    torch.cuda.set_device(0)
    model, _, opt = model.partition("+")
    if os.environ.get("SLAB_CU_MASKS"):      # two ranks on ONE GPU with disjoint halves of its CUs (the cross-rank persistent solve needs both kernels resident)
        os.environ["PSGSDF_CU_MASK"] = os.environ["SLAB_CU_MASKS"].split(",")[rank]
    if os.environ.get("SLAB_FAULT_HALO") and rank == 1:      # this rank's n-th halo exchange pushes nothing (the neighbours' waits are bounded: tests)
        os.environ["PSGSDF_FAULT_HALO"] = os.environ["SLAB_FAULT_HALO"]
    if os.environ.get("SLAB_FAULT_HALO"):                    # (fault injection exists in the development build only: every rank of such a run loads it)
        os.environ["PSGSDF_USE_DEV_LIB"] = "1"
    # SLAB_FRAMES=F[:W:H]: another keyframe count (> 64: two visibility words per voxel) / image size than the default scene of these tests
    fr = [int(x) for x in os.environ.get("SLAB_FRAMES", "").split(":") if x]
    sc = synth.make_scene(N=N, F=fr[0] if fr else (5 if mode == "optimize" else 6), W=fr[1] if len(fr) > 1 else 160, H=fr[2] if len(fr) > 2 else 120, model=model)
    kw = {"reg_weight_rho": 0.02} if opt == "reg" else {}      # "+reg": the albedo regulariser ("reg albedo")
    if mode == "optimize":
        kw.update(upsample=1, max_it=n_iters, conv_threshold=0.0, damping=10.0)      # through the 2x refinement after iteration 5 (tests/test_parity_gpu.py test_optimize_matches_oracle's recipe)
        if model == "LED":
            kw.update(reg_weight_n=0.1, reg_weight_l=5.0)
    st = capi.default_settings(sc.model_id, **kw)
    eng = capi.load_engine(sc, sc.K, st, 0)
    tr = None
    if transport == "rccl":
        assert world == 1
        eng.comm_init(rank, world, capi.comm_unique_id())
    elif transport == "rccldup":      # dry run of the RCCL path with several ranks on ONE device: RCCL must refuse, the engine must report it (no hang)
        import time
        idf = out + ".id"
        if rank == 0:
            open(idf + ".tmp", "wb").write(capi.comm_unique_id()); os.replace(idf + ".tmp", idf)
        t0 = time.time()
        while not os.path.exists(idf) and time.time() - t0 < 60:
            time.sleep(0.05)
        try:
            eng.comm_init(rank, world, open(idf, "rb").read())
        except capi.PsgsdfError as ex:
            print("COMM_ERROR", ex, flush=True)
            sys.exit(3)
        print("COMM_OK", flush=True)
        sys.exit(0)
    elif transport == "sockets":      # the engine's built-in node-local transport over the socket pairs the test made (psgsdf_comm_init_sockets)
        eng.comm_init_sockets([int(x) for x in os.environ["SLAB_FDS"].split(",")], rank, world)
    else:
        import torch.distributed as dist
        from _gloo_transport import GlooTransport
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        tr = GlooTransport(dist)
        eng.comm_init_ext(tr.ops, rank, world)
    track = None
    if mode.startswith("fuse"):   # slab-parallel front end: every rank fuses every frame into the planes it holds, tracks one frame against its slab
        eng.volume_init(sc.F)
        for f in range(sc.F):
            eng.integrate_frame(sc.images[f], sc.depth[f], eng.estimate_normals(sc.depth[f]), sc.poses_gt[f], f, z_min=0.05, z_max=10.0)
        P, iters, conv = eng.track(sc.depth[1], sc.poses_gt[0], z_min=0.05, z_max=10.0, num_iterations=3)
        track = np.concatenate([P.reshape(-1), [iters, conv]])
        cut_before = eng.mg_info() if world > 1 else None
        if mode == "fuse_rebalance":
            eng.rebalance_slabs()
        fused = eng.download_volume(); fused_vis = eng.download_vis_seq(1)
        eng.set_keyframes(np.arange(sc.F, dtype=np.int32), sc.images, sc.poses); eng.init()
    elif mode == "iterate_slab":    # slab-local upload: this rank only ever hands over its own planes (+ halo) of the volume
        eng.load_scene_slab(sc, rank, world)
    else:
        eng.load_scene(sc)
    eng.init_albedo()
    e0 = eng.normalize_weights()
    conv = -1
    if mode == "optimize":        # the product loop with its stop decisions (and, on multi-rank contexts too, the speculative start of the next iteration)
        recs, conv = eng.optimize(capi.ALL)
    else:
        recs = eng.iterate(capi.ALL, n_iters)
    if mode == "refine":          # the 2x refinement of PsOptimizer.cpp:386-409 between iterations: gather, refine, new partition
        eng.upsample2x()
        recs += eng.iterate(capi.ALL, 1)
    info = eng.mg_info()
    v = eng.download_volume()          # whole-volume arrays, NaN outside the z-planes this rank owns
    xs = {}
    if os.environ.get("SLAB_EXTRACT"):      # this rank's share of the writers' geometry (collective calls) + a host-side sum over the ranks
        mx, mc = eng.extract_mesh(); p0, c0 = eng.extract_pointcloud(0); p1, c1 = eng.extract_pointcloud(1); lo, dim, blk = eng.extract_sdf()
        cnt = np.zeros(world); cnt[rank] = len(mx)
        xs = dict(x_mesh_xyz=mx, x_mesh_rgb=mc, x_pc0=p0, x_pc0_rgb=c0, x_pc1=p1, x_pc1_rgb=c1, x_sdf_lo=lo, x_sdf_dim=dim, x_sdf=blk, x_counts=eng.comm_allreduce_host(cnt), x_weight=v["weight"])
    np.savez(out + f".rank{rank}.npz", dist=v["dist"], rgb=v["rgb"], grad=v["grad"], poses=eng.download_poses(), light=eng.download_light(),
             e_total=[r["e_total"] for r in recs], cg=[r["cg_iters"] for r in recs], e0=e0, band=eng.download_band(info["row1"] - info["row0"]),
             info=[info["row0"], info["row1"], info["halo"], info["S"], info["need_lo"], info["need_hi"], info["z0"], info["z1"], info["rows"]],
             track=track if track is not None else np.zeros(0), fused_weight=fused["weight"] if mode.startswith("fuse") else np.zeros(0), fused_dist=fused["dist"] if mode.startswith("fuse") else np.zeros(0),
             fused_vis=fused_vis if mode.startswith("fuse") else np.zeros(0), cut_before=[cut_before["z0"], cut_before["z1"]] if mode.startswith("fuse") and world > 1 else [0, 0],
             halo_pushes=eng.debug_sync_stats()["halo_pushes"], conv=conv, spec=[eng.debug_sync_stats()[k] for k in ("speculative_starts", "speculative_undos")], upsampled=[int(r["upsampled"]) for r in recs],
             **xs, ncoll=eng.comm_stats(), dim=list(eng.info().dim), n_band=eng.info().n_band, xr=[eng.debug_sync_stats()[k] for k in ("cross_rank_ready", "cross_rank_solves", "persist_fallbacks", "cross_rank_mem_kind", "probe_stale", "probe_timeouts")])
    eng.close()
    if tr is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), sys.argv[8], sys.argv[9])
